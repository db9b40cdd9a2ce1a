"""Train / eval batch construction for the SASRec family (reference REC/data/dataset/{trainset,evalset,
collate_fn}.py) plus a vectorised whole-batch builder.

Reference semantics kept:
  * SEQTrainDataset item = (items [2, L+1], masked_index [L]): positives left-padded with 0; ONE negative per
    target position, uniform over [1, N-1] rejecting the sequence's own items (trainset.py:40-44), right-aligned
    (neg[0] is always 0); masked_index = 1 on the len-1 target positions (trainset.py:46-75);
  * SeqEvalDataset item = (history tensor, last-L left-padded item_seq, target): valid = [:-2] / [-2],
    test = [:-1] / [-1] (evalset.py:17-34); seq_eval_collate stacks them and emits the (history_u, history_i)
    pairs used to mask the whole history at eval (collate_fn.py:6-32).
What is new: `SeqTrainBatcher` builds whole batches with numpy (the per-sample Python `random` loop in 10 worker
processes is the bottleneck once a step takes ~2 ms on the GPU).
"""
from __future__ import annotations

import random

import numpy as np
import torch
from torch.utils.data import Dataset


class SEQTrainDataset(Dataset):
    def __init__(self, config, dataload):
        self.dataload = dataload
        self.item_num = dataload.item_num
        self.train_seq = dataload.train_feat["item_seq"]
        self.length = len(self.train_seq)
        self.max_seq_length = config["MAX_ITEM_LIST_LENGTH"] + 1

    def __len__(self):
        return self.length

    def _neg_sample(self, item_set):
        item = random.randint(1, self.item_num - 1)
        while item in item_set:
            item = random.randint(1, self.item_num - 1)
        return item

    @staticmethod
    def _pad(sequence, max_length):
        sequence = [0] * (max_length - len(sequence)) + list(sequence)
        return torch.tensor(sequence[-max_length:], dtype=torch.long)

    def __getitem__(self, index):
        item_seq = self.train_seq[index]
        n = len(item_seq)
        neg = [self._neg_sample(item_seq) for _ in range(n - 1)]
        mask = [1] * (n - 1)
        return (torch.stack((self._pad(list(item_seq), self.max_seq_length), self._pad(neg, self.max_seq_length))),
                self._pad(mask, self.max_seq_length - 1))


class SeqTrainBatcher:
    """Vectorised equivalent of DataLoader(SEQTrainDataset, sampler=DistributedSampler): yields whole
    (items [B,2,L+1], masked_index [B,L]) int64 batches.  The sample ORDER reproduces torch's DistributedSampler
    (shuffle with generator seed = seed + epoch, pad to a multiple of world by wrapping, rank-strided)."""

    def __init__(self, config, dataload, rank=0, world=1, seed=0, drop_last=False):
        self.dataload = dataload
        self.item_num = dataload.item_num
        self.W = config["MAX_ITEM_LIST_LENGTH"] + 1
        self.batch_size = config["train_batch_size"]
        seqs = dataload.train_feat["item_seq"]
        self.n = len(seqs)
        self.windows = np.zeros((self.n, self.W), dtype=np.int64)
        self.lens = np.zeros(self.n, dtype=np.int64)
        for i, s in enumerate(seqs):
            k = len(s)
            self.windows[i, self.W - k:] = s
            self.lens[i] = k
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0
        self.num_samples = -(-self.n // world)
        self.drop_last = drop_last
        self.neg_seed = int(config["seed"] or 0)
        # device_sampler: yield only the positive windows ([B, L+1] int64, pinned-copy friendly); the negatives and
        # the mask are then drawn on the GPU by ops.sample_negatives (pxr_sample_negatives_i64) -- the host work per
        # batch drops from ~0.8 ms of numpy to one fancy-index gather
        try:
            ds = config["device_sampler"]       # Config returns None for missing keys, plain dicts raise
        except KeyError:
            ds = None
        self.device_sampler = bool(ds) if ds is not None else False
        self._batch_counter = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples // self.batch_size if self.drop_last else -(-self.num_samples // self.batch_size)

    def _indices(self):
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        idx = torch.randperm(self.n, generator=g).tolist()
        total = self.num_samples * self.world
        pad = total - len(idx)
        if pad > 0:
            idx += (idx * ((pad + len(idx) - 1) // len(idx) + 1))[:pad]
        return np.asarray(idx[self.rank:total:self.world], dtype=np.int64)

    def make_batch(self, rows, rng):
        pos = self.windows[rows]                                   # [B, W]
        lens = self.lens[rows]
        B, W = pos.shape
        col = np.arange(W)[None, :]
        tgt = col >= (W - lens[:, None] + 1)                       # target positions: the last len-1 columns
        neg = rng.integers(1, self.item_num, size=(B, W))
        if self.item_num > 2 * W:                                  # rejection of the sequence's own items
            for _ in range(64):
                clash = (neg[:, :, None] == pos[:, None, :]).any(-1) & tgt
                if not clash.any():
                    break
                neg[clash] = rng.integers(1, self.item_num, size=int(clash.sum()))
        neg = np.where(tgt, neg, 0)
        items = np.stack((pos, neg), axis=1)
        return items, tgt[:, 1:].astype(np.int64)

    def estimate_exchange_rows(self, n_batches: int = 64, margin: float = 1.25) -> int:
        """A row capacity for the data-parallel exchange of the sparse table gradient (`dp_exchange_rows: auto`): the
        largest number of distinct item ids among the first `n_batches` batches of this rank's epoch-0 order (negatives
        drawn like the host sampler does), times `margin`, rounded up to 256 and capped at the worst case B*(2L+1).  It is
        a statistical bound -- a later batch that exceeds it sets the status word and the trainer raises at its next host
        sync (pxr_merge_split_rows_f32) -- and it costs no training state: the batcher's random streams are functions of
        (seed, epoch, rank) only."""
        worst = self.batch_size * (2 * self.W - 1)
        idx = self._indices()
        rng = np.random.default_rng([self.neg_seed, 0, self.rank, 12345])
        top = 0
        for b in range(min(n_batches, len(self))):
            rows = idx[b * self.batch_size:(b + 1) * self.batch_size]
            items, _ = self.make_batch(rows, rng)
            top = max(top, int(np.count_nonzero(np.unique(items))))
        return int(min(worst, (int(top * margin) + 255) // 256 * 256))

    def __iter__(self):
        idx = self._indices()
        rng = np.random.default_rng([self.neg_seed, self.epoch, self.rank])
        nb = len(self)
        for b in range(nb):
            rows = idx[b * self.batch_size:(b + 1) * self.batch_size]
            if self.device_sampler:
                # (windows, batch id): the consumer turns them into (items, mask) on the device
                yield torch.from_numpy(self.windows[rows]), (self.neg_seed, (self.epoch << 32) | (self.rank << 24) | b)
                continue
            items, mask = self.make_batch(rows, rng)
            yield torch.from_numpy(items), torch.from_numpy(mask)


class SeqEvalDataset(Dataset):
    def __init__(self, config, dataload, phase="valid"):
        self.dataload = dataload
        self.max_item_list_length = config["MAX_ITEM_LIST_LENGTH"]
        self.user_seq = list(dataload.user_seq.values())
        self.phase = phase
        self.length = len(self.user_seq)
        self.item_num = dataload.item_num

    def __len__(self):
        return self.length

    def _padding_sequence(self, sequence, max_length):
        sequence = [0] * (max_length - len(sequence)) + list(sequence)
        return sequence[-max_length:]

    def __getitem__(self, index):
        seq = self.user_seq[index]
        if self.phase == "valid":
            history_seq, item_target = seq[:-2], seq[-2]
        else:
            history_seq, item_target = seq[:-1], seq[-1]
        item_seq = self._padding_sequence(history_seq, self.max_item_list_length)
        return torch.tensor(np.asarray(history_seq, dtype=np.int64)), item_seq, int(item_target)


def seq_eval_collate(batch):
    history_i = [item[0] for item in batch]
    item_seq = torch.tensor([item[1] for item in batch], dtype=torch.long)
    item_target = torch.tensor([item[2] for item in batch], dtype=torch.long)
    history_u = torch.cat([torch.full_like(h, i) for i, h in enumerate(history_i)])
    history_i = torch.cat(history_i)
    positive_u = torch.arange(item_seq.shape[0])
    return item_seq, (history_u, history_i), positive_u, item_target


class SeqEvalBatcher:
    """Vectorised equivalent of DataLoader(SeqEvalDataset, sampler=NonConsecutiveSequentialDistributedSampler,
    collate_fn=seq_eval_collate) (reference evalset.py:4-36, collate_fn.py:6-32, data/utils.py:134-159): yields the same
    `(item_seq [b,L], (history_u, history_i), positive_u [b], item_target [b])` batches, built from a CSR image of the
    user sequences with numpy index arithmetic instead of one Python __getitem__ per user (200 K users: 1.6 s -> ms).
    Duck-types what the Trainer reads from a DataLoader: iteration, `len()`, `.dataset.dataload`, `.sampler.dataset`."""

    def __init__(self, config, dataload, phase="valid", rank=0, world=1):
        self.dataset = SeqEvalDataset(config, dataload, phase=phase)       # keeps the per-user API (and the tests) alive
        self.sampler = type("Sampler", (), {"dataset": self.dataset})()
        self.batch_size = config["eval_batch_size"]
        self.L = config["MAX_ITEM_LIST_LENGTH"]
        seqs = self.dataset.user_seq
        lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
        self.offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
        np.cumsum(lens, out=self.offsets[1:])
        self.flat = np.concatenate([np.asarray(s, dtype=np.int64) for s in seqs]) if len(seqs) else np.zeros(0, np.int64)
        self.cut = lens - (2 if phase == "valid" else 1)                   # history = seq[:cut], target = seq[cut]
        self.users = np.arange(rank, len(seqs), world, dtype=np.int64)     # rank r takes users r, r+W, ... (no padding)

    def __len__(self):
        return -(-len(self.users) // self.batch_size)

    def __iter__(self):
        L = self.L
        for b0 in range(0, len(self.users), self.batch_size):
            u = self.users[b0:b0 + self.batch_size]
            start, cut = self.offsets[u], self.cut[u]
            target = self.flat[start + cut]
            # history pairs (row in batch, item) for every past interaction
            hist_u = np.repeat(np.arange(len(u), dtype=np.int64), cut)
            first = np.cumsum(cut) - cut                                   # first pair of each user
            within = np.arange(int(cut.sum()), dtype=np.int64) - np.repeat(first, cut)
            hist_i = self.flat[np.repeat(start, cut) + within]
            # the last L history items, left-padded with 0
            col = np.arange(L, dtype=np.int64)[None, :]
            src = cut[:, None] - L + col                                   # position inside the history
            ok = src >= 0
            item_seq = np.where(ok, self.flat[np.where(ok, start[:, None] + src, 0)], 0)
            yield (torch.from_numpy(item_seq), (torch.from_numpy(hist_u), torch.from_numpy(hist_i)),
                   torch.arange(len(u)), torch.from_numpy(target))
