"""Synthetic inputs of the reference's shapes (numpy only; no HIP, no oracle).

Real Pixel200K/1M/8M CSVs are not available offline, so bench.py, smoke() and the
parity tests draw batches with the statistics SURVEY.md §8(d) prescribes:
  * train batch = (items int64 [B,2,L+1], masked_index int64 [B,L]) exactly as SEQTrainDataset emits them
    (reference trainset.py:46-75): positives left-padded with 0, negatives one per target position,
    right-aligned, 0 elsewhere (so neg[:,0] is always 0), masked_index = 1 on the len-1 target positions;
  * ~70 % of sequences have the full length L+1, the rest uniform in [3, L];
  * positive ids ~ Zipf(s=1.0) over ranks 1..N-1 mapped through a fixed permutation (long-tail popularity as
    in dataset/statistics/item_rank.png); negatives uniform in [1, N-1] (trainset.py:40-44; the rejection of
    the sequence's own items is applied too).
"""
from __future__ import annotations

import numpy as np


class ZipfItems:
    def __init__(self, n_items: int, s: float = 1.0, seed: int = 2020):
        self.n_items = n_items
        ranks = np.arange(1, n_items, dtype=np.float64)
        w = ranks ** (-s)
        self.cdf = np.cumsum(w) / w.sum()
        rng = np.random.default_rng(seed)
        self.perm = rng.permutation(n_items - 1).astype(np.int64) + 1  # rank r (0-based) -> item id in [1, N-1]

    def sample(self, rng: np.random.Generator, size) -> np.ndarray:
        u = rng.random(size)
        r = np.searchsorted(self.cdf, u, side="right")
        r = np.minimum(r, self.n_items - 2)
        return self.perm[r]


def train_batch(n_items: int, B: int, L: int, rng: np.random.Generator, zipf: ZipfItems | None = None,
                full_frac: float = 0.7, uniform_ids: bool = False, min_len: int = 3):
    """Returns (items [B,2,L+1] int64, masked_index [B,L] int64)."""
    items = np.zeros((B, 2, L + 1), dtype=np.int64)
    mask = np.zeros((B, L), dtype=np.int64)
    lo = max(min(min_len, L), 2)              # shortest window: 2 items (one input, one target) when L is tiny
    full = rng.random(B) < full_frac          # (draw order kept: the fixtures / benches depend on this stream)
    short = rng.integers(lo, L + 1, size=B) if lo < L + 1 else np.full(B, L + 1)
    lens = np.where(full, L + 1, short)
    for b in range(B):
        n = int(lens[b])
        if uniform_ids or zipf is None:
            seq = rng.integers(1, n_items, size=n)
        else:
            seq = zipf.sample(rng, n)
        neg = rng.integers(1, n_items, size=n - 1)
        if n_items > 4 * n:  # rejection of the sequence's own items (trainset.py:40-44)
            own = set(seq.tolist())
            for j in range(n - 1):
                while int(neg[j]) in own:
                    neg[j] = rng.integers(1, n_items)
        items[b, 0, L + 1 - n:] = seq
        items[b, 1, L + 1 - (n - 1):] = neg
        mask[b, L - (n - 1):] = 1
    return items, mask


def eval_batch(n_items: int, B: int, L: int, rng: np.random.Generator, zipf: ZipfItems | None = None,
               hist_lo: int = 5, hist_hi: int = 60):
    """Returns (item_seq [B,L], history_u [H], history_i [H], positive_i [B]) shaped like seq_eval_collate's
    output (reference collate_fn.py:6-32): the FULL history is masked, the last L items form the input."""
    item_seq = np.zeros((B, L), dtype=np.int64)
    hu, hi, pos = [], [], np.zeros(B, dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(hist_lo, hist_hi + 1))
        hist = zipf.sample(rng, n) if zipf is not None else rng.integers(1, n_items, size=n)
        tail = hist[-L:]
        item_seq[b, L - len(tail):] = tail
        hu.append(np.full(n, b, dtype=np.int64))
        hi.append(hist.astype(np.int64))
        pos[b] = rng.integers(1, n_items)
    return item_seq, np.concatenate(hu), np.concatenate(hi), pos
