"""On-device negative sampling (pxr_sample_negatives_i64) against the reference's semantics (trainset.py:40-63):
positives copied, len-1 negatives left-padded, never one of the window's own items, uniform over [1, N-1], mask =
len-1 ones left-padded; stateless and reproducible; and the Trainer-side path (batcher -> prefetcher)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _windows(B, W, n_items, rng, small_pool=None):
    pos = np.zeros((B, W), dtype=np.int64)
    lens = rng.integers(2, W + 1, size=B)
    lens[0], lens[1] = W, 2
    for b in range(B):
        pool = small_pool if small_pool is not None else n_items - 1
        pos[b, W - lens[b]:] = rng.integers(1, pool + 1, size=lens[b])
    return pos, lens


@pytest.mark.parametrize("B,L,n_items", [(64, 50, 400_001), (7, 10, 500), (300, 20, 96_001)])
def test_semantics(B, L, n_items):
    from pixelrec_amd import ops

    W = L + 1
    rng = np.random.default_rng(B)
    pos, lens = _windows(B, W, n_items, rng)
    items, mask = ops.sample_negatives(torch.from_numpy(pos).cuda(), n_items, seed=2020, batch_counter=5)
    items, mask = items.cpu().numpy(), mask.cpu().numpy()
    assert np.array_equal(items[:, 0], pos)
    col = np.arange(W)[None, :]
    tgt = col >= (W - lens[:, None] + 1)
    neg = items[:, 1]
    assert np.all(neg[~tgt] == 0) and np.all(neg[tgt] >= 1) and np.all(neg[tgt] <= n_items - 1)
    assert np.array_equal(mask, tgt[:, 1:].astype(np.int64))
    assert mask.sum(1).tolist() == (lens - 1).tolist()
    for b in range(B):                                                  # never one of the window's own items
        assert not np.isin(neg[b][tgt[b]], pos[b][pos[b] > 0]).any()
    again, _ = ops.sample_negatives(torch.from_numpy(pos).cuda(), n_items, seed=2020, batch_counter=5)
    other, _ = ops.sample_negatives(torch.from_numpy(pos).cuda(), n_items, seed=2020, batch_counter=6)
    assert np.array_equal(again.cpu().numpy(), items) and not np.array_equal(other.cpu().numpy()[:, 1], neg)


def test_rejection_and_uniformity():
    from pixelrec_amd import ops

    n_items, L, B = 201, 40, 4096                                       # windows hold ~20 % of a small catalogue
    W = L + 1
    rng = np.random.default_rng(1)
    pos, lens = _windows(B, W, n_items, rng)
    counts = np.zeros(n_items, dtype=np.int64)
    for c in range(8):
        items, _ = ops.sample_negatives(torch.from_numpy(pos).cuda(), n_items, seed=7, batch_counter=c)
        neg = items[:, 1].cpu().numpy()
        for b in range(0, B, 97):
            own = pos[b][pos[b] > 0]
            assert not np.isin(neg[b][neg[b] > 0], own).any()
        counts += np.bincount(neg[neg > 0], minlength=n_items)
    assert counts[0] == 0
    freq = counts[1:] / counts[1:].sum()
    # own-item rejection makes the marginal only approximately uniform; every id must be within 25 % of 1/(N-1)
    assert np.all(np.abs(freq * (n_items - 1) - 1.0) < 0.25)


def test_trainer_pipeline_with_device_sampler(tmp_path):
    import os

    from pixelrec_amd.data import Data, SeqTrainBatcher
    from pixelrec_amd.data.utils import _TrainLoader
    from pixelrec_amd.trainer.trainer import _Prefetcher
    from pixelrec_amd.utils.enum_type import InputType

    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cfg = {"data_path": G, "dataset": "TinyInter", "MAX_ITEM_LIST_LENGTH": 10, "MODEL_INPUT_TYPE": InputType.SEQ,
           "train_batch_size": 8, "seed": 2020, "device_sampler": True}
    d = Data(cfg)
    d.build()
    loader = _TrainLoader(SeqTrainBatcher(cfg, d))
    seen = 0
    for items, mask in _Prefetcher(loader, torch.device("cuda")):
        assert items.is_cuda and items.shape[1:] == (2, 11) and mask.shape[1] == 10
        it, mk = items.cpu().numpy(), mask.cpu().numpy()
        lens = (it[:, 0] > 0).sum(1)
        assert (mk.sum(1) == lens - 1).all() and ((it[:, 1] > 0).sum(1) == lens - 1).all()
        seen += items.shape[0]
    assert seen == len(d.train_feat["item_seq"])
