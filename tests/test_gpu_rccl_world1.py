"""The RCCL code path on the one GPU the development box has: a 1-rank "nccl" (= RCCL) process group with the
collectives FORCED on.  all-reduce(SUM) / all-gather over one rank are identities and the merge of an already unique
row list reproduces it, so k training steps through the RCCL branch (all_gather_into_tensor, async handles on RCCL's
stream, merge kernel, lazy AdamW on the merged rows) must be bit-identical to k steps without any collective.
Also replays the same step from a hipGraph that CONTAINS the collectives."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

CFG = {"n_layers": 2, "n_heads": 2, "embedding_size": 64, "inner_size": 2, "hidden_dropout_prob": 0.1,
       "attn_dropout_prob": 0.1, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
       "MAX_ITEM_LIST_LENGTH": 10, "seed": 2020}
N, B, STEPS = 500, 8, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(force, graphed=False, defer=False, clip=None):
    from pixelrec_amd import synth
    from pixelrec_amd.model import SASRec
    from pixelrec_amd.optim import PxrAdamW, clip_grad_norm_
    from pixelrec_amd.parallel import DataParallel

    class DL:
        item_num = N

    torch.manual_seed(7)
    model = SASRec(CFG, DL()).cuda().train()
    dp = DataParallel(model, force_collectives=force)
    assert dp.grad_sync.active == force
    opt = PxrAdamW(model, lr=1e-3, weight_decay=0.1)
    rng = np.random.default_rng(3)
    z = synth.ZipfItems(N, seed=1)
    batches = [tuple(torch.from_numpy(a).cuda() for a in synth.train_batch(N, B, 10, rng, z)) for _ in range(STEPS)]
    losses = []
    if graphed:
        from pixelrec_amd.graph import GraphedTrainStep

        gstep = GraphedTrainStep(dp, opt, *batches[0], warmup=0)
        for b in batches:
            losses.append(float(gstep(*b)))
    else:
        for b in batches:
            opt.zero_grad()
            loss = dp(b)
            loss.backward()
            dp.sync_gradients(defer_flat=defer)
            assert bool(model._flat_grad_waits) == (defer and force)
            if clip:
                clip_grad_norm_(dp, max_norm=clip)
            opt.step()
            assert not model._flat_grad_waits
            losses.append(float(loss.detach()))
    opt.flush()
    torch.cuda.synchronize()
    return losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


@pytest.fixture(scope="module")
def rccl_world1():
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


def test_forced_rccl_collectives_are_identity(rccl_world1):
    l0, sd0 = _run(force=False)
    l1, sd1 = _run(force=True)
    assert l0 == l1
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


@pytest.mark.parametrize("clip", [None, 0.05])
def test_deferred_flat_allreduce_wait(rccl_world1, clip):
    """sync_gradients(defer_flat=True): the flat all-reduce is waited for by its consumer (PxrAdamW.step after the
    table-row update, or clip_grad_norm_) instead of by sync -- same result, bit for bit."""
    l0, sd0 = _run(force=False, clip=clip)
    l1, sd1 = _run(force=True, defer=True, clip=clip)
    assert l0 == l1
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


def test_rccl_collectives_replay_from_hipgraph(rccl_world1, monkeypatch):
    # bit identity of capture + replay with the collectives inside: both runs on per-step exact h2 scales (a captured step would
    # otherwise write its gradient planes under the recent steps' scales: tests/test_gpu_h2_stale.py)
    monkeypatch.setenv("PXR_SEQ_H2_STALE", "0")
    l0, sd0 = _run(force=False)
    l1, sd1 = _run(force=True, graphed=True)
    assert l0 == l1
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


def test_row_sharded_table_through_rccl(rccl_world1):
    """The sharded step's collectives (id all-gather, reduce_scatter_tensor of the served rows, shard all-gather for
    evaluation) on the real RCCL backend with one rank: identical to the plain model."""
    from pixelrec_amd import synth
    from pixelrec_amd.model import SASRec, ShardedDataParallel, ShardedSASRec
    from pixelrec_amd.optim import PxrAdamW
    from pixelrec_amd.parallel import DataParallel

    class DL:
        item_num = N

    rng = np.random.default_rng(3)
    z = synth.ZipfItems(N, seed=1)
    batches = [tuple(torch.from_numpy(a).cuda() for a in synth.train_batch(N, B, 10, rng, z)) for _ in range(STEPS)]

    def run(sharded):
        torch.manual_seed(7)
        if sharded:
            m = ShardedSASRec(CFG, DL()).cuda().train()
            dp = ShardedDataParallel(m, force_collectives=True)
        else:
            m = SASRec(CFG, DL()).cuda().train()
            dp = DataParallel(m)
        opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1)
        losses = []
        for b in batches:
            opt.zero_grad()
            loss = dp(b)
            loss.backward()
            dp.sync_gradients()
            opt.step()
            losses.append(float(loss.detach()))
        return losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}

    l0, sd0 = run(False)
    l1, sd1 = run(True)
    assert l0 == l1
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k
