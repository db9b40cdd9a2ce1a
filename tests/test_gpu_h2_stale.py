"""Round 6: h2 gradient planes under the PREVIOUS step's scales (csrc/h2.hip "stale scales", ops.H2Sites, seqcore).  The producers
(LayerNorm backward, attention backward) write the planes themselves under a device exponent that exists before they run:
  * the planes are, bit for bit, the h2 split of the fp32 gradient the exact path writes, under that exponent;
  * the partial maxima they leave are the exact path's; pxr_h2_sites_update turns them into the next step's exponents / bounds;
  * a value beyond the headroom is saturated (no inf) and raises PXR_STATUS_H2_STALE;
  * the training trajectory on stale scales holds the reference's golden bars (4 AdamW steps) and stays as close to the exact-scale
    trajectory as the goldens' tolerance."""
import ctypes
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _split_ref(x, e):
    """h2 planes of x * 2^e through the plain split entry (host exponent)."""
    from pixelrec_amd import lib as _l
    from pixelrec_amd import ops

    o = ops.Planes.alloc(x.shape[0], x.shape[1], x.device, fmt=1)
    P, I64, I32 = ctypes.c_void_p * 1, ctypes.c_int64 * 1, ctypes.c_int * 1
    _l.check(_l.load().pxr_split_h2_multi_f32(1, P(x.data_ptr()), I64(x.shape[0]), I64(x.shape[1]), I64(x.stride(0)), P(o.ptr().value),
                                              I64(o.ps), I64(o.pr), I32(e), _l.stream_ptr()), "pxr_split_h2_multi_f32")
    return o


@pytest.mark.parametrize("p_drop", [0.0, 0.2])
@pytest.mark.parametrize("rows,D", [(120, 128), (3200, 512), (77, 768)])
def test_ln_bwd_planes_under_a_given_exponent_are_the_split_of_the_exact_gradient(rows, D, p_drop):
    from pixelrec_amd import ops

    torch.manual_seed(rows + D)
    dev = "cuda"
    dy, xh = torch.randn(rows, D, device=dev) * 1e-4, torch.randn(rows, D, device=dev)
    rs, gamma = torch.rand(rows, device=dev) + 0.5, torch.rand(D, device=dev) + 0.5
    dg, db = torch.empty(D, device=dev), torch.empty(D, device=dev)
    n_parts = ops.ln_bwd_stat_parts(rows)
    st0 = torch.empty(max(n_parts, 64), device=dev)
    dz0, dx0 = ops.ln_bwd(0, dy.view(1, rows, D), xh.view(1, rows, D), rs, gamma, dg, db, p_drop, 7, 3, need_dx=p_drop > 0, stat=st0)
    want = (dx0 if dx0 is not None else dz0).view(rows, D)
    sites = ops.H2Sites(3, dev)
    e = 14 - ops.H2_STALE_HEADROOM - math.frexp(float(want.abs().max()))[1]
    sites.exps[1] = e
    st1 = torch.empty_like(st0)
    dg1, db1 = torch.empty(D, device=dev), torch.empty(D, device=dev)
    dz1, gp, _ = ops.ln_bwd_h2s(dy.view(1, rows, D), xh.view(1, rows, D), rs, gamma, dg1, db1, sites, 1, st1, p_drop, 7, 3)
    ref = _split_ref(want.contiguous(), e)
    assert torch.equal(dz1.view(rows, D), dz0.view(rows, D)) and torch.equal(dg1, dg) and torch.equal(db1, db)
    assert torch.equal(st1[:n_parts], st0[:n_parts])
    assert torch.equal(gp.buf.view(torch.int16), ref.buf.view(torch.int16))
    assert int(gp.exp_dev.item()) == e
    ops.raise_on_bad_indices(dev)


def test_attention_backward_planes_under_a_given_exponent():
    from pixelrec_amd import ops

    torch.manual_seed(5)
    dev = "cuda"
    for (B, H, L, d) in ((6, 4, 50, 128), (5, 2, 10, 32), (3, 4, 20, 64)):
        D = H * d
        qkv = torch.randn(B, L, 3 * D, device=dev)
        km = (torch.rand(B, L, device=dev) > 0.2).long()
        ctx, probs = ops.attn_fwd(qkv, km, L, B, H, L, d, 0.1, 9, 1)
        dctx = torch.randn(B, L, D, device=dev) * 1e-3
        st0 = torch.zeros(ops.ATTN_STAT_SLOTS, device=dev)
        dqkv = ops.attn_bwd(dctx, qkv, probs, B, H, L, d, 0.1, 9, 1, stat=st0)
        sites = ops.H2Sites(2, dev)
        e = 14 - ops.H2_STALE_HEADROOM - math.frexp(float(dqkv.abs().max()))[1]
        sites.exps[0] = e
        st1 = torch.zeros(ops.ATTN_STAT_SLOTS, device=dev)
        gp = ops.attn_bwd_h2s(dctx, qkv, probs, B, H, L, d, sites, 0, st1, 0.1, 9, 1)
        ref = _split_ref(dqkv.view(B * L, 3 * D), e)
        assert torch.equal(st1, st0)
        assert torch.equal(gp.buf.view(torch.int16), ref.buf.view(torch.int16)), (B, H, L, d)
    ops.raise_on_bad_indices(dev)


def test_a_gradient_beyond_the_headroom_is_saturated_and_flagged():
    from pixelrec_amd import ops

    dev = "cuda"
    rows, D = 64, 128
    torch.manual_seed(1)
    dy, xh = torch.randn(rows, D, device=dev), torch.randn(rows, D, device=dev)
    rs, gamma = torch.ones(rows, device=dev), torch.ones(D, device=dev)
    sites = ops.H2Sites(1, dev)
    sites.exps[0] = 20                       # |dz| ~ 1 -> 2^20: far beyond 65504
    st = torch.empty(ops.ln_bwd_stat_parts(rows), device=dev)
    ops.raise_on_bad_indices(dev)
    dz, gp, _ = ops.ln_bwd_h2s(dy.view(1, rows, D), xh.view(1, rows, D), rs, gamma, torch.empty(D, device=dev), torch.empty(D, device=dev),
                               sites, 0, st)
    assert torch.isfinite(gp.buf.float()).all()                    # saturated, never inf
    assert float(gp.buf.float().abs().max()) == 65504.0
    with pytest.raises(RuntimeError, match="PXR_SEQ_H2_STALE"):
        ops.raise_on_bad_indices(dev)
    ops.raise_on_bad_indices(dev)                                   # the word was cleared


def test_sites_update_derives_next_steps_exponents_and_bounds():
    from pixelrec_amd import ops

    dev = "cuda"
    H = ops.H2_STALE_HEADROOM
    sites = ops.H2Sites(3, dev)
    sites.exps.fill_(99); sites.bexp.fill_(77); sites.stats.fill_(-1.0)
    parts = [torch.rand(200, device=dev) * 3e-5, torch.zeros(64, device=dev), torch.rand(1024, device=dev) * 0.7]
    W = ops.split_h2_auto([torch.randn(128, 64, device=dev) * 0.05], col_stats=True)[0]
    sites.update(parts, [200, 64, 1024], [3200, 3200, 3200], [W, None, None], 1.7)
    torch.cuda.synchronize()
    for s, (p, n) in enumerate(zip(parts, [200, 64, 1024])):
        mx = float(p[:n].max())
        if mx == 0.0:                                               # a site without gradient keeps its entries
            assert int(sites.exps[s]) == 99 and float(sites.stats[s, 0]) == -1.0
            continue
        assert int(sites.exps[s]) == 14 - H - math.frexp(mx)[1]
        assert 2.0 ** (13 - H) <= mx * 2.0 ** int(sites.exps[s]) < 2.0 ** (14 - H)
        assert float(sites.stats[s, 0]) == np.float32(mx) * 2.0 ** H and float(sites.stats[s, 1]) == np.float32(3200.0) * (np.float32(mx) * np.float32(2.0 ** H))
    bound = np.float32(np.float32(float(parts[0][:200].max())) * np.float32(2.0 ** H)) * np.float32(float(W.stats[1])) * np.float32(1.7)
    assert int(sites.bexp[0]) == 15 - math.frexp(float(bound))[1]
    assert int(sites.bexp[1]) == 77 and int(sites.bexp[2]) == 77    # no weight behind these sites
    # the scale follows a DECAYING maximum of the steps' maxima: a smaller step maximum only lowers it by the decay factor ...
    first = [float(p[:n].max()) for p, n in zip(parts, [200, 64, 1024])]
    small = [p * 0.01 for p in parts]
    sites.update(small, [200, 64, 1024], [3200, 3200, 3200], [W, None, None], 1.7)
    assert float(sites.run_max[0]) == np.float32(first[0]) * np.float32(ops.H2_STALE_DECAY)
    assert int(sites.exps[0]) == 14 - H - math.frexp(float(sites.run_max[0]))[1]
    # ... and a larger one raises it at once
    big = [p * 100.0 for p in parts]
    sites.update(big, [200, 64, 1024], [3200, 3200, 3200], [W, None, None], 1.7)
    assert float(sites.run_max[2]) == float(big[2].max()) and int(sites.exps[2]) == 14 - H - math.frexp(float(big[2].max()))[1]


@pytest.fixture
def stale_env():
    prev = {k: os.environ.get(k) for k in ("PXR_SEQ_H2_STALE", "PXR_SEQ_H2", "PXR_PLANES")}
    yield
    for k, v in prev.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("case", ["tiny", "ns"])
def test_four_adamw_steps_on_stale_scales_hold_the_golden_bars(case, stale_env):
    """tests/test_gpu_sasrec.py::test_adamw_four_steps with the sequence block on h2 operands and the gradient planes of steps 2-4
    written under the previous step's scales: the same bars against the reference's goldens; and the exact-scale run of the same
    steps ends within those bars of it."""
    from tests import golden_util as T
    from tests.test_gpu_sasrec import _model
    from pixelrec_amd import ops
    from pixelrec_amd.optim import PxrAdamW

    os.environ["PXR_PLANES"], os.environ["PXR_SEQ_H2"] = "1", "1"
    prev_mode = ops.set_gemm_mode("bf16x3")
    try:
        finals = {}
        for stale in ("1", "0"):
            os.environ["PXR_SEQ_H2_STALE"] = stale
            meta, z = T.load_case(case)
            m, p = _model(meta)
            m.train()
            opt = PxrAdamW(m, lr=1e-4, weight_decay=0.1)
            rows = torch.from_numpy(z["adamw.watch_rows"])
            for s in range(4):
                loss = m((torch.from_numpy(z["adamw.items"][s]).cuda(), torch.from_numpy(z["adamw.masks"][s]).cuda()))
                loss.backward()
                opt.step()
                assert abs(float(loss) - float(z[f"adamw.loss{s}"])) <= 3e-5 * max(1.0, abs(float(loss)))
                sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
                T.compare(z, f"adamw.step{s}.table_rows", sd["item_embedding.weight"][rows], 1e-5)
                for k in ("position_embedding.weight", "LayerNorm.weight", "trm_encoder.layer.0.multi_head_attention.query.weight",
                          "trm_encoder.layer.1.feed_forward.dense_2.weight", "trm_encoder.layer.1.feed_forward.dense_1.bias"):
                    T.compare(z, f"adamw.step{s}." + k, sd[k], 1e-5)
            if stale == "1":
                assert m._h2_sites is not None and m._h2_sites.seeded_for is not None     # the stale path really ran
            else:
                assert m._h2_sites is None
            finals[stale] = sd
            ops.raise_on_bad_indices("cuda")
        for k in finals["1"]:
            if k.endswith("key.bias"):      # mathematically zero gradient: its AdamW update is the sign of rounding noise
                continue
            assert (finals["1"][k] - finals["0"][k]).abs().max().item() <= 2e-6, k
    finally:
        ops.set_gemm_mode(prev_mode)


def test_captured_steps_on_stale_scales_track_the_eager_exact_steps(stale_env):
    """GraphedTrainStep turns the stale scales on: 12 replayed steps stay within the parity budget of 12 eager steps on exact
    per-step scales (same batches, dropout on -- the masks are a stateless hash), and raise nothing."""
    from pixelrec_amd import ops, synth
    from pixelrec_amd.graph import GraphedTrainStep
    from pixelrec_amd.model import SASRec
    from pixelrec_amd.optim import PxrAdamW
    from pixelrec_amd.parallel import DataParallel

    cfg = {"n_layers": 2, "n_heads": 2, "embedding_size": 128, "inner_size": 2, "hidden_dropout_prob": 0.1, "attn_dropout_prob": 0.1,
           "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02, "MAX_ITEM_LIST_LENGTH": 20, "seed": 2020}
    N, B = 3000, 32

    class DL:
        item_num = N

    rng = np.random.default_rng(4)
    z = synth.ZipfItems(N, seed=2)
    batches = [tuple(torch.from_numpy(a).cuda() for a in synth.train_batch(N, B, 20, rng, z)) for _ in range(12)]
    out = {}
    for mode in ("graph", "eager"):
        os.environ.pop("PXR_SEQ_H2_STALE", None)
        torch.manual_seed(3)
        m = SASRec(cfg, DL()).cuda().train()
        opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1)
        if mode == "graph":
            g = GraphedTrainStep(DataParallel(m), opt, *batches[0], warmup=0)
            losses = [float(g(*b)) for b in batches]
            assert m.h2_stale_scales and m._h2_sites is not None and m._h2_sites.seeded_for is not None
        else:
            losses = []
            for b in batches:
                loss = m(b)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            assert m._h2_sites is None
        opt.flush()
        ops.raise_on_bad_indices("cuda")
        out[mode] = (losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    for a, b in zip(out["graph"][0], out["eager"][0]):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b))
    for k, v in out["eager"][1].items():
        if k.endswith("key.bias"):
            continue
        assert (out["graph"][1][k] - v).abs().max().item() <= 2e-5, k       # 12 steps at lr 1e-3: 2 % of one step's movement


def test_trainer_epoch_loss_survives_a_mid_epoch_recapture(tmp_path, monkeypatch):
    """A stale-scale overflow drops the captured step in the middle of an epoch and a new one is captured (twice: exact scales, then the
    stale ones again): the epoch's loss is still the sum over ALL its steps -- equal, to rounding, to an eager run of the same epoch
    (the reference's per-epoch total, trainer.py:105-128)."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_eval import _harness
    from pixelrec_amd import ops

    def epoch(force_fallback_at, use_graph):
        torch.manual_seed(3)                      # the same initial weights in both runs (the batcher is seeded by the config)
        config, dataload, (train, valid, test), model, DataParallel, Trainer = _harness(tmp_path, 32, 6, 2, 2, 2, epochs=1, p_drop=0.0)
        config["use_hip_graph"] = use_graph
        trainer = Trainer(config, DataParallel(model.to(config["device"])))
        trainer.H2_STALE_BACKOFF = (2, 64)
        if force_fallback_at is not None:
            orig = trainer._h2_stale_resume

            def hook():
                if trainer._steps_done == force_fallback_at:
                    trainer._h2_stale_fallback(ops.H2StaleOverflow("forced by the test"))
                orig()
            trainer._h2_stale_resume = hook
        return trainer._train_epoch(train, 0), len(train), trainer

    base, n, _ = epoch(None, use_graph=False)
    got, n2, tr = epoch(2, use_graph=True)
    assert n == n2 and n >= 6
    assert tr.model.module.h2_stale_scales is True and getattr(tr, "_stale_resume_at", None) is None      # fell back at step 2, resumed at step 4
    assert abs(got - base) <= 2e-4 * abs(base), (got, base)
