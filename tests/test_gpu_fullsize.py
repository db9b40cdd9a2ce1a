"""BASELINE.json's full sizes on the GPU: configs[1] (N=400 001, D=512, L=50, B=64) and configs[0]'s shape
(D=128, L=20, Pixel200K-sized catalogue) -- direct oracle comparison of one training step where the CPU oracle still
finishes in seconds, plus size-independent properties (run-to-run bit reproducibility, lazy == dense schedule,
eval-mode loss == training loss with dropout off, ragged/empty edge cases)."""
import numpy as np
import pytest
import torch

from oracle import sasrec_oracle as O

pytestmark = pytest.mark.gpu


def _cfg(D, L, H, p=0.0):
    return {"n_layers": 2, "n_heads": H, "embedding_size": D, "inner_size": 2, "hidden_dropout_prob": p,
            "attn_dropout_prob": p, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
            "MAX_ITEM_LIST_LENGTH": L, "seed": 2020}


def _batch(N, B, L, seed):
    from pixelrec_amd import synth

    rng = np.random.default_rng(seed)
    it, mk = synth.train_batch(N, B, L, rng, synth.ZipfItems(N, seed=seed))
    return torch.from_numpy(it), torch.from_numpy(mk)


def _model(N, D, L, H, params=None, p=0.0):
    from pixelrec_amd.model import SASRec

    class DL:
        item_num = N

    m = SASRec(_cfg(D, L, H, p), DL())
    if params is not None:
        m.load_state_dict(params, strict=True)
    return m.cuda().train()


# B=64 takes the fused sort (n = 3*B*L <= 65536 occurrences: one launch per pass), B=1200 the multi-launch radix sort
@pytest.mark.parametrize("N,D,L,H,B", [(400_001, 512, 50, 4, 64), (96_001, 128, 20, 4, 64), (96_001, 128, 20, 4, 1200)])
def test_one_step_at_full_size_matches_oracle(N, D, L, H, B):
    from pixelrec_amd.optim import PxrAdamW

    params = O.synth_params(N, D, L, 2, 2, seed=21, perturb=True)
    m = _model(N, D, L, H, params)
    opt = PxrAdamW(m, lr=1e-4, weight_decay=0.1)
    items, mask = _batch(N, B, L, 5)
    loss = m((items.cuda(), mask.cuda()))
    loss.backward()
    tr = O.OracleTrainer(params, {"n_layers": 2, "n_heads": H, "layer_norm_eps": 1e-12}, lr=1e-4, weight_decay=0.1)
    ref_loss, g = O.loss_and_grads(tr.p, items, mask, tr.cfg)
    assert abs(float(loss.detach()) - float(ref_loss)) <= 3e-5 * max(1.0, abs(float(ref_loss)))
    sp = m.sparse_table_grad
    n = sp.count()
    idx = sp.idx[:n].cpu()
    ref_rows = g["item_embedding.weight"][idx]
    assert (sp.rows[:n].cpu() - ref_rows).abs().max().item() <= 3e-6 + 2e-4 * ref_rows.abs().max().item()
    touched = torch.zeros(N, dtype=torch.bool); touched[idx] = True
    assert float(g["item_embedding.weight"][~touched].abs().max()) == 0.0          # nothing missed
    for k in ("position_embedding.weight", "trm_encoder.layer.0.multi_head_attention.query.weight",
              "trm_encoder.layer.1.feed_forward.dense_2.weight", "LayerNorm.weight"):
        got = dict(m.named_parameters())[k].grad.cpu()
        assert (got - g[k]).abs().max().item() <= 5e-6 + 3e-4 * g[k].abs().max().item(), k
    opt.step()
    tr.step(items, mask)
    sd = m.state_dict()                                                            # flushes the lazy table
    rows = torch.cat([idx[:200], torch.tensor([0, 1, 2, N - 1])])
    assert (sd["item_embedding.weight"][rows.cuda()].cpu() - tr.p["item_embedding.weight"][rows]).abs().max().item() < 1e-5
    diff = (sd["item_embedding.weight"].cpu() - tr.p["item_embedding.weight"]).abs()
    assert diff[~touched].max().item() < 1e-7                      # untouched rows: weight decay only (lazy replay exact)
    # touched rows: Adam moves an entry whose gradient is rounding noise by +-lr whatever its sign, so a few entries
    # may differ by 2*lr; everything else agrees to 1e-5
    assert diff.max().item() <= 2.1e-4 and (diff[touched] > 1e-5).float().mean().item() < 1e-3
    for k in ("position_embedding.weight", "trm_encoder.layer.1.feed_forward.dense_2.weight"):
        assert (sd[k].cpu() - tr.p[k]).abs().max().item() < 1e-5, k


def test_bit_reproducible_and_schedule_independent(monkeypatch):
    """Same inputs -> same bits, run to run (no float atomics anywhere) and across lazy / dense table schedules and
    grouped / per-layer weight-gradient launches."""
    monkeypatch.setenv("PXR_LAZY_REPLAY", "exact")      # lazy == dense bit for bit is a statement about the exact replay
    from pixelrec_amd.optim import PxrAdamW

    N, D, L, H, B = 50_001, 256, 50, 4, 32
    params = O.synth_params(N, D, L, 2, 2, seed=8)
    batches = [_batch(N, B, L, s) for s in range(6)]

    def run(schedule, grouped, p):
        m = _model(N, D, L, H, params, p)
        m.group_weight_grads = grouped
        opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1, table_update=schedule)
        for it, mk in batches:
            loss = m((it.cuda(), mk.cuda())); loss.backward(); opt.step()
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    a = run("lazy", True, 0.1)
    b = run("lazy", True, 0.1)
    c = run("dense", True, 0.1)
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert torch.equal(a[k], c[k]), k
    # grouped vs per-layer weight-gradient launches use different GEMM kernels (no split-K vs split-K): same math,
    # rounding-level differences in ONE backward (trained weights would amplify them through Adam's sign-like step)
    grads = []
    for grouped in (True, False):
        m = _model(N, D, L, H, params, 0.0)
        m.group_weight_grads = grouped
        it, mk = batches[0]
        m((it.cuda(), mk.cuda())).backward()
        grads.append({k: v.grad.detach().clone() for k, v in m.named_parameters() if v.grad is not None})
    for k in grads[0]:
        ref = grads[0][k].abs().max().item()
        assert (grads[0][k] - grads[1][k]).abs().max().item() <= 1e-7 + 2e-5 * ref, k


def test_edge_cases_all_padding_and_min_length():
    """A batch holding an all-padding row, a 2-item row (one target) and L = 64 (the widest sequence one wave-wide
    softmax handles)."""
    N, D, L, H = 300, 64, 64, 2
    params = O.synth_params(N, D, L, 2, 2, seed=2)
    m = _model(N, D, L, H, params)
    items = torch.zeros(3, 2, L + 1, dtype=torch.int64)
    mask = torch.zeros(3, L, dtype=torch.int64)
    items[1, 0, -2:] = torch.tensor([4, 9]); items[1, 1, -1] = 17; mask[1, -1] = 1
    g = torch.Generator().manual_seed(0)
    items[2, 0] = torch.randint(1, N, (L + 1,), generator=g); items[2, 1, 1:] = torch.randint(1, N, (L,), generator=g)
    mask[2] = 1
    loss = m((items.cuda(), mask.cuda()))
    loss.backward()
    ref, gr = O.loss_and_grads(params, items, mask, {"n_layers": 2, "n_heads": H, "layer_norm_eps": 1e-12})
    assert torch.isfinite(loss) and abs(float(loss.detach()) - float(ref)) < 3e-5 * max(1.0, abs(float(ref)))
    dense = m.sparse_table_grad.to_dense(N).cpu()
    assert (dense - gr["item_embedding.weight"]).abs().max().item() < 5e-6 + 2e-4 * gr["item_embedding.weight"].abs().max().item()
    with pytest.raises(Exception):      # head size must be a multiple of 4 (16-byte rows per head): a loud error
        _model(N, 36, 10, 6)((torch.zeros(1, 2, 11, dtype=torch.int64).cuda(), torch.zeros(1, 10, dtype=torch.int64).cuda()))


def test_wide_embedding_4096_and_chunked_attention():
    """emb 4096 (BASELINE configs[3]'s width; head size 1024 -> the d-chunked attention kernels, VEC=16 LayerNorm /
    lazy-AdamW paths) on a small catalogue: one step against the oracle."""
    from pixelrec_amd.optim import PxrAdamW

    N, D, L, H, B = 600, 4096, 12, 4, 3
    params = O.synth_params(N, D, L, 2, 2, seed=31)
    m = _model(N, D, L, H, params)
    opt = PxrAdamW(m, lr=1e-4, weight_decay=0.1)
    items, mask = _batch(N, B, L, 9)
    loss = m((items.cuda(), mask.cuda()))
    loss.backward()
    ocfg = {"n_layers": 2, "n_heads": H, "layer_norm_eps": 1e-12}
    ref, g = O.loss_and_grads(params, items, mask, ocfg)
    assert abs(float(loss.detach()) - float(ref)) <= 5e-5 * max(1.0, abs(float(ref)))
    dense = m.sparse_table_grad.to_dense(N).cpu()
    assert (dense - g["item_embedding.weight"]).abs().max().item() <= 1e-5 + 3e-4 * g["item_embedding.weight"].abs().max().item()
    for k in ("trm_encoder.layer.0.multi_head_attention.key.weight", "trm_encoder.layer.1.feed_forward.LayerNorm.weight",
              "LayerNorm.bias", "position_embedding.weight"):
        got = dict(m.named_parameters())[k].grad.cpu()
        assert (got - g[k]).abs().max().item() <= 1e-5 + 5e-4 * g[k].abs().max().item(), k
    opt.step()
    tr = O.OracleTrainer(params, ocfg, lr=1e-4, weight_decay=0.1)
    tr.step(items, mask)
    sd = m.state_dict()
    diff = (sd["item_embedding.weight"].cpu() - tr.p["item_embedding.weight"]).abs()
    # AdamW's first step moves an element by lr * g/(|g|+eps): where |g| is at rounding-noise level the SIGN is noise
    # too (in the reference as well), so such elements may differ by up to 2*lr; everywhere else the match is tight
    solid = g["item_embedding.weight"].abs() > 1e-6
    assert diff[solid].max().item() < 2e-5
    assert diff.max().item() <= 2.1e-4 and (diff > 2e-5).float().mean().item() < 1e-3


@pytest.mark.parametrize("N,D,L,H,B", [(97, 96, 7, 4, 5), (300, 80, 13, 4, 3), (150, 64, 64, 1, 2), (5000, 256, 64, 8, 9),
                                        (60, 32, 2, 2, 4), (1000, 1024, 33, 16, 2), (211, 192, 50, 3, 7),
                                        (300, 64, 100, 2, 3), (500, 512, 128, 4, 2), (200, 96, 65, 3, 2), (400, 256, 127, 1, 2),
                                        # beyond the fused attention kernels: batched-GEMM attention (any length; and
                                        # 65..128 positions with a head size that is not a multiple of 8)
                                        (300, 64, 129, 2, 2), (400, 128, 200, 4, 3), (200, 40, 70, 2, 2), (250, 64, 300, 1, 2)])
def test_unusual_shapes_match_oracle(N, D, L, H, B):
    """Head sizes that are not powers of two (24, 20), one head, the longest single-wave sequence (64), the two-keys-per-lane kernels
    (65..128 positions), two positions, 16 heads: loss, every gradient, predict scores and one AdamW step against the oracle."""
    from pixelrec_amd.optim import PxrAdamW

    params = O.synth_params(N, D, L, 2, 2, seed=N + D, perturb=True)
    m = _model(N, D, L, H, params)
    opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1)
    items, mask = _batch(N, B, L, 3)
    loss = m((items.cuda(), mask.cuda()))
    loss.backward()
    cfg = {"n_layers": 2, "n_heads": H, "layer_norm_eps": 1e-12}
    tr = O.OracleTrainer(params, cfg, lr=1e-3, weight_decay=0.1)
    ref_loss, g = O.loss_and_grads(tr.p, items, mask, tr.cfg)
    assert abs(float(loss.detach()) - float(ref_loss)) <= 3e-5 * max(1.0, abs(float(ref_loss)))
    dense = m.sparse_table_grad.to_dense(N).cpu()
    assert (dense - g["item_embedding.weight"]).abs().max().item() <= 3e-6 + 3e-4 * g["item_embedding.weight"].abs().max().item()
    for k, v in m.named_parameters():
        if k != "item_embedding.weight":
            assert (v.grad.cpu() - g[k]).abs().max().item() <= 5e-6 + 3e-4 * g[k].abs().max().item(), k
    m.eval()                                # logits +-1e-4 on identical parameters (before the optimizer moves them)
    seq = items[:, 0, 1:].contiguous().cuda()
    with torch.no_grad():
        scores = m.predict(seq, m.compute_item_all()).cpu()
    ref_scores = O.predict(tr.p, seq.cpu(), tr.p["item_embedding.weight"], tr.cfg)
    assert (scores - ref_scores).abs().max().item() <= 1e-4
    m.train()
    opt.step()
    tr.step(items, mask)
    sd = m.state_dict()
    for k in ("item_embedding.weight", "position_embedding.weight", "trm_encoder.layer.1.feed_forward.dense_1.weight"):
        d = (sd[k].cpu() - tr.p[k]).abs()
        solid = g[k].abs() > 1e-6          # Adam's sign(g) is rounding noise where |g| ~ 0 (moves such entries by +-lr)
        assert d[solid].max().item() < 2e-5 if solid.any() else True, k
        assert d.max().item() <= 2.1e-3, k


@pytest.mark.parametrize("N,D,L,H,B", [(300, 64, 100, 2, 3), (500, 256, 128, 4, 2), (300, 64, 150, 2, 2)])
def test_long_sequence_dropout_parity(N, D, L, H, B):
    """65..128 positions with dropout ON: the long attention kernels regenerate the same counter-hash masks in forward
    and backward; the masks, restated in numpy, are injected into the oracle."""
    from oracle import dropout_rng as R

    params = O.synth_params(N, D, L, 2, 2, seed=L, perturb=True)
    m = _model(N, D, L, H, params, p=0.1)
    m.train()
    items, mask = _batch(N, B, L, 9)
    seed = (m._drop_seed * 1000003 + m._step_counter) & 0xFFFFFFFFFFFFFFFF
    loss = m((items.cuda(), mask.cuda()))
    loss.backward()
    drop = R.sasrec_masks(seed, B, L, D, H, 2, 0.1, 0.1)
    cfg = {"n_layers": 2, "n_heads": H, "layer_norm_eps": 1e-12, "hidden_dropout_prob": 0.1, "attn_dropout_prob": 0.1}
    ref_loss, g = O.loss_and_grads(params, items, mask, cfg, drop)
    assert abs(float(loss.detach()) - float(ref_loss)) <= 3e-5 * max(1.0, abs(float(ref_loss)))
    for k, v in m.named_parameters():
        got = m.sparse_table_grad.to_dense(N).cpu() if k == "item_embedding.weight" else v.grad.detach().cpu()
        assert (got - g[k]).abs().max().item() <= 5e-6 + 3e-4 * g[k].abs().max().item(), k
