"""The lazy (exact catch-up) table optimizer == the dense sweep: same dense-AdamW semantics, no O(N*D) pass."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _replay_mode(request, monkeypatch):
    """The bit-identity statements of this module (lazy == dense sweep) are made in the EXACT replay mode; the default fast
    replay (carried sqrt / Newton reciprocal) has its own tolerance tests below, marked fast_replay."""
    if request.node.get_closest_marker("fast_replay") is None:
        monkeypatch.setenv("PXR_LAZY_REPLAY", "exact")
    else:
        monkeypatch.delenv("PXR_LAZY_REPLAY", raising=False)


def _setup(n_items, D=64, L=10, B=4, seed=3):
    from oracle import sasrec_oracle as O
    from pixelrec_amd import synth
    from pixelrec_amd.model import SASRec

    cfg = {"n_layers": 1, "n_heads": 2, "embedding_size": D, "inner_size": 2, "hidden_dropout_prob": 0.0,
           "attn_dropout_prob": 0.0, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
           "MAX_ITEM_LIST_LENGTH": L, "seed": 2020}

    class DL:
        item_num = n_items

    params = O.synth_params(n_items, D, L, 1, 2, seed=seed)

    def make():
        m = SASRec(cfg, DL())
        m.load_state_dict(params, strict=True)
        return m.cuda().train()

    rng = np.random.default_rng(seed)
    zipf = synth.ZipfItems(n_items, seed=seed)
    return make, rng, zipf, synth


def _run(make, batches, mode, flush_every=0):
    from pixelrec_amd.optim import PxrAdamW

    m = make()
    opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1, table_update=mode)
    for i, (it, mk) in enumerate(batches):
        loss = m((it, mk))
        loss.backward()
        opt.step()
        if flush_every and (i + 1) % flush_every == 0:
            opt.flush()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}          # state_dict() flushes
    return sd, opt


def test_lazy_equals_dense_bitwise_short_gaps():
    make, rng, zipf, synth = _setup(n_items=1500)
    batches = [tuple(torch.from_numpy(x).cuda() for x in synth.train_batch(1500, 4, 10, rng, zipf)) for _ in range(60)]
    dense, _ = _run(make, batches, "dense")
    lazy, opt = _run(make, batches, "lazy")
    for k in dense:
        assert torch.equal(dense[k], lazy[k]), k                              # incl. rows never touched in 60 steps
    assert torch.equal(opt._tm, _run(make, batches, "dense")[1]._tm)
    assert int((opt._last != opt.step_count).sum()) == 0                     # flushed
    lazy2, _ = _run(make, batches, "lazy", flush_every=7)                    # intermediate flushes change nothing
    for k in dense:
        assert torch.equal(dense[k], lazy2[k]), k


def test_lazy_long_gaps_closed_form_tail():
    """Optimizer kernels alone (no model feedback, which would chaotically amplify rounding through Adam's
    sign-like update on noise-level gradients): identical sparse gradient sequences into the dense sweep and the
    lazy replay, with gaps far beyond the 256 exactly-replayed steps.  There only weight decay acts and the closed
    form p *= exp(sum log decay) replaces the sequential product: deviation is accumulated-rounding level."""
    from pixelrec_amd import ops

    N, D, T, lr, wd, b1, b2, eps = 4000, 64, 900, 1e-3, 0.1, 0.9, 0.999, 1e-8
    g = torch.Generator().manual_seed(0)
    p0 = (torch.randn(N, D, generator=g) * 0.02).cuda()
    pd, md, vd = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pl, ml, vl = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    slot = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    last = torch.zeros(N, dtype=torch.int32, device="cuda")
    hyper = torch.zeros(T + 8, 4, device="cuda")
    cumlog = torch.zeros(T + 8, dtype=torch.float64, device="cuda")
    cap = 16
    max_gap = 0
    seen = {}
    for t in range(1, T + 1):
        n = int(torch.randint(1, cap + 1, (1,), generator=g))
        # rows 0..39 are hot (touched often), the rest are hit rarely => gaps of hundreds of steps
        hot = torch.randint(1, 40, (n // 2 + 1,), generator=g)
        cold = torch.randint(40, N, (n,), generator=g)
        idx = torch.unique(torch.cat([hot, cold]))[:cap]
        for r in idx.tolist():
            max_gap = max(max_gap, t - seen.get(r, 0))
            seen[r] = t
        sp = ops.SparseRows(cap, D, "cuda")
        sp.idx[:len(idx)] = idx.cuda()
        sp.rows[:len(idx)] = (torch.randn(len(idx), D, generator=g) * 1e-3).cuda()
        sp.n[0] = len(idx)
        ops.adamw_table(pd, md, vd, slot, sp, lr, b1, b2, eps, wd, t)
        ops.adamw_hyper_append(hyper, cumlog, t, lr, b1, b2, eps, wd)
        ops.adamw_rows(pl, ml, vl, last, hyper, cumlog, t - 1, t, b1, b2, eps, rows=sp.idx, n_rows=sp.n, max_rows=cap,
                       grows=sp.rows)
    assert max_gap > 600
    ops.adamw_rows(pl, ml, vl, last, hyper, cumlog, T, 0, b1, b2, eps)              # flush
    assert int((last != T).sum()) == 0 and int((slot != -1).sum()) == 0
    assert (pd - pl).abs().max().item() < 2e-7                                      # parity budget is 1e-5
    assert (md - ml).abs().max().item() < 1e-9 and (vd - vl).abs().max().item() < 1e-10
    # rows whose gaps never exceeded the exact window are bit-identical
    hot_rows = torch.arange(1, 40).cuda()
    assert torch.equal(pd[hot_rows], pl[hot_rows])


def test_graph_replay_equals_eager_steps():
    """hipGraph replay of the whole step == the eager sequence, step for step (dropout on: the seed offset and the
    optimizer step number live on the device), including a capture that starts without consuming a batch."""
    from pixelrec_amd.graph import GraphedTrainStep
    from pixelrec_amd.optim import PxrAdamW

    make, rng, zipf, synth = _setup(n_items=3000)
    batches = [tuple(torch.from_numpy(x).cuda() for x in synth.train_batch(3000, 4, 10, rng, zipf)) for _ in range(12)]

    def run(graph):
        m = make()
        m.hidden_dropout_prob = m.attn_dropout_prob = 0.1
        opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1)
        losses = []
        g = None
        for i, (it, mk) in enumerate(batches):
            if graph and i >= 2:
                if g is None:
                    g = GraphedTrainStep(m, opt, it, mk, warmup=0)
                losses.append(float(g(it, mk)))
            else:
                loss = m((it, mk)); loss.backward(); opt.step()
                losses.append(float(loss.detach()))
        return losses, {k: v.detach().clone() for k, v in m.state_dict().items()}, opt.step_count

    le, se, ne = run(False)
    lg, sg, ng = run(True)
    assert ne == ng == len(batches)
    assert le == lg                                   # identical losses, step by step (same kernels, same seeds)
    for k in se:
        assert torch.equal(se[k], sg[k]), k


@pytest.mark.parametrize("graph", [False, True])
def test_weight_gradients_beside_the_tail_of_the_step(graph):
    """weight_grad_mode "fork_tail" (the grouped weight-gradient launch on a side stream beside the reductions, the
    segmented sum and the row update) == the one-stream schedule, bit for bit, eagerly and replayed from a hipGraph."""
    from pixelrec_amd.graph import GraphedTrainStep
    from pixelrec_amd.optim import PxrAdamW

    make, rng, zipf, synth = _setup(n_items=3000)
    batches = [tuple(torch.from_numpy(x).cuda() for x in synth.train_batch(3000, 4, 10, rng, zipf)) for _ in range(8)]

    def run(mode):
        m = make()
        m.weight_grad_mode = mode
        m.defer_weight_grad_join = True
        m.hidden_dropout_prob = m.attn_dropout_prob = 0.1
        opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1)
        losses, g = [], None
        for i, (it, mk) in enumerate(batches):
            if graph and i >= 2:
                if g is None:
                    g = GraphedTrainStep(m, opt, it, mk, warmup=0)
                losses.append(float(g(it, mk)))
            else:
                loss = m((it, mk)); loss.backward(); opt.step()
                losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        return losses, {k: v.detach().clone() for k, v in m.state_dict().items()}

    l0, s0 = run("grouped")
    l1, s1 = run("fork_tail")
    assert l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_checkpoint_round_trip_resumes_bit_identically():
    """k steps -> state_dict (model + optimizer, lazy table flushed) -> fresh model / optimizer -> k more steps must equal
    2k uninterrupted steps bit for bit (dropout on: the seed offset is part of the resumed state through step_count)."""
    import numpy as np

    from pixelrec_amd import synth
    from pixelrec_amd.model import SASRec
    from pixelrec_amd.optim import PxrAdamW

    N, D, L, B, K = 3000, 64, 10, 16, 5
    cfg = {"n_layers": 2, "n_heads": 2, "embedding_size": D, "inner_size": 2, "hidden_dropout_prob": 0.1,
           "attn_dropout_prob": 0.1, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
           "MAX_ITEM_LIST_LENGTH": L, "seed": 2020}

    class DL:
        item_num = N

    rng = np.random.default_rng(1)
    z = synth.ZipfItems(N, seed=2)
    batches = [tuple(torch.from_numpy(a).cuda() for a in synth.train_batch(N, B, L, rng, z)) for _ in range(2 * K)]

    def fresh():
        torch.manual_seed(3)
        m = SASRec(cfg, DL()).cuda().train()
        return m, PxrAdamW(m, lr=1e-3, weight_decay=0.1)

    def run(m, opt, bs):
        for b in bs:
            opt.zero_grad()
            m(b).backward()
            opt.step()

    m_ref, o_ref = fresh()
    run(m_ref, o_ref, batches)
    ref = {k: v.detach().clone() for k, v in m_ref.state_dict().items()}

    m1, o1 = fresh()
    run(m1, o1, batches[:K])
    ckpt = {"state_dict": {k: v.detach().cpu().clone() for k, v in m1.state_dict().items()},
            "dropout_step": m1.dropout_step(),
            "optimizer": {k: (v.detach().cpu().clone() if torch.is_tensor(v) else v) for k, v in o1.state_dict().items()}}
    m2, o2 = fresh()
    m2.load_state_dict(ckpt["state_dict"], strict=True)
    o2.load_state_dict({k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ckpt["optimizer"].items()})
    m2.set_dropout_step(ckpt["dropout_step"])
    assert ckpt["dropout_step"] == K
    run(m2, o2, batches[K:])
    got = m2.state_dict()
    for k in ref:
        assert torch.equal(ref[k], got[k]), k


def test_load_state_dict_while_rows_lag():
    """Loading weights while the lazy optimizer still owes updates (no flush since the last steps): the owed updates
    belong to the REPLACED rows -- the loaded table must come back from state_dict() untouched, and training on from it
    must equal the dense-sweep optimizer doing the same thing."""
    make, rng, zipf, synth = _setup(n_items=1500)
    batches = [tuple(torch.from_numpy(x).cuda() for x in synth.train_batch(1500, 4, 10, rng, zipf)) for _ in range(12)]

    def scenario(mode):
        from pixelrec_amd.optim import PxrAdamW

        m = make()
        opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1, table_update=mode)
        start = {k: v.detach().clone() for k, v in m.state_dict().items()}
        for it, mk in batches[:6]:
            m((it, mk)).backward()
            opt.step()
        if mode == "lazy":
            assert opt._dirty                                     # rows lag behind the step counter right now
        m.load_state_dict(start, strict=True)                     # e.g. evaluate(load_best_model=True) mid-run
        after_load = {k: v.detach().clone() for k, v in m.state_dict().items()}
        for k in start:
            assert torch.equal(start[k], after_load[k]), (mode, k)
        for it, mk in batches[6:]:
            m((it, mk)).backward()
            opt.step()
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    dense, lazy = scenario("dense"), scenario("lazy")
    for k in dense:
        assert torch.equal(dense[k], lazy[k]), k


def test_lookahead_catch_up_changes_no_bit():
    """SASRec.set_next_batch (the next batch's table rows are caught up on a side stream beside this step's GEMMs and
    advanced through the step by the optimizer) is a pure schedule hint: with it, without it, with a WRONG hint and
    with the dense sweep the parameters end up bit-identical -- eagerly and through the captured step graph."""
    from pixelrec_amd.graph import GraphedTrainStep
    from pixelrec_amd.optim import PxrAdamW

    n_items = 3000
    make, rng, zipf, synth = _setup(n_items=n_items, D=64, L=10, B=6)
    batches = [tuple(torch.from_numpy(x).cuda() for x in synth.train_batch(n_items, 6, 10, rng, zipf)) for _ in range(40)]

    def run(mode, hint, graph=False):
        m = make()
        opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1, table_update=mode)
        g = GraphedTrainStep(m, opt, *batches[0], warmup=0, lookahead=hint != "none") if graph else None
        for i, (it, mk) in enumerate(batches):
            nxt = None
            if hint == "next" and i + 1 < len(batches):
                nxt = batches[i + 1][0]
            elif hint == "wrong":
                nxt = batches[(i * 7 + 3) % len(batches)][0]
            if g is not None:
                g(it, mk, next_items=nxt)
                continue
            if nxt is not None:
                m.set_next_batch(nxt)
            loss = m((it, mk))
            loss.backward()
            opt.step()
        return {k: v.detach().clone() for k, v in m.state_dict().items()}

    ref = run("dense", "none")
    for mode, hint, graph in (("lazy", "none", False), ("lazy", "next", False), ("lazy", "wrong", False),
                              ("lazy", "next", True), ("lazy", "none", True)):
        got = run(mode, hint, graph)
        for k in ref:
            assert torch.equal(ref[k], got[k]), (mode, hint, graph, k)


def test_sort_beside_the_forward_pass_changes_no_bit(monkeypatch):
    """Opt-in schedule PXR_SORT_OVERLAP=1 (rows claimed from the raw id tensor, the id sort on a second stream beside the
    forward pass) against the default serial one (sort -> catch-up of the unique list -> forward): same bits everywhere."""
    make, rng, zipf, synth = _setup(n_items=1500)
    batches = [tuple(torch.from_numpy(x).cuda() for x in synth.train_batch(1500, 4, 10, rng, zipf)) for _ in range(40)]
    a, opt_a = _run(make, batches, "lazy")
    monkeypatch.setenv("PXR_SORT_OVERLAP", "1")
    b, opt_b = _run(make, batches, "lazy")
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(opt_a._tm, opt_b._tm) and torch.equal(opt_a._tv, opt_b._tv) and torch.equal(opt_a._last, opt_b._last)


def test_catch_up_from_a_raw_id_list_equals_the_unique_list():
    """pxr_adamw_rows_ids_f32: duplicates, padding zeros and out-of-range ids in the list; the rows named are replayed exactly
    once, bit-identically to pxr_adamw_rows_f32 on the sorted unique list, and nothing else moves."""
    from pixelrec_amd.optim import PxrAdamW
    N = 900
    make, rng, zipf, synth = _setup(n_items=N)
    batches = [tuple(torch.from_numpy(x).cuda() for x in synth.train_batch(N, 4, 10, rng, zipf)) for _ in range(25)]
    ms, opts = [], []
    for _ in range(2):
        m = make()
        opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1, table_update="lazy")
        for it, mk in batches:
            m((it, mk)).backward()
            opt.step()
        ms.append(m); opts.append(opt)
    g = torch.Generator().manual_seed(5)
    valid = torch.randint(1, N, (300,), generator=g)
    raw = torch.cat([valid, valid[:120], torch.tensor([0, 0, -3, N, N + 17, 2 ** 40])])[torch.randperm(426, generator=g)].cuda()
    uniq = torch.unique(valid).cuda()
    before = ms[0].item_embedding.weight.data.clone()
    assert opts[0].catch_up_ids(raw.contiguous())
    opts[1].catch_up_rows(uniq, torch.tensor([uniq.numel()], dtype=torch.int32, device="cuda"), uniq.numel())
    torch.cuda.synchronize()
    for x, y in ((ms[0].item_embedding.weight.data, ms[1].item_embedding.weight.data), (opts[0]._tm, opts[1]._tm),
                 (opts[0]._tv, opts[1]._tv), (opts[0]._last, opts[1]._last)):
        assert torch.equal(x, y)
    moved = (ms[0].item_embedding.weight.data != before).any(dim=1).nonzero().flatten().cpu()
    assert set(moved.tolist()) <= set(uniq.cpu().tolist()) and moved.numel() > 0


def _optimizer_only_run(T=700, N=3000, D=64, seed=0, lr=1e-3, truth=False):
    """Identical sparse gradient sequences into the dense sweep and the lazy replay (no model feedback)."""
    from pixelrec_amd import ops
    wd, b1, b2, eps = 0.1, 0.9, 0.999, 1e-8
    g = torch.Generator().manual_seed(seed)
    p0 = (torch.randn(N, D, generator=g) * 0.02).cuda()
    pd, md, vd = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pl, ml, vl = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    slot = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    last = torch.zeros(N, dtype=torch.int32, device="cuda")
    hyper = torch.zeros(T + 8, 4, device="cuda")
    cumlog = torch.zeros(T + 8, dtype=torch.float64, device="cuda")
    cap = 24
    # the same recurrence in float64 (per-step scalars as the fp32 table holds them): what BOTH fp32 schedules approximate
    p64, m64, v64 = p0.double(), torch.zeros_like(p0, dtype=torch.float64), torch.zeros_like(p0, dtype=torch.float64)
    b1_64 = 1.0 - float(torch.tensor(1.0 - b1, dtype=torch.float32))     # the factor m - m * float(1 - b1) applies
    b2_64, omb2_64 = float(torch.tensor(b2, dtype=torch.float32)), float(torch.tensor(1.0 - b2, dtype=torch.float32))
    for t in range(1, T + 1):
        hot = torch.randint(1, 60, (8,), generator=g)                       # short gaps
        warm = torch.randint(60, 600, (8,), generator=g)                    # gaps of tens .. ~200 steps: the replayed range
        cold = torch.randint(600, N, (8,), generator=g)                     # beyond the replay window
        idx = torch.unique(torch.cat([hot, warm, cold]))[:cap]
        sp = ops.SparseRows(cap, D, "cuda")
        sp.idx[:len(idx)] = idx.cuda()
        # gradients with a wide dynamic range, including components far below eps-scale
        sp.rows[:len(idx)] = (torch.randn(len(idx), D, generator=g) * torch.logspace(-9, -2, D)).cuda()
        sp.n[0] = len(idx)
        ops.adamw_table(pd, md, vd, slot, sp, lr, b1, b2, eps, wd, t)
        ops.adamw_hyper_append(hyper, cumlog, t, lr, b1, b2, eps, wd)
        ops.adamw_rows(pl, ml, vl, last, hyper, cumlog, t - 1, t, b1, b2, eps, rows=sp.idx, n_rows=sp.n, max_rows=cap,
                       grows=sp.rows)
        if truth:
            g64 = torch.zeros_like(p64)
            g64[idx.cuda()] = sp.rows[:len(idx)].double()
            dec, ss, isb = (float(x) for x in hyper[t, :3].double().cpu())
            p64 = p64 * dec
            m64 = m64 + (g64 - m64) * (1.0 - b1_64)
            v64 = v64 * b2_64 + omb2_64 * g64 * g64
            p64 = p64 - ss * m64 / (v64.sqrt() * isb + eps)
    ops.adamw_rows(pl, ml, vl, last, hyper, cumlog, T, 0, b1, b2, eps)      # flush
    if truth:
        return (pd, md, vd), (pl, ml, vl), p64
    return (pd, md, vd), (pl, ml, vl)


@pytest.mark.fast_replay
def test_fast_replay_is_the_default_and_stays_inside_the_parity_budget():
    """Default replay (round 6: gaps summed in closed form, csrc/adamw.hip "SERIES replay"; short gaps and the first ~200 optimizer
    steps: sqrt(v) carried as a product, 1/denominator by Newton steps) against the dense sweep on identical gradient sequences, both
    against the same recurrence in float64.  The closed form rounds ONCE per gap where the sweep rounds once per step, so it sits
    CLOSER to the float64 recurrence than the sweep does, and the distance between the two fp32 schedules is the sweep's own
    rounding walk: p within 5e-7 of the sweep (the parity budget of the step is 1e-5), m / v to rounding."""
    (pd, md, vd), (pl, ml, vl), p64 = _optimizer_only_run(truth=True)
    assert not torch.equal(pd, pl)                                          # it IS the approximate path ...
    assert (pd - pl).abs().max().item() < 5e-7                              # ... well inside the budget
    e_lazy, e_dense = (pl.double() - p64).abs(), (pd.double() - p64).abs()
    assert e_lazy.max().item() < 2e-7 and e_lazy.max().item() <= e_dense.max().item(), (e_lazy.max(), e_dense.max())
    assert e_lazy.mean().item() <= e_dense.mean().item(), (e_lazy.mean(), e_dense.mean())
    assert (md - ml).abs().max().item() <= 2e-6 * md.abs().max().item()
    assert ((vd - vl).abs() <= 4e-5 * vd.abs() + 1e-30).all()


@pytest.mark.fast_replay
def test_carried_product_replay_alone_stays_within_1e_7_of_the_sweep(monkeypatch):
    """PXR_LAZY_SERIES=0 (every gap through the step-by-step loop, the default up to round 5): its roundings follow the sweep's."""
    monkeypatch.setenv("PXR_LAZY_SERIES", "0")
    (pd, md, vd), (pl, ml, vl) = _optimizer_only_run()
    assert not torch.equal(pd, pl)
    assert (pd - pl).abs().max().item() < 1e-7


def test_exact_replay_mode_is_bit_identical_on_the_same_sequences():
    (pd, md, vd), (pl, ml, vl) = _optimizer_only_run(T=240, N=800)
    assert torch.equal(pd, pl) and torch.equal(md, ml) and torch.equal(vd, vl)


@pytest.mark.fast_replay
def test_fast_replay_training_tracks_the_dense_schedule():
    """Whole model, default replay: a few steps of lazy against dense stay within the per-step parity budget (over long runs
    ANY rounding-level difference is amplified by Adam's sign-like update on noise-level gradients -- the GEMM modes diverge
    from each other the same way -- so the many-step statement is the optimizer-only one above)."""
    make, rng, zipf, synth = _setup(n_items=1500)
    batches = [tuple(torch.from_numpy(x).cuda() for x in synth.train_batch(1500, 4, 10, rng, zipf)) for _ in range(8)]
    dense, _ = _run(make, batches, "dense")
    lazy, _ = _run(make, batches, "lazy")
    for k in dense:
        assert (dense[k] - lazy[k]).abs().max().item() < 1e-5, k
