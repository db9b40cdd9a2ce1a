"""Series replay of the lazy table AdamW (csrc/adamw.hip, adamw_rows_kernel "SERIES replay"): a gap of zero-gradient steps summed in
closed form from six wave-reduced moments of the per-step scalars, against the step-by-step replays of the same rows -- the exact one
(PXR_LAZY_REPLAY=exact: the dense sweep's own arithmetic, torch.optim.AdamW semantics of the reference trainer.py:66-103,125) and the
carried-product loop it replaces (PXR_LAZY_SERIES=0)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

B1, B2, EPS, WD = 0.9, 0.999, 1e-8, 0.1


def _table(T, lr_of):
    from pixelrec_amd import ops
    hyper = torch.zeros(T + 8, 4, device="cuda")
    cumlog = torch.zeros(T + 8, dtype=torch.float64, device="cuda")
    for t in range(1, T + 1):
        ops.adamw_hyper_append(hyper, cumlog, t, lr_of(t), B1, B2, EPS, WD)
    return hyper, cumlog


def _rows(N, D, gaps, T, seed):
    """Rows in every state a trained table holds: moments from gradients over seven decades (components far below eps included),
    rows that never saw a gradient (m = v = 0), and a few with a single recent gradient (|m| / sqrt(v) at its maximum)."""
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(N, D, generator=g) * 0.02
    gr = torch.randn(N, D, generator=g) * torch.logspace(-9, -2, D)
    m = gr * torch.rand(N, 1, generator=g)
    v = gr * gr * torch.rand(N, 1, generator=g).clamp_min(1e-3)
    m[::7] = 0.0
    v[::7] = 0.0
    m[3::11] = 0.1 * gr[3::11]
    v[3::11] = 0.001 * gr[3::11] ** 2
    last = torch.tensor([T - gaps[i % len(gaps)] for i in range(N)], dtype=torch.int32)
    return p.cuda(), m.cuda(), v.cuda(), last.cuda()


def _truth64(state, hyper, T):
    """The zero-gradient recurrence of the rows in float64, per-step scalars as the fp32 table holds them."""
    p, m, v = (t[1:].double().clone() for t in state[:3])
    last = state[3][1:]
    b1 = 1.0 - float(torch.tensor(1.0 - B1, dtype=torch.float32))        # the factor m - m * float(1 - b1) applies
    b2 = float(torch.tensor(B2, dtype=torch.float32))
    h = hyper.double().cpu()
    for t in range(int(last.min()) + 1, T + 1):
        act = (last < t).view(-1, 1)
        dec, ss, isb = float(h[t, 0]), float(h[t, 1]), float(h[t, 2])
        m2, v2 = m * b1, v * b2
        p2 = p * dec - ss * m2 / (v2.sqrt() * isb + EPS)
        p, m, v = torch.where(act, p2, p), torch.where(act, m2, m), torch.where(act, v2, v)
    return p


def _catch_up(monkeypatch, mode, state, hyper, cumlog, T):
    from pixelrec_amd import ops
    monkeypatch.delenv("PXR_LAZY_REPLAY", raising=False)
    monkeypatch.delenv("PXR_LAZY_SERIES", raising=False)
    if mode == "exact":
        monkeypatch.setenv("PXR_LAZY_REPLAY", "exact")
    elif mode == "loop":
        monkeypatch.setenv("PXR_LAZY_SERIES", "0")
    p, m, v, last = (t.clone() for t in state)
    N = p.shape[0]
    idx = torch.arange(N, dtype=torch.int64, device="cuda")
    # (row 0 is the padding row of a table: the row list skips id 0 -- the comparison below leaves it out)
    ops.adamw_rows(p, m, v, last, hyper, cumlog, T, 0, B1, B2, EPS, rows=idx, n_rows=torch.tensor([N], dtype=torch.int32, device="cuda"),
                   max_rows=N)
    torch.cuda.synchronize()
    assert int(last[1:].min()) == T
    return p[1:], m[1:], v[1:]


@pytest.mark.parametrize("T,lr_of", [(1500, lambda t: 1e-3), (700, lambda t: 1e-3),
                                      (900, lambda t: 1e-3 * (0.55 + 0.45 * math.cos(t / 40.0)))],
                         ids=["step1500", "step700", "moving_lr"])
def test_series_replay_against_the_exact_and_the_carried_product_replays(monkeypatch, T, lr_of):
    hyper, cumlog = _table(T, lr_of)
    gaps = [1, 2, 3, 5, 6, 7, 9, 16, 31, 50, 64, 65, 80, 100, 127, 128]
    state = _rows(1 + 16 * 24, 512, gaps, T, seed=T)
    ex = _catch_up(monkeypatch, "exact", state, hyper, cumlog, T)
    lp = _catch_up(monkeypatch, "loop", state, hyper, cumlog, T)
    se = _catch_up(monkeypatch, "series", state, hyper, cumlog, T)
    gap_of = torch.tensor([gaps[i % len(gaps)] for i in range(1, 1 + 16 * 24)], device="cuda")
    short = gap_of < 6
    # gaps below the threshold take the loop in both settings: the same bits; longer gaps take the series (not the same bits)
    assert torch.equal(se[0][short], lp[0][short])
    assert not torch.equal(se[0][~short], lp[0][~short])
    # p (|p| ~ 0.02-0.08, summed updates up to ~0.1 at lr 1e-3).  The exact replay rounds once per step -- a walk of ~1e-7 over
    # a 128-step gap; the series rounds once per gap: it is held to 4e-8 of the recurrence in float64, must be CLOSER to it than the
    # exact fp32 replay is (maximum and mean), and within that walk of the exact replay
    tr = _truth64(state, hyper, T)
    e_se, e_ex = (se[0].double() - tr).abs(), (ex[0].double() - tr).abs()
    assert e_se.max().item() < 4e-8, (e_se.max().item(), e_ex.max().item())
    assert e_se.max().item() <= e_ex.max().item() and e_se.mean().item() <= 0.5 * e_ex.mean().item()
    assert (se[0] - ex[0]).abs().max().item() < 2.5e-7
    # m: b1^gap in one rounding of exp2 against gap roundings; v likewise
    assert (se[1] - ex[1]).abs().max().item() <= 4e-6 * ex[1].abs().max().item()
    assert ((se[2] - ex[2]).abs() <= 4e-5 * ex[2].abs() + 1e-30).all()


def test_series_replay_beyond_the_window_and_with_long_gaps(monkeypatch):
    """Gaps past the windows: the series sums 256 steps (the exact mode's window; the loop stops at 128 and drops up to 4.7e-4 lr of
    Adam terms), then the closed-form decay tail takes over as in every mode."""
    T = 1400
    hyper, cumlog = _table(T, lambda t: 1e-3)
    gaps = [129, 200, 256, 257, 400, 1000]
    state = _rows(1 + 6 * 16, 512, gaps, T, seed=3)
    ex = _catch_up(monkeypatch, "exact", state, hyper, cumlog, T)
    lp = _catch_up(monkeypatch, "loop", state, hyper, cumlog, T)
    se = _catch_up(monkeypatch, "series", state, hyper, cumlog, T)
    tr = _truth64(state, hyper, T)
    e_se, e_ex, e_lp = ((x[0].double() - tr).abs() for x in (se, ex, lp))
    assert e_se.max().item() < 6e-8 and e_se.max().item() <= e_ex.max().item() and e_se.max().item() <= e_lp.max().item()
    assert (se[0] - ex[0]).abs().max().item() < 4e-7         # (the exact replay's rounding walk over 256 steps)
    assert (se[1] - ex[1]).abs().max().item() <= 4e-6 * ex[1].abs().max().item() + 1e-30
    assert ((se[2] - ex[2]).abs() <= 8e-5 * ex[2].abs() + 1e-30).all()


@pytest.mark.parametrize("T", [40, 130, 260])
def test_early_steps_fall_back_where_the_bound_says_so(monkeypatch, T):
    """In the first optimizer steps the bias correction moves the denominator by percents per step: the per-row truncation bound
    sends those gaps to the loop (same bits as PXR_LAZY_SERIES=0); whatever path a row takes, it stays on the exact replay."""
    hyper, cumlog = _table(T, lambda t: 1e-3)
    gaps = [g for g in (6, 12, 30, 60, 100, 128) if g < T]
    state = _rows(1 + len(gaps) * 16, 256, gaps, T, seed=T)
    lp = _catch_up(monkeypatch, "loop", state, hyper, cumlog, T)
    se = _catch_up(monkeypatch, "series", state, hyper, cumlog, T)
    tr = _truth64(state, hyper, T)
    assert (se[0].double() - tr).abs().max().item() <= max((lp[0].double() - tr).abs().max().item(), 4e-8)
    if T == 40:
        # the 30-step gaps start at optimizer step 11, where the bias correction moves the denominator by ~4 % per step: fell back
        long_gap = torch.tensor([gaps[i % len(gaps)] == 30 for i in range(1, 1 + len(gaps) * 16)], device="cuda")
        assert torch.equal(se[0][long_gap], lp[0][long_gap])


def test_other_betas_keep_the_loop(monkeypatch):
    """b1 / sqrt(b2) > 0.95: the update terms do not die out inside the window, the launch does not arm the series."""
    from pixelrec_amd import ops
    T = 600
    hyper = torch.zeros(T + 8, 4, device="cuda")
    cumlog = torch.zeros(T + 8, dtype=torch.float64, device="cuda")
    for t in range(1, T + 1):
        ops.adamw_hyper_append(hyper, cumlog, t, 1e-3, 0.99, 0.999, EPS, WD)
    p, m, v, last = _rows(1 + 64, 256, [20, 90], T, seed=9)
    outs = []
    for series in ("0", "1"):
        monkeypatch.setenv("PXR_LAZY_SERIES", series)
        pp, mm, vv, ll = p.clone(), m.clone(), v.clone(), last.clone()
        idx = torch.arange(65, dtype=torch.int64, device="cuda")
        ops.adamw_rows(pp, mm, vv, ll, hyper, cumlog, T, 0, 0.99, 0.999, EPS, rows=idx,
                       n_rows=torch.tensor([65], dtype=torch.int32, device="cuda"), max_rows=65)
        outs.append(pp)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
