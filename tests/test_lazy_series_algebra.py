"""The algebra behind the series replay of the lazy table AdamW (csrc/adamw.hip "SERIES replay"), in numpy float64 on the
CPU: the closed form  p_n = p_0 P_0 - (m_0 / D_1) sum_q (-w)^q T_q  against the zero-gradient AdamW recurrence it replaces
(torch.optim.AdamW with g = 0, reference trainer.py:66-103,125), and the per-row remainder bound the kernel tests before it trusts
the truncated series."""
import numpy as np
import pytest

B1, B2, EPS, WD = 0.9, 0.999, 1e-8, 0.1


def _table(T, lr):
    t = np.arange(0, T + 1, dtype=np.float64)
    dec = np.full(T + 1, 1.0 - lr * WD)
    dec[0] = 1.0
    ss = np.zeros(T + 1)
    ss[1:] = lr / (1.0 - B1 ** t[1:])
    isb = np.ones(T + 1)
    isb[1:] = 1.0 / np.sqrt(1.0 - B2 ** t[1:])
    return dec, ss, isb


def _recurrence(p, m, v, s, n, tab):
    dec, ss, isb = tab
    p, m, v = p.copy(), m.copy(), v.copy()
    for k in range(s, s + n):
        p *= dec[k]
        m *= B1
        v *= B2
        p -= ss[k] * m / (np.sqrt(v) * isb[k] + EPS)
    return p


def _series(p, m, v, s, n, tab, Q=5):
    """Returns (p_n, remainder bound relative to |T_0|) -- the kernel's formulas, term for term."""
    dec, ss, isb = tab
    j = np.arange(1, n + 1)
    k = s + j - 1
    cl = np.concatenate([[0.0], np.cumsum(np.log(dec[s:s + n]))])
    P_j, P_0 = np.exp(cl[n] - cl[j]), np.exp(cl[n])
    c = np.sqrt(B2) ** j * isb[k]
    a = ss[k] * B1 ** j * P_j
    sg = c / c[0] - 1.0
    T = [np.sum(a * sg ** q) for q in range(Q + 1)]
    bound = np.sum(np.abs(a) * np.abs(sg) ** (Q + 1) / (1.0 - np.abs(sg))) / abs(T[0])
    d1 = np.sqrt(v) * c[0] + EPS
    w = np.sqrt(v) * c[0] / d1
    poly = T[Q] + 0.0 * w
    for q in range(Q - 1, -1, -1):
        poly = T[q] - w * poly
    return p * P_0 - (m / d1) * poly, bound


@pytest.mark.parametrize("T", [260, 700, 1500, 20000])
@pytest.mark.parametrize("n", [6, 30, 128, 256])
def test_series_equals_the_recurrence_up_to_its_own_remainder_bound(T, n):
    rng = np.random.default_rng(T + n)
    D = 2048
    gr = rng.standard_normal(D) * np.logspace(-9, -2, D)       # gradients over seven decades, far below eps included
    m, v, p = gr * 0.5, gr * gr * 0.05, rng.standard_normal(D) * 0.02
    m[::9], v[::9] = 0.0, 0.0                                  # rows that never saw a gradient
    tab = _table(T + 8, 1e-3)
    s = T - n + 1
    want = _recurrence(p, m, v, s, n, tab)
    got, bound = _series(p, m, v, s, n, tab)
    upd = np.abs(want - p * np.prod(tab[0][s:s + n])).max()     # size of the summed Adam terms
    err = np.abs(got - want).max()
    assert err <= 1.05 * bound * upd + 3e-16, (err, bound, upd)      # (+ float64 rounding of |p| ~ 0.05 over the loop)
    if bound <= 2e-7:                                          # the rows the kernel lets through
        assert err <= 2.2e-7 * upd + 3e-16


def test_the_bound_rejects_the_first_optimizer_steps_and_accepts_a_running_epoch():
    z = np.zeros(4)
    assert _series(z, z, z + 1e-8, 11, 30, _table(64, 1e-3))[1] > 2e-7        # bias correction still moving by percents per step
    assert _series(z, z, z + 1e-8, 30, 100, _table(160, 1e-3))[1] > 2e-7
    assert _series(z, z, z + 1e-8, 493, 128, _table(640, 1e-3))[1] < 2e-7     # the bench stream's state (optimizer aged 620 steps)
    assert _series(z, z, z + 1e-8, 365, 256, _table(640, 1e-3))[1] < 2e-7
    assert _series(z, z, z + 1e-8, 40, 6, _table(64, 1e-3))[1] < 2e-7         # a short gap passes early on: its sigmas are small
