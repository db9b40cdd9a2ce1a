"""The arithmetic behind the bf16x3 GEMM mode (csrc/gemm_b3.cuh), restated in numpy: the three-term bf16 split of an
fp32 number is EXACT, and a dot product formed from six of the nine cross products differs from the exact one by a
few 2^-25 |a||b| per term -- the error class of an fp32 FMA chain.  (The kernels themselves are held to fp64 on the GPU
in tests/test_gpu_gemm_b3.py; this pins the scheme where no GPU is needed.)"""
import numpy as np


def bf16_rne(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 -- what v_cvt_pk_bf16_f32 does for finite inputs."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    hi = bf16_rne(x)
    r1 = (x - hi).astype(np.float32)
    mid = bf16_rne(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = bf16_rne(r2)
    return hi, mid, lo, r1, r2


def test_three_bf16_terms_carry_all_24_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-40, 40, 200000))).astype(np.float32)
    x = np.concatenate([x, np.float32([0.0, 1.0, -1.0, 3.0, 65504.0, 1e-30, -2.5e-20, 16777215.0, 0.1, 1.0 + 2.0 ** -23])])
    hi, mid, lo, r1, r2 = split3(x)
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))      # remainders are exact
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - mid.astype(np.float64))
    assert np.array_equal(lo, r2)                                                                   # lo needs <= 8 bits
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    ax = np.abs(x.astype(np.float64))
    assert (np.abs(mid) <= ax * 2.0 ** -8).all() and (np.abs(lo) <= ax * 2.0 ** -16).all()


def test_six_products_are_fp32_class():
    rng = np.random.default_rng(1)
    for K in (32, 512, 4096):
        a = (rng.standard_normal((64, K)) * np.exp(rng.uniform(-6, 6, (64, 1)))).astype(np.float32)
        b = (rng.standard_normal((64, K)) * np.exp(rng.uniform(-6, 6, (64, 1)))).astype(np.float32)
        ah, am, al, _, _ = split3(a)
        bh, bm, bl, _, _ = split3(b)
        d = lambda p, q: (p.astype(np.float64) * q.astype(np.float64)).sum(1)       # bf16 x bf16 products are exact
        six = d(ah, bh) + (d(ah, bm) + d(am, bh)) + (d(am, bm) + d(ah, bl) + d(al, bh))
        exact = d(a, b)
        scale = (np.abs(a.astype(np.float64)) * np.abs(b.astype(np.float64))).sum(1)
        dropped = np.abs(six - exact) / scale
        assert dropped.max() < 2.0 ** -24                                           # < one fp32 ulp of sum |a||b|
        # an fp32 FMA chain on the same data, for scale
        acc = np.zeros(64, dtype=np.float32)
        for k in range(K):
            acc = (acc.astype(np.float64) + a[:, k].astype(np.float64) * b[:, k].astype(np.float64)).astype(np.float32)
        chain = np.abs(acc.astype(np.float64) - exact) / scale
        assert dropped.max() <= max(chain.max(), 2.0 ** -26) * 4
