"""CPU checks of the two-plane fp16 operand format against its numpy restatement (oracle/h2_oracle.py): representation error,
the three-product GEMM against fp64 and against the six-product 3 x bf16 scheme on the same operands, the scale rules (host
exponent, rigorous bound for an input gradient), and pixelrec_amd.ops.h2_exponent == the oracle's."""
import numpy as np
import pytest

from oracle import h2_oracle as H


def test_exponent_puts_the_maximum_below_two_to_the_14():
    from pixelrec_amd.ops import h2_exponent

    rng = np.random.default_rng(0)
    for v in list(10.0 ** rng.uniform(-30, 30, 200)) + [1.0, 2.0 ** 13, 2.0 ** 14, 65504.0, 1e-45, 0.0, float("inf")]:
        e = H.exponent(v)
        assert e == h2_exponent(v)
        if 0.0 < v < float("inf") and -60 < e < 60:
            assert 2.0 ** 13 <= v * 2.0 ** e < 2.0 ** 14


@pytest.mark.parametrize("scale", [1.0, 0.02, 3e-5, 300.0])
def test_two_planes_hold_22_bits(scale):
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((200, 96)) * scale).astype(np.float32)
    e = H.exponent(float(np.abs(x).max()))
    hi, lo = H.split(x, e)
    assert np.isfinite(hi.astype(np.float32)).all()
    err = np.abs(H.dense(hi, lo, e) - x.astype(np.float64))
    # |d| <= max(2^-22 |x|, 2^-25 / 2^e): relative 2^-22 down to 2^-17 of the maximum, an absolute floor below
    assert (err <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -25 * 2.0 ** -e) * 1.0001).all()


def test_unscaled_small_operands_hit_the_fp16_floor_and_the_scale_removes_it():
    rng = np.random.default_rng(2)
    x = (rng.standard_normal((64, 64)) * 1e-6).astype(np.float32)
    rel = lambda e: float(np.sqrt(np.mean((H.dense(*H.split(x, e), e) - x) ** 2) / np.mean(x.astype(np.float64) ** 2)))
    assert rel(0) > 2.0 ** -8                      # 2^-25 absolute on 1e-6 values
    assert rel(H.exponent(float(np.abs(x).max()))) < 2.0 ** -21


@pytest.mark.parametrize("M,N,K,ws", [(128, 96, 512, 0.05), (64, 64, 3072, 0.02), (96, 32, 64, 1.0)])
def test_three_products_are_as_accurate_as_six(M, N, K, ws):
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = (rng.standard_normal((N, K)) * ws).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64).T
    ea, eb = H.exponent(float(np.abs(a).max())), H.exponent(float(np.abs(b).max()))
    c3 = H.gemm3(H.split(a, ea), ea, H.split(b, eb), eb)
    c6 = H.gemm6(H.split_bf16x3(a), H.split_bf16x3(b))
    rms = lambda c: float(np.sqrt(np.mean((c - ref) ** 2) / np.mean(ref ** 2)))
    e3, e6 = rms(c3), rms(c6)
    assert e3 < 2.0 ** -21, (e3, e6)               # 22-bit operands, the lo*lo product dropped
    assert e3 < 8.0 * e6 + 2.0 ** -24, (e3, e6)    # the six-product scheme drops terms of 2^-24: same class
    assert np.abs(c3 - ref).max() <= 2.0 ** -20 * np.sqrt(K) * np.abs(a).max() * np.abs(b).max()


def test_the_input_gradient_bound_never_overflows_and_is_not_absurdly_loose():
    rng = np.random.default_rng(5)
    for trial in range(20):
        T, N, K = 300, 128, 64
        dy = (rng.standard_normal((T, N)) * 10.0 ** rng.uniform(-7, -2)).astype(np.float32)
        W = (rng.standard_normal((N, K)) * 10.0 ** rng.uniform(-3, 0)).astype(np.float32)
        mul = rng.uniform(0.0, 1.13, (T, K)).astype(np.float32)
        du = (dy.astype(np.float64) @ W.astype(np.float64)) * mul
        e = H.bound_exponent(float(np.abs(dy).max()), float(np.abs(W).sum(0).max()), 1.13)
        top = np.abs(du).max() * 2.0 ** e
        assert top < 2.0 ** 15 < 65504.0           # rigorous: no overflow whatever the data
        assert top > 2.0 ** 3                      # loose by sqrt(N)-vs-N at most (N = 128: <= 2^12 here with the random mul)
        hi, lo = H.split(du.astype(np.float32), e)
        assert np.isfinite(hi.astype(np.float32)).all()
        # rows*max in place of the column sums (the sequence block's weights): still a bound, at most rows/1 looser
        e2 = H.bound_exponent(float(np.abs(dy).max()), float(N * np.abs(W).max()), 1.13)
        assert e2 <= e and np.abs(du).max() * 2.0 ** e2 < 2.0 ** 15
