"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the library's two-plane fp16 operand format and of the three-product GEMM
built on it (pixelrec_amd/csrc/planes.cuh "h2", gemm_p4.cuh P4Cfg<..., HALF>, h2.hip):

    x * 2^e = hi + lo + d,   hi = fp16_rne(x 2^e),  lo = fp16_rne(x 2^e - hi),   |d| <= max(2^-22 |x 2^e|, 2^-25)
    A B^T  ~=  2^-(ea + eb) (hi_a hi_b^T + lo_a hi_b^T + hi_a lo_b^T)            (lo_a lo_b^T, 2^-22 relative, is dropped)

so that the accuracy claims of DESIGN.md (as accurate against fp64 as the six-product 3 x bf16 kernels) and the scale rules
(host: e = 14 - ceil(log2 max|x|); the rigorous bound e = 15 - ceil(log2(max|dy| * max column sum |W| * factor)) for a gradient that
is written before its maximum can be known) can be checked on the CPU, without the GPU.  The products are accumulated in fp64
here: what is restated is the OPERAND format and the choice of products, not the MFMA's fp32 accumulation order."""
import math

import numpy as np


def exponent(max_abs: float, top: int = 14) -> int:
    """e with max_abs * 2^e in [2^(top-1), 2^top)   (pixelrec_amd.ops.h2_exponent / h2.hip::h2_exp_for)."""
    if not (max_abs > 0.0) or math.isinf(max_abs):
        return 0
    return max(-60, min(60, top - math.frexp(max_abs)[1]))


def split(x: np.ndarray, e: int = 0):
    """-> (hi, lo) float16 planes of x * 2^e (fp32 arithmetic like the kernels: the remainder is exact)."""
    xs = (np.asarray(x, dtype=np.float32) * np.float32(2.0 ** e)).astype(np.float32)
    hi = xs.astype(np.float16)
    lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def dense(hi, lo, e: int = 0) -> np.ndarray:
    return (hi.astype(np.float64) + lo.astype(np.float64)) * 2.0 ** -e


def gemm3(a_planes, ea: int, b_planes, eb: int) -> np.ndarray:
    """A [M,K] x B[N,K]^T from planes: the three products the kernels execute."""
    ah, al = (p.astype(np.float64) for p in a_planes)
    bh, bl = (p.astype(np.float64) for p in b_planes)
    return (ah @ bh.T + al @ bh.T + ah @ bl.T) * 2.0 ** -(ea + eb)


def split_bf16x3(x: np.ndarray):
    """The exact three-term bf16 split of the six-product kernels (planes.cuh::p3_split2), for comparison."""
    def bf16(v):
        u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000          # round to nearest even on the upper 16 bits
        return u.astype(np.uint32).view(np.float32)
    x = np.asarray(x, dtype=np.float32)
    h = bf16(x)
    r1 = (x - h).astype(np.float32)
    m = bf16(r1)
    r2 = (r1 - m).astype(np.float32)
    lo = bf16(r2)
    return h, m, lo


def gemm6(a3, b3) -> np.ndarray:
    ah, am, al = (p.astype(np.float64) for p in a3)
    bh, bm, bl = (p.astype(np.float64) for p in b3)
    return ah @ bh.T + am @ bh.T + ah @ bm.T + al @ bh.T + ah @ bl.T + am @ bm.T


def bound_exponent(dy_max: float, w_colsum_max: float, factor: float = 1.0) -> int:
    """pxr_h2_bound_exp: e with |dy W| * factor * 2^e < 2^15 for every element."""
    return exponent(float(np.float32(dy_max) * np.float32(w_colsum_max) * np.float32(factor)), 15)
