// gemm_f32.cuh -- LDS-tiled fp32 GEMM main loop on the gfx950 f32-input MFMA (v_mfma_f32_32x32x2_f32).
//
// Why f32 MFMA: the parity bar is +-1e-4 on logits / +-1e-5 on post-step parameters against the fp32
// reference (SURVEY.md §7 hard part 1).  v_mfma_f32_32x32x2_f32 is bit-for-bit an fp32 fmaf chain and runs
// at the fp32 vector peak (157 TFLOP/s) from ONE wave per SIMD; gfx950 has no xf32/TF32.
//
// Computes C[M,N] = A_op[M,K] * B_op[K,N] for operands stored either
//   "KC" (k-contiguous):  A as [M][K] row-major (lda = row stride),  B as [N][K] row-major  (nn.Linear weight)
//   "XC" (x-contiguous):  A as [K][M] row-major,                      B as [K][N] row-major
// which covers   Y = X W^T            (KC,KC)   reference layers.py:586-588,613,666,669 and sasrec.py:112
//                dX = dY W            (KC,XC)   autograd of the above
//                dW = dY^T X          (XC,XC)   autograd of the above (reduction over tokens, split-K)
//
// Tiling: 256 threads = 4 waves arranged 2x2; block tile BM x BN, K step BK = 32; each wave owns
// (BM/2)x(BN/2) made of 32x32 MFMA blocks.  LDS images need no transposes in either flavour:
//   KC operand -> LDS [row][k] with row stride BK+4 floats (ds_write_b128 / ds_read_b128 conflict-free:
//                 36*r mod 64 is a bijection on 16-byte slots for the 16 rows of a b128 lane group);
//                 a lane reads 4 consecutive k; lanes 0-31 take k 0..3, lanes 32-63 take k 4..7 of the
//                 8-wide k sub-step, so register t feeds MFMA t with k = 4*(lane>>5)+t on BOTH operands.
//   XC operand -> LDS [k][x] with row stride BM (or BN) floats; per MFMA one ds_read_b32 at
//                 [4*(lane>>5)+t][x0 + (lane&31)] (32 consecutive floats per half-wave: conflict-free).
// Double-buffered: global loads for tile kt+1 are issued into registers before the MFMAs of tile kt and
// written to the other LDS buffer afterwards; one barrier per K tile.
#pragma once
#include "pxr_common.h"

namespace pxr {

constexpr int GEMM_BK = 32;
constexpr int GEMM_THREADS = 256;

// KW = number of wave groups that split the k-steps of each K tile between them (intra-workgroup split-K): KW=2
// doubles the waves per output tile, which is what hides the LDS/global latency bubbles when a GEMM has too few
// output tiles to fill the chip (M = B*L = 3200 tokens); the KW partial accumulators are added through LDS.
template <int BM, int BN, bool A_KC, bool B_KC, int KW = 1>
struct GemmCfg {
  static constexpr int BK = GEMM_BK;
  static constexpr int NT = GEMM_THREADS * KW;
  static constexpr int LDA = A_KC ? (BK + 4) : BM;
  static constexpr int LDB = B_KC ? (BK + 4) : BN;
  static constexpr int A_STAGE = (A_KC ? BM : BK) * LDA;  // floats
  static constexpr int B_STAGE = (B_KC ? BN : BK) * LDB;
  static constexpr int STAGE = A_STAGE + B_STAGE;
  static constexpr int LDS_BYTES = 2 * STAGE * 4;
  static constexpr int WM = BM / 2, WN = BN / 2;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static constexpr int A_LD4 = BM * BK / 4 / NT;  // float4 loads per thread per tile
  static constexpr int B_LD4 = BN * BK / 4 / NT;
  static_assert(A_LD4 >= 1 && B_LD4 >= 1, "tile too small for this many threads");
  static_assert(BM % 64 == 0 && BN % 64 == 0, "tile must be a multiple of 64");
  struct Acc {
    f32x16 v[TM][TN];
  };
};

// ---- global -> register tile fetch (zero-filled outside the matrix) ------------------------------------
// KC operand: matrix [X][K] (row stride ld), tile rows x0.., k range k0..k0+31
template <int BX, int NLD, int NT>
__device__ __forceinline__ void fetch_kc(float4 (&r)[NLD], const float* __restrict__ g, int64_t ld, int x0,
                                         int X, int k0, int kend, int tid) {
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    const int f = tid + p * NT;  // float4 index in tile: row = f/8, q = f%8
    const int row = f >> 3, q = f & 7;
    const int gx = x0 + row, gk = k0 + q * 4;
    if (gx < X && gk < kend)
      r[p] = *reinterpret_cast<const float4*>(g + (int64_t)gx * ld + gk);
    else
      r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int BX, int NLD, int NT>
__device__ __forceinline__ void stash_kc(const float4 (&r)[NLD], float* lds, int tid) {
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    const int f = tid + p * NT;
    const int row = f >> 3, q = f & 7;
    *reinterpret_cast<float4*>(lds + row * (GEMM_BK + 4) + q * 4) = r[p];
  }
}
// XC operand: matrix [K][X] (row stride ld), tile k rows k0..k0+31, x range x0..x0+BX-1
template <int BX, int NLD, int NT>
__device__ __forceinline__ void fetch_xc(float4 (&r)[NLD], const float* __restrict__ g, int64_t ld, int x0,
                                         int X, int k0, int kend, int tid) {
  constexpr int Q = BX / 4;  // float4 per k row
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    const int f = tid + p * NT;
    const int kk = f / Q, q = f % Q;
    const int gk = k0 + kk, gx = x0 + q * 4;
    if (gk < kend && gx < X)
      r[p] = *reinterpret_cast<const float4*>(g + (int64_t)gk * ld + gx);
    else
      r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int BX, int NLD, int NT>
__device__ __forceinline__ void stash_xc(const float4 (&r)[NLD], float* lds, int tid) {
  constexpr int Q = BX / 4;
#pragma unroll
  for (int p = 0; p < NLD; ++p) {
    const int f = tid + p * NT;
    const int kk = f / Q, q = f % Q;
    *reinterpret_cast<float4*>(lds + kk * BX + q * 4) = r[p];
  }
}

// ---- the main loop ---------------------------------------------------------------------------------------
// acc[bi][bj] element `reg` of lane l is C[m0 + wm*WM + bi*32 + (reg&3) + 8*(reg>>2) + 4*(l>>5)]
//                                         [n0 + wn*WN + bj*32 + (l&31)]          (gfx950 32x32 C/D map)
// CS (only with an XC A operand): additionally accumulate, per thread, the sum over k of the A float4s it stages
// (column sums of the stored [K][M] matrix = bias gradient when A = dY) into *cs; the caller reduces across threads.
template <int BM, int BN, bool A_KC, bool B_KC, bool CS = false, int KW = 1>
__device__ __forceinline__ void gemm_mainloop(typename GemmCfg<BM, BN, A_KC, B_KC, KW>::Acc& accs,
                                              const float* __restrict__ A, int64_t lda,
                                              const float* __restrict__ B, int64_t ldb, int M, int N,
                                              int kbeg, int kend, int m0, int n0, float* smem,
                                              float4* cs = nullptr) {
  using Cfg = GemmCfg<BM, BN, A_KC, B_KC, KW>;
  constexpr int NT = Cfg::NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 2, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  const int h = lane >> 5, r = lane & 31;
  auto& acc = accs.v;

#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[Cfg::A_LD4], rb[Cfg::B_LD4];
  const int nk = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;
  if (nk <= 0) return;

  auto fetch = [&](int kt) {
    const int k0 = kbeg + kt * GEMM_BK;
    if constexpr (A_KC) fetch_kc<BM, Cfg::A_LD4, NT>(ra, A, lda, m0, M, k0, kend, tid);
    else fetch_xc<BM, Cfg::A_LD4, NT>(ra, A, lda, m0, M, k0, kend, tid);
    if constexpr (CS) {
      static_assert(!A_KC, "column sums are taken over an x-contiguous A operand");
#pragma unroll
      for (int p = 0; p < Cfg::A_LD4; ++p) { cs->x += ra[p].x; cs->y += ra[p].y; cs->z += ra[p].z; cs->w += ra[p].w; }
    }
    if constexpr (B_KC) fetch_kc<BN, Cfg::B_LD4, NT>(rb, B, ldb, n0, N, k0, kend, tid);
    else fetch_xc<BN, Cfg::B_LD4, NT>(rb, B, ldb, n0, N, k0, kend, tid);
  };
  auto stash = [&](int buf) {
    float* sa = smem + buf * Cfg::STAGE;
    float* sb = sa + Cfg::A_STAGE;
    if constexpr (A_KC) stash_kc<BM, Cfg::A_LD4, NT>(ra, sa, tid);
    else stash_xc<BM, Cfg::A_LD4, NT>(ra, sa, tid);
    if constexpr (B_KC) stash_kc<BN, Cfg::B_LD4, NT>(rb, sb, tid);
    else stash_xc<BN, Cfg::B_LD4, NT>(rb, sb, tid);
  };

  fetch(0);
  stash(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1 < nk);
    if (more) fetch(kt + 1);
    const float* sa = smem + (kt & 1) * Cfg::STAGE;
    const float* sb = sa + Cfg::A_STAGE;
#pragma unroll
    for (int ks0 = 0; ks0 < GEMM_BK / 8; ks0 += KW) {
      const int ks = ks0 + (KW > 1 ? wk : 0);
      float a[Cfg::TM][4], b[Cfg::TN][4];
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        const int row = wm * Cfg::WM + i * 32 + r;
        if constexpr (A_KC) {
          const float4 v = *reinterpret_cast<const float4*>(sa + row * Cfg::LDA + ks * 8 + h * 4);
          a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) a[i][t] = sa[(ks * 8 + h * 4 + t) * Cfg::LDA + row];
        }
      }
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const int col = wn * Cfg::WN + j * 32 + r;
        if constexpr (B_KC) {
          const float4 v = *reinterpret_cast<const float4*>(sb + col * Cfg::LDB + ks * 8 + h * 4);
          b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) b[j][t] = sb[(ks * 8 + h * 4 + t) * Cfg::LDB + col];
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
          for (int j = 0; j < Cfg::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
    if (more) stash((kt + 1) & 1);
    __syncthreads();
  }
  if constexpr (KW > 1) {
    // add the partial accumulators of wave group 1 into wave group 0 through LDS (the staging buffers are free:
    // the loop ended with a barrier).  Layout [w4][element][lane] => conflict-free 4-byte accesses.
    static_assert(KW == 2, "intra-workgroup split-K is built for 2 wave groups");
    constexpr int NE = Cfg::TM * Cfg::TN * 16;
    static_assert(4 * NE * 64 <= 2 * Cfg::STAGE, "accumulator exchange does not fit in the staging LDS");
    float* ex = smem + (w4 * NE) * 64 + lane;
    if (wk == 1) {
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) ex[((i * Cfg::TN + j) * 16 + e) * 64] = acc[i][j][e];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += ex[((i * Cfg::TN + j) * 16 + e) * 64];
    }
  }
}

// XCD-aware bijective remap of a linear block id so that each of the 8 XCDs (block b runs on XCD b % 8)
// works on one contiguous chunk of the tile sequence (cdna_hip_programming.md T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int xcd = bid & 7, q = nblk >> 3, rem = nblk & 7;
  const int start = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  return start + (bid >> 3);
}

}  // namespace pxr
