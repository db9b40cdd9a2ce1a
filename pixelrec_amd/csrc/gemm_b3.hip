// gemm_b3.hip -- fp32 GEMM kernels on the bf16 matrix pipe through the exact 3 x bf16 operand split (gemm_b3.cuh):
// the counterparts of gemm_kernel / grouped_dw_kernel of gemm_f32.hip, reached through pxr_gemm_f32 /
// pxr_grouped_linear_bwd_weight_f32 when the GEMM mode is "bf16x3" (the default; PXR_GEMM_MODE=f32 selects the
// f32-input MFMA kernels).
//
// Reference call sites replaced: the same as gemm_f32.hip (layers.py:586-588,613,666,669, sasrec.py:112, and the
// autograd transposes of all of them).
#include "gemm_b3.cuh"

#include <cstdlib>

namespace pxr {

template <int BM, int BN, bool A_KC, bool B_KC, int FINE, int EPI, int STAGES = 2>
__global__ void __launch_bounds__((B3Cfg<BM, BN, FINE>::NT))
gemm_b3_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
               float* __restrict__ C, int64_t ldc, int M, int N, int K, const float* __restrict__ bias,
               float* __restrict__ aux, int64_t ldaux, int tiles_m, int tiles_n, int ksplit_len,
               int64_t split_stride, int n_fastest, GemmBatch bt) {
  using Cfg = B3Cfg<BM, BN, FINE, STAGES>;
  using F = typename Cfg::F;
  __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];
  if (bt.nb2 > 0) {   // batched launch (block-uniform)
    const int z1 = blockIdx.z / bt.nb2, z2 = blockIdx.z % bt.nb2;
    A += z1 * bt.a1 + z2 * bt.a2;
    B += z1 * bt.b1 + z2 * bt.b2;
    C += z1 * bt.c1 + z2 * bt.c2;
  }
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);      // tile order: see gemm_kernel
  const int tm = n_fastest ? t / tiles_n : t % tiles_m, tn = n_fastest ? t % tiles_n : t / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = blockIdx.y * ksplit_len;
  const int kend = min(K, kbeg + ksplit_len);
  C += (int64_t)blockIdx.y * split_stride;
  const LanePos lp = lane_pos<F>();
  constexpr bool DIRECT = (Cfg::NT == 1024);     // 16-wave tiles: 128 VGPRs per lane, none to hold `aux` across the loop
  AuxRegs<F, EPI, DIRECT> ar;
  epi_prefetch_aux<F, EPI, DIRECT>(ar, aux, ldaux, M, N, m0, n0, lp);
  typename F::Acc accs;
  gemm_b3_mainloop<BM, BN, A_KC, B_KC, FINE, false, STAGES>(accs, A, lda, B, ldb, M, N, kbeg, kend, m0, n0, smem);
  epi_store<F, EPI, DIRECT>(accs, ar, C, ldc, M, N, bias, aux, ldaux, m0, n0, lp, bt.act);
}

// All dW[N,K] = dY[M,N]^T X[M,K] (+ db[N] = column sums of dY) of a backward pass in ONE launch: see grouped_dw_kernel.
template <int STAGES>
__global__ void __launch_bounds__(GEMM_THREADS) grouped_dw_b3_kernel(DwGroup g) {
  using Cfg = B3Cfg<64, 64, 0, STAGES>;
  using F = typename Cfg::F;
  __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];
  const int t = xcd_remap(blockIdx.x, g.total_tiles);
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < g.n; ++i)
    if (t >= g.p[i].tile_begin) pi = i;
  const DwProblem& P = g.p[pi];
  const int local = t - P.tile_begin;
  const int tm = local % P.tiles_m, tn = local / P.tiles_m;
  const int m0 = tm * 64, n0 = tn * 64;
  typename F::Acc accs;
  float cs[1] = {0.f};
  const bool do_bias = (P.db != nullptr) && (tn == 0);
  if (do_bias)
    gemm_b3_mainloop<64, 64, false, false, 0, true, STAGES>(accs, P.dy, P.N, P.x, P.K, P.N, P.K, 0, P.M, m0, n0, smem, cs);
  else
    gemm_b3_mainloop<64, 64, false, false, 0, false, STAGES>(accs, P.dy, P.N, P.x, P.K, P.N, P.K, 0, P.M, m0, n0, smem);
  auto& acc = accs.v[0][0];
  const LanePos lp = lane_pos<F>();
  const int col = n0 + lp.wn * 32 + lp.r;
  if (col < P.K) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + lp.wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lp.h;
      if (row < P.N) P.dW[(int64_t)row * P.K + col] = acc[e];
    }
  }
  if (do_bias) {
    // thread tid staged column (tid % 64) of this 64-column block for the k rows of chunk tid / 64: add the 4 chunk
    // owners in fixed order (the main loop ended with a barrier, smem is free)
    float* red = reinterpret_cast<float*>(smem);
    red[threadIdx.x] = cs[0];
    __syncthreads();
    if (threadIdx.x < 64) {
      const float s = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
      const int c = m0 + threadIdx.x;
      if (c < P.N) P.db[c] = s;
    }
  }
}

// The same with 128x128 tiles (16 waves): the two layers of the step at D = 512 give exactly 256 tiles -- one per CU, four
// waves per SIMD -- and the tile runs the long token reduction 33 % faster than the 64x64 one (tools/diag/dw_tile_test.py:
// 180 vs 133 TFLOP/s at 256 tiles).  `tile_begin` / `tiles_m` of the group are in 128-tiles here.  Every tile takes the
// column sums (8 adds per staged chunk); the tn == 0 tiles store them: one copy of the main loop in the kernel.
__global__ void __launch_bounds__(1024) grouped_dw_b3_kernel128(DwGroup g) {
  using Cfg = B3Cfg<128, 128, 1>;
  using F = typename Cfg::F;
  __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];
  const int t = xcd_remap(blockIdx.x, g.total_tiles);
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < g.n; ++i)
    if (t >= g.p[i].tile_begin) pi = i;
  // block-uniform problem fields, pinned into scalar registers (the kernel has 128 VGPRs per lane)
  const DwProblem& P = g.p[pi];
  const float* dy = P.dy;
  const float* x = P.x;
  float* dW = P.dW;
  float* db = P.db;
  const int PM = __builtin_amdgcn_readfirstlane(P.M), PN = __builtin_amdgcn_readfirstlane(P.N);
  const int PK = __builtin_amdgcn_readfirstlane(P.K), ptm = __builtin_amdgcn_readfirstlane(P.tiles_m);
  const int local = t - __builtin_amdgcn_readfirstlane(P.tile_begin);
  const int tm = local % ptm, tn = local / ptm;
  const int m0 = tm * 128, n0 = tn * 128;
  typename F::Acc accs;
  float cs[1] = {0.f};
  gemm_b3_mainloop<128, 128, false, false, 1, true>(accs, dy, PN, x, PK, PN, PK, 0, PM, m0, n0, smem, cs);
  auto& acc = accs.v[0][0];
  const LanePos lp = lane_pos<F>();
  const int col = n0 + lp.wn * 32 + lp.r;
  if (col < PK) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + lp.wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * lp.h;
      if (row < PN) dW[(int64_t)row * PK + col] = acc[e];
    }
  }
  if (db != nullptr && tn == 0) {
    // threads 0..511 staged A: thread tid held column tid % 128 for the k rows of chunk tid / 128 (fixed-order add)
    float* red = reinterpret_cast<float*>(smem);
    if (threadIdx.x < 512) red[threadIdx.x] = cs[0];
    __syncthreads();
    if (threadIdx.x < 128) {
      const float s = (red[threadIdx.x] + red[threadIdx.x + 128]) + (red[threadIdx.x + 256] + red[threadIdx.x + 384]);
      const int c = m0 + threadIdx.x;
      if (c < PN) db[c] = s;
    }
  }
}

static int b3_stages() {
  static const int st = getenv("PXR_B3_STAGES") ? atoi(getenv("PXR_B3_STAGES")) : 2;    // LDS stages of the 64x64 tile
  return st == 3 ? 3 : 2;
}

template <int BM, int BN, bool A_KC, bool B_KC, int FINE, int EPI, int STAGES = 2>
static int launch_b3(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
                     const float* bias, float* aux, int64_t ldaux, int splits, int ksplit_len, int64_t split_stride,
                     const GemmBatch& bt, int batch, hipStream_t st) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  static const int xcd_env = getenv("PXR_GEMM_XCD") ? atoi(getenv("PXR_GEMM_XCD")) : -1;
  const int n_fastest = xcd_env >= 0 ? xcd_env : (M > N ? 1 : 0);
  hipLaunchKernelGGL((gemm_b3_kernel<BM, BN, A_KC, B_KC, FINE, EPI, STAGES>), dim3(tiles_m * tiles_n, splits, batch),
                     dim3(B3Cfg<BM, BN, FINE>::NT), 0, st, A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, tiles_m, tiles_n,
                     ksplit_len, split_stride, n_fastest, bt);
  return pxr_check_launch("pxr_gemm_f32(bf16x3)");
}

template <bool A_KC, bool B_KC, int EPI>
static int tile_b3(int tile, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N,
                   int K, const float* bias, float* aux, int64_t ldaux, int splits, int ksplit_len, int64_t split_stride,
                   const GemmBatch& bt, int batch, hipStream_t st) {
  if (tile == 1281)
    return launch_b3<128, 128, A_KC, B_KC, 1, EPI>(A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, splits, ksplit_len,
                                                   split_stride, bt, batch, st);
  if (b3_stages() == 3)
    return launch_b3<64, 64, A_KC, B_KC, 0, EPI, 3>(A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, splits, ksplit_len,
                                                    split_stride, bt, batch, st);
  return launch_b3<64, 64, A_KC, B_KC, 0, EPI>(A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, splits, ksplit_len,
                                               split_stride, bt, batch, st);
}

int gemm_b3_launch(int a_kc, int b_kc, int epilogue, int tile, const float* A, int64_t lda, const float* B, int64_t ldb,
                   float* C, int64_t ldc, int M, int N, int K, const float* bias, float* aux, int64_t ldaux, int splits,
                   int ksplit_len, int64_t split_stride, const GemmBatch& bt, int batch, hipStream_t st) {
#define PXR_B3(AK, BK_, E)                                                                                            \
  case E: return tile_b3<AK, BK_, E>(tile, A, lda, B, ldb, C, ldc, M, N, K, bias, aux, ldaux, splits, ksplit_len,     \
                                     split_stride, bt, batch, st)
  if (a_kc && b_kc) {
    switch (epilogue) {
      PXR_B3(true, true, EPI_NONE); PXR_B3(true, true, EPI_BIAS); PXR_B3(true, true, EPI_BIAS_GELU);
      PXR_B3(true, true, EPI_BIAS_GELU_GRAD); PXR_B3(true, true, EPI_BIAS_ADD); PXR_B3(true, true, EPI_BIAS_QGELU_GRAD);
      PXR_B3(true, true, EPI_BIAS_RELU); PXR_B3(true, true, EPI_BIAS_ACT_GRAD);
    }
  } else if (a_kc && !b_kc) {
    switch (epilogue) {
      PXR_B3(true, false, EPI_NONE); PXR_B3(true, false, EPI_MUL_DGELU); PXR_B3(true, false, EPI_ADD);
      PXR_B3(true, false, EPI_MUL);
    }
  } else if (!a_kc && !b_kc) {
    switch (epilogue) { PXR_B3(false, false, EPI_NONE); }
  }
#undef PXR_B3
  pxr_set_error("pxr_gemm_f32(bf16x3): operand flavour (%d,%d) / epilogue %d is not instantiated", a_kc, b_kc, epilogue);
  return PXR_ERR_BAD_ARG;
}

int grouped_dw_b3_launch(const DwGroup& g, hipStream_t st) {
  // 128x128 tiles when they fill (most of) the chip at least once
  int64_t t128 = 0;
  DwGroup g2 = g;
  for (int i = 0; i < g.n; ++i) {
    g2.p[i].tile_begin = (int)t128;
    g2.p[i].tiles_m = (g.p[i].N + 127) / 128;
    t128 += (int64_t)g2.p[i].tiles_m * ((g.p[i].K + 127) / 128);
  }
  static const int dw128 = getenv("PXR_B3_DW128") ? atoi(getenv("PXR_B3_DW128")) : 1;
  if (dw128 && t128 >= 192 && t128 < (1ll << 30)) {
    g2.total_tiles = (int)t128;
    hipLaunchKernelGGL(grouped_dw_b3_kernel128, dim3((unsigned)t128), dim3(1024), 0, st, g2);
    return pxr_check_launch("pxr_grouped_linear_bwd_weight_f32(bf16x3, 128x128)");
  }
  if (b3_stages() == 3) hipLaunchKernelGGL(grouped_dw_b3_kernel<3>, dim3(g.total_tiles), dim3(GEMM_THREADS), 0, st, g);
  else hipLaunchKernelGGL(grouped_dw_b3_kernel<2>, dim3(g.total_tiles), dim3(GEMM_THREADS), 0, st, g);
  return pxr_check_launch("pxr_grouped_linear_bwd_weight_f32(bf16x3)");
}

}  // namespace pxr
