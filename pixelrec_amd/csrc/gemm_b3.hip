// gemm_b3.hip -- fp32 GEMM on the bf16 matrix pipe through an exact 3 x bf16 operand split (gemm_b3.cuh), and the
// kernel that prepares the pre-split (optionally transposed) operand planes.
//
// Reference call sites replaced: the same nn.Linear forwards / input gradients as gemm_f32.hip (layers.py:586-588,613,
// 666,669 and their autograd transposes) and the full-catalog scoring product (sasrec.py:112).
#include "gemm_b3.cuh"

#include <cstdlib>

namespace pxr {

template <int BM, int BN, int FINE, int EPI, int KW = 1, int HINT = 0>
__global__ void __launch_bounds__((B3Cfg<BM, BN, FINE, KW>::NT))
gemm_b3_kernel(const float* __restrict__ A, int64_t lda, const __bf16* __restrict__ Bp, int64_t ldb, int64_t bplane,
               float* __restrict__ C, int64_t ldc, int M, int N, int K, const float* __restrict__ bias,
               float* __restrict__ aux, int64_t ldaux, int tiles_m, int tiles_n, int n_fastest, int act) {
  using Cfg = B3Cfg<BM, BN, FINE, KW>;
  using F = typename Cfg::F;
  __shared__ __attribute__((aligned(16))) char smem[Cfg::LDS_BYTES];
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = n_fastest ? t / tiles_n : t % tiles_m, tn = n_fastest ? t % tiles_n : t / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const LanePos lp = lane_pos<F>();
  const bool second_group = KW > 1 && (int)(threadIdx.x >> 6) >= F::G;
  AuxRegs<F, EPI> ar;
  if (!second_group) epi_prefetch_aux<F, EPI>(ar, aux, ldaux, M, N, m0, n0, lp);
  typename F::Acc accs;
  gemm_b3_mainloop<BM, BN, FINE, KW, HINT>(accs, A, lda, Bp, ldb, bplane, M, N, 0, K, m0, n0, smem);
  if (second_group) return;   // handed its partial sums over in the main loop
  epi_store<F, EPI>(accs, ar, C, ldc, M, N, bias, aux, ldaux, m0, n0, lp, act);
}

// ---- operand planes ------------------------------------------------------------------------------------------------------
// dst planes [3][R][ldd] (bf16; ldd >= C rounded up to 8, pad columns written as zeros) of the matrix
//   transpose == 0:  dst[r][k] = src[r * ld + k]     (R x C = the stored matrix)
//   transpose == 1:  dst[r][k] = src[k * ld + r]     (R x C = its transpose)
// Up to 16 matrices per launch (all weights of the sequence block, both orientations).
constexpr int SPLIT_MAX = 32;
struct SplitProblem {
  const float* src; __bf16* dst;
  int64_t ld, ldd, plane;
  int R, C, transpose;
  int block_begin;
};
struct SplitGroup {
  SplitProblem p[SPLIT_MAX];
  int n;
};

__global__ void __launch_bounds__(256) split_planes_kernel(SplitGroup g) {
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < g.n; ++i)
    if ((int)blockIdx.x >= g.p[i].block_begin) pi = i;
  const SplitProblem& P = g.p[pi];
  const int nch = (int)(P.ldd >> 3);
  const int64_t item = (int64_t)(blockIdx.x - P.block_begin) * 256 + threadIdx.x;
  if (item >= (int64_t)P.R * nch) return;
  int r, c;
  if (P.transpose) { c = (int)(item / P.R); r = (int)(item % P.R); }      // lanes run along r: coalesced column reads
  else { r = (int)(item / nch); c = (int)(item % nch); }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = c * 8 + j;
    v[j] = k < P.C ? (P.transpose ? P.src[(int64_t)k * P.ld + r] : P.src[(int64_t)r * P.ld + k]) : 0.f;
  }
  u32x4v ph, pm, pl;
  b3_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), ph, pm, pl);
  const int64_t o = (int64_t)r * P.ldd + c * 8;
  *reinterpret_cast<u32x4v*>(P.dst + o) = ph;
  *reinterpret_cast<u32x4v*>(P.dst + P.plane + o) = pm;
  *reinterpret_cast<u32x4v*>(P.dst + 2 * P.plane + o) = pl;
}

template <int BM, int BN, int FINE, int EPI, int KW = 1, int HINT = 0>
static int launch_b3(const float* A, int64_t lda, const __bf16* Bp, int64_t ldb, int64_t bplane, float* C, int64_t ldc, int M,
                     int N, int K, const float* bias, float* aux, int64_t ldaux, int act, hipStream_t st) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  static const int xcd_env = getenv("PXR_GEMM_XCD") ? atoi(getenv("PXR_GEMM_XCD")) : -1;
  const int n_fastest = xcd_env >= 0 ? xcd_env : (M > N ? 1 : 0);
  hipLaunchKernelGGL((gemm_b3_kernel<BM, BN, FINE, EPI, KW, HINT>), dim3(tiles_m * tiles_n), dim3(B3Cfg<BM, BN, FINE, KW>::NT), 0, st, A, lda, Bp,
                     ldb, bplane, C, ldc, M, N, K, bias, aux, ldaux, tiles_m, tiles_n, n_fastest, act);
  return pxr_check_launch("pxr_gemm_b3_f32");
}

template <int EPI>
static int dispatch_b3(int tile, const float* A, int64_t lda, const __bf16* Bp, int64_t ldb, int64_t bplane, float* C,
                       int64_t ldc, int M, int N, int K, const float* bias, float* aux, int64_t ldaux, int act,
                       hipStream_t st) {
  if (tile == 1281) return launch_b3<128, 128, 1, EPI>(A, lda, Bp, ldb, bplane, C, ldc, M, N, K, bias, aux, ldaux, act, st);
  if (tile == 642) return launch_b3<64, 64, 0, EPI, 2>(A, lda, Bp, ldb, bplane, C, ldc, M, N, K, bias, aux, ldaux, act, st);
  if constexpr (EPI == EPI_BIAS) {      // experimental issue-order variants (sweeps only)
    if (tile == 641) return launch_b3<64, 64, 0, EPI, 1, 1>(A, lda, Bp, ldb, bplane, C, ldc, M, N, K, bias, aux, ldaux, act, st);
    if (tile == 643) return launch_b3<64, 64, 0, EPI, 2, 1>(A, lda, Bp, ldb, bplane, C, ldc, M, N, K, bias, aux, ldaux, act, st);
    if (tile == 12811) return launch_b3<128, 128, 1, EPI, 1, 1>(A, lda, Bp, ldb, bplane, C, ldc, M, N, K, bias, aux, ldaux, act, st);
  }
  return launch_b3<64, 64, 0, EPI>(A, lda, Bp, ldb, bplane, C, ldc, M, N, K, bias, aux, ldaux, act, st);
}

}  // namespace pxr

using namespace pxr;

// Planes of up to 32 matrices in one launch.  dst[i] holds 3 planes of R[i] x ldd[i] bf16, plane[i] elements apart.
extern "C" int pxr_split_bf16x3_f32(int n, const float* const* src, const int64_t* ld, const int* R, const int* Ccols,
                                    const int* transpose, void* const* dst, const int64_t* ldd, const int64_t* plane,
                                    void* stream) {
  PXR_REQUIRE(n >= 1 && n <= SPLIT_MAX && src && ld && R && Ccols && transpose && dst && ldd && plane,
              "pxr_split_bf16x3_f32: bad args (n=%d, max %d)", n, SPLIT_MAX);
  SplitGroup g{};
  g.n = n;
  int64_t blocks = 0;
  for (int i = 0; i < n; ++i) {
    PXR_REQUIRE(src[i] && dst[i] && R[i] > 0 && Ccols[i] > 0 && ldd[i] % 8 == 0 && ldd[i] >= ((Ccols[i] + 7) & ~7) &&
                    plane[i] >= (int64_t)R[i] * ldd[i] && plane[i] % 8 == 0 && ((uintptr_t)dst[i] & 15) == 0,
                "pxr_split_bf16x3_f32: matrix %d has a bad shape / stride / alignment", i);
    SplitProblem& P = g.p[i];
    P.src = src[i]; P.dst = (__bf16*)dst[i]; P.ld = ld[i]; P.ldd = ldd[i]; P.plane = plane[i];
    P.R = R[i]; P.C = Ccols[i]; P.transpose = transpose[i];
    P.block_begin = (int)blocks;
    blocks += ((int64_t)R[i] * (ldd[i] >> 3) + 255) / 256;
    PXR_REQUIRE(blocks < (1ll << 31), "pxr_split_bf16x3_f32: too many blocks");
  }
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g);
  return pxr_check_launch("pxr_split_bf16x3_f32");
}

// C[M,N] = epilogue(A[M,K] * B^T) with A fp32 row-major and B given as bf16x3 planes [3][N][ldb] (pxr_split_bf16x3_f32).
// Epilogues as pxr_gemm_f32 (all of them: the forward ones on weight planes, ADD / MUL / MUL_DGELU on planes of the
// TRANSPOSED weight for the input gradients); act = activation code of EPI_BIAS_ACT_GRAD.
// tile_hint: 0 heuristic, 64 / 1281 force the 64x64 (4 waves) / 128x128 (16 waves) tile.
extern "C" int pxr_gemm_b3_f32(int M, int N, int K, const float* A, int64_t lda, const void* Bp, int64_t ldb, int64_t bplane,
                               float* C, int64_t ldc, int epilogue, int act, const float* bias, float* aux, int64_t ldaux,
                               int tile_hint, void* stream) {
  PXR_REQUIRE(A && Bp && C, "pxr_gemm_b3_f32: null operand");
  PXR_REQUIRE(M >= 0 && N >= 0 && K >= 0, "pxr_gemm_b3_f32: negative dim");
  if (M == 0 || N == 0) return PXR_OK;
  PXR_REQUIRE((lda % 4) == 0 && (K % 4) == 0 && (((uintptr_t)A) & 15) == 0, "pxr_gemm_b3_f32: A must be 16-byte aligned with K, lda multiples of 4");
  PXR_REQUIRE((ldb % 8) == 0 && ldb >= ((K + 7) & ~7) && (bplane % 8) == 0 && (((uintptr_t)Bp) & 15) == 0,
              "pxr_gemm_b3_f32: planes need ldb %% 8 == 0, ldb >= K rounded up to 8, 16-byte alignment");
  PXR_REQUIRE(epilogue >= 0 && epilogue <= EPI_LAST, "pxr_gemm_b3_f32: bad epilogue %d", epilogue);
  PXR_REQUIRE(!(epilogue == EPI_BIAS || epilogue == EPI_BIAS_GELU || epilogue == EPI_BIAS_GELU_GRAD || epilogue == EPI_BIAS_ADD ||
                epilogue == EPI_BIAS_QGELU_GRAD || epilogue == EPI_BIAS_RELU || epilogue == EPI_BIAS_ACT_GRAD) || bias,
              "pxr_gemm_b3_f32: epilogue needs bias");
  PXR_REQUIRE(!(epilogue >= EPI_BIAS_GELU && epilogue != EPI_BIAS_RELU) || aux, "pxr_gemm_b3_f32: epilogue needs aux");
  hipStream_t st = (hipStream_t)stream;
  const int64_t t128 = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
  int tile = (t128 >= 384 || (t128 >= 192 && t128 <= 256)) ? 1281 : 64;
  if (tile_hint == 64 || tile_hint == 1281 || tile_hint == 641 || tile_hint == 642 || tile_hint == 643 || tile_hint == 12811) tile = tile_hint;
  const float* bA = A;
  switch (epilogue) {
#define PXR_B3_CASE(E) \
  case E: return dispatch_b3<E>(tile, bA, lda, (const __bf16*)Bp, ldb, bplane, C, ldc, M, N, K, bias, aux, ldaux, act, st)
    PXR_B3_CASE(EPI_NONE);
    PXR_B3_CASE(EPI_BIAS);
    PXR_B3_CASE(EPI_BIAS_GELU);
    PXR_B3_CASE(EPI_MUL_DGELU);
    PXR_B3_CASE(EPI_ADD);
    PXR_B3_CASE(EPI_BIAS_GELU_GRAD);
    PXR_B3_CASE(EPI_MUL);
    PXR_B3_CASE(EPI_BIAS_ADD);
    PXR_B3_CASE(EPI_BIAS_QGELU_GRAD);
    PXR_B3_CASE(EPI_BIAS_RELU);
    PXR_B3_CASE(EPI_BIAS_ACT_GRAD);
#undef PXR_B3_CASE
  }
  pxr_set_error("pxr_gemm_b3_f32: epilogue %d", epilogue);
  return PXR_ERR_BAD_ARG;
}
