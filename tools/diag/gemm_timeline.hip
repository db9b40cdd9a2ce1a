// Diagnostic (not part of libpxr.so): the 64x64 fp32-MFMA GEMM tile kernel with per-workgroup timestamps, to see where a
// launch's time goes at M = 3200 tokens: shader-clock cycles (s_memtime) AND constant-rate time (s_memrealtime, 100 MHz)
// at workgroup start / after the main loop / at the end, plus the XCD / CU the workgroup ran on.
//   => effective clock (cycles per wall microsecond), launch ramp (first start -> last start), per-CU tile counts
//      (grid quantisation), main-loop cycles per K tile, tail.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -I../../pixelrec_amd/csrc gemm_timeline.hip -o libgemm_timeline.so
#include "gemm_f32.cuh"

using namespace pxr;

struct Stamp {
  unsigned long long c0, c1, c2;   // shader cycles: start, after main loop, end
  unsigned long long r0, r2;       // realtime (100 MHz) at start / end
  unsigned int xcc, hwid, pad0, pad1;
};

template <int BM, int BN, bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256) timeline_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ B, int64_t ldb,
                                                       float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                       int tiles_m, int tiles_n, int n_fastest, Stamp* st) {
  using Cfg = GemmCfg<BM, BN, A_KC, B_KC>;
  __shared__ __attribute__((aligned(16))) float smem[2 * Cfg::STAGE];
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const int t = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int tm = n_fastest ? t / tiles_n : t % tiles_m, tn = n_fastest ? t % tiles_n : t / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  typename Cfg::Acc accs;
  gemm_mainloop<BM, BN, A_KC, B_KC, false, 1, 2, 2, 0>(accs, A, lda, B, ldb, M, N, 0, K, m0, n0, smem);
  const unsigned long long c1 = __builtin_readcyclecounter();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN, h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int col = n0 + wn * Cfg::WN + j * 32 + r;
    if (col >= N) continue;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * Cfg::WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < M) C[(int64_t)row * ldc + col] = accs.v[i][j][e];
      }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Stamp s;
    s.c0 = c0; s.c1 = c1; s.c2 = __builtin_readcyclecounter();
    s.r0 = r0; s.r2 = __builtin_amdgcn_s_memrealtime();
    unsigned int xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    s.xcc = xcc; s.hwid = hwid; s.pad0 = blockIdx.x; s.pad1 = (unsigned)t;
    st[blockIdx.x] = s;
  }
}

extern "C" int gemm_timeline(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda, const float* B,
                             int64_t ldb, float* C, int64_t ldc, int n_fastest, void* stamps, void* stream) {
  const int tiles_m = (M + 63) / 64, tiles_n = (N + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
  if (a_kc && b_kc)
    hipLaunchKernelGGL((timeline_kernel<64, 64, true, true>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, B, ldb, C,
                       ldc, M, N, K, tiles_m, tiles_n, n_fastest, (Stamp*)stamps);
  else if (a_kc && !b_kc)
    hipLaunchKernelGGL((timeline_kernel<64, 64, true, false>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, B, ldb,
                       C, ldc, M, N, K, tiles_m, tiles_n, n_fastest, (Stamp*)stamps);
  else
    hipLaunchKernelGGL((timeline_kernel<64, 64, false, false>), dim3(tiles_m * tiles_n), dim3(256), 0, st, A, lda, B, ldb,
                       C, ldc, M, N, K, tiles_m, tiles_n, n_fastest, (Stamp*)stamps);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
