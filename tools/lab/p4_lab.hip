// p4_lab.hip -- standalone A/B harness for the big-tile planes GEMM main loops (gemm_p3.cuh lockstep vs gemm_p4.cuh ping-pong).
// Not part of the product: builds to an executable (tools/lab/build.sh), runs a matrix of (variant, shape, dbg) on one GPU and
// prints TF/s (six bf16 products per fp32 multiply counted) + the difference from the gemm_p3 result on the same operands.
//   usage: p4_lab [iters] [shape-substring] [h2]      (h2: the fp16 two-plane experiment)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "gemm_w4.cuh"

void pxr_set_error(const char*, ...) {}
int32_t* pxr_status_word(void) { return nullptr; }

using namespace pxr;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__global__ void fill_normal(float* x, int64_t n, uint32_t seed, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t a = pxr_hash32(seed, 1, (uint64_t)i), b = pxr_hash32(seed, 2, (uint64_t)i);
  const float u1 = ((a >> 8) + 1) * (1.0f / 16777217.0f), u2 = (b >> 8) * (1.0f / 16777216.0f);
  x[i] = scale * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
}
__global__ void split_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols8, P3Mat out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols8) return;
  const int64_t row = i / cols8;
  const int c = (int)(i % cols8) * 8;
  const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c);
  const float4 b = *reinterpret_cast<const float4*>(x + row * ldx + c + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  p3_store8(out, row, c, v);
}
// X[R][C] -> X^T planes (rows = C, cols = R): the x-contiguous storage of an operand
__global__ void split_t_kernel(const float* __restrict__ x, int64_t R, int64_t C, P3Mat out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;       // over C x (R / 8)
  const int64_t r8 = R / 8;
  if (i >= C * r8) return;
  const int64_t c = i / r8;
  const int r0 = (int)(i % r8) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = x[(int64_t)(r0 + e) * C + c];
  p3_store8(out, c, r0, v);
}
__global__ void diff_kernel(const float* a, const float* b, int64_t n, float* out /* [0] max |a-b|, [1] max |a| */) {
  float d = 0.f, m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    d = fmaxf(d, fabsf(a[i] - b[i]));
    m = fmaxf(m, fabsf(a[i]));
  }
  d = wave_max(d); m = wave_max(m);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(d));
    atomicMax(reinterpret_cast<int*>(out) + 1, __float_as_int(m));
  }
}

__global__ void mismatch_kernel(const float* a, const float* b, int64_t n, unsigned long long* out /* [0] count, [1] first index */) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (a[i] != b[i]) {
      atomicAdd(out, 1ull);
      atomicMin(out + 1, (unsigned long long)i);
    }
}


// ---- fp16 two-plane ("h2") experiment: split with a power-of-two scale, sampled fp64 check ------------------------------------
__global__ void split_h2_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols8, P3Mat out, float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols8) return;
  const int64_t row = i / cols8;
  const int c = (int)(i % cols8) * 8;
  const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c);
  const float4 b = *reinterpret_cast<const float4*>(x + row * ldx + c + 4);
  const float v[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale, b.x * scale, b.y * scale, b.z * scale, b.w * scale};
  h2_store8(out, row, c, v);
}
// sample s -> element (i, j) of C; out[0] += err^2, out[1] += ref^2, out[2] = max err (as double bits via atomicMax on u64)
__global__ void sample_err_kernel(const float* A, const float* B, const float* C, int M, int N, int K, double cscale, int ns, double* out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  const int i = (int)(pxr_hash32(77u, 1, (uint64_t)s) % (uint32_t)M), j = (int)(pxr_hash32(77u, 2, (uint64_t)s) % (uint32_t)N);
  double r = 0.0;
  for (int k = 0; k < K; ++k) r += (double)A[(int64_t)i * K + k] * (double)B[(int64_t)j * K + k];
  const double e = fabs((double)C[(int64_t)i * N + j] * cscale - r);
  atomicAdd(out, e * e);
  atomicAdd(out + 1, r * r);
  atomicMax(reinterpret_cast<unsigned long long*>(out + 2), (unsigned long long)__double_as_longlong(e));
}

// Cfg::W4 exists only on the lab's own configurations (gemm_w4.cuh)
template <class C, class = void>
struct lab_is_w4 : std::false_type {};
template <class C>
struct lab_is_w4<C, std::void_t<decltype(C::W4)>> : std::true_type {};

struct LabArgs {
  P3Mat A, B;
  float* C;
  int M, N, K, tiles_m, tiles_n, dbg;
  unsigned long long* clk;     // [0] shader cycles, [1] 100 MHz ticks spent by workgroup 0 (effective clock under this kernel's load)
};

template <class Cfg, bool A_KC, bool B_KC, int DBG>
__device__ __forceinline__ void lab_mainloop(typename Cfg::Acc& accs, const LabArgs& g, int m0, int n0, char* smem) {
  if constexpr (lab_is_w4<Cfg>::value) gemm_w4_mainloop<Cfg, A_KC, B_KC>(accs, g.A, g.B, g.K, m0, n0, smem);
  else if constexpr (Cfg::PINGPONG) gemm_p4_mainloop<Cfg, A_KC, B_KC, false, (DBG & 31)>(accs, g.A, g.B, g.K, m0, n0, smem, nullptr);
  else gemm_p3_mainloop<Cfg, A_KC, B_KC, false>(accs, g.A, g.B, g.K, m0, n0, smem, nullptr, DBG & 31);
}

template <class Cfg, bool A_KC, bool B_KC, int DBG>
__global__ void __launch_bounds__(Cfg::NT) lab_kernel(const LabArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  const int t = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const bool n_fastest = g.M > g.N;
  const int tm = n_fastest ? t / g.tiles_n : t % g.tiles_m, tn = n_fastest ? t % g.tiles_n : t / g.tiles_m;
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
  typename Cfg::Acc accs;
  lab_mainloop<Cfg, A_KC, B_KC, DBG>(accs, g, m0, n0, smem);
  constexpr bool nostore = (DBG & 32) != 0;
  if constexpr (!lab_is_w4<Cfg>::value && Cfg::BM * Cfg::EPI_LD * 4 <= Cfg::LDS_BYTES && (!Cfg::PINGPONG || Cfg::BN <= 128)) {
    p3_row_epilogue<Cfg>(accs, smem, g.M, g.N, m0, n0, [&](int, int row, int col, int nv, float (&v)[8]) {
      if (nostore && v[0] != 1.2345e38f) return;
      float* cp = g.C + (int64_t)row * g.N + col;
      if (nv == 8) {
        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        for (int e = 0; e < nv; ++e) cp[e] = v[e];
      }
    });
  } else {
    // direct stores from the accumulators through a buffer descriptor (rows beyond M are dropped by the bounds check; N % BN == 0)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN, h = lane >> 5, r = lane & 31;
    const bufrsrc rc = make_rsrc(g.C, (int64_t)g.M * g.N * 4);
    const unsigned voff = (unsigned)(((int64_t)(m0 + wm * Cfg::WM + 4 * h) * g.N + n0 + wn * Cfg::WN + r) * 4);
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (nostore && accs.v[i][j][e] != 1.2345e38f) continue;
          if constexpr ((DBG & 64) != 0) {
            const int row = m0 + wm * Cfg::WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, col = n0 + wn * Cfg::WN + j * 32 + r;
            if (row < g.M && col < g.N) g.C[(int64_t)row * g.N + col] = accs.v[i][j][e];
            continue;
          }
          const unsigned so = (unsigned)((i * 32 + (e & 3) + 8 * (e >> 2)) * g.N + j * 32) * 4u;
          const float val = accs.v[i][j][e];      // (bit_cast of a vector ELEMENT lvalue reads element 0: clang quirk)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rc, voff, so, 0);
        }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && g.clk) {
    g.clk[0] = __builtin_readcyclecounter() - c0;
    g.clk[1] = wall_clock64() - r0;
  }
}

struct Planes {
  __bf16* p = nullptr;
  P3Mat m{};
};
static Planes make_planes(const float* x, int64_t R, int64_t C, bool transposed) {
  // planes of X (rows R, cols C) or of X^T
  const int64_t rows = transposed ? C : R, cols = transposed ? R : C;
  const int64_t pr = (rows + 31) / 32 * 32, ps = pr * cols;
  Planes P;
  CK(hipMalloc(&P.p, (size_t)ps * 3 * 2 + 4096));
  CK(hipMemset(P.p, 0, (size_t)ps * 3 * 2 + 4096));
  P.m = P3Mat{P.p, ps, pr};
  if (!transposed) {
    const int64_t n = R * (C / 8);
    hipLaunchKernelGGL(split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, C, R, (int)(C / 8), P.m);
  } else {
    const int64_t n = C * (R / 8);
    hipLaunchKernelGGL(split_t_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, R, C, P.m);
  }
  CK(hipDeviceSynchronize());
  return P;
}


static Planes make_planes_h2(const float* x, int64_t R, int64_t C, float scale) {
  const int64_t pr = (R + 31) / 32 * 32, ps = pr * C;
  Planes P;
  CK(hipMalloc(&P.p, (size_t)ps * 2 * 2 + 4096));
  CK(hipMemset(P.p, 0, (size_t)ps * 2 * 2 + 4096));
  P.m = P3Mat{P.p, ps, pr};
  const int64_t n = R * (C / 8);
  hipLaunchKernelGGL(split_h2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, x, C, R, (int)(C / 8), P.m, scale);
  CK(hipDeviceSynchronize());
  return P;
}
static void sample_err(const char* what, const float* A, const float* B, const float* C, int M, int N, int K, double cscale) {
  double* d;
  CK(hipMalloc(&d, 24));
  CK(hipMemset(d, 0, 24));
  const int ns = 16384;
  hipLaunchKernelGGL(sample_err_kernel, dim3(ns / 256), dim3(256), 0, 0, A, B, C, M, N, K, cscale, ns, d);
  double h[3];
  CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
  printf("    %-30s vs fp64 on %d samples: rms err %.3e  max err %.3e  (rms |c| %.3e; rms err / rms c = 2^%.1f)\n", what, ns, sqrt(h[0] / ns), h[2],
         sqrt(h[1] / ns), log2(sqrt(h[0] / h[1])));
  CK(hipFree(d));
}

template <class Cfg, bool A_KC, bool B_KC, int DBG = 0>
static float run(const char* name, LabArgs g, int iters, const float* ref, float* diffbuf, bool quiet = false) {
  g.tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM;
  g.tiles_n = (g.N + Cfg::BN - 1) / Cfg::BN;
  g.dbg = DBG;
  auto kern = lab_kernel<Cfg, A_KC, B_KC, DBG>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES));
  const dim3 grid((unsigned)(g.tiles_m * g.tiles_n)), blk(Cfg::NT);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, grid, blk, Cfg::LDS_BYTES, 0, g);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, blk, Cfg::LDS_BYTES, 0, g);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double tf6 = 12.0 * g.M * g.N * (double)g.K / us / 1e6;
  unsigned long long hc[2] = {0, 1};
  if (g.clk) CK(hipMemcpy(hc, g.clk, 16, hipMemcpyDeviceToHost));
  const double ghz = (double)hc[0] / ((double)hc[1] / 100e6) / 1e9;
  float hd[2] = {0.f, 0.f};
  if (ref && (g.dbg & 63) == 0) {
    CK(hipMemset(diffbuf, 0, 8));
    hipLaunchKernelGGL(diff_kernel, dim3(1024), dim3(256), 0, 0, (const float*)g.C, ref, (int64_t)g.M * g.N, diffbuf);
    CK(hipMemcpy(hd, diffbuf, 8, hipMemcpyDeviceToHost));
  }
  if (hd[0] != 0.f) {
    unsigned long long* mm;
    CK(hipMalloc(&mm, 16));
    unsigned long long init[2] = {0ull, ~0ull}, res[2];
    CK(hipMemcpy(mm, init, 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mismatch_kernel, dim3(1024), dim3(256), 0, 0, (const float*)g.C, ref, (int64_t)g.M * g.N, mm);
    CK(hipMemcpy(res, mm, 16, hipMemcpyDeviceToHost));
    float va = 0, vb = 0;
    CK(hipMemcpy(&va, g.C + res[1], 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&vb, ref + res[1], 4, hipMemcpyDeviceToHost));
    printf("    MISMATCH: %llu of %lld elements; first at row %llu col %llu: got %g want %g\n", res[0], (long long)g.M * g.N, res[1] / g.N,
           res[1] % g.N, va, vb);
    CK(hipFree(mm));
  }
  if (!quiet)
    printf("  %-34s dbg=%-2d %9.1f us  %7.1f TF(6p)  frac %.3f  clk %.2f GHz  maxdiff %.3g (max|c| %.3g)\n", name, g.dbg, us, tf6,
           tf6 / 2500.0, ghz, hd[0], hd[1]);
  fflush(stdout);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return (float)us;
}

#ifndef LAB_VARIANTS
#define LAB_VARIANTS 1
#endif

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10;
  const char* only = argc > 2 ? argv[2] : "";
  struct Shape { const char* name; int M, N, K; };
  const Shape shapes[] = {
      {"seq fc1 (B=64)", 3200, 1024, 512},
      {"seq qkv (B=64)", 3200, 1536, 512},
      {"seq out (B=64)", 3200, 512, 512},
      {"seq fc2 (B=64)", 3200, 512, 1024},
      {"scoring  items x users", 400128, 1024, 512},
      {"vit fc1", 69344, 3072, 768},
      {"vit fc2", 69344, 768, 3072},
      {"vit qkv", 69344, 2304, 768},
  };
  float* diffbuf;
  CK(hipMalloc(&diffbuf, 8));
  unsigned long long* clk;
  CK(hipMalloc(&clk, 16));

  if (argc > 3 && !strcmp(argv[3], "h2")) {
    // the fp16 two-plane product against the six-product bf16 one: time, clock, error against fp64
    struct H2Shape { const char* name; int M, N, K; float sa, sb; int ea, eb; };
    const H2Shape hs[] = {
        {"seq fc1 (B=64)", 3200, 1024, 512, 1.0f, 0.05f, 0, 8},
        {"seq qkv (B=64)", 3200, 1536, 512, 1.0f, 0.05f, 0, 8},
        {"seq out (B=64)", 3200, 512, 512, 1.0f, 0.05f, 0, 8},
        {"seq fc2 (B=64)", 3200, 512, 1024, 1.0f, 0.05f, 0, 8},
        {"vit fc1", 69344, 3072, 768, 1.0f, 0.05f, 0, 8},
        {"vit fc2", 69344, 768, 3072, 1.0f, 0.05f, 0, 8},
        {"vit qkv", 69344, 2304, 768, 1.0f, 0.05f, 0, 8},
        {"scoring  items x users", 400128, 1024, 512, 0.05f, 1.0f, 8, 0},
        {"tiny-b (1e-6 weights, unscaled: fp16 subnormals)", 8192, 1024, 512, 1.0f, 1e-6f, 0, 0},
        {"tiny-b scaled 2^16", 8192, 1024, 512, 1.0f, 1e-6f, 0, 16},
    };
    for (const H2Shape& s : hs) {
      if (only[0] && !strstr(s.name, only)) continue;
      printf("== %s  M=%d N=%d K=%d\n", s.name, s.M, s.N, s.K);
      float *A, *B, *C;
      CK(hipMalloc(&A, (size_t)s.M * s.K * 4));
      CK(hipMalloc(&B, (size_t)s.N * s.K * 4));
      CK(hipMalloc(&C, (size_t)s.M * s.N * 4));
      hipLaunchKernelGGL(fill_normal, dim3((unsigned)(((int64_t)s.M * s.K + 255) / 256)), dim3(256), 0, 0, A, (int64_t)s.M * s.K, 11u, s.sa);
      hipLaunchKernelGGL(fill_normal, dim3((unsigned)(((int64_t)s.N * s.K + 255) / 256)), dim3(256), 0, 0, B, (int64_t)s.N * s.K, 12u, s.sb);
      CK(hipDeviceSynchronize());
      LabArgs g{};
      g.M = s.M; g.N = s.N; g.K = s.K; g.clk = clk; g.C = C;
      {
        Planes Ap = make_planes(A, s.M, s.K, false), Bp = make_planes(B, s.N, s.K, false);
        g.A = Ap.m; g.B = Bp.m;
        run<P3Cfg<256, 128, 4, 2, 2>, true, true>("bf16x3 p3 256x128 lockstep", g, iters, nullptr, diffbuf);
        sample_err("bf16x3 six products (3 sets)", A, B, C, s.M, s.N, s.K, 1.0);
        if (s.M <= 4096) {     // the product's lockstep tiles at 3 200 tokens
          run<P3Cfg<128, 64, 2, 2, 2>, true, true>("bf16x3 p3 128x64 s2 (product, N >= 1024)", g, iters, nullptr, diffbuf);
          run<P3Cfg<64, 64, 2, 2, 3>, true, true>("bf16x3 p3 64x64 s3 (product, N < 1024)", g, iters, nullptr, diffbuf);
          sample_err("bf16x3 six products (64x64)", A, B, C, s.M, s.N, s.K, 1.0);
        }
        run<P4Cfg<256, 128, 4, 2, 3, 3, 0>, true, true>("bf16x3 p4 256x128 acc3", g, iters, nullptr, diffbuf);
        if (s.N % 256 == 0) {
          run<P4Cfg<256, 256, 4, 2, 3, 1, 0, 3>, true, true, 64>("bf16x3 p4 256x256 acc1", g, iters, nullptr, diffbuf);
          sample_err("bf16x3 six products (1 set)", A, B, C, s.M, s.N, s.K, 1.0);
        }
        CK(hipFree(Ap.p)); CK(hipFree(Bp.p));
      }
      {
        Planes Ap = make_planes_h2(A, s.M, s.K, ldexpf(1.0f, s.ea)), Bp = make_planes_h2(B, s.N, s.K, ldexpf(1.0f, s.eb));
        g.A = Ap.m; g.B = Bp.m;
        const double cs = ldexp(1.0, -(s.ea + s.eb));
        if (s.M <= 4096) {     // the same lockstep tiles on two fp16 planes (gemm_p3.cuh P3Cfg<..., HALF>)
          run<P3Cfg<128, 64, 2, 2, 2, true>, true, true>("h2 p3 128x64 s2", g, iters, nullptr, diffbuf);
          run<P3Cfg<128, 64, 2, 2, 3, true>, true, true>("h2 p3 128x64 s3", g, iters, nullptr, diffbuf);
          run<P3Cfg<128, 64, 2, 2, 4, true>, true, true>("h2 p3 128x64 s4", g, iters, nullptr, diffbuf);
          run<P3Cfg<64, 64, 2, 2, 3, true>, true, true>("h2 p3 64x64 s3", g, iters, nullptr, diffbuf);
          run<P3Cfg<64, 64, 2, 2, 4, true>, true, true>("h2 p3 64x64 s4", g, iters, nullptr, diffbuf);
          sample_err("h2 three products (64x64)", A, B, C, s.M, s.N, s.K, cs);
          run<P3Cfg<128, 128, 2, 2, 3, true>, true, true>("h2 p3 128x128 s3", g, iters, nullptr, diffbuf);
        }
        run<P4Cfg<256, 128, 4, 2, 4, 2, 0, 2, true>, true, true>("h2 p4 256x128 ns4 acc2", g, iters, nullptr, diffbuf);
        sample_err("h2 three products (2 sets)", A, B, C, s.M, s.N, s.K, cs);
        run<P4Cfg<256, 128, 4, 2, 4, 1, 0, 2, true>, true, true>("h2 p4 256x128 ns4 acc1", g, iters, nullptr, diffbuf);
        sample_err("h2 three products (1 set)", A, B, C, s.M, s.N, s.K, cs);
        run<P4Cfg<256, 128, 4, 2, 6, 1, 0, 2, true>, true, true>("h2 p4 256x128 ns6 acc1", g, iters, nullptr, diffbuf);
        if (s.N % 256 == 0) {
          run<P4Cfg<256, 256, 4, 2, 4, 1, 0, 2, true>, true, true, 64>("h2 p4 256x256 ns4 acc1", g, iters, nullptr, diffbuf);
          sample_err("h2 three products (1 set, 256x256)", A, B, C, s.M, s.N, s.K, cs);
          run<P4Cfg<256, 256, 4, 2, 3, 1, 0, 2, true>, true, true, 64>("h2 p4 256x256 ns3 acc1", g, iters, nullptr, diffbuf);
          run<W4Cfg<4, 2, true>, true, true, 64>("h2 w4 256x256 ns4", g, iters, nullptr, diffbuf);
          sample_err("h2 three products (w4)", A, B, C, s.M, s.N, s.K, cs);
          run<W4Cfg<3, 2, true>, true, true, 64>("h2 w4 256x256 ns3", g, iters, nullptr, diffbuf);
          run<W4Cfg<5, 2, true>, true, true, 64>("h2 w4 256x256 ns5", g, iters, nullptr, diffbuf);
          run<W4Cfg<4, 2, true>, true, true, 32 + 64>("h2 w4 256x256 ns4 nostore", g, iters, nullptr, diffbuf);
          run<W4Cfg<4, 2, true, 4, 2>, true, true, 64>("h2 f8 256x256 ns4", g, iters, nullptr, diffbuf);
          run<W4Cfg<4, 2, true, 4, 2>, true, true, 32 + 64>("h2 f8 256x256 ns4 nostore", g, iters, nullptr, diffbuf);
          run<P4Cfg<256, 256, 4, 2, 4, 1, 0, 2, true>, true, true, 32 + 64>("h2 p4 256x256 ns4 acc1 nostore", g, iters, nullptr, diffbuf);
          run<P4Cfg<256, 256, 4, 2, 4, 1, 0, 2, true>, true, true, 32 + 4>("h2 p4 256x256 ns4 acc1 nostore nomfma", g, iters, nullptr, diffbuf);
        }
        CK(hipMemset(Ap.p, 0, (size_t)Ap.m.ps * 4)); CK(hipMemset(Bp.p, 0, (size_t)Bp.m.ps * 4));
        run<P4Cfg<256, 128, 4, 2, 4, 1, 0, 2, true>, true, true>("h2 p4 256x128 ns4 acc1  ZERO operands", g, iters, nullptr, diffbuf);
        CK(hipFree(Ap.p)); CK(hipFree(Bp.p));
      }
      CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
    }
    return 0;
  }
  for (const Shape& s : shapes) {
    if (only[0] && !strstr(s.name, only)) continue;
    printf("== %s  M=%d N=%d K=%d\n", s.name, s.M, s.N, s.K);
    float *A, *B, *C, *Cref;
    CK(hipMalloc(&A, (size_t)s.M * s.K * 4));
    CK(hipMalloc(&B, (size_t)s.N * s.K * 4));
    CK(hipMalloc(&C, (size_t)s.M * s.N * 4));
    CK(hipMalloc(&Cref, (size_t)s.M * s.N * 4));
    hipLaunchKernelGGL(fill_normal, dim3((unsigned)(((int64_t)s.M * s.K + 255) / 256)), dim3(256), 0, 0, A, (int64_t)s.M * s.K, 11u, 1.0f);
    hipLaunchKernelGGL(fill_normal, dim3((unsigned)(((int64_t)s.N * s.K + 255) / 256)), dim3(256), 0, 0, B, (int64_t)s.N * s.K, 12u, 0.05f);
    CK(hipDeviceSynchronize());
    Planes Ap = make_planes(A, s.M, s.K, false), Bp = make_planes(B, s.N, s.K, false);
    Planes Bt = make_planes(B, s.N, s.K, true);      // B^T: [K][N] storage -> the XC flavour of the same product
    LabArgs g{};
    g.A = Ap.m; g.B = Bp.m; g.M = s.M; g.N = s.N; g.K = s.K; g.dbg = 0; g.clk = clk;
    // reference: the lockstep 256x128 tile of gemm_p3.cuh
    g.C = Cref;
    run<P3Cfg<256, 128, 4, 2, 2>, true, true>("p3 256x128 s2 (baseline)", g, iters, nullptr, diffbuf);
    g.C = C;
    const bool n256 = s.N % 256 == 0;
#define RUN(NAME, DBG, ...) run<__VA_ARGS__, true, true, DBG>(NAME, g, iters, Cref, diffbuf)
    RUN("p4 256x128 ns3 acc3 dma0", 0, P4Cfg<256, 128, 4, 2, 3, 3, 0>);
    if (s.M > 4096)
    RUN("p4 256x128 ns3 acc3 nostore", 32, P4Cfg<256, 128, 4, 2, 3, 3, 0>);
    RUN("p4 256x256 ns3 acc1 npl3 nostore", 32, P4Cfg<256, 256, 4, 2, 3, 1, 0, 3>);
    RUN("p4 256x256 ns4 acc1 npl2 nostore", 32, P4Cfg<256, 256, 4, 2, 4, 1, 0, 2>);
    if (n256) {
      // gemm_w4.cuh: one wave per SIMD, 128 x 128 wave tiles; reference for bit-identity = the ping-pong one-set tile of the same planes
      g.C = Cref;
      run<P4Cfg<256, 256, 4, 2, 3, 1, 0, 3>, true, true, 64>("p4 256x256 ns3 acc1 npl3 (w4 reference)", g, iters, nullptr, diffbuf);
      g.C = C;
      RUN("w4 256x256 ns3 npl3", 64, W4Cfg<3, 3>);
      RUN("w4 256x256 ns3 npl3 nostore", 32 + 64, W4Cfg<3, 3>);
      g.C = Cref;
      run<P4Cfg<256, 256, 4, 2, 4, 1, 0, 2>, true, true, 64>("p4 256x256 ns4 acc1 npl2 (w4 reference)", g, iters, nullptr, diffbuf);
      g.C = C;
      RUN("w4 256x256 ns4 npl2", 64, W4Cfg<4, 2>);
      RUN("w4 256x256 ns3 npl2", 64, W4Cfg<3, 2>);
      RUN("w4 256x256 ns5 npl2", 64, W4Cfg<5, 2>);
      RUN("w4 256x256 ns4 npl2 nostore", 32 + 64, W4Cfg<4, 2>);
      RUN("f8 256x256 ns4 npl2 (8 waves free-running)", 64, W4Cfg<4, 2, false, 4, 2>);
      RUN("f8 256x256 ns3 npl2", 64, W4Cfg<3, 2, false, 4, 2>);
      RUN("f8 256x256 ns4 npl2 wg2x4", 64, W4Cfg<4, 2, false, 2, 4>);
      RUN("f8 256x256 ns4 npl2 nostore", 32 + 64, W4Cfg<4, 2, false, 4, 2>);
      RUN("f8 256x256 ns3 npl3", 64, W4Cfg<3, 3, false, 4, 2>);
      g.C = Cref;
      run<P3Cfg<256, 128, 4, 2, 2>, true, true>("p3 256x128 s2 (baseline again)", g, iters, nullptr, diffbuf);
      g.C = C;
    }
    RUN("p4 256x256 ns4 acc1 npl2 nostore nomfma", 36, P4Cfg<256, 256, 4, 2, 4, 1, 0, 2>);
    RUN("p4 256x256 ns4 acc1 npl2 nostore nodma", 34, P4Cfg<256, 256, 4, 2, 4, 1, 0, 2>);
    RUN("p4 256x256 ns4 acc1 npl2 nostore noreads", 48, P4Cfg<256, 256, 4, 2, 4, 1, 0, 2>);
    RUN("p4 256x256 ns4 acc1 npl2 nostore nodma noreads", 50, P4Cfg<256, 256, 4, 2, 4, 1, 0, 2>);
    RUN("p4 256x256 ns8 acc1 npl1 nostore", 32, P4Cfg<256, 256, 4, 2, 8, 1, 0, 1>);
    RUN("p4 256x256 ns8 acc1 npl1 nostore nomfma", 36, P4Cfg<256, 256, 4, 2, 8, 1, 0, 1>);
    RUN("p4 256x256 ns8 acc1 npl1 nostore nodma", 34, P4Cfg<256, 256, 4, 2, 8, 1, 0, 1>);
    RUN("p4 256x256 ns8 acc1 npl1 nostore noreads", 48, P4Cfg<256, 256, 4, 2, 8, 1, 0, 1>);
    RUN("p4 256x256 ns8 acc1 npl1 nostore nodma noreads nomfma", 54, P4Cfg<256, 256, 4, 2, 8, 1, 0, 1>);
    RUN("p4 256x256 ns8 acc1 npl1 nostore nostagger", 33, P4Cfg<256, 256, 4, 2, 8, 1, 0, 1>);
    if (s.M <= 4096) {
      run<P3Cfg<128, 64, 2, 2, 2>, true, true, 0>("p3 128x64 s2 (product, N>=1024)", g, iters, Cref, diffbuf);
      run<P3Cfg<64, 64, 2, 2, 3>, true, true, 0>("p3 64x64 s3 (product, N<1024; non-early)", g, iters, Cref, diffbuf);
      RUN("p4 128x128 wg2x4 ns3 acc3", 0, P4Cfg<128, 128, 2, 4, 3, 3, 0>);
      RUN("p4 128x128 wg4x2 ns3 acc3", 0, P4Cfg<128, 128, 4, 2, 3, 3, 0>);
      RUN("p4 128x128 wg2x4 ns4 acc3", 0, P4Cfg<128, 128, 2, 4, 4, 3, 0>);
      RUN("p4 128x192 wg4x2 ns3 acc3", 0, P4Cfg<128, 192, 4, 2, 3, 3, 0>);
      RUN("p4 128x64 wg4x2 ns3 acc3", 0, P4Cfg<128, 64, 4, 2, 3, 3, 0>);
      RUN("p4 64x128 wg2x4 ns3 acc3", 0, P4Cfg<64, 128, 2, 4, 3, 3, 0>);
      RUN("p4 128x128 wg2x4 ns3 acc3 nostore", 32, P4Cfg<128, 128, 2, 4, 3, 3, 0>);
    }
    RUN("p3 128x128 s3 4 waves (2 WG/CU)", 0, P3Cfg<128, 128, 2, 2, 3>);
    // zero-filled operands: the same instruction stream at the clock the power budget allows without data toggling
    CK(hipMemset(Ap.p, 0, (size_t)Ap.m.ps * 6)); CK(hipMemset(Bp.p, 0, (size_t)Bp.m.ps * 6));
    g.B = Bp.m;
    run<P3Cfg<256, 128, 4, 2, 2>, true, true>("p3 256x128 s2  ZERO operands", g, iters, nullptr, diffbuf);
    run<P4Cfg<256, 128, 4, 2, 3, 3, 0>, true, true>("p4 256x128 ns3 acc3  ZERO operands", g, iters, nullptr, diffbuf);

    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(Cref));
    CK(hipFree(Ap.p)); CK(hipFree(Bp.p)); CK(hipFree(Bt.p));
  }
  return 0;
}
