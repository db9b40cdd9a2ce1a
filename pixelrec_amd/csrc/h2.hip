// h2.hip -- producers of the TWO-plane fp16 operand format (planes.cuh "h2") whose power-of-two scale is chosen ON THE DEVICE:
// tensors that change every step (the gradients of a backward pass, the weights of trainable blocks), where a host-side maximum
// would cost a synchronisation per tensor.
//
//   pxr_h2_split_auto_multi_f32   per matrix: max |x| (and, for weights, the largest column sum of |w|) by atomic maxima, then the
//                                 split with e = 14 - ceil(log2 max): max |x| 2^e in [2^13, 2^14); e and the statistics stay in
//                                 device memory, the GEMMs read e from there (pxr_gemm_h2_f32 / pxr_grouped_dw_h2_f32 *_exp_dev)
//   pxr_h2_bound_exp              the exponent of a tensor that is written as planes BEFORE its maximum can be known (an input
//                                 gradient leaving a GEMM epilogue): from the rigorous bound
//                                     |sum_k dy[t,k] w[k,j] f| <= max |dy| * max_j sum_k |w[k,j]| * max |f|
//                                 e = 15 - ceil(log2 bound): no overflow whatever the data; the bound is loose by the usual
//                                 sqrt(K)-vs-K factor (2^5 .. 2^6 for the tower), which costs that many of fp16's 30 binades of
//                                 headroom below 2^15, not accuracy of the values that matter (see DESIGN.md "fp16 two-plane")
#include "gemm_p3.cuh"

namespace pxr {

// non-negative floats order like their bit patterns: atomicMax on the int view (NaN: larger than everything -> propagates as "huge")
__device__ __forceinline__ void h2_atomic_max(float* p, float v) { atomicMax(reinterpret_cast<int*>(p), __float_as_int(v)); }

// e with max_abs * 2^e in [2^(top-1), 2^top)
__device__ __forceinline__ int h2_exp_for(float max_abs, int top) {
  if (!(max_abs > 0.f) || isinf(max_abs)) return 0;
  int ex;
  (void)frexpf(max_abs, &ex);          // max_abs = f 2^ex, f in [0.5, 1)
  const int e = top - ex;
  return e < -60 ? -60 : (e > 60 ? 60 : e);
}

// stat[0] = max |x|   (four independent 16-byte loads in flight per thread: the tensors are hundreds of MB).  FLAT: contiguous rows --
// no 64-bit division per load (a select between the two address forms made the compiler evaluate both: 1 TB/s instead of 4)
template <bool FLAT>
__global__ void __launch_bounds__(256) h2_max_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols4, float* stat) {
  const int64_t n = rows * cols4, stride = (int64_t)gridDim.x * 256;
  float m = 0.f;
  bool bad = false;
  auto take = [&](const float4& v) {
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    bad |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
  };
  auto at = [&](int64_t i) {
    if constexpr (FLAT) return *reinterpret_cast<const float4*>(x + i * 4);
    else return *reinterpret_cast<const float4*>(x + (i / cols4) * ldx + (i % cols4) * 4);
  };
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const float4 a = at(i), b = at(i + stride), c = at(i + 2 * stride), d = at(i + 3 * stride);
    take(a); take(b); take(c); take(d);
  }
  for (; i < n; i += stride) take(at(i));
  if (bad) m = __int_as_float(0x7f800000);       // NaN -> inf
  // one atomic per WORKGROUP, and only when it can raise the maximum: 16 384 same-address atomics (one per wave) serialise in one L2
  // channel -- ~200 us on top of a 50 us read (measured)
  __shared__ float wm[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (m > __hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) h2_atomic_max(stat, m);
  }
}
// stat[0] = max |w|, stat[1] = max over columns j of sum_k |w[k][j]|   (w [rows][cols]).  A workgroup owns 32 columns: 8 row lanes x
// 32 columns of threads walk the rows (128-byte coalesced segments), the 8 partial sums of a column are added in a fixed order
// (deterministic: the exponent derived from the bound must not depend on the launch)
__global__ void __launch_bounds__(256) h2_colstat_kernel(const float* __restrict__ w, int64_t ldw, int rows, int cols, float* stat) {
  __shared__ float ps[8][33], pm[8][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + cx;
  float m = 0.f, s0 = 0.f, s1 = 0.f;
  bool bad = false;
  if (j < cols) {
    int k = ry;
    for (; k + 8 < rows; k += 16) {
      const float a = fabsf(w[(int64_t)k * ldw + j]), b = fabsf(w[(int64_t)(k + 8) * ldw + j]);
      m = fmaxf(m, fmaxf(a, b)); s0 += a; s1 += b;
      bad |= (a != a) | (b != b);
    }
    if (k < rows) {
      const float a = fabsf(w[(int64_t)k * ldw + j]);
      m = fmaxf(m, a); s0 += a;
      bad |= (a != a);
    }
  }
  if (bad) { m = __int_as_float(0x7f800000); s0 = m; }
  ps[ry][cx] = s0 + s1; pm[ry][cx] = m;
  __syncthreads();
  if (ry == 0) {
    float s = 0.f, mm = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { s += ps[q][cx]; mm = fmaxf(mm, pm[q][cx]); }
    if (mm > 0.f) { h2_atomic_max(stat, mm); h2_atomic_max(stat + 1, s); }
  }
}

// max |x| of up to 16 CONTIGUOUS matrices in one launch (the weights of the sequence block: 8 small matrices, where one launch per
// matrix costs more than the reads): H2_BPM workgroups per matrix, one atomic each
constexpr int H2_BPM = 16;
struct H2MaxMulti {
  const float* x[16];
  int64_t n4[16];          // float4 elements
  int n;
  float* stats;            // [n][2]
};
__global__ void __launch_bounds__(256) h2_max_multi_kernel(const H2MaxMulti m) {
  const int mat = blockIdx.x / H2_BPM, part = blockIdx.x % H2_BPM;
  const float4* x = reinterpret_cast<const float4*>(m.x[mat]);
  const int64_t n = m.n4[mat];
  float mx = 0.f;
  bool bad = false;
  for (int64_t i = (int64_t)part * 256 + threadIdx.x; i < n; i += (int64_t)H2_BPM * 256) {
    const float4 v = x[i];
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    bad |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
  }
  if (bad) mx = __int_as_float(0x7f800000);
  __shared__ float wm[4];
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (mx > 0.f) h2_atomic_max(m.stats + 2 * mat, mx);
  }
}

// (a kernel, not hipMemsetAsync: inside a captured step the memset node was seen to run out of order with the atomic maxima that
// follow it -- intermittently zeroing finished statistics; a kernel node is ordered like every other launch of the stream)
__global__ void h2_zero_kernel(float* p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0.f;
}

struct H2SplitAuto {
  const float* x[16]; int64_t ldx[16]; int rows[16], cols8[16]; P3Mat out[16];
  int64_t begin[17];
  int n;
  float* stats;            // [n][2]
  int* exps;               // [n]
  int32_t* status;
  int fill_colsum;         // no column statistics were gathered: stats[2 i + 1] = rows * max |x| (>= every column sum of |x|)
  int top;                 // the largest scaled value lands in [2^(top-1), 2^top): 14 by default (two binades below the fp16 limit)
  const float* parts;      // n == 1 only: max |x| = the maximum of these n_parts partial maxima (one per workgroup of the producer:
  int n_parts;             // pxr_ln_bwd_stat_f32 / pxr_attn_bwd_stat_f32); every workgroup reduces them itself (<= 1024 words from L2)
  // n == 1 with parts: also the exponent of the INPUT GRADIENT the next GEMM writes as planes (pxr_h2_bound_exp's rule, saving its
  // one-thread launch): *bound_out = 15 - ceil(log2(max |x| * bound_b[0] * bound_factor)); bound_out == null: none
  const float* bound_b; float bound_factor; int* bound_out;
};
__global__ void __launch_bounds__(256) h2_split_auto_kernel(const H2SplitAuto m) {
  __shared__ float pmax[4];
  float part_max = 0.f;
  if (m.parts) {
    for (int q = threadIdx.x; q < m.n_parts; q += 256) part_max = fmaxf(part_max, m.parts[q]);
    part_max = wave_max(part_max);
    if ((threadIdx.x & 63) == 0) pmax[threadIdx.x >> 6] = part_max;
    __syncthreads();
    part_max = fmaxf(fmaxf(pmax[0], pmax[1]), fmaxf(pmax[2], pmax[3]));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      m.stats[0] = part_max;
      if (m.bound_out) *m.bound_out = h2_exp_for(part_max * m.bound_b[0] * m.bound_factor, 15);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < m.n) {
    const float mx = m.parts ? part_max : m.stats[2 * threadIdx.x];
    m.exps[threadIdx.x] = h2_exp_for(mx, m.top);
    if (m.fill_colsum) m.stats[2 * threadIdx.x + 1] = (float)m.rows[threadIdx.x] * mx;
  }
  // (grid-stride form; launched with one chunk per thread, see h2_split_blocks)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m.begin[m.n]; i += (int64_t)gridDim.x * 256) {
    int pi = 0;
#pragma unroll 1
    for (int k = 1; k < m.n; ++k)
      if (i >= m.begin[k]) pi = k;
    const float sc = ldexpf(1.0f, h2_exp_for(m.parts ? part_max : m.stats[2 * pi], m.top));
    const int64_t li = i - m.begin[pi];
    const int64_t row = li / m.cols8[pi];
    const int c = (int)(li % m.cols8[pi]) * 8;
    const float* src = m.x[pi] + row * m.ldx[pi] + c;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    const float v[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, b.x * sc, b.y * sc, b.z * sc, b.w * sc};
    px_store8(m.out[pi], PXR_PLANES_H2, m.status, row, c, v);
  }
}

__global__ void h2_bound_exp_kernel(const float* a_max, const float* b_colsum, float factor, int* exp_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *exp_out = h2_exp_for(a_max[0] * b_colsum[0] * factor, 15);
}

// ---- stale scales (round 6).  The gradients a backward pass hands to its GEMMs (LayerNorm backward outputs, dq | dk | dv) change
// slowly from one training step to the next, and the h2 format tolerates a scale that is off by many binades (planes.cuh).  So their
// producers write the planes THEMSELVES under the exponent this kernel derived from the PREVIOUS step's maximum, HEADROOM binades
// below the usual placement (the recent maximum in [2^(13-H), 2^(14-H)): a value may exceed it 2^(H+2)-fold before it leaves the fp16
// range -- and is then saturated and flagged, never inf), and leave this step's partial maxima for the next update: six split launches and six fp32
// round trips per step disappear.  One workgroup per site: max over its partials -> exps[s], stats[2 s .. 2 s + 1] = (max 2^H, rows max
// 2^H) (what a consumer may assume about the NEXT step's values) and, for a site whose planes feed a GEMM that writes its own output
// as planes before knowing it (du), bexp[s] = 15 - ceil(log2(max 2^H * colsum |W| * factor)).  A site without gradient (max 0 or
// non-finite) keeps its old entries.
struct H2Sites {
  const float* parts[16]; int n_parts[16]; int rows[16];
  const float* bound_b[16];         // &W.stats[1] (largest column sum of |w|) or null
  int n, headroom;
  float bound_factor, decay;
  int* exps; float* stats; int* bexp;
  float* run_max;                   // [n] decaying maximum of the step maxima: run = max(step max, run * decay)
};
__global__ void __launch_bounds__(256) h2_sites_update_kernel(const H2Sites m) {
  __shared__ float pmax[4];
  const int s = blockIdx.x;
  float mx = 0.f;
  for (int q = threadIdx.x; q < m.n_parts[s]; q += 256) mx = fmaxf(mx, m.parts[s][q]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) pmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmaxf(fmaxf(pmax[0], pmax[1]), fmaxf(pmax[2], pmax[3]));
    if (mx > 0.f && !isinf(mx)) {
      // the maxima of these gradients are heavy-tailed (one sharp softmax row of one sequence sets max |dqkv|: 10 x from step to
      // step is ordinary, 50 x was seen within 220 steps of the bench stream), so the scale follows a DECAYING maximum of the last
      // steps' maxima (half-life ln 2 / (1 - decay) steps), not the last step alone
      mx = fmaxf(mx, m.run_max[s] * m.decay);
      m.run_max[s] = mx;
      const float grown = ldexpf(mx, m.headroom);
      m.exps[s] = h2_exp_for(mx, 14 - m.headroom);
      m.stats[2 * s] = grown;
      m.stats[2 * s + 1] = (float)m.rows[s] * grown;
      if (m.bound_b[s]) m.bexp[s] = h2_exp_for(grown * m.bound_b[s][0] * m.bound_factor, 15);
    }
  }
}

// workgroups of the split: one 8-element chunk per thread (the kernel's loop is a grid-stride one, so any cap works; capping at
// 8 192 workgroups -- three chunks per thread on 102 400-token tensors -- measured SLOWER: 162 vs 138 us per launch, profiles/r05)
static inline int64_t h2_split_blocks(int64_t chunks) {
  const int64_t b = (chunks + 255) / 256;
  return b < 1 ? 1 : (b > 0x7fffffff ? 0x7fffffff : b);
}
}  // namespace pxr

using namespace pxr;

extern "C" int pxr_h2_split_auto_multi_f32(int n, const float* const* x, const int64_t* rows, const int64_t* cols, const int64_t* ldx,
                                           void* const* planes, const int64_t* plane_stride, const int64_t* panel_rows, int col_stats,
                                           float* stats, int* exps, void* stream) {
  PXR_REQUIRE(n >= 1 && n <= 16 && x && rows && cols && ldx && planes && plane_stride && panel_rows && stats && exps,
              "pxr_h2_split_auto_multi_f32: bad args");
  hipStream_t st = (hipStream_t)stream;
  H2SplitAuto m{};
  m.n = n;
  m.stats = stats; m.exps = exps; m.status = pxr_status_word();
  // col_stats: bits 0-7 the mode (below); bits 8-15 optionally `top` (8 .. 15; 0 = the default 14): tensors that are rewritten in
  // place between splits by someone who reuses the exponent (the optimizer's weight planes) ask for more headroom
  m.top = (col_stats >> 8) & 0xFF;
  col_stats &= 0xFF;
  PXR_REQUIRE(m.top == 0 || (m.top >= 8 && m.top <= 15), "pxr_h2_split_auto_multi_f32: top must be 8 .. 15");
  if (m.top == 0) m.top = 14;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    PXR_REQUIRE(x[i] && planes[i] && rows[i] > 0 && rows[i] < (1ll << 31) && cols[i] > 0 && cols[i] % 32 == 0 && ldx[i] % 4 == 0 &&
                    (((uintptr_t)x[i]) & 15) == 0 && p3_mat_ok(planes[i], plane_stride[i], panel_rows[i], rows[i], cols[i]),
                "pxr_h2_split_auto_multi_f32: matrix %d is bad", i);
    m.x[i] = x[i]; m.ldx[i] = ldx[i]; m.rows[i] = (int)rows[i]; m.cols8[i] = (int)(cols[i] / 8);
    m.out[i] = P3Mat{reinterpret_cast<__bf16*>(planes[i]), plane_stride[i], panel_rows[i]};
    m.begin[i] = total;
    total += rows[i] * (cols[i] / 8);
  }
  m.begin[n] = total;
  PXR_REQUIRE((total + 255) / 256 < (1ll << 31), "pxr_h2_split_auto_multi_f32: too large");
  // col_stats: 0 = max |x| only (stats[2 i + 1] is then filled with rows * max, a bound on every column sum), 1 = max and the largest
  // column sum of |x|, 2 = the caller's producers already gathered stats[2 i] (atomic maxima into a zeroed slot): no statistics pass
  m.fill_colsum = (col_stats != 1);
  if (col_stats != 2) {
    hipLaunchKernelGGL(h2_zero_kernel, dim3(1), dim3(64), 0, st, stats, 2 * n);
    bool contiguous = true;
    for (int i = 0; i < n; ++i) contiguous = contiguous && (ldx[i] == cols[i]);
    if (col_stats == 0 && contiguous && n > 1) {
      H2MaxMulti mm{};
      mm.n = n; mm.stats = stats;
      for (int i = 0; i < n; ++i) { mm.x[i] = x[i]; mm.n4[i] = rows[i] * (cols[i] / 4); }
      hipLaunchKernelGGL(h2_max_multi_kernel, dim3((unsigned)(n * H2_BPM)), dim3(256), 0, st, mm);
    } else {
      for (int i = 0; i < n; ++i) {
        if (col_stats == 1) {
          hipLaunchKernelGGL(h2_colstat_kernel, dim3((unsigned)((cols[i] + 31) / 32)), dim3(256), 0, st, x[i], ldx[i], (int)rows[i], (int)cols[i],
                             stats + 2 * i);
        } else {
          const int64_t work = rows[i] * (cols[i] / 4);
          const int64_t blocks = (work + 1023) / 1024;          // >= 4 loads per thread; few workgroups = few same-address atomics
          const dim3 grid((unsigned)(blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks)));
          if (ldx[i] == cols[i]) hipLaunchKernelGGL(h2_max_kernel<true>, grid, dim3(256), 0, st, x[i], ldx[i], rows[i], (int)(cols[i] / 4), stats + 2 * i);
          else hipLaunchKernelGGL(h2_max_kernel<false>, grid, dim3(256), 0, st, x[i], ldx[i], rows[i], (int)(cols[i] / 4), stats + 2 * i);
        }
      }
    }
  }
  hipLaunchKernelGGL(h2_split_auto_kernel, dim3((unsigned)h2_split_blocks(total)), dim3(256), 0, st, m);
  return pxr_check_launch("pxr_h2_split_auto_multi_f32");
}

// One matrix whose maximum arrives as `n_parts` PARTIAL maxima (one word per workgroup of the kernel that produced x: the LayerNorm
// backward's per-workgroup values, the attention backward's spread slots): no statistics launch, no atomics on a single word --
// every workgroup of the split reduces the partials itself.  stats[0] = max |x|, stats[1] = rows * max |x|, exps[0] as above.
// bound_exp_out (optional): also the exponent pxr_h2_bound_exp(stats, bound_b_colsum, bound_factor) would compute for the input gradient
// the next GEMM forms from x and a weight whose largest column sum of |w| is *bound_b_colsum.
extern "C" int pxr_h2_split_parts_f32(const float* x, int64_t rows, int64_t cols, int64_t ldx, void* planes, int64_t plane_stride,
                                      int64_t panel_rows, const float* parts, int n_parts, float* stats, int* exps,
                                      const float* bound_b_colsum, float bound_factor, int* bound_exp_out, void* stream) {
  PXR_REQUIRE(x && planes && parts && stats && exps && n_parts >= 1 && n_parts <= 1024, "pxr_h2_split_parts_f32: bad args (1..1024 partials)");
  PXR_REQUIRE(!bound_exp_out || (bound_b_colsum && bound_factor > 0.f), "pxr_h2_split_parts_f32: the bound needs the weight's column-sum statistic");
  PXR_REQUIRE(rows > 0 && rows < (1ll << 31) && cols > 0 && cols % 32 == 0 && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 &&
                  p3_mat_ok(planes, plane_stride, panel_rows, rows, cols), "pxr_h2_split_parts_f32: bad matrix");
  H2SplitAuto m{};
  m.n = 1; m.stats = stats; m.exps = exps; m.status = pxr_status_word();
  m.x[0] = x; m.ldx[0] = ldx; m.rows[0] = (int)rows; m.cols8[0] = (int)(cols / 8);
  m.out[0] = P3Mat{reinterpret_cast<__bf16*>(planes), plane_stride, panel_rows};
  m.begin[0] = 0; m.begin[1] = rows * (cols / 8);
  m.fill_colsum = 1; m.parts = parts; m.n_parts = n_parts; m.top = 14;
  m.bound_b = bound_b_colsum; m.bound_factor = bound_factor; m.bound_out = bound_exp_out;
  PXR_REQUIRE((m.begin[1] + 255) / 256 < (1ll << 31), "pxr_h2_split_parts_f32: too large");
  hipLaunchKernelGGL(h2_split_auto_kernel, dim3((unsigned)h2_split_blocks(m.begin[1])), dim3(256), 0, (hipStream_t)stream, m);
  return pxr_check_launch("pxr_h2_split_parts_f32");
}

extern "C" int pxr_h2_bound_exp(const float* a_max, const float* b_colsum, float factor, int* exp_out, void* stream) {
  PXR_REQUIRE(a_max && b_colsum && exp_out && factor > 0.f, "pxr_h2_bound_exp: bad args");
  hipLaunchKernelGGL(h2_bound_exp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a_max, b_colsum, factor, exp_out);
  return pxr_check_launch("pxr_h2_bound_exp");
}

// exps / stats / bexp / run_max: persistent device arrays of n (<= 16) sites (run_max zero-initialised; decay: the per-step factor of the
// decaying maximum the scales follow, 0 = the last step alone) (the caller seeds them once from an exact pass: pxr_h2_split_parts_f32
// writes the same quantities; a site's exponent in use is exps[s], i.e. the headroom is already in it).  parts[s]: n_parts[s] partial
// maxima of site s left by this step's producers; rows[s]: rows of the site's matrix; bound_b[s]: NULL or the address of the largest
// column sum of |W| of the weight behind the site (Planes.stats + 1).
extern "C" int pxr_h2_sites_update(int n, const float* const* parts, const int* n_parts, const int* rows, const float* const* bound_b,
                                   float bound_factor, int headroom, float decay, float* run_max, int* exps, float* stats, int* bexp,
                                   void* stream) {
  PXR_REQUIRE(n >= 1 && n <= 16 && parts && n_parts && rows && bound_b && exps && stats && bexp && run_max && headroom >= 0 &&
                  headroom <= 8 && bound_factor > 0.f && decay >= 0.f && decay <= 1.f,
              "pxr_h2_sites_update: bad args (1..16 sites, headroom 0..8, decay in [0, 1])");
  H2Sites m{};
  m.n = n; m.headroom = headroom; m.bound_factor = bound_factor; m.exps = exps; m.stats = stats; m.bexp = bexp;
  m.decay = decay; m.run_max = run_max;
  for (int i = 0; i < n; ++i) {
    PXR_REQUIRE(parts[i] && n_parts[i] >= 1 && n_parts[i] <= 1024 && rows[i] > 0, "pxr_h2_sites_update: site %d is bad", i);
    m.parts[i] = parts[i]; m.n_parts[i] = n_parts[i]; m.rows[i] = rows[i]; m.bound_b[i] = bound_b[i];
  }
  hipLaunchKernelGGL(h2_sites_update_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, m);
  return pxr_check_launch("pxr_h2_sites_update");
}
