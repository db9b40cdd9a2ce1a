// pxr_common.h -- shared device/host helpers for the pixelrec_amd HIP kernels (gfx950 / CDNA4 only).
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes (hard-coded, see guides: warpSize folds to 64 on gfx950);
//   * every C-ABI entry point takes a hipStream_t (as void*) and returns an int status
//     (0 = ok, <0 = PXR_ERR_*); nothing here allocates or frees device memory;
//   * all arithmetic is fp32 (the reference runs fp32 end to end, code/run.py has no AMP).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PXR_OK 0
#define PXR_ERR_BAD_ARG (-1)
#define PXR_ERR_LAUNCH (-2)
#define PXR_ERR_WORKSPACE (-3)

#define PXR_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Record the most recent error string for pxr_last_error(); thread-local so that the autograd thread
// and the main thread do not clobber each other.
void pxr_set_error(const char* fmt, ...);

// Device status word registered with pxr_set_status_word (null when none): see api.cpp.  PXR_STATUS_BAD_INDEX is set
// by kernels that met an embedding id outside the table.
int32_t* pxr_status_word(void);
int pxr_cu_count(void);          // api.cpp: compute units of the current device (0 if unknown)
#define PXR_STATUS_BAD_INDEX 1
#define PXR_STATUS_TOPK_OVERFLOW 4   /* the candidate buffer of the two-pass top-k overflowed (results may miss items) */
#define PXR_STATUS_GEMM_TIMEOUT 8    /* a stream-K GEMM worker gave up waiting for a partial tile (results are wrong) */
#define PXR_STATUS_ROWS_OVERFLOW 2   /* a rank's unique-row count exceeded the capacity of a reduced row exchange */
#define PXR_STATUS_TOPK_UNDERFLOW 32 /* a user of the two-pass top-k ended with fewer than K candidates (non-finite scores / thresholds?) */
#define PXR_STATUS_H2_RANGE 64       /* a producer of fp16 two-plane operands (planes.cuh "h2") met a value outside the fp16 range */
#define PXR_STATUS_H2_STALE 128      /* h2 planes written under the previous step's scale: a value outgrew its headroom and was saturated */
#define PXR_STATUS_SHARD_OVERFLOW 16 /* row-sharded table: more hit rows owned by ONE rank than the per-pair request capacity */

static inline int pxr_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pxr_set_error("%s: %s", what, hipGetErrorString(e));
    return PXR_ERR_LAUNCH;
  }
  return PXR_OK;
}

#define PXR_REQUIRE(cond, ...)       \
  do {                               \
    if (!(cond)) {                   \
      pxr_set_error(__VA_ARGS__);    \
      return PXR_ERR_BAD_ARG;        \
    }                                \
  } while (0)

#ifdef __HIPCC__
// ---------------------------------------------------------------- streaming (non-temporal) 16-byte accesses
// For data that is touched once per launch (table sweeps, big gathers, score matrices): keeps the stream out of
// L2 / Infinity Cache write-allocate.  Measured on the 428 MB row gather: 5.7 -> 6.8 TB/s with NT stores.
typedef float pxr_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 pxr_ld_stream(const float* p) {
  const pxr_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pxr_f32x4*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void pxr_st_stream(float* p, const float4& v) {
  pxr_f32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<pxr_f32x4*>(p));
}

// ---------------------------------------------------------------- XCD placement
namespace pxr {
// XCD-aware bijective remap of a linear block id so that each of the 8 XCDs (block b runs on XCD b % 8)
// works on one contiguous chunk of the tile sequence (cdna_hip_programming.md T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int xcd = bid & 7, q = nblk >> 3, rem = nblk & 7;
  const int start = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  return start + (bid >> 3);
}
}  // namespace pxr
// Row-parallel producers (LayerNorm, attention, the plane split) use the SAME map as the GEMM tiles that read their output
// next: with the M tile index slowest in the GEMM's tile order, XCD x multiplies the rows [x M/8, (x+1) M/8) -- when the
// producer's workgroups on XCD x wrote exactly those rows, the operand is still in that XCD's L2 (measured on the
// 3200 x 512 x 512 projection: 17.0 us behind an aligned producer, 20.1 us behind an interleaved one, 16.2 us L2-hot;
// tools/xcd_align_probe.py).  Speed only: nothing depends on where a workgroup actually runs.

// ---------------------------------------------------------------- wave / block reductions
// Wave64 reductions on the VALU's DPP path instead of six ds_bpermute round trips through the LDS crossbar
// (__shfl_xor): quad swaps, mirror within 8 / 16 lanes, then the gfx9 row broadcasts -- lane 63 ends up with the total
// and v_readlane hands it to every lane.  ~6 dependent VALU ops (tens of cycles) vs ~6 x 100+ cycles; the row softmax
// of the attention kernels does 2 reductions per query row.  Must be called with all 64 lanes active.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float pxr_dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                              CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += pxr_dpp<0xB1, 0xf>(0.f, v);    // quad_perm [1,0,3,2]
  v += pxr_dpp<0x4E, 0xf>(0.f, v);    // quad_perm [2,3,0,1]
  v += pxr_dpp<0x141, 0xf>(0.f, v);   // row_half_mirror
  v += pxr_dpp<0x140, 0xf>(0.f, v);   // row_mirror: every lane of a 16-lane row holds the row sum
  v += pxr_dpp<0x142, 0xa>(0.f, v);   // row_bcast:15 into rows 1, 3
  v += pxr_dpp<0x143, 0xc>(0.f, v);   // row_bcast:31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, pxr_dpp<0xB1, 0xf>(v, v));
  v = fmaxf(v, pxr_dpp<0x4E, 0xf>(v, v));
  v = fmaxf(v, pxr_dpp<0x141, 0xf>(v, v));
  v = fmaxf(v, pxr_dpp<0x140, 0xf>(v, v));
  v = fmaxf(v, pxr_dpp<0x142, 0xa>(v, v));
  v = fmaxf(v, pxr_dpp<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// ---------------------------------------------------------------- fixed-order reduction of per-block partials
// out[n] = sum_p part[p][n] for n < N.  Block = PXR_RED_CX columns x PXR_RED_CY row-lanes; row-lane r sums
// p = r, r+CY, ... with 8 independent loads in flight (a plain loop is one dependent L2 round trip per partial row),
// then the CY lanes are combined in a fixed tree => deterministic.  Columns n < split go to out_a[n], the rest to
// out_b[n - split] (LayerNorm dgamma | dbeta share one partial buffer); pass split = N for a single output.
constexpr int PXR_RED_CX = 16, PXR_RED_CY = 16;
static_assert(PXR_RED_CX * PXR_RED_CY == 256, "reduction kernels run 256 threads");

__device__ __forceinline__ float pxr_strided_column_sum(const float* __restrict__ part, int P, int N, int col, int ty) {
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  int p = ty;
  for (; p + 7 * PXR_RED_CY < P; p += 8 * PXR_RED_CY) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(p + PXR_RED_CY * u) * N + col];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] += v[u];
  }
  for (int u = 0; p < P; p += PXR_RED_CY, ++u) acc[u & 7] += part[(int64_t)p * N + col];
  return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}
// block-level tail shared by the reduction kernels: returns (on row-lane 0) the total of the CY row-lane sums
__device__ __forceinline__ float pxr_combine_row_lanes(float s, float (&red)[PXR_RED_CY][PXR_RED_CX + 1], int tx, int ty) {
  red[ty][tx] = s;
  __syncthreads();
  float v = 0.f;
  if (ty == 0) {
    float t[PXR_RED_CY];
#pragma unroll
    for (int r = 0; r < PXR_RED_CY; ++r) t[r] = red[r][tx];
#pragma unroll
    for (int w = PXR_RED_CY / 2; w >= 1; w >>= 1)
#pragma unroll
      for (int r = 0; r < w; ++r) t[r] = t[2 * r] + t[2 * r + 1];
    v = t[0];
  }
  return v;
}

static __global__ void __launch_bounds__(256) pxr_reduce_partials_kernel(const float* __restrict__ part, int P, int N,
                                                                  float* __restrict__ out_a,
                                                                  float* __restrict__ out_b, int split) {
  __shared__ float red[PXR_RED_CY][PXR_RED_CX + 1];
  const int tx = threadIdx.x % PXR_RED_CX, ty = threadIdx.x / PXR_RED_CX;
  const int col = blockIdx.x * PXR_RED_CX + tx;
  const float s = (col < N) ? pxr_strided_column_sum(part, P, N, col, ty) : 0.f;
  const float v = pxr_combine_row_lanes(s, red, tx, ty);
  if (ty == 0 && col < N) {
    if (col < split) out_a[col] = v;
    else out_b[col - split] = v;
  }
}

// ---------------------------------------------------------------- stateless dropout RNG
// keep(elem) for Bernoulli(1-p) dropout: a counter-based hash of (seed, stream, element index), so the
// backward kernel regenerates the forward mask instead of storing it.  tests/ restates this in numpy
// (oracle/dropout_rng.py) to feed identical masks to the CPU oracle.  The reference uses ATen's Philox
// stream (nn.Dropout, sasrec.py:46, layers.py:575,579,641) which cannot be reproduced bit-for-bit;
// training-mode parity is checked with this mask injected into the oracle.
__device__ __host__ __forceinline__ uint32_t pxr_fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __host__ __forceinline__ uint32_t pxr_hash32(uint64_t seed, uint32_t stream, uint64_t idx) {
  // 32-bit-only mixing (64-bit multiplies are multi-instruction on the VALU): two murmur3 finalizer rounds.
  const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  uint32_t h = pxr_fmix32(lo ^ (uint32_t)seed);
  h += hi * 0x9E3779B1u + stream * 0x85EBCA77u + (uint32_t)(seed >> 32);
  return pxr_fmix32(h);
}
// threshold = floor(p * 2^32); keep when hash >= threshold  (p == 0 -> threshold 0 -> always keep)
__device__ __host__ __forceinline__ uint32_t pxr_drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t <= 0.0) return 0u;
  if (t >= 4294967295.0) return 4294967295u;
  return (uint32_t)t;
}
__device__ __host__ __forceinline__ bool pxr_keep(uint64_t seed, uint32_t stream, uint64_t idx, uint32_t thr) {
  return pxr_hash32(seed, stream, idx) >= thr;
}
#endif  // __HIPCC__
