// planes.cuh -- the pre-split operand format of the planes GEMMs (gemm_p3.cuh): an fp32 matrix as three bf16 planes
// hi | mid | lo (x = hi + mid + lo exactly) in PANEL layout.  Shared by the GEMM kernels and by every producer that writes
// its output straight in this format (LayerNorm, attention, GEMM epilogues, the optimizer).
//
// Panel layout of a matrix X[R][C] (C % 32 == 0; `pr` >= R rows allocated per panel, pr % 32 == 0): plane q starts
// q * ps elements after the base; inside a plane, panel cb = c / 32 holds columns 32 cb .. 32 cb + 31 of ALL rows:
//     element (r, c)  at  ((cb * pr + r) * 32 + (((c >> 3) & 3) ^ ((r >> 2) & 3)) * 8 + (c & 7))      [elements]
// i.e. a 64-byte row segment per (row, panel) whose four 16-byte chunks are XOR-swizzled by the row: the LDS image of a GEMM
// tile is a byte copy of 1 KiB runs of a panel (16 rows), conflict-free for ds_read_b128 and ds_read_b64_tr_b16.
#pragma once
#include "pxr_common.h"

namespace pxr {

typedef unsigned p3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned p3_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 p3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float p3_f32x2 __attribute__((ext_vector_type(2)));

// ---- the exact 3-term split (same arithmetic as gemm_b3.cuh::b3_split2: hi = bf16_rne(x), mid = bf16_rne(x - hi),
// lo = x - hi - mid; both remainders are exact in fp32 and lo has <= 8 significant bits) -----------------------------
__device__ __forceinline__ void p3_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const p3_f32x2 v = {a, b};
  const p3_bf16x2 h = __builtin_convertvector(v, p3_bf16x2);
  const p3_f32x2 r1 = v - __builtin_convertvector(h, p3_f32x2);
  const p3_bf16x2 m = __builtin_convertvector(r1, p3_bf16x2);
  const p3_f32x2 r2 = r1 - __builtin_convertvector(m, p3_f32x2);
  const p3_bf16x2 l = __builtin_convertvector(r2, p3_bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  mid = __builtin_bit_cast(unsigned, m);
  lo = __builtin_bit_cast(unsigned, l);
}
// A planes matrix as a kernel argument.
struct P3Mat {
  __bf16* p;       // plane 0 (nullptr: "no planes wanted" for optional outputs)
  int64_t ps;      // plane stride, elements
  int64_t pr;      // rows per panel (allocated), multiple of 32
};
// element offset of (row r, 8-column chunk starting at column c, c % 8 == 0) inside a plane
__device__ __forceinline__ int64_t p3_chunk_index(int64_t pr, int64_t r, int c) {
  return (((int64_t)(c >> 5) * pr + r) << 5) + ((((c >> 3) & 3) ^ ((int)(r >> 2) & 3)) << 3);
}
// store 8 consecutive values x[r][c .. c+7] (c % 8 == 0) into the three planes: one 16-byte store per plane
__device__ __forceinline__ void p3_store8(const P3Mat& m, int64_t r, int c, const float (&v)[8]) {
  p3_u32x4 p[3];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h, mi, l;
    p3_split2(v[2 * j], v[2 * j + 1], h, mi, l);
    p[0][j] = h; p[1][j] = mi; p[2][j] = l;
  }
  __bf16* dst = m.p + p3_chunk_index(m.pr, r, c);
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<p3_u32x4*>(dst + q * m.ps) = p[q];
}
// store 4 consecutive values x[r][c .. c+3] (c % 4 == 0): one 8-byte store per plane (row-per-wave kernels, float4 per lane)
__device__ __forceinline__ void p3_store4(const P3Mat& m, int64_t r, int c, const float4& v) {
  unsigned h0, m0, l0, h1, m1, l1;
  p3_split2(v.x, v.y, h0, m0, l0);
  p3_split2(v.z, v.w, h1, m1, l1);
  __bf16* dst = m.p + p3_chunk_index(m.pr, r, c & ~7) + (c & 4);
  *reinterpret_cast<p3_u32x2*>(dst) = p3_u32x2{h0, h1};
  *reinterpret_cast<p3_u32x2*>(dst + m.ps) = p3_u32x2{m0, m1};
  *reinterpret_cast<p3_u32x2*>(dst + 2 * m.ps) = p3_u32x2{l0, l1};
}
// ---- the TWO-plane fp16 format ("h2"): x * 2^e = hi + lo + d with hi = fp16_rne(x 2^e), lo = fp16_rne(x 2^e - hi) (the remainder is
// exact in fp32), |d| <= max(2^-22 |x 2^e|, 2^-25) -- 22 significant bits in two planes, so a product needs three MFMAs (hi*hi,
// lo*hi, hi*lo; the dropped lo*lo is 2^-22 relative) instead of the six of the 3 x bf16 split.  fp16 has 5 exponent bits: the
// producer picks the power-of-two scale 2^e (exact) that puts its values inside [2^-24, 65504]; consumers undo it in their
// epilogue.  Same panel layout as the bf16 planes (2-byte elements), planes 0 and 1 of a P3Mat.
typedef _Float16 p3_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 p3_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void h2_split2(float a, float b, unsigned& hi, unsigned& lo) {
  const p3_f32x2 v = {a, b};
  const p3_f16x2 h = __builtin_convertvector(v, p3_f16x2);
  const p3_f32x2 r1 = v - __builtin_convertvector(h, p3_f32x2);
  const p3_f16x2 l = __builtin_convertvector(r1, p3_f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
// store 8 consecutive values (already scaled) into the two fp16 planes: one 16-byte store per plane
__device__ __forceinline__ void h2_store8(const P3Mat& m, int64_t r, int c, const float (&v)[8]) {
  p3_u32x4 p[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h, l;
    h2_split2(v[2 * j], v[2 * j + 1], h, l);
    p[0][j] = h; p[1][j] = l;
  }
  __bf16* dst = m.p + p3_chunk_index(m.pr, r, c);
#pragma unroll
  for (int q = 0; q < 2; ++q) *reinterpret_cast<p3_u32x4*>(dst + q * m.ps) = p[q];
}
__device__ __forceinline__ void h2_store4(const P3Mat& m, int64_t r, int c, const float4& v) {
  unsigned h0, l0, h1, l1;
  h2_split2(v.x, v.y, h0, l0);
  h2_split2(v.z, v.w, h1, l1);
  __bf16* dst = m.p + p3_chunk_index(m.pr, r, c & ~7) + (c & 4);
  *reinterpret_cast<p3_u32x2*>(dst) = p3_u32x2{h0, h1};
  *reinterpret_cast<p3_u32x2*>(dst + m.ps) = p3_u32x2{l0, l1};
}
// ---- format-dispatching stores (fmt: 0 = three bf16 planes, 1 = two fp16 planes; wave-uniform).  The fp16 format has a finite
// range: a value beyond it sets PXR_STATUS_H2_RANGE in the status word (the host raises at its next check) instead of silently
// becoming inf.  `scale` (a power of two, h2 only) is applied before the split.
#define PXR_PLANES_BF16X3 0
#define PXR_PLANES_H2 1
__device__ __forceinline__ void px_store8(const P3Mat& m, int fmt, int32_t* status, int64_t r, int c, const float (&v)[8]) {
  if (fmt == PXR_PLANES_H2) {
    bool bad = false;          // per element: fmaxf would DROP a NaN and let it into the planes unflagged (advisor r4)
#pragma unroll
    for (int e = 0; e < 8; ++e) bad |= !(fabsf(v[e]) <= 65504.f);                // false for NaN and for |x| beyond the range
    if (bad && status) atomicOr(status, PXR_STATUS_H2_RANGE);
    h2_store8(m, r, c, v);
  } else {
    p3_store8(m, r, c, v);
  }
}
__device__ __forceinline__ void px_store4(const P3Mat& m, int fmt, int32_t* status, int64_t r, int c, const float4& v) {
  if (fmt == PXR_PLANES_H2) {
    const bool bad = !(fabsf(v.x) <= 65504.f) | !(fabsf(v.y) <= 65504.f) | !(fabsf(v.z) <= 65504.f) | !(fabsf(v.w) <= 65504.f);
    if (bad && status) atomicOr(status, PXR_STATUS_H2_RANGE);                    // NaN included (every comparison is false)
    h2_store4(m, r, c, v);
  } else {
    p3_store4(m, r, c, v);
  }
}
// ---- h2 planes under a scale that was NOT derived from the values being stored (round 6: the previous step's maximum of the same
// gradient with PXR_H2_STALE_HEADROOM binades of room, seqcore "stale scales"): v * scale is range-checked -- a value beyond the fp16
// range raises PXR_STATUS_H2_STALE (a NaN PXR_STATUS_H2_RANGE as everywhere) -- and then SATURATED to +-65504, so that an outlier
// step clips its largest gradient elements instead of writing inf into the planes (and NaN into the weights behind them).
__device__ __forceinline__ float h2_sat(float x) { return fminf(fmaxf(x, -65504.f), 65504.f); }
__device__ __forceinline__ void px_store4_h2s(const P3Mat& m, int32_t* status, int64_t r, int c, float4 v, float scale) {
  v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
  const bool nan = (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
  const bool big = (fabsf(v.x) > 65504.f) | (fabsf(v.y) > 65504.f) | (fabsf(v.z) > 65504.f) | (fabsf(v.w) > 65504.f);
  if ((nan | big) && status) atomicOr(status, nan ? PXR_STATUS_H2_RANGE : PXR_STATUS_H2_STALE);
  if (big) v = make_float4(h2_sat(v.x), h2_sat(v.y), h2_sat(v.z), h2_sat(v.w));
  h2_store4(m, r, c, v);
}
__device__ __forceinline__ void px_store8_h2s(const P3Mat& m, int32_t* status, int64_t r, int c, const float (&x)[8], float scale) {
  float v[8];
  bool nan = false, big = false;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] = x[e] * scale;
    nan |= v[e] != v[e];
    big |= fabsf(v[e]) > 65504.f;
  }
  if ((nan | big) && status) atomicOr(status, nan ? PXR_STATUS_H2_RANGE : PXR_STATUS_H2_STALE);
  if (big) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = h2_sat(v[e]);
  }
  h2_store8(m, r, c, v);
}
// px_store4 with streaming (non-temporal) stores on request: for planes far larger than the L2 that their reader -- a GEMM launched
// after this kernel has ended -- streams from HBM / MALL anyway (stream: wave-uniform)
__device__ __forceinline__ void px_store4s(const P3Mat& m, int fmt, int32_t* status, int64_t r, int c, const float4& v, bool stream) {
  if (!stream) { px_store4(m, fmt, status, r, c, v); return; }
  __bf16* dst = m.p + p3_chunk_index(m.pr, r, c & ~7) + (c & 4);
  if (fmt == PXR_PLANES_H2) {
    const bool bad = !(fabsf(v.x) <= 65504.f) | !(fabsf(v.y) <= 65504.f) | !(fabsf(v.z) <= 65504.f) | !(fabsf(v.w) <= 65504.f);
    if (bad && status) atomicOr(status, PXR_STATUS_H2_RANGE);
    unsigned h0, l0, h1, l1;
    h2_split2(v.x, v.y, h0, l0);
    h2_split2(v.z, v.w, h1, l1);
    __builtin_nontemporal_store(p3_u32x2{h0, h1}, reinterpret_cast<p3_u32x2*>(dst));
    __builtin_nontemporal_store(p3_u32x2{l0, l1}, reinterpret_cast<p3_u32x2*>(dst + m.ps));
  } else {
    unsigned h0, m0, l0, h1, m1, l1;
    p3_split2(v.x, v.y, h0, m0, l0);
    p3_split2(v.z, v.w, h1, m1, l1);
    __builtin_nontemporal_store(p3_u32x2{h0, h1}, reinterpret_cast<p3_u32x2*>(dst));
    __builtin_nontemporal_store(p3_u32x2{m0, m1}, reinterpret_cast<p3_u32x2*>(dst + m.ps));
    __builtin_nontemporal_store(p3_u32x2{l0, l1}, reinterpret_cast<p3_u32x2*>(dst + 2 * m.ps));
  }
}
static inline bool p3_mat_ok(const void* p, int64_t ps, int64_t pr, int64_t rows, int64_t cols) {
  return p == nullptr || (cols % 32 == 0 && pr % 32 == 0 && pr >= rows && ps >= pr * cols && ps % 8 == 0 && ((uintptr_t)p & 15) == 0);
}

}  // namespace pxr
