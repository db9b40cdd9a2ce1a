// tower_attn.hip -- fused self-attention of the visual item tower (CLIP ViT blocks: no mask, no dropout, head size 64),
// forward.  Replaces, per block, the three launches S = Q K^T (batched GEMM) -> softmax rows -> O = P V (batched GEMM)
// of model/vit_native.py::_attn_fwd, i.e. HF CLIPAttention.forward as the reference calls it through
// REC/model/modules.py (item tower) -- and with them the [images x heads, T, T] score matrix in HBM (665 MB per ViT-B/16
// block at 352 images: written once, read twice, rewritten once).
//
// One workgroup per (image, head); wave w owns the 32 queries 32 w .. 32 w + 31 (T = 197 -> 7 waves, T = 257 -> 9), keys
// stream through LDS in chunks of 32 shared by all waves.  Both contractions run on v_mfma_f32_32x32x16_bf16 with the exact
// bf16x3 split of their fp32 operands (hi | mid | lo, the 6 significant cross products, fp32 accumulation -- the same
// arithmetic as the GEMMs of this library), TRANSPOSED so that the softmax never leaves the registers:
//     S^T = K_chunk Q^T     C layout: lane = query, its 16 accumulator entries = 16 keys  -> the row max / row sum of a query
//                           are reductions over a lane's own registers plus ONE exchange with lane ^ 32
//     O^T += V_chunk^T P^T  P^T is needed as the B operand: lane = query, 8 keys per 16-key step -- exactly the entries the
//                           lane already holds, up to a fixed permutation of the keys inside a 16-key step, which is applied
//                           to V when it is staged (the contraction index order is free).  P never touches LDS or HBM.
// Online softmax (running max m, running sum l per query, O^T rescaled by 2^(m_old - m_new) per chunk) in the log2 domain.
// K and V are split into planes once per workgroup while they are staged (VALU on 2 x T x 64 values, against T^2 work);
// Q fragments are split once per wave and stay in registers.  The context leaves as fp32 and/or as bf16x3 planes in the
// panel layout the out-projection GEMM consumes (planes.cuh).
#include <math.h>

#include "planes.cuh"

namespace pxr {

typedef __bf16 ta_bf16x8 __attribute__((ext_vector_type(8)));
typedef float ta_f32x16 __attribute__((ext_vector_type(16)));
typedef float ta_f32x4 __attribute__((ext_vector_type(4)));   // (HIP's float4 is a struct of unions: conditional loads of it end up in scratch)

constexpr int TA_D = 64;        // head size
constexpr int TA_KC = 32;       // keys per chunk
constexpr int TA_KLD = 72;      // K chunk row stride (bf16): [key][dh], 144 B -> conflict-free 16-byte reads by lane = key
constexpr int TA_VLD = 40;      // V^T chunk row stride (bf16): [dh][key slot], 80 B
constexpr int TA_KPLANE = TA_KC * TA_KLD;        // elements per plane
constexpr int TA_VPLANE = TA_D * TA_VLD;
constexpr int TA_ITEMS = 256 + 128;              // staging work items per chunk: 256 for K (8 dh of a key), 128 for V (4 keys x 4 dh)

struct TowerAttnArgs {
  const float* q; const float* k; const float* v;   // element (image b, token t, head h, c) at p[(b*T + t)*ld + h*64 + c]
  int64_t ld;
  float* ctx; int64_t ld_ctx;                       // fp32 context [images*T, ld_ctx] (head h at column 64 h) or null
  P3Mat op;                                         // context as planes of the [images*T, heads*64] matrix, or p == null
  int op_fmt;                                       // PXR_PLANES_BF16X3 | PXR_PLANES_H2 (planes.cuh)
  int32_t* status;                                  // status word for the fp16 range check of the h2 format, or null
  float* lse;                                       // [images*heads, T] natural-log sum of exp of the scaled scores, or null
  int heads, T;
  float scale_log2;                                 // head_size^-0.5 * log2(e)
};

struct TaStage {          // one thread's share of a K/V chunk in flight between global memory and LDS
  ta_f32x4 r0, r1, r2, r3;
};

template <int NT>
__device__ __forceinline__ void ta_load(TaStage (&st)[(TA_ITEMS + NT - 1) / NT], const TowerAttnArgs& a, int64_t base, int key0) {
  constexpr int ITER = (TA_ITEMS + NT - 1) / NT;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int item = threadIdx.x + it * NT;
    const ta_f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (item < 256) {
      const int key = key0 + (item >> 3), c = (item & 7) * 8;
      const bool ok = key < a.T;
      const float* src = a.k + base + (int64_t)key * a.ld + c;
      st[it].r0 = ok ? *reinterpret_cast<const ta_f32x4*>(src) : z;
      st[it].r1 = ok ? *reinterpret_cast<const ta_f32x4*>(src + 4) : z;
      st[it].r2 = z; st[it].r3 = z;
    } else {
      const int j = (item - 256) & 127, kg = j & 7, c = (j >> 3) * 4;       // items >= TA_ITEMS load (and later drop) a duplicate
      const int key = key0 + kg * 4;
      const float* src = a.v + base + (int64_t)key * a.ld + c;
      st[it].r0 = (key < a.T) ? *reinterpret_cast<const ta_f32x4*>(src) : z;
      st[it].r1 = (key + 1 < a.T) ? *reinterpret_cast<const ta_f32x4*>(src + a.ld) : z;
      st[it].r2 = (key + 2 < a.T) ? *reinterpret_cast<const ta_f32x4*>(src + 2 * a.ld) : z;
      st[it].r3 = (key + 3 < a.T) ? *reinterpret_cast<const ta_f32x4*>(src + 3 * a.ld) : z;
    }
  }
}

// H2: the operands are split into TWO fp16 planes (planes.cuh "h2", unit scale: q, k, v are O(1) projections, p <= 1) and multiplied
// with three products on v_mfma_f32_32x32x16_f16 instead of six on the bf16 pipe -- the blocks whose GEMMs run on h2 operands
template <int NT, bool H2>
__device__ __forceinline__ void ta_store(const TaStage (&st)[(TA_ITEMS + NT - 1) / NT], __bf16* sK, __bf16* sV) {
  constexpr int ITER = (TA_ITEMS + NT - 1) / NT;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int item = threadIdx.x + it * NT;
    if (item < 256) {
      const int key = item >> 3, c = (item & 7) * 8;
      const float v[8] = {st[it].r0.x, st[it].r0.y, st[it].r0.z, st[it].r0.w, st[it].r1.x, st[it].r1.y, st[it].r1.z, st[it].r1.w};
      p3_u32x4 p[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned h, m, l = 0;
        if constexpr (H2) h2_split2(v[2 * e], v[2 * e + 1], h, m);
        else p3_split2(v[2 * e], v[2 * e + 1], h, m, l);
        p[0][e] = h; p[1][e] = m; p[2][e] = l;
      }
      __bf16* dst = sK + key * TA_KLD + c;
#pragma unroll
      for (int q = 0; q < (H2 ? 2 : 3); ++q) *reinterpret_cast<p3_u32x4*>(dst + q * TA_KPLANE) = p[q];
    } else if (item < TA_ITEMS) {
      // keys 4 kg .. 4 kg + 3 of dh c .. c + 3, transposed: per dh one 8-byte run of 4 keys.  Key groups are stored in the
      // order the P^T fragments hold them: inside a 16-key step  [0-3 | 8-11 | 4-7 | 12-15]  (slot group = kg with its two
      // low bits swapped).
      // (consecutive lanes = consecutive key groups: their 8-byte transposed writes land 2-4 words apart instead of 16 banks apart)
      const int j = item - 256, kg = j & 7, c = (j >> 3) * 4;
      const int sg = (kg & 4) | ((kg & 1) << 1) | ((kg >> 1) & 1);
      __bf16* dst = sV + c * TA_VLD + sg * 4;
      const TaStage& t = st[it];
#define PXR_TA_VROW(i, f)                                                            \
  {                                                                                  \
    unsigned h0, m0, l0 = 0, h1, m1, l1 = 0;                                         \
    if constexpr (H2) {                                                              \
      h2_split2(t.r0.f, t.r1.f, h0, m0);                                             \
      h2_split2(t.r2.f, t.r3.f, h1, m1);                                             \
    } else {                                                                         \
      p3_split2(t.r0.f, t.r1.f, h0, m0, l0);                                         \
      p3_split2(t.r2.f, t.r3.f, h1, m1, l1);                                         \
    }                                                                                \
    *reinterpret_cast<p3_u32x2*>(dst + (i) * TA_VLD) = p3_u32x2{h0, h1};             \
    *reinterpret_cast<p3_u32x2*>(dst + (i) * TA_VLD + TA_VPLANE) = p3_u32x2{m0, m1}; \
    if constexpr (!H2) *reinterpret_cast<p3_u32x2*>(dst + (i) * TA_VLD + 2 * TA_VPLANE) = p3_u32x2{l0, l1}; \
  }
      PXR_TA_VROW(0, x) PXR_TA_VROW(1, y) PXR_TA_VROW(2, z) PXR_TA_VROW(3, w)
#undef PXR_TA_VROW
    }
  }
}

// acc += sum over the 6 significant products of (a0 + a1 + a2) (b0 + b1 + b2), smallest terms first
__device__ __forceinline__ void ta_mma6(ta_f32x16& acc, const ta_bf16x8 (&a)[3], const ta_bf16x8 (&b)[3]) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
}

__device__ __forceinline__ void ta_split8(const float (&v)[8], ta_bf16x8 (&out)[3]) {
  p3_u32x4 p[3];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned h, m, l;
    p3_split2(v[2 * e], v[2 * e + 1], h, m, l);
    p[0][e] = h; p[1][e] = m; p[2][e] = l;
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) out[q] = __builtin_bit_cast(ta_bf16x8, p[q]);
}
// the fp16 two-plane flavour: (a0 + a1)(b0 + b1) without a1 b1 (2^-22 relative), small terms first; out[2] unused
__device__ __forceinline__ void ta_mma3h(ta_f32x16& acc, const ta_bf16x8 (&a)[3], const ta_bf16x8 (&b)[3]) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(p3_f16x8, a[1]), __builtin_bit_cast(p3_f16x8, b[0]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(p3_f16x8, a[0]), __builtin_bit_cast(p3_f16x8, b[1]), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(p3_f16x8, a[0]), __builtin_bit_cast(p3_f16x8, b[0]), acc, 0, 0, 0);
}
__device__ __forceinline__ void ta_split8h(const float (&v)[8], ta_bf16x8 (&out)[3]) {
  p3_u32x4 p[2];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned h, l;
    h2_split2(v[2 * e], v[2 * e + 1], h, l);
    p[0][e] = h; p[1][e] = l;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) out[q] = __builtin_bit_cast(ta_bf16x8, p[q]);
  out[2] = out[1];
}

template <int NW, bool H2 = false>
__global__ void __launch_bounds__(NW * 64) tower_attn_fwd_kernel(TowerAttnArgs a) {
  constexpr int NP = H2 ? 2 : 3;
  constexpr int NT = NW * 64;
  constexpr int ITER = (TA_ITEMS + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) __bf16 sK[2][3 * TA_KPLANE];
  __shared__ __attribute__((aligned(16))) __bf16 sV[2][3 * TA_VPLANE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.heads, hd = bh - b * a.heads;
  const int T = a.T;
  const int64_t base = (int64_t)b * T * a.ld + (int64_t)hd * TA_D;
  const int nchunk = (T + TA_KC - 1) / TA_KC;

  TaStage st[ITER];
  ta_load<NT>(st, a, base, 0);

  // this wave's queries as B fragments of S^T = K Q^T: lane = query r, k slots = dh 16 ks + 8 hh .. + 7
  const int q_row = wave * 32 + r;
  ta_bf16x8 qf[4][3];
  {
    const bool ok = q_row < T;
    const float* src = a.q + base + (int64_t)q_row * a.ld + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ta_f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
      if (ok) { x0 = *reinterpret_cast<const ta_f32x4*>(src + ks * 16); x1 = *reinterpret_cast<const ta_f32x4*>(src + ks * 16 + 4); }
      const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      if constexpr (H2) ta_split8h(v, qf[ks]);
      else ta_split8(v, qf[ks]);
    }
  }
  ta_store<NT, H2>(st, sK[0], sV[0]);
  __syncthreads();

  ta_f32x16 accO[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) accO[j][e] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;     // l_run: this lane's keys only (the two half-waves are added at the end)

  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) ta_load<NT>(st, a, base, (c + 1) * TA_KC);
    // ---- S^T chunk: rows = keys (A operand from LDS: lane = key r), columns = this wave's queries
    ta_f32x16 s;
#pragma unroll
    for (int e = 0; e < 16; ++e) s[e] = 0.f;
    const __bf16* kb = sK[buf] + r * TA_KLD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ta_bf16x8 kf[3];
#pragma unroll
      for (int q = 0; q < NP; ++q) kf[q] = *reinterpret_cast<const ta_bf16x8*>(kb + q * TA_KPLANE + ks * 16);
      if constexpr (H2) ta_mma3h(s, kf, qf[ks]);
      else ta_mma6(s, kf, qf[ks]);
    }
    // ---- online softmax of the lane's 16 keys: key(e) = 32 c + (e & 3) + 8 (e >> 2) + 4 hh
    float mx = -INFINITY;
    const bool tail = (c + 1) * TA_KC > T;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float x = s[e] * a.scale_log2;
      if (tail && c * TA_KC + (e & 3) + 8 * (e >> 2) + 4 * hh >= T) x = -INFINITY;
      s[e] = x;
      mx = fmaxf(mx, x);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = __builtin_amdgcn_exp2f(s[e] - m_new);
      psum += s[e];
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) accO[j][e] *= alpha;
    // ---- O^T += V^T P^T: entries 8 ks2 .. 8 ks2 + 7 are the lane's 8 key slots of 16-key step ks2
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const float pv[8] = {s[8 * ks2], s[8 * ks2 + 1], s[8 * ks2 + 2], s[8 * ks2 + 3],
                           s[8 * ks2 + 4], s[8 * ks2 + 5], s[8 * ks2 + 6], s[8 * ks2 + 7]};
      ta_bf16x8 pf[3];
      if constexpr (H2) ta_split8h(pv, pf);
      else ta_split8(pv, pf);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ta_bf16x8 vf[3];
        const __bf16* vb = sV[buf] + (j * 32 + r) * TA_VLD + ks2 * 16 + hh * 8;
#pragma unroll
        for (int q = 0; q < NP; ++q) vf[q] = *reinterpret_cast<const ta_bf16x8*>(vb + q * TA_VPLANE);
        if constexpr (H2) ta_mma3h(accO[j], vf, pf);
        else ta_mma6(accO[j], vf, pf);
      }
    }
    if (c + 1 < nchunk) ta_store<NT, H2>(st, sK[buf ^ 1], sV[buf ^ 1]);
    __syncthreads();
  }

  // ---- finish: O[q][dh] = O^T / l.  Entry e of block j is dh = 32 j + (e & 3) + 8 (e >> 2) + 4 hh: the two half-waves swap
  // 4-value groups so that each lane owns 8 consecutive dh of two 8-groups (g = 2 t + hh), one 32-byte run / 16-byte plane chunk
  const float l = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l;
  const bool ok = q_row < T;
  if (a.lse && ok && hh == 0) a.lse[(int64_t)bh * T + q_row] = (m_run + log2f(l)) * 0.6931471805599453f;
  const int64_t orow = (int64_t)b * T + q_row;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float own[4], got[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo_g = accO[j][4 * (2 * t) + i] * inv, hi_g = accO[j][4 * (2 * t + 1) + i] * inv;
        own[i] = hh ? hi_g : lo_g;
        got[i] = __shfl_xor(hh ? lo_g : hi_g, 32);
      }
      float v8[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { v8[i] = hh ? got[i] : own[i]; v8[4 + i] = hh ? own[i] : got[i]; }
      const int col = hd * TA_D + j * 32 + 8 * (2 * t + hh);
      if (ok) {
        if (a.ctx) {
          float* dst = a.ctx + orow * a.ld_ctx + col;
          *reinterpret_cast<ta_f32x4*>(dst) = ta_f32x4{v8[0], v8[1], v8[2], v8[3]};
          *reinterpret_cast<ta_f32x4*>(dst + 4) = ta_f32x4{v8[4], v8[5], v8[6], v8[7]};
        }
        if (a.op.p) px_store8(a.op, a.op_fmt, a.status, orow, col, v8);
      }
    }
}

// =====================================================================================================================
// Backward (the trainable blocks of the tower): dQ, dK, dV from dO, recomputing P from the saved log-sum-exp -- the
// [images x heads, T, T] probability matrix is neither saved by the forward nor written by the backward.
//   delta_q = sum_c dO[q,c] O[q,c]          (= sum_k P[q,k] dP[q,k]: the softmax Jacobian's row term)
//   P = exp(scale S - lse),  dP = dO V^T,  dS = P o (dP - delta) * scale,   dQ = dS K,  dK = dS^T Q,  dV = P^T dO
// Two launches, both with the forward's structure (one workgroup per (image, head), a wave owns a block of 32 rows, the other
// side streams through LDS in chunks of 32 rows, every contraction on the bf16x3 MFMA path, C-layout accumulators re-used
// as B operands so that P / dS never leave the registers):
//   tower_attn_bwd_dq_kernel   wave = 32 QUERIES (as in the forward: S^T, dP^T with lane = query), keys stream;
//                              dQ^T += K^T dS^T.  Also writes delta.
//   tower_attn_bwd_dkv_kernel  wave = 32 KEYS (S, dP with lane = key), queries stream with their lse / delta;
//                              dV^T += dO^T P,  dK^T += Q^T dS.
// Cost: 72 + 96 MFMA groups per (32 x 32) block pair against 48 in the forward.
struct TowerBwdArgs {
  const float* q; const float* k; const float* v; int64_t ld;      // as in the forward
  const float* dctx; const float* ctx; int64_t ld_c;                // dO and O: [images*T, ld_c], head h at column 64 h
  const float* lse;                                                 // [images*heads, T] from the forward
  float* delta;                                                     // [images*heads, T] workspace: written by dq, read by dkv
  float* dq; float* dk; float* dv; int64_t ld_d;                    // outputs, element (b, t, h, c) at p[(b*T + t)*ld_d + 64 h + c]
  int heads, T;
  float scale, scale_log2;
};

struct TaBlk { ta_f32x4 r0, r1, r2, r3; };     // rows 4 kg .. 4 kg + 3, columns c .. c + 3 of a 32 x 64 chunk

__device__ __forceinline__ void ta_blk_load(TaBlk& t, const float* src, int64_t ld, int row0, int T, int j) {
  const int kg = j & 7, c = (j >> 3) * 4, row = row0 + 4 * kg;
  const float* p = src + (int64_t)row * ld + c;
  const ta_f32x4 z = {0.f, 0.f, 0.f, 0.f};
  t.r0 = (row < T) ? *reinterpret_cast<const ta_f32x4*>(p) : z;
  t.r1 = (row + 1 < T) ? *reinterpret_cast<const ta_f32x4*>(p + ld) : z;
  t.r2 = (row + 2 < T) ? *reinterpret_cast<const ta_f32x4*>(p + 2 * ld) : z;
  t.r3 = (row + 3 < T) ? *reinterpret_cast<const ta_f32x4*>(p + 3 * ld) : z;
}
// kc: [row][dh] planes (A operand, k = dh);  tr: [dh][row slot] planes (A operand, k = rows in fragment order); either may be null
__device__ __forceinline__ void ta_blk_store(const TaBlk& t, int j, __bf16* kc, __bf16* tr) {
  const int kg = j & 7, c = (j >> 3) * 4;       // lanes of a wave: 8 row groups x 8 column quads (transposed writes conflict-free)
  if (kc) {
    __bf16* dst = kc + (4 * kg) * TA_KLD + c;
#define PXR_TA_KROW(i, R)                                                                    \
  {                                                                                          \
    unsigned h0, m0, l0, h1, m1, l1;                                                         \
    p3_split2(t.R.x, t.R.y, h0, m0, l0);                                                     \
    p3_split2(t.R.z, t.R.w, h1, m1, l1);                                                     \
    *reinterpret_cast<p3_u32x2*>(dst + (i) * TA_KLD) = p3_u32x2{h0, h1};                     \
    *reinterpret_cast<p3_u32x2*>(dst + (i) * TA_KLD + TA_KPLANE) = p3_u32x2{m0, m1};         \
    *reinterpret_cast<p3_u32x2*>(dst + (i) * TA_KLD + 2 * TA_KPLANE) = p3_u32x2{l0, l1};     \
  }
    PXR_TA_KROW(0, r0) PXR_TA_KROW(1, r1) PXR_TA_KROW(2, r2) PXR_TA_KROW(3, r3)
#undef PXR_TA_KROW
  }
  if (tr) {
    const int sg = (kg & 4) | ((kg & 1) << 1) | ((kg >> 1) & 1);
    __bf16* dst = tr + c * TA_VLD + sg * 4;
#define PXR_TA_TROW(i, f)                                                                    \
  {                                                                                          \
    unsigned h0, m0, l0, h1, m1, l1;                                                         \
    p3_split2(t.r0.f, t.r1.f, h0, m0, l0);                                                   \
    p3_split2(t.r2.f, t.r3.f, h1, m1, l1);                                                   \
    *reinterpret_cast<p3_u32x2*>(dst + (i) * TA_VLD) = p3_u32x2{h0, h1};                     \
    *reinterpret_cast<p3_u32x2*>(dst + (i) * TA_VLD + TA_VPLANE) = p3_u32x2{m0, m1};         \
    *reinterpret_cast<p3_u32x2*>(dst + (i) * TA_VLD + 2 * TA_VPLANE) = p3_u32x2{l0, l1};     \
  }
    PXR_TA_TROW(0, x) PXR_TA_TROW(1, y) PXR_TA_TROW(2, z) PXR_TA_TROW(3, w)
#undef PXR_TA_TROW
  }
}

// B fragments of a wave's own 32 rows (lane = row r, k slots = dh 16 ks + 8 hh ..): rows >= T are zero
__device__ __forceinline__ void ta_row_frags(ta_bf16x8 (&f)[4][3], const float* src, bool ok, int hh) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    ta_f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
    if (ok) { x0 = *reinterpret_cast<const ta_f32x4*>(src + ks * 16 + hh * 8); x1 = *reinterpret_cast<const ta_f32x4*>(src + ks * 16 + hh * 8 + 4); }
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    ta_split8(v, f[ks]);
  }
}

// acc (two 32 x 32 blocks of a TRANSPOSED result: lane = output row, entries = 64 columns) -> dst[row][0 .. 63] fp32
__device__ __forceinline__ void ta_store_rows(const ta_f32x16 (&acc)[2], float* dst, bool ok, int hh) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float own[4], got[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo_g = acc[j][4 * (2 * t) + i], hi_g = acc[j][4 * (2 * t + 1) + i];
        own[i] = hh ? hi_g : lo_g;
        got[i] = __shfl_xor(hh ? lo_g : hi_g, 32);
      }
      if (ok) {
        float* d = dst + j * 32 + 8 * (2 * t + hh);
        *reinterpret_cast<ta_f32x4*>(d) = hh ? ta_f32x4{got[0], got[1], got[2], got[3]} : ta_f32x4{own[0], own[1], own[2], own[3]};
        *reinterpret_cast<ta_f32x4*>(d + 4) = hh ? ta_f32x4{own[0], own[1], own[2], own[3]} : ta_f32x4{got[0], got[1], got[2], got[3]};
      }
    }
}

constexpr int TB_ITEMS = 256;                                             // two matrices x 128 blocks of 4 x 4 per chunk
constexpr int TB_DQ_BUF = 2 * 3 * TA_KPLANE + 3 * TA_VPLANE;              // K kc | V kc | K tr        (bf16 elements)
constexpr int TB_DKV_BUF = 2 * 3 * TA_KPLANE + 2 * 3 * TA_VPLANE + 128;   // Q kc | dO kc | Q tr | dO tr | lse2[32] delta[32] (as 128 bf16 slots)

template <int NW>
__global__ void __launch_bounds__(NW * 64) tower_attn_bwd_dq_kernel(TowerBwdArgs a) {
  constexpr int NT = NW * 64;
  constexpr int ITER = (TB_ITEMS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char ta_smem[];
  __bf16* sm = reinterpret_cast<__bf16*>(ta_smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.heads, hd = bh - b * a.heads;
  const int T = a.T;
  const int64_t base = (int64_t)b * T * a.ld + (int64_t)hd * TA_D;
  const int64_t cbase = (int64_t)b * T * a.ld_c + (int64_t)hd * TA_D;
  const int nchunk = (T + TA_KC - 1) / TA_KC;
  auto kkc = [&](int buf) { return sm + buf * TB_DQ_BUF; };
  auto vkc = [&](int buf) { return sm + buf * TB_DQ_BUF + 3 * TA_KPLANE; };
  auto ktr = [&](int buf) { return sm + buf * TB_DQ_BUF + 6 * TA_KPLANE; };
  TaBlk st[ITER];
  auto load = [&](int key0) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int item = threadIdx.x + it * NT;
      if (item < 128) ta_blk_load(st[it], a.k + base, a.ld, key0, T, item);
      else if (item < TB_ITEMS) ta_blk_load(st[it], a.v + base, a.ld, key0, T, item - 128);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int item = threadIdx.x + it * NT;
      if (item < 128) ta_blk_store(st[it], item, kkc(buf), ktr(buf));
      else if (item < TB_ITEMS) ta_blk_store(st[it], item - 128, vkc(buf), nullptr);
    }
  };
  load(0);

  const int q_row = wave * 32 + r;
  const bool ok = q_row < T;
  ta_bf16x8 qf[4][3], dof[4][3];
  ta_row_frags(qf, a.q + base + (int64_t)q_row * a.ld, ok, hh);
  ta_row_frags(dof, a.dctx + cbase + (int64_t)q_row * a.ld_c, ok, hh);
  float dsum = 0.f;
  if (ok) {
    const float* po = a.ctx + cbase + (int64_t)q_row * a.ld_c + hh * 8;
    const float* pd = a.dctx + cbase + (int64_t)q_row * a.ld_c + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const ta_f32x4 o0 = *reinterpret_cast<const ta_f32x4*>(po + ks * 16), o1 = *reinterpret_cast<const ta_f32x4*>(po + ks * 16 + 4);
      const ta_f32x4 d0 = *reinterpret_cast<const ta_f32x4*>(pd + ks * 16), d1 = *reinterpret_cast<const ta_f32x4*>(pd + ks * 16 + 4);
      dsum += (o0.x * d0.x + o0.y * d0.y) + (o0.z * d0.z + o0.w * d0.w) + (o1.x * d1.x + o1.y * d1.y) + (o1.z * d1.z + o1.w * d1.w);
    }
  }
  const float delta = dsum + __shfl_xor(dsum, 32);
  const float lse2 = ok ? a.lse[(int64_t)bh * T + q_row] * 1.4426950408889634f : 0.f;
  if (ok && hh == 0) a.delta[(int64_t)bh * T + q_row] = delta;
  store(0);
  __syncthreads();

  ta_f32x16 accQ[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) accQ[j][e] = 0.f;

  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) load((c + 1) * TA_KC);
    ta_f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
    const __bf16* kb = kkc(buf) + r * TA_KLD + hh * 8;
    const __bf16* vb = vkc(buf) + r * TA_KLD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ta_bf16x8 f[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) f[q] = *reinterpret_cast<const ta_bf16x8*>(kb + q * TA_KPLANE + ks * 16);
      ta_mma6(s, f, qf[ks]);
#pragma unroll
      for (int q = 0; q < 3; ++q) f[q] = *reinterpret_cast<const ta_bf16x8*>(vb + q * TA_KPLANE + ks * 16);
      ta_mma6(dp, f, dof[ks]);
    }
    // dS^T of the lane's query against its 16 keys: key(e) = 32 c + (e & 3) + 8 (e >> 2) + 4 hh
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const bool real = c * TA_KC + (e & 3) + 8 * (e >> 2) + 4 * hh < T;
      const float p = real ? __builtin_amdgcn_exp2f(fmaf(s[e], a.scale_log2, -lse2)) : 0.f;
      s[e] = p * (dp[e] - delta) * a.scale;
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const float dv8[8] = {s[8 * ks2], s[8 * ks2 + 1], s[8 * ks2 + 2], s[8 * ks2 + 3],
                            s[8 * ks2 + 4], s[8 * ks2 + 5], s[8 * ks2 + 6], s[8 * ks2 + 7]};
      ta_bf16x8 dsf[3];
      ta_split8(dv8, dsf);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ta_bf16x8 f[3];
        const __bf16* tb = ktr(buf) + (j * 32 + r) * TA_VLD + ks2 * 16 + hh * 8;
#pragma unroll
        for (int q = 0; q < 3; ++q) f[q] = *reinterpret_cast<const ta_bf16x8*>(tb + q * TA_VPLANE);
        ta_mma6(accQ[j], f, dsf);
      }
    }
    if (c + 1 < nchunk) store(buf ^ 1);
    __syncthreads();
  }
  ta_store_rows(accQ, a.dq + ((int64_t)b * T + q_row) * a.ld_d + (int64_t)hd * TA_D, ok, hh);
}

template <int NW>
__global__ void __launch_bounds__(NW * 64) tower_attn_bwd_dkv_kernel(TowerBwdArgs a) {
  constexpr int NT = NW * 64;
  constexpr int ITER = (TB_ITEMS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char ta_smem[];
  __bf16* sm = reinterpret_cast<__bf16*>(ta_smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.heads, hd = bh - b * a.heads;
  const int T = a.T;
  const int64_t base = (int64_t)b * T * a.ld + (int64_t)hd * TA_D;
  const int64_t cbase = (int64_t)b * T * a.ld_c + (int64_t)hd * TA_D;
  const int nchunk = (T + TA_KC - 1) / TA_KC;
  auto qkc = [&](int buf) { return sm + buf * TB_DKV_BUF; };
  auto dkc = [&](int buf) { return sm + buf * TB_DKV_BUF + 3 * TA_KPLANE; };
  auto qtr = [&](int buf) { return sm + buf * TB_DKV_BUF + 6 * TA_KPLANE; };
  auto dtr = [&](int buf) { return sm + buf * TB_DKV_BUF + 6 * TA_KPLANE + 3 * TA_VPLANE; };
  auto stat = [&](int buf) { return reinterpret_cast<float*>(sm + buf * TB_DKV_BUF + 6 * TA_KPLANE + 6 * TA_VPLANE); };   // lse2[32] | delta[32]
  TaBlk st[ITER];
  float st_stat = 0.f;
  auto load = [&](int q0) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int item = threadIdx.x + it * NT;
      if (item < 128) ta_blk_load(st[it], a.q + base, a.ld, q0, T, item);
      else if (item < TB_ITEMS) ta_blk_load(st[it], a.dctx + cbase, a.ld_c, q0, T, item - 128);
    }
    if (threadIdx.x < 64) {      // lane < 32: lse (as log2, +inf for rows beyond T => P = 0); lanes 32..63: delta
      const int q = q0 + (threadIdx.x & 31);
      st_stat = (threadIdx.x < 32) ? (q < T ? a.lse[(int64_t)bh * T + q] * 1.4426950408889634f : INFINITY)
                                   : (q < T ? a.delta[(int64_t)bh * T + q] : 0.f);
    }
  };
  auto store = [&](int buf) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int item = threadIdx.x + it * NT;
      if (item < 128) ta_blk_store(st[it], item, qkc(buf), qtr(buf));
      else if (item < TB_ITEMS) ta_blk_store(st[it], item - 128, dkc(buf), dtr(buf));
    }
    if (threadIdx.x < 64) stat(buf)[threadIdx.x] = st_stat;
  };
  load(0);

  const int k_row = wave * 32 + r;
  const bool ok = k_row < T;
  ta_bf16x8 kf[4][3], vf[4][3];
  ta_row_frags(kf, a.k + base + (int64_t)k_row * a.ld, ok, hh);
  ta_row_frags(vf, a.v + base + (int64_t)k_row * a.ld, ok, hh);
  store(0);
  __syncthreads();

  ta_f32x16 accK[2], accV[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) { accK[j][e] = 0.f; accV[j][e] = 0.f; }

  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunk) load((c + 1) * TA_KC);
    // S and dP of the lane's key against the chunk's 32 queries: query(e) = (e & 3) + 8 (e >> 2) + 4 hh
    ta_f32x16 s, dp;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
    const __bf16* qb = qkc(buf) + r * TA_KLD + hh * 8;
    const __bf16* db = dkc(buf) + r * TA_KLD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      ta_bf16x8 f[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) f[q] = *reinterpret_cast<const ta_bf16x8*>(qb + q * TA_KPLANE + ks * 16);
      ta_mma6(s, f, kf[ks]);
#pragma unroll
      for (int q = 0; q < 3; ++q) f[q] = *reinterpret_cast<const ta_bf16x8*>(db + q * TA_KPLANE + ks * 16);
      ta_mma6(dp, f, vf[ks]);
    }
    const float* stq = stat(buf);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int q = (e & 3) + 8 * (e >> 2) + 4 * hh;
      const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[e], a.scale_log2, -stq[q])) : 0.f;
      s[e] = p;
      dp[e] = p * (dp[e] - stq[32 + q]) * a.scale;
    }
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const float p8[8] = {s[8 * ks2], s[8 * ks2 + 1], s[8 * ks2 + 2], s[8 * ks2 + 3], s[8 * ks2 + 4], s[8 * ks2 + 5], s[8 * ks2 + 6], s[8 * ks2 + 7]};
      const float d8[8] = {dp[8 * ks2], dp[8 * ks2 + 1], dp[8 * ks2 + 2], dp[8 * ks2 + 3], dp[8 * ks2 + 4], dp[8 * ks2 + 5], dp[8 * ks2 + 6], dp[8 * ks2 + 7]};
      ta_bf16x8 pf[3], dsf[3];
      ta_split8(p8, pf);
      ta_split8(d8, dsf);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        ta_bf16x8 f[3];
        const __bf16* tb = dtr(buf) + (j * 32 + r) * TA_VLD + ks2 * 16 + hh * 8;
#pragma unroll
        for (int q = 0; q < 3; ++q) f[q] = *reinterpret_cast<const ta_bf16x8*>(tb + q * TA_VPLANE);
        ta_mma6(accV[j], f, pf);                                        // dV^T += dO^T P
        const __bf16* tq = qtr(buf) + (j * 32 + r) * TA_VLD + ks2 * 16 + hh * 8;
#pragma unroll
        for (int q = 0; q < 3; ++q) f[q] = *reinterpret_cast<const ta_bf16x8*>(tq + q * TA_VPLANE);
        ta_mma6(accK[j], f, dsf);                                       // dK^T += Q^T dS
      }
    }
    if (c + 1 < nchunk) store(buf ^ 1);
    __syncthreads();
  }
  const int64_t orow = ((int64_t)b * T + k_row) * a.ld_d + (int64_t)hd * TA_D;
  ta_store_rows(accK, a.dk + orow, ok, hh);
  ta_store_rows(accV, a.dv + orow, ok, hh);
}

}  // namespace pxr

using namespace pxr;

constexpr int TA_MAXW = 9;

// 1 when pxr_tower_attn_fwd_f32 serves this shape
extern "C" int pxr_tower_attn_supported(int T, int d) { return (d == TA_D && T >= 1 && T <= 32 * TA_MAXW) ? 1 : 0; }

// ctx[b*T + t, 64 h .. 64 h + 63] = softmax_t'(scale * q_t . k_t') v_t'   per (image b, head h); no mask, no dropout.
// q/k/v: fp32, element (b, t, h, c) at p[(b*T + t)*ld + 64 h + c] (the fused projection output is passed with three base
// pointers).  Outputs: ctx fp32 and/or ctx planes (planes.cuh) -- at least one; lse optional ([images*heads, T]).
static int tower_attn_fwd_impl(const float* q, const float* k, const float* v, int64_t ld, int64_t images, int heads, int T, int d,
                               float scale, float* ctx, int64_t ld_ctx, void* ctx_planes, int64_t c_ps, int64_t c_pr, int c_fmt,
                               float* lse, void* stream);
extern "C" int pxr_tower_attn_fwd_f32(const float* q, const float* k, const float* v, int64_t ld, int64_t images, int heads,
                                      int T, int d, float scale, float* ctx, int64_t ld_ctx, void* ctx_planes, int64_t c_ps,
                                      int64_t c_pr, float* lse, void* stream) {
  return tower_attn_fwd_impl(q, k, v, ld, images, heads, T, d, scale, ctx, ld_ctx, ctx_planes, c_ps, c_pr, PXR_PLANES_BF16X3, lse, stream);
}
// the same with the context planes in the two-plane fp16 format (planes.cuh "h2", unit scale; the operand of pxr_gemm_h2_f32)
extern "C" int pxr_tower_attn_fwd_h2_f32(const float* q, const float* k, const float* v, int64_t ld, int64_t images, int heads,
                                         int T, int d, float scale, float* ctx, int64_t ld_ctx, void* ctx_planes, int64_t c_ps,
                                         int64_t c_pr, float* lse, void* stream) {
  PXR_REQUIRE(ctx_planes, "pxr_tower_attn_fwd_h2_f32: no planes");
  return tower_attn_fwd_impl(q, k, v, ld, images, heads, T, d, scale, ctx, ld_ctx, ctx_planes, c_ps, c_pr, PXR_PLANES_H2, lse, stream);
}
static int tower_attn_fwd_impl(const float* q, const float* k, const float* v, int64_t ld, int64_t images, int heads, int T, int d,
                               float scale, float* ctx, int64_t ld_ctx, void* ctx_planes, int64_t c_ps, int64_t c_pr, int c_fmt,
                               float* lse, void* stream) {
  PXR_REQUIRE(q && k && v && images >= 0 && heads > 0, "pxr_tower_attn_fwd_f32: bad args");
  PXR_REQUIRE(pxr_tower_attn_supported(T, d), "pxr_tower_attn_fwd_f32: head size %d / %d tokens not supported (64, <= %d)", d,
              T, 32 * TA_MAXW);
  PXR_REQUIRE(ctx || ctx_planes, "pxr_tower_attn_fwd_f32: no output");
  PXR_REQUIRE(ld % 4 == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "pxr_tower_attn_fwd_f32: q/k/v alignment");
  PXR_REQUIRE(!ctx || (ld_ctx % 4 == 0 && ((uintptr_t)ctx & 15) == 0 && ld_ctx >= (int64_t)heads * d),
              "pxr_tower_attn_fwd_f32: ctx layout");
  PXR_REQUIRE(p3_mat_ok(ctx_planes, c_ps, c_pr, images * T, (int64_t)heads * d), "pxr_tower_attn_fwd_f32: ctx planes layout");
  if (images == 0) return PXR_OK;
  PXR_REQUIRE(images * heads < (1ll << 31), "pxr_tower_attn_fwd_f32: too many (image, head) pairs");
  TowerAttnArgs a;
  a.q = q; a.k = k; a.v = v; a.ld = ld;
  a.ctx = ctx; a.ld_ctx = ld_ctx;
  a.op = P3Mat{(__bf16*)ctx_planes, c_ps, c_pr};
  a.op_fmt = c_fmt;
  a.status = pxr_status_word();
  a.lse = lse;
  a.heads = heads; a.T = T;
  a.scale_log2 = scale * 1.4426950408889634f;
  const dim3 grid((unsigned)(images * heads));
  hipStream_t s = (hipStream_t)stream;
  // h2 context planes = a block whose GEMMs run on fp16 two-plane operands: so do the two contractions here (PXR_TOWER_ATTN_H2=0:
  // only the output format changes, the contractions stay on the six bf16 products)
  static const int env_h2 = getenv("PXR_TOWER_ATTN_H2") ? atoi(getenv("PXR_TOWER_ATTN_H2")) : 1;
  const bool h2_math = (c_fmt == PXR_PLANES_H2) && env_h2;
  switch ((T + 31) / 32) {
#define PXR_TA_CASE(NW)                                                                                           \
  case NW:                                                                                                        \
    if (h2_math) hipLaunchKernelGGL((tower_attn_fwd_kernel<NW, true>), grid, dim3(NW * 64), 0, s, a);             \
    else hipLaunchKernelGGL((tower_attn_fwd_kernel<NW, false>), grid, dim3(NW * 64), 0, s, a);                    \
    break;
    PXR_TA_CASE(1) PXR_TA_CASE(2) PXR_TA_CASE(3) PXR_TA_CASE(4) PXR_TA_CASE(5) PXR_TA_CASE(6) PXR_TA_CASE(7) PXR_TA_CASE(8)
    PXR_TA_CASE(9)
#undef PXR_TA_CASE
    default: PXR_REQUIRE(false, "pxr_tower_attn_fwd_f32: unreachable");
  }
  return pxr_check_launch("pxr_tower_attn_fwd_f32");
}

// d(q | k | v) of pxr_tower_attn_fwd_f32 from dctx, the forward's ctx and lse (both fp32).  delta_ws: [images*heads, T] floats
// of scratch.  dq / dk / dv use the q / k / v addressing with row stride ld_d (they may be three column ranges of one matrix).
extern "C" int pxr_tower_attn_bwd_f32(const float* q, const float* k, const float* v, int64_t ld, const float* dctx,
                                      const float* ctx, int64_t ld_c, const float* lse, int64_t images, int heads, int T, int d,
                                      float scale, float* dq, float* dk, float* dv, int64_t ld_d, float* delta_ws, void* stream) {
  PXR_REQUIRE(q && k && v && dctx && ctx && lse && dq && dk && dv && delta_ws && images >= 0 && heads > 0,
              "pxr_tower_attn_bwd_f32: bad args");
  PXR_REQUIRE(pxr_tower_attn_supported(T, d), "pxr_tower_attn_bwd_f32: head size %d / %d tokens not supported (64, <= %d)", d,
              T, 32 * TA_MAXW);
  PXR_REQUIRE(ld % 4 == 0 && ld_c % 4 == 0 && ld_d % 4 == 0, "pxr_tower_attn_bwd_f32: row strides must be multiples of 4");
  PXR_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dctx | (uintptr_t)ctx | (uintptr_t)dq | (uintptr_t)dk |
                (uintptr_t)dv) & 15) == 0, "pxr_tower_attn_bwd_f32: 16-byte alignment");
  if (images == 0) return PXR_OK;
  PXR_REQUIRE(images * heads < (1ll << 31), "pxr_tower_attn_bwd_f32: too many (image, head) pairs");
  TowerBwdArgs a;
  a.q = q; a.k = k; a.v = v; a.ld = ld; a.dctx = dctx; a.ctx = ctx; a.ld_c = ld_c; a.lse = lse; a.delta = delta_ws;
  a.dq = dq; a.dk = dk; a.dv = dv; a.ld_d = ld_d; a.heads = heads; a.T = T;
  a.scale = scale; a.scale_log2 = scale * 1.4426950408889634f;
  const dim3 grid((unsigned)(images * heads));
  hipStream_t s = (hipStream_t)stream;
  constexpr int lds_dq = 2 * TB_DQ_BUF * 2, lds_dkv = 2 * TB_DKV_BUF * 2;      // two buffers of bf16
  switch ((T + 31) / 32) {
#define PXR_TB_CASE(NW)                                                                                                     \
  case NW: {                                                                                                                \
    static bool attr = false;                                                                                               \
    if (!attr) {                                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(tower_attn_bwd_dq_kernel<NW>),                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_dq) != hipSuccess ||                          \
          hipFuncSetAttribute(reinterpret_cast<const void*>(tower_attn_bwd_dkv_kernel<NW>),                                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_dkv) != hipSuccess) {                         \
        (void)hipGetLastError();                                                                                            \
        pxr_set_error("pxr_tower_attn_bwd_f32: cannot reserve %d bytes of LDS", lds_dkv);                                   \
        return PXR_ERR_LAUNCH;                                                                                              \
      }                                                                                                                     \
      attr = true;                                                                                                          \
    }                                                                                                                       \
    hipLaunchKernelGGL(tower_attn_bwd_dq_kernel<NW>, grid, dim3(NW * 64), lds_dq, s, a);                                    \
    hipLaunchKernelGGL(tower_attn_bwd_dkv_kernel<NW>, grid, dim3(NW * 64), lds_dkv, s, a);                                  \
  } break;
    PXR_TB_CASE(1) PXR_TB_CASE(2) PXR_TB_CASE(3) PXR_TB_CASE(4) PXR_TB_CASE(5) PXR_TB_CASE(6) PXR_TB_CASE(7) PXR_TB_CASE(8)
    PXR_TB_CASE(9)
#undef PXR_TB_CASE
    default: PXR_REQUIRE(false, "pxr_tower_attn_bwd_f32: unreachable");
  }
  return pxr_check_launch("pxr_tower_attn_bwd_f32");
}
