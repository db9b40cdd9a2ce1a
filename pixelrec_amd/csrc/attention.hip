// attention.hip -- SASRec masked multi-head self-attention core, forward and backward, one workgroup per
// (batch, head).  Replaces layers.py:590-612 (scores = QK^T / sqrt(d) + mask; softmax; dropout; PV; head merge)
// together with the mask construction of sasrec.py:119-126, and their autograd.
//
// Shapes are tiny (L <= 128 keys, d = D/H per head): the work is latency/LDS-bound (0.33 GFLOP per layer at B=64,
// SURVEY.md §8 a6).  Kernel families in this file, picked by pxr_attn_{fwd,bwd}_f32:
//   attn_{fwd,bwd}_mfma1_kernel<NW>  L <= 64, d <= 128: every operand staged once, back-to-back MFMA phases (default, 8 waves)
//   attn_{fwd,bwd}_mfma_kernel      L <= 64, any d % 8 == 0: d-chunked (emb 4096: d = 1024)
//   attn_{fwd,bwd}_long_kernel      65 <= L <= 128, d % 8 == 0: 128-row tiles, two keys per lane
//   attn_{fwd,bwd}_kernel<NW>       L <= 64, d % 4 == 0: VALU fallback (PXR_ATTN_MFMA=0 or d % 8 != 0)
// Common design:
//   * q/k/v are read in place from the fused QKV projection output [B, L, 3D] (row stride `ld`), the context
//     is written head-merged into [B, L, D]: no permute/contiguous copies (layers.py:590-592,610-612);
//   * Q/K (then V) tiles are staged through LDS in d-chunks of <= 128 columns, so any head size works
//     (d = 32 for emb 128, 128 for emb 512, 1024 for emb 4096) with a fixed LDS footprint;
//   * a wave owns query rows {w, w+NW, ...}; lane j owns key j (and j + 64 in the long kernels): the row softmax is a
//     pair of wave reductions on the DPP path (pxr_common.h) -- no LDS round trip;
//   * mask semantics are the reference's ADDITIVE -1e9 in fp32 (not -inf): score + (-1e9) == -1e9 exactly, a
//     fully masked (left-padded) query row therefore becomes a uniform 1/L distribution over ALL L keys
//     (SURVEY.md §7 hard part 4) -- reproduced by doing the same arithmetic;
//   * attention-prob dropout (layers.py:608) uses the counter-hash mask, regenerated in backward;
//     the un-dropped probabilities are saved ([B,H,L,L]) for the backward pass.
#include <stdlib.h>

#include "planes.cuh"

namespace pxr {

constexpr int ATT_MAXL = 64;   // keys per sequence handled by one wave-wide softmax
constexpr int ATT_DC = 128;    // d-chunk staged through LDS
constexpr int ATT_KLD = ATT_DC + 4;  // padded row stride of the "key-side" tile: conflict-free ds_read_b128 by lane=row

#define PXR_ATTN_STAT_SLOTS 64     /* words of the backward's `stat` buffer (a power of two) */
struct AttnArgs {
  const float* q; const float* k; const float* v;  // element (b,t,h,c) at p[(b*L+t)*ld + h*d + c]
  int64_t ld;
  const int64_t* keymask; int64_t km_bstride;      // key j of batch b is real iff keymask[b*km_bstride+j] != 0
  float* ctx; int64_t ld_ctx;                      // [B*L, ld_ctx], head h at column h*d
  float* probs;                                    // [B,H,L,L] (softmax output before dropout) or null
  // backward only
  const float* dctx; float* dq; float* dk; float* dv; int64_t ld_d;
  int B, H, L, d;
  float sqrt_d;
  float p_drop; uint32_t drop_thr; uint32_t stream; uint64_t seed;
  const int64_t* step_dev;  // optional device counter added to the seed
  // optional outputs as bf16x3 planes (planes.cuh), attn_*_mfma1_kernel only: ctx as the [B*L, H*d] matrix; dq | dk | dv as
  // column ranges (starting at pcol[0..2]) of one [B*L, .] matrix.  With planes the fp32 outputs may be null.
  P3Mat op;
  int pcol[3];
  int op_fmt;               // PXR_PLANES_BF16X3 | PXR_PLANES_H2 (the forward's ctx planes; planes.cuh)
  int32_t* status;          // status word for the fp16 range check of h2 planes, or null
  const int* gexp;          // backward: null = the gradient planes are three bf16 planes; else two fp16 planes of gradient *
                            // 2^gexp[0], a DEVICE exponent chosen before the launch (previous step's statistics), saturating stores
  float* stat;              // backward (attn_bwd_mfma1_kernel): max of |dq|, |dk|, |dv| into PXR_ATTN_STAT_SLOTS zeroed words, or null
};

// stage tile[row][0..w) <- src[(row)*ld + 0..w) for row < L; all threads of the block, float4 accesses
__device__ __forceinline__ void stage_tile(float* tile, int tstride, const float* src, int64_t ld, int L, int w) {
  const int q4 = w >> 2;
  for (int f = threadIdx.x; f < L * q4; f += blockDim.x) {
    const int row = f / q4, c = (f - row * q4) * 4;
    *reinterpret_cast<float4*>(tile + row * tstride + c) = *reinterpret_cast<const float4*>(src + (int64_t)row * ld + c);
  }
}

// acc[g] += sum_c A[i_g][c] * Bt[lane][c]  for the wave's rows i_g = wave + NW*g   (A stride ATT_DC, Bt stride ATT_KLD)
template <int NW>
__device__ __forceinline__ void score_accumulate(float (&acc)[64 / NW], const float* sA, const float* sB, int L, int w,
                                                 int wave, int lane) {
  const float* brow = sB + lane * ATT_KLD;
#pragma unroll
  for (int gg = 0; gg < 16 / NW; ++gg) {
    const int i0 = wave + 4 * NW * gg;
    if (i0 >= L) break;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const float* r0 = sA + (i0)*ATT_DC;
    const float* r1 = sA + (i0 + NW) * ATT_DC;
    const float* r2 = sA + (i0 + 2 * NW) * ATT_DC;
    const float* r3 = sA + (i0 + 3 * NW) * ATT_DC;
    for (int c = 0; c < w; c += 4) {
      const float4 kb = *reinterpret_cast<const float4*>(brow + c);
      const float4 q0 = *reinterpret_cast<const float4*>(r0 + c);
      const float4 q1 = *reinterpret_cast<const float4*>(r1 + c);
      const float4 q2 = *reinterpret_cast<const float4*>(r2 + c);
      const float4 q3 = *reinterpret_cast<const float4*>(r3 + c);
      a0 += q0.x * kb.x + q0.y * kb.y + q0.z * kb.z + q0.w * kb.w;
      a1 += q1.x * kb.x + q1.y * kb.y + q1.z * kb.z + q1.w * kb.w;
      a2 += q2.x * kb.x + q2.y * kb.y + q2.z * kb.z + q2.w * kb.w;
      a3 += q3.x * kb.x + q3.y * kb.y + q3.z * kb.z + q3.w * kb.w;
    }
    acc[gg * 4 + 0] += a0; acc[gg * 4 + 1] += a1; acc[gg * 4 + 2] += a2; acc[gg * 4 + 3] += a3;
  }
}

// out[r][c] = sum_t W(r,t) * X[t][c]  for the wave's rows r = wave + NW*g, columns c = lane, lane+64 (< w).
// TRANS=false: W(r,t) = sW[r*64 + t];  TRANS=true: W(r,t) = sW[t*64 + r].   X stride ATT_DC.
template <int NW, bool TRANS>
__device__ __forceinline__ void rowmix_store(const float* sW, const float* sX, int L, int w, int wave, int lane,
                                             float* out, int64_t ld_out) {
  for (int gg = 0; gg < 16 / NW; ++gg) {
    const int i0 = wave + 4 * NW * gg;
    if (i0 >= L) break;
    float o[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r][0] = o[r][1] = 0.f;
    for (int t = 0; t < L; ++t) {
      const float x0 = sX[t * ATT_DC + lane];
      const float x1 = sX[t * ATT_DC + 64 + lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + NW * r;  // rows >= L read zeroed LDS rows and are never stored
        const float wv = TRANS ? sW[t * 64 + i] : sW[i * 64 + t];
        o[r][0] += wv * x0;
        o[r][1] += wv * x1;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + NW * r;
      if (i < L) {
        if (lane < w) out[(int64_t)i * ld_out + lane] = o[r][0];
        if (lane + 64 < w) out[(int64_t)i * ld_out + 64 + lane] = o[r][1];
      }
    }
  }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64) attn_fwd_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  __shared__ __attribute__((aligned(16))) float sA[ATT_MAXL * ATT_DC];    // Q chunk, later V chunk
  __shared__ __attribute__((aligned(16))) float sB[ATT_MAXL * ATT_KLD];   // K chunk
  __shared__ __attribute__((aligned(16))) float sP[ATT_MAXL * 64];        // (dropped) probabilities
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;   // one XCD: consecutive sequences
  const int L = a.L, d = a.d;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;

  // LDS rows >= L are read (never used) by lanes/rows beyond L: zero them once so no NaN garbage circulates
  for (int f = threadIdx.x; f < ATT_MAXL * ATT_DC; f += NW * 64) sA[f] = 0.f;
  for (int f = threadIdx.x; f < ATT_MAXL * ATT_KLD; f += NW * 64) sB[f] = 0.f;
  for (int f = threadIdx.x; f < ATT_MAXL * 64; f += NW * 64) sP[f] = 0.f;
  __syncthreads();

  float acc[64 / NW];
#pragma unroll
  for (int g = 0; g < 64 / NW; ++g) acc[g] = 0.f;
  for (int dc0 = 0; dc0 < d; dc0 += ATT_DC) {
    const int w = min(ATT_DC, d - dc0);
    stage_tile(sA, ATT_DC, a.q + base + dc0, a.ld, L, w);
    stage_tile(sB, ATT_KLD, a.k + base + dc0, a.ld, L, w);
    __syncthreads();
    score_accumulate<NW>(acc, sA, sB, L, w, wave, lane);
    __syncthreads();
  }

  const bool key_real = (lane < L) && (a.keymask[(int64_t)b * a.km_bstride + lane] != 0);
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
#pragma unroll
  for (int g = 0; g < 64 / NW; ++g) {
    const int i = wave + NW * g;  // acc[gg*4+r] holds row wave + NW*(4*gg + r)
    if (i < L) {
      // reference arithmetic: scores / sqrt(d) + (-1e9 | 0)   (layers.py:597,601; sasrec.py:125)
      float s = acc[g] / a.sqrt_d + ((key_real && lane <= i) ? 0.0f : -1e9f);
      if (lane >= L) s = -INFINITY;
      const float m = wave_max(s);
      const float e = (lane < L) ? expf(s - m) : 0.f;
      const float sum = wave_sum(e);
      const float p = e / sum;
      if (lane < L) {
        const int64_t pi = (((int64_t)b * a.H + h) * L + i) * L + lane;
        if (a.probs) a.probs[pi] = p;
        float pd = p;
        if (drop) pd = pxr_keep(a.seed, a.stream, (uint64_t)pi, a.drop_thr) ? p * inv_keep : 0.f;
        sP[i * 64 + lane] = pd;
      }
    }
  }
  __syncthreads();

  float* out = a.ctx + (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  for (int dc0 = 0; dc0 < d; dc0 += ATT_DC) {
    const int w = min(ATT_DC, d - dc0);
    stage_tile(sA, ATT_DC, a.v + base + dc0, a.ld, L, w);
    __syncthreads();
    rowmix_store<NW, false>(sP, sA, L, w, wave, lane, out + dc0, a.ld_ctx);
    __syncthreads();
  }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64) attn_bwd_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  __shared__ __attribute__((aligned(16))) float sA[ATT_MAXL * ATT_DC];
  __shared__ __attribute__((aligned(16))) float sB[ATT_MAXL * ATT_KLD];
  __shared__ __attribute__((aligned(16))) float sP[ATT_MAXL * 64];   // dropped probabilities Pd
  __shared__ __attribute__((aligned(16))) float sS[ATT_MAXL * 64];   // dS / sqrt(d)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;   // one XCD: consecutive sequences
  const int L = a.L, d = a.d;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;
  const int64_t cbase = (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  const int64_t dbase = (int64_t)b * L * a.ld_d + (int64_t)h * d;

  for (int f = threadIdx.x; f < ATT_MAXL * ATT_DC; f += NW * 64) sA[f] = 0.f;
  for (int f = threadIdx.x; f < ATT_MAXL * ATT_KLD; f += NW * 64) sB[f] = 0.f;
  for (int f = threadIdx.x; f < ATT_MAXL * 64; f += NW * 64) { sP[f] = 0.f; sS[f] = 0.f; }
  __syncthreads();

  // dPd[i][j] = sum_c dctx[i][c] * V[j][c]
  float acc[64 / NW];
#pragma unroll
  for (int g = 0; g < 64 / NW; ++g) acc[g] = 0.f;
  for (int dc0 = 0; dc0 < d; dc0 += ATT_DC) {
    const int w = min(ATT_DC, d - dc0);
    stage_tile(sA, ATT_DC, a.dctx + cbase + dc0, a.ld_ctx, L, w);
    stage_tile(sB, ATT_KLD, a.v + base + dc0, a.ld, L, w);
    __syncthreads();
    score_accumulate<NW>(acc, sA, sB, L, w, wave, lane);
    __syncthreads();
  }
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
#pragma unroll
  for (int g = 0; g < 64 / NW; ++g) {
    const int i = wave + NW * g;
    if (i < L) {
      float p = 0.f, pd = 0.f, dp = 0.f;
      if (lane < L) {
        const int64_t pi = (((int64_t)b * a.H + h) * L + i) * L + lane;
        p = a.probs[pi];
        const bool keep = !drop || pxr_keep(a.seed, a.stream, (uint64_t)pi, a.drop_thr);
        const float kf = drop ? (keep ? inv_keep : 0.f) : 1.f;
        pd = p * kf;
        dp = acc[g] * kf;
      }
      const float t = wave_sum(dp * p);
      if (lane < L) {
        sP[i * 64 + lane] = pd;
        sS[i * 64 + lane] = p * (dp - t) / a.sqrt_d;   // softmax backward, then the 1/sqrt(d) of layers.py:597
      }
    }
  }
  __syncthreads();

  for (int dc0 = 0; dc0 < d; dc0 += ATT_DC) {
    const int w = min(ATT_DC, d - dc0);
    // dV[j][c] = sum_i Pd[i][j] * dctx[i][c]
    stage_tile(sA, ATT_DC, a.dctx + cbase + dc0, a.ld_ctx, L, w);
    __syncthreads();
    rowmix_store<NW, true>(sP, sA, L, w, wave, lane, a.dv + dbase + dc0, a.ld_d);
    __syncthreads();
    // dQ[i][c] = sum_j dS[i][j] * K[j][c]
    stage_tile(sA, ATT_DC, a.k + base + dc0, a.ld, L, w);
    __syncthreads();
    rowmix_store<NW, false>(sS, sA, L, w, wave, lane, a.dq + dbase + dc0, a.ld_d);
    __syncthreads();
    // dK[j][c] = sum_i dS[i][j] * Q[i][c]
    stage_tile(sA, ATT_DC, a.q + base + dc0, a.ld, L, w);
    __syncthreads();
    rowmix_store<NW, true>(sS, sA, L, w, wave, lane, a.dk + dbase + dc0, a.ld_d);
    __syncthreads();
  }
}

// =================================================================================================================
// MFMA variant (default when d % 8 == 0): the four small matrix products of the attention core run on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32, exact fp32) from LDS-resident operands; 4 waves per (batch, head), each owning one 32x32
// block of the 64x64 score tile and a 32x64 slab of every 64x128 output chunk.  The softmax stays the wave-shuffle
// row softmax above (lane = key) reading the score tile from LDS, so mask / dropout / saved-probability semantics are
// byte-for-byte the same code.  Fragment reads follow gemm_f32.cuh: "KC" operands are [row][k] with a padded stride
// (conflict-free ds_read_b128, k split 0-3 / 4-7 between half-waves), "XC" operands are [k][x] (ds_read_b32).
constexpr int ATT_SLD = 64 + 4;        // score / probability tile stride (KC operand of P.V and dS.K)

// acc[j] += A_op[32 rows @ m_base] x B_op[32 cols @ n_base + 32 j],  K multiple of 8
template <bool A_KC, bool B_KC, int TN>
__device__ __forceinline__ void lds_mma(f32x16 (&acc)[TN], const float* sA, int lda, const float* sB, int ldb, int m_base,
                                        int n_base, int K, int lane) {
  const int h = lane >> 5, r = lane & 31;
  for (int k0 = 0; k0 < K; k0 += 8) {
    float a[4], b[TN][4];
    if constexpr (A_KC) {
      const float4 v = *reinterpret_cast<const float4*>(sA + (m_base + r) * lda + k0 + h * 4);
      a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = sA[(k0 + h * 4 + t) * lda + m_base + r];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (B_KC) {
        const float4 v = *reinterpret_cast<const float4*>(sB + (n_base + j * 32 + r) * ldb + k0 + h * 4);
        b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) b[j][t] = sB[(k0 + h * 4 + t) * ldb + n_base + j * 32 + r];
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[j][t], acc[j], 0, 0, 0);
  }
}

template <int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TN]) {
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
}

// global out[row][col] = acc  for row < L, col < w   (block at m_base, n_base + 32 j)
template <int TN>
__device__ __forceinline__ void store_acc(const f32x16 (&acc)[TN], float* out, int64_t ld, int m_base, int n_base, int L,
                                          int w, int lane) {
  const int h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n_base + j * 32 + r;
    if (col >= w) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m_base + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row < L) out[(int64_t)row * ld + col] = acc[j][e];
    }
  }
}

// the same block(s) as bf16x3 planes: through an LDS tile ([64][ATT_KLD], free at the call site) so that every thread owns 8
// consecutive columns of a row = one 16-byte store per plane.  ALL threads of the workgroup must call it (two barriers).
template <int TN, int NT>
__device__ __forceinline__ void store_acc_planes(const f32x16 (&acc)[TN], bool mine, float* tile, int m_base, int n_base, int L,
                                                 int w, int lane, const P3Mat& P, int64_t row0, int col0, int fmt = PXR_PLANES_BF16X3,
                                                 int32_t* status = nullptr, float stale_scale = 0.f) {
  const int h = lane >> 5, r = lane & 31;
  if (mine) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        tile[(m_base + (e & 3) + 8 * (e >> 2) + 4 * h) * ATT_KLD + n_base + j * 32 + r] = acc[j][e];
  }
  __syncthreads();
  const int cpr = w >> 3;
  for (int q = threadIdx.x; q < L * cpr; q += NT) {
    const int row = q / cpr, c8 = (q - row * cpr) * 8;
    const float4 x0 = *reinterpret_cast<const float4*>(tile + row * ATT_KLD + c8);
    const float4 x1 = *reinterpret_cast<const float4*>(tile + row * ATT_KLD + c8 + 4);
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    if (stale_scale != 0.f) px_store8_h2s(P, status, row0 + row, col0 + c8, v, stale_scale);
    else px_store8(P, fmt, status, row0 + row, col0 + c8, v);
  }
  __syncthreads();
}

// stage rows < L of a [L][w] global tile into LDS with the given stride; rows L..63 are zero-filled.
// All of a thread's global loads are issued BEFORE its first LDS store: with one workgroup per CU a
// load -> wait -> ds_write loop would serialise ~8 HBM round trips per tile (measured: 20 of the kernel's 25 us).
template <int NT = 256>
__device__ __forceinline__ void stage_tile_z(float* tile, int tstride, const float* src, int64_t ld, int L, int w) {
  const int q4 = w >> 2;
  const int total = ATT_MAXL * q4;            // <= 64 * 32 = 2048 float4 => <= 8 per thread at 256 threads
  constexpr int NP = 2048 / NT;
  float4 v[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int f = threadIdx.x + p * NT;
    v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < total) {
      const int row = f / q4, c = (f - row * q4) * 4;
      if (row < L) v[p] = *reinterpret_cast<const float4*>(src + (int64_t)row * ld + c);
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int f = threadIdx.x + p * NT;
    if (f < total) {
      const int row = f / q4, c = (f - row * q4) * 4;
      *reinterpret_cast<float4*>(tile + row * tstride + c) = v[p];
    }
  }
}

__global__ void __launch_bounds__(256) attn_fwd_mfma_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  __shared__ __attribute__((aligned(16))) float sQ[ATT_MAXL * ATT_KLD];   // Q chunk (KC); later V chunk (XC, stride 128)
  __shared__ __attribute__((aligned(16))) float sK[ATT_MAXL * ATT_KLD];   // K chunk (KC)
  __shared__ __attribute__((aligned(16))) float sS[ATT_MAXL * ATT_SLD];   // scores -> dropped probabilities
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;   // one XCD: consecutive sequences
  const int L = a.L, d = a.d;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;

  f32x16 accS[1];
  zero_acc<1>(accS);
  for (int dc0 = 0; dc0 < d; dc0 += ATT_DC) {
    const int w = min(ATT_DC, d - dc0);
    stage_tile_z(sQ, ATT_KLD, a.q + base + dc0, a.ld, L, w);
    stage_tile_z(sK, ATT_KLD, a.k + base + dc0, a.ld, L, w);
    __syncthreads();
    lds_mma<true, true, 1>(accS, sQ, ATT_KLD, sK, ATT_KLD, wm * 32, wn * 32, w, lane);
    __syncthreads();
  }
  {
    const int hh = lane >> 5, r = lane & 31;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      sS[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * ATT_SLD + wn * 32 + r] = accS[0][e];
  }
  __syncthreads();

  const bool key_real = (lane < L) && (a.keymask[(int64_t)b * a.km_bstride + lane] != 0);
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  for (int i = wave; i < L; i += 4) {
    // reference arithmetic: scores / sqrt(d) + (-1e9 | 0)   (layers.py:597,601; sasrec.py:125)
    float s = sS[i * ATT_SLD + lane] / a.sqrt_d + ((key_real && lane <= i) ? 0.0f : -1e9f);
    if (lane >= L) s = -INFINITY;
    const float m = wave_max(s);
    const float e = (lane < L) ? expf(s - m) : 0.f;
    const float sum = wave_sum(e);
    const float p = e / sum;
    float pd = 0.f;
    if (lane < L) {
      const int64_t pi = (((int64_t)b * a.H + h) * L + i) * L + lane;
      if (a.probs) a.probs[pi] = p;
      pd = p;
      if (drop) pd = pxr_keep(a.seed, a.stream, (uint64_t)pi, a.drop_thr) ? p * inv_keep : 0.f;
    }
    sS[i * ATT_SLD + lane] = pd;   // lanes >= L: exact zeros, so the padded V rows never contribute
  }
  __syncthreads();

  float* out = a.ctx + (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  for (int dc0 = 0; dc0 < d; dc0 += ATT_DC) {
    const int w = min(ATT_DC, d - dc0);
    stage_tile_z(sQ, ATT_DC, a.v + base + dc0, a.ld, L, w);
    __syncthreads();
    f32x16 accO[2];
    zero_acc<2>(accO);
    if (wn * 64 < w) lds_mma<true, false, 2>(accO, sS, ATT_SLD, sQ, ATT_DC, wm * 32, wn * 64, 64, lane);
    store_acc<2>(accO, out + dc0, a.ld_ctx, wm * 32, wn * 64, L, w, lane);
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) attn_bwd_mfma_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  __shared__ __attribute__((aligned(16))) float sQ[ATT_MAXL * ATT_KLD];   // dO chunk (KC) / generic XC tile (stride 128)
  __shared__ __attribute__((aligned(16))) float sK[ATT_MAXL * ATT_KLD];   // V chunk (KC)
  __shared__ __attribute__((aligned(16))) float sP[ATT_MAXL * ATT_SLD];   // dropped probabilities Pd
  __shared__ __attribute__((aligned(16))) float sS[ATT_MAXL * ATT_SLD];   // dP, then dS / sqrt(d)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;   // one XCD: consecutive sequences
  const int L = a.L, d = a.d;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;
  const int64_t cbase = (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  const int64_t dbase = (int64_t)b * L * a.ld_d + (int64_t)h * d;

  for (int f = threadIdx.x; f < ATT_MAXL * ATT_SLD; f += 256) sP[f] = 0.f;

  // dPd[i][j] = sum_c dctx[i][c] * V[j][c]
  f32x16 accP[1];
  zero_acc<1>(accP);
  for (int dc0 = 0; dc0 < d; dc0 += ATT_DC) {
    const int w = min(ATT_DC, d - dc0);
    stage_tile_z(sQ, ATT_KLD, a.dctx + cbase + dc0, a.ld_ctx, L, w);
    stage_tile_z(sK, ATT_KLD, a.v + base + dc0, a.ld, L, w);
    __syncthreads();
    lds_mma<true, true, 1>(accP, sQ, ATT_KLD, sK, ATT_KLD, wm * 32, wn * 32, w, lane);
    __syncthreads();
  }
  {
    const int hh = lane >> 5, r = lane & 31;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      sS[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * ATT_SLD + wn * 32 + r] = accP[0][e];
  }
  __syncthreads();
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  for (int i = wave; i < L; i += 4) {
    float p = 0.f, pd = 0.f, dp = 0.f;
    if (lane < L) {
      const int64_t pi = (((int64_t)b * a.H + h) * L + i) * L + lane;
      p = a.probs[pi];
      const bool keep = !drop || pxr_keep(a.seed, a.stream, (uint64_t)pi, a.drop_thr);
      const float kf = drop ? (keep ? inv_keep : 0.f) : 1.f;
      pd = p * kf;
      dp = sS[i * ATT_SLD + lane] * kf;
    }
    const float t = wave_sum(dp * p);
    sP[i * ATT_SLD + lane] = pd;
    sS[i * ATT_SLD + lane] = (lane < L) ? p * (dp - t) / a.sqrt_d : 0.f;   // softmax backward, then 1/sqrt(d)
  }
  __syncthreads();

  for (int dc0 = 0; dc0 < d; dc0 += ATT_DC) {
    const int w = min(ATT_DC, d - dc0);
    const bool mine = wn * 64 < w;
    f32x16 acc[2];
    // dV[j][c] = sum_i Pd[i][j] * dctx[i][c]
    stage_tile_z(sQ, ATT_DC, a.dctx + cbase + dc0, a.ld_ctx, L, w);
    __syncthreads();
    zero_acc<2>(acc);
    if (mine) lds_mma<false, false, 2>(acc, sP, ATT_SLD, sQ, ATT_DC, wm * 32, wn * 64, 64, lane);
    store_acc<2>(acc, a.dv + dbase + dc0, a.ld_d, wm * 32, wn * 64, L, w, lane);
    __syncthreads();
    // dQ[i][c] = sum_j dS[i][j] * K[j][c]
    stage_tile_z(sQ, ATT_DC, a.k + base + dc0, a.ld, L, w);
    __syncthreads();
    zero_acc<2>(acc);
    if (mine) lds_mma<true, false, 2>(acc, sS, ATT_SLD, sQ, ATT_DC, wm * 32, wn * 64, 64, lane);
    store_acc<2>(acc, a.dq + dbase + dc0, a.ld_d, wm * 32, wn * 64, L, w, lane);
    __syncthreads();
    // dK[j][c] = sum_i dS[i][j] * Q[i][c]
    stage_tile_z(sQ, ATT_DC, a.q + base + dc0, a.ld, L, w);
    __syncthreads();
    zero_acc<2>(acc);
    if (mine) lds_mma<false, false, 2>(acc, sS, ATT_SLD, sQ, ATT_DC, wm * 32, wn * 64, 64, lane);
    store_acc<2>(acc, a.dk + dbase + dc0, a.ld_d, wm * 32, wn * 64, L, w, lane);
    __syncthreads();
  }
}

// ---- single-phase MFMA kernels for d <= 128 (every shipped config: d = 32 .. 128) ---------------------------------
// With ONE workgroup per CU nothing hides a global->LDS staging round trip, and the chunked kernels above pay one per
// operand (3 forward, 5 backward).  Here every operand of the (batch, head) problem is fetched ONCE, up front, into
// its own [64][132] LDS tile -- a padded stride serves both fragment flavours (ds_read_b128 along k for "KC" uses,
// ds_read_b32 along x for "XC" uses) -- so the kernel is one load phase followed by back-to-back MFMA phases.
// Backward keeps dS in registers while the probability tile is used for dV, so one 64x68 tile suffices:
// 4 x 33 KB + 17 KB = 152 KB of the CU's 160 KB.
// NW = 4 or 8 waves.  The 64x64 score tile is always computed by waves 0-3 (one 32x32 block each); with 8 waves the
// row softmax handles 7 instead of 13 rows per wave, the staging issues half as many loads per thread and the
// 64x128 output is 8 blocks of 32x32 (one per wave) instead of 4 slabs of 32x64.
template <int NW>
__global__ void __launch_bounds__(64 * NW) attn_fwd_mfma1_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  constexpr int NT = 64 * NW;
  __shared__ __attribute__((aligned(16))) float sQ[ATT_MAXL * ATT_KLD];
  __shared__ __attribute__((aligned(16))) float sK[ATT_MAXL * ATT_KLD];
  __shared__ __attribute__((aligned(16))) float sV[ATT_MAXL * ATT_KLD];
  __shared__ __attribute__((aligned(16))) float sS[ATT_MAXL * ATT_SLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;   // one XCD: consecutive sequences
  const int L = a.L, d = a.d;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;
  stage_tile_z<NT>(sQ, ATT_KLD, a.q + base, a.ld, L, d);
  stage_tile_z<NT>(sK, ATT_KLD, a.k + base, a.ld, L, d);
  stage_tile_z<NT>(sV, ATT_KLD, a.v + base, a.ld, L, d);
  const bool key_real = (lane < L) && (a.keymask[(int64_t)b * a.km_bstride + lane] != 0);
  __syncthreads();

  if (wave < 4) {
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 accS[1];
    zero_acc<1>(accS);
    lds_mma<true, true, 1>(accS, sQ, ATT_KLD, sK, ATT_KLD, wm * 32, wn * 32, d, lane);
    const int hh = lane >> 5, r = lane & 31;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      sS[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * ATT_SLD + wn * 32 + r] = accS[0][e];
  }
  __syncthreads();
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  for (int i = wave; i < L; i += NW) {
    float s = sS[i * ATT_SLD + lane] / a.sqrt_d + ((key_real && lane <= i) ? 0.0f : -1e9f);
    if (lane >= L) s = -INFINITY;
    const float m = wave_max(s);
    const float e = (lane < L) ? expf(s - m) : 0.f;
    const float sum = wave_sum(e);
    const float p = e / sum;
    float pd = 0.f;
    if (lane < L) {
      const int64_t pi = (((int64_t)b * a.H + h) * L + i) * L + lane;
      if (a.probs) a.probs[pi] = p;
      pd = p;
      if (drop) pd = pxr_keep(a.seed, a.stream, (uint64_t)pi, a.drop_thr) ? p * inv_keep : 0.f;
    }
    sS[i * ATT_SLD + lane] = pd;
  }
  __syncthreads();
  float* ctx = a.ctx + (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  if constexpr (NW == 4) {
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 accO[2];
    zero_acc<2>(accO);
    if (wn * 64 < d) lds_mma<true, false, 2>(accO, sS, ATT_SLD, sV, ATT_KLD, wm * 32, wn * 64, 64, lane);
    if (a.ctx) store_acc<2>(accO, ctx, a.ld_ctx, wm * 32, wn * 64, L, d, lane);
    if (a.op.p) store_acc_planes<2, NT>(accO, wn * 64 < d, sQ, wm * 32, wn * 64, L, d, lane, a.op, (int64_t)b * L, h * d, a.op_fmt, a.status);
  } else {
    const int wm = wave >> 2, wn = wave & 3;
    f32x16 accO[1];
    zero_acc<1>(accO);
    if (wn * 32 < d) lds_mma<true, false, 1>(accO, sS, ATT_SLD, sV, ATT_KLD, wm * 32, wn * 32, 64, lane);
    if (a.ctx) store_acc<1>(accO, ctx, a.ld_ctx, wm * 32, wn * 32, L, d, lane);
    if (a.op.p) store_acc_planes<1, NT>(accO, wn * 32 < d, sQ, wm * 32, wn * 32, L, d, lane, a.op, (int64_t)b * L, h * d, a.op_fmt, a.status);
  }
}

template <int NW>
__global__ void __launch_bounds__(64 * NW) attn_bwd_mfma1_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  constexpr int NT = 64 * NW;
  constexpr int ROWS = 64 / NW;   // query rows per wave (row i = wave + NW * g)
  __shared__ __attribute__((aligned(16))) float sQ[ATT_MAXL * ATT_KLD];
  __shared__ __attribute__((aligned(16))) float sK[ATT_MAXL * ATT_KLD];
  __shared__ __attribute__((aligned(16))) float sV[ATT_MAXL * ATT_KLD];
  __shared__ __attribute__((aligned(16))) float sO[ATT_MAXL * ATT_KLD];   // dctx
  __shared__ __attribute__((aligned(16))) float sS[ATT_MAXL * ATT_SLD];   // dP -> Pd -> dS/sqrt(d)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;   // one XCD: consecutive sequences
  const int L = a.L, d = a.d;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;
  const int64_t cbase = (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  const int64_t dbase = (int64_t)b * L * a.ld_d + (int64_t)h * d;
  stage_tile_z<NT>(sO, ATT_KLD, a.dctx + cbase, a.ld_ctx, L, d);
  stage_tile_z<NT>(sV, ATT_KLD, a.v + base, a.ld, L, d);
  stage_tile_z<NT>(sK, ATT_KLD, a.k + base, a.ld, L, d);
  stage_tile_z<NT>(sQ, ATT_KLD, a.q + base, a.ld, L, d);
  // this wave's probability rows (saved by the forward pass) -- issued before the barrier so they overlap the staging
  float prow[ROWS];
#pragma unroll
  for (int g = 0; g < ROWS; ++g) {
    const int i = wave + NW * g;
    prow[g] = (i < L && lane < L) ? a.probs[(((int64_t)b * a.H + h) * L + i) * L + lane] : 0.f;
  }
  __syncthreads();

  if (wave < 4) {
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 accP[1];
    zero_acc<1>(accP);
    lds_mma<true, true, 1>(accP, sO, ATT_KLD, sV, ATT_KLD, wm * 32, wn * 32, d, lane);   // dPd = dctx V^T
    const int hh = lane >> 5, r = lane & 31;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      sS[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * ATT_SLD + wn * 32 + r] = accP[0][e];
  }
  __syncthreads();
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  float dsrow[ROWS];
#pragma unroll
  for (int g = 0; g < ROWS; ++g) {
    const int i = wave + NW * g;
    dsrow[g] = 0.f;
    if (i < L) {   // wave-uniform
      const float p = prow[g];
      float pd = 0.f, dp = 0.f;
      if (lane < L) {
        const int64_t pi = (((int64_t)b * a.H + h) * L + i) * L + lane;
        const bool keep = !drop || pxr_keep(a.seed, a.stream, (uint64_t)pi, a.drop_thr);
        const float kf = drop ? (keep ? inv_keep : 0.f) : 1.f;
        pd = p * kf;
        dp = sS[i * ATT_SLD + lane] * kf;
      }
      const float t = wave_sum(dp * p);
      dsrow[g] = (lane < L) ? p * (dp - t) / a.sqrt_d : 0.f;   // softmax backward, then the 1/sqrt(d) of layers.py:597
      sS[i * ATT_SLD + lane] = pd;                              // tile now holds Pd (rows >= L stay 0: dctx rows are 0)
    }
  }
  __syncthreads();
  // output blocks: NW = 4 -> 32x64 slab per wave; NW = 8 -> one 32x32 block per wave
  constexpr int TN = (NW == 4) ? 2 : 1;
  const int wm = (NW == 4) ? (wave >> 1) : (wave >> 2);
  const int n_base = (NW == 4) ? (wave & 1) * 64 : (wave & 3) * 32;
  const bool mine = n_base < d;
  f32x16 acc[TN];
  zero_acc<TN>(acc);
  const float gsc = a.gexp ? ldexpf(1.0f, a.gexp[0]) : 0.f;      // != 0: stale-scale fp16 planes (store_acc_planes)
  float gmax = 0.f;         // max |dq|, |dk|, |dv| over this thread's entries INSIDE the [L, d] block (the tiles' padding holds
  auto take = [&]() {       // whatever the LDS held: store_acc masks it, so must the statistics)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool col_ok = n_base + j * 32 + (lane & 31) < d;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (col_ok && row < L) gmax = fmaxf(gmax, fabsf(acc[j][e]));
      }
    }
  };
  if (mine) lds_mma<false, false, TN>(acc, sS, ATT_SLD, sO, ATT_KLD, wm * 32, n_base, 64, lane);   // dV = Pd^T dctx
  if (a.stat && mine) take();
  if (a.dv) store_acc<TN>(acc, a.dv + dbase, a.ld_d, wm * 32, n_base, L, d, lane);
  // (sV was last read by the dPd product, two barriers ago: free for the plane staging)
  if (a.op.p) store_acc_planes<TN, NT>(acc, mine, sV, wm * 32, n_base, L, d, lane, a.op, (int64_t)b * L, a.pcol[2] + h * d, PXR_PLANES_BF16X3, a.status, gsc);
  __syncthreads();
#pragma unroll
  for (int g = 0; g < ROWS; ++g) {
    const int i = wave + NW * g;
    if (i < L) sS[i * ATT_SLD + lane] = dsrow[g];
  }
  __syncthreads();
  zero_acc<TN>(acc);
  if (mine) lds_mma<true, false, TN>(acc, sS, ATT_SLD, sK, ATT_KLD, wm * 32, n_base, 64, lane);    // dQ = dS K
  if (a.stat && mine) take();
  if (a.dq) store_acc<TN>(acc, a.dq + dbase, a.ld_d, wm * 32, n_base, L, d, lane);
  if (a.op.p) store_acc_planes<TN, NT>(acc, mine, sV, wm * 32, n_base, L, d, lane, a.op, (int64_t)b * L, a.pcol[0] + h * d, PXR_PLANES_BF16X3, a.status, gsc);
  zero_acc<TN>(acc);
  if (mine) lds_mma<false, false, TN>(acc, sS, ATT_SLD, sQ, ATT_KLD, wm * 32, n_base, 64, lane);   // dK = dS^T Q
  if (a.stat && mine) take();
  if (a.dk) store_acc<TN>(acc, a.dk + dbase, a.ld_d, wm * 32, n_base, L, d, lane);
  if (a.op.p) store_acc_planes<TN, NT>(acc, mine, sV, wm * 32, n_base, L, d, lane, a.op, (int64_t)b * L, a.pcol[1] + h * d, PXR_PLANES_BF16X3, a.status, gsc);
  if (a.stat) {
    // one atomic per WORKGROUP, spread over PXR_ATTN_STAT_SLOTS words (the caller -- or the LayerNorm backward launch in front of
    // this one -- zeroed them; pxr_h2_split_parts_f32 reduces them): round 4 raised ONE word once per wave, 2 048 same-address
    // atomics at B = 64 that all found it still low
    __shared__ float smax[NW];
    gmax = wave_max(gmax);
    if (lane == 0) smax[wave] = gmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = smax[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) m = fmaxf(m, smax[w]);
      float* slot = a.stat + (bh & (PXR_ATTN_STAT_SLOTS - 1));
      if (m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<int*>(slot), __float_as_int(m));
    }
  }
}

// ---- two workgroups per CU (round 5): the single-phase kernels again, in 72 KB of LDS instead of 119 / 152 KB ---------------------
// With ONE workgroup per CU every phase of a (batch, head) problem -- the staging round trip, the score product, the row softmax,
// the output products, the plane transposition, the drain of the stores -- is exposed: 12.5 (forward) / 21 us (backward) per problem
// for 3 / 6 us of MFMA work, and a grid of B*H > 256 problems runs at that serial rate.  Two co-resident workgroups fill each
// other's gaps: 8.4 / 14.4 us per problem and CU at B*H = 8 192 (tools/attn_bench.py), ~3.4 TB/s of operand traffic.  What makes
// them fit:
//   * tiles of TR = 52 rows instead of 64: row L is a ZERO row and every fragment read of a row >= L is clamped to it (the MFMA
//     blocks still cover 64 rows; the padding rows they multiply are the same zeros as before, so every sum is the same sum);
//   * TWO operand tiles instead of three (forward) / four (backward): an operand that is needed later is FETCHED up front with
//     the others -- its loads stay in flight in registers -- and written over a tile whose operand has been consumed
//     (forward: V over Q after the score product; backward: K over V after dP, Q over dctx after dV);
//   * k loops stop at the first multiple of 8 >= L.
// Same products in the same order as attn_*_mfma1_kernel<8>: results are bit-identical (tests/test_gpu_attention.py).
constexpr int ATT2_TR = 52;                       // tile rows: L <= 51
constexpr int ATT2_NT = 512;
constexpr int ATT2_NP = (ATT2_TR * (ATT_DC / 4) + ATT2_NT - 1) / ATT2_NT;   // float4 per thread and tile (4)

// rows 0..L of a tile (row L = zeros) as NP float4 per thread: the loads ...
__device__ __forceinline__ void tile2_load(float4 (&v)[ATT2_NP], const float* src, int64_t ld, int L, int w) {
  const int q4 = w >> 2;
#pragma unroll
  for (int p = 0; p < ATT2_NP; ++p) {
    const int f = threadIdx.x + p * ATT2_NT;
    const int row = f / q4, c = (f - row * q4) * 4;
    v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < L) v[p] = *reinterpret_cast<const float4*>(src + (int64_t)row * ld + c);
  }
}
// ... and the LDS stores (any time later)
__device__ __forceinline__ void tile2_store(float* tile, const float4 (&v)[ATT2_NP], int L, int w) {
  const int q4 = w >> 2;
#pragma unroll
  for (int p = 0; p < ATT2_NP; ++p) {
    const int f = threadIdx.x + p * ATT2_NT;
    const int row = f / q4, c = (f - row * q4) * 4;
    if (row <= L) *reinterpret_cast<float4*>(tile + row * ATT_KLD + c) = v[p];
  }
}

// lds_mma with the row index of either operand clamped to za / zb (the zero row of a TR-row tile; 63 = no clamp: the score tile)
template <bool A_KC, bool B_KC, int TN>
__device__ __forceinline__ void lds_mma_z(f32x16 (&acc)[TN], const float* sA, int lda, int za, const float* sB, int ldb, int zb,
                                          int m_base, int n_base, int K, int lane) {
  const int h = lane >> 5, r = lane & 31;
  const int arow = min(m_base + r, za);
  int brow[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) brow[j] = min(n_base + j * 32 + r, zb);
  for (int k0 = 0; k0 < K; k0 += 8) {
    float a[4], b[TN][4];
    if constexpr (A_KC) {
      const float4 v = *reinterpret_cast<const float4*>(sA + arow * lda + k0 + h * 4);
      a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = sA[min(k0 + h * 4 + t, za) * lda + m_base + r];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (B_KC) {
        const float4 v = *reinterpret_cast<const float4*>(sB + brow[j] * ldb + k0 + h * 4);
        b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) b[j][t] = sB[min(k0 + h * 4 + t, zb) * ldb + n_base + j * 32 + r];
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[j][t], acc[j], 0, 0, 0);
  }
}

// one 32x32 accumulator block -> rows < L of a TR-row tile (the first half of store_acc_planes)
__device__ __forceinline__ void acc_to_tile2(const f32x16& acc, bool mine, float* tile, int m_base, int n_base, int L, int lane) {
  if (!mine) return;
  const int h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = m_base + (e & 3) + 8 * (e >> 2) + 4 * h;
    if (row < L) tile[row * ATT_KLD + n_base + r] = acc[e];
  }
}
// ... and the second: rows < L of the tile as planes, 8 consecutive columns per thread and store
__device__ __forceinline__ void tile2_to_planes(const float* tile, int L, int w, const P3Mat& P, int64_t row0, int col0, int fmt,
                                                int32_t* status, float stale_scale = 0.f) {
  const int cpr = w >> 3;
  for (int q = threadIdx.x; q < L * cpr; q += ATT2_NT) {
    const int row = q / cpr, c8 = (q - row * cpr) * 8;
    const float4 x0 = *reinterpret_cast<const float4*>(tile + row * ATT_KLD + c8);
    const float4 x1 = *reinterpret_cast<const float4*>(tile + row * ATT_KLD + c8 + 4);
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    if (stale_scale != 0.f) px_store8_h2s(P, status, row0 + row, col0 + c8, v, stale_scale);
    else px_store8(P, fmt, status, row0 + row, col0 + c8, v);
  }
}

__global__ void __launch_bounds__(ATT2_NT, 4) attn_fwd_mfma2_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  constexpr int NW = 8;
  __shared__ __attribute__((aligned(16))) float T0[ATT2_TR * ATT_KLD];   // Q, then V
  __shared__ __attribute__((aligned(16))) float T1[ATT2_TR * ATT_KLD];   // K, then the plane staging
  __shared__ __attribute__((aligned(16))) float sS[ATT_MAXL * ATT_SLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;
  const int L = a.L, d = a.d, Kp = (L + 7) & ~7;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;
  float4 rq[ATT2_NP], rk[ATT2_NP], rv[ATT2_NP];
  tile2_load(rq, a.q + base, a.ld, L, d);
  tile2_load(rk, a.k + base, a.ld, L, d);
  tile2_load(rv, a.v + base, a.ld, L, d);
  const bool key_real = (lane < L) && (a.keymask[(int64_t)b * a.km_bstride + lane] != 0);
  tile2_store(T0, rq, L, d);
  tile2_store(T1, rk, L, d);
  __syncthreads();
  if (wave < 4) {
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 accS[1];
    zero_acc<1>(accS);
    lds_mma_z<true, true, 1>(accS, T0, ATT_KLD, L, T1, ATT_KLD, L, wm * 32, wn * 32, d, lane);
    const int hh = lane >> 5, r = lane & 31;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      sS[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * ATT_SLD + wn * 32 + r] = accS[0][e];
  }
  __syncthreads();
  tile2_store(T0, rv, L, d);                     // V over Q (row L stays zero: the loads left zeros there)
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  for (int i = wave; i < L; i += NW) {
    float s = sS[i * ATT_SLD + lane] / a.sqrt_d + ((key_real && lane <= i) ? 0.0f : -1e9f);
    if (lane >= L) s = -INFINITY;
    const float m = wave_max(s);
    const float e = (lane < L) ? expf(s - m) : 0.f;
    const float sum = wave_sum(e);
    const float p = e / sum;
    float pd = 0.f;
    if (lane < L) {
      const int64_t pi = (((int64_t)b * a.H + h) * L + i) * L + lane;
      if (a.probs) a.probs[pi] = p;
      pd = p;
      if (drop) pd = pxr_keep(a.seed, a.stream, (uint64_t)pi, a.drop_thr) ? p * inv_keep : 0.f;
    }
    sS[i * ATT_SLD + lane] = pd;
  }
  __syncthreads();
  float* ctx = a.ctx + (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  const int wm = wave >> 2, wn = wave & 3;
  const bool mine = wn * 32 < d;
  f32x16 accO[1];
  zero_acc<1>(accO);
  if (mine) lds_mma_z<true, false, 1>(accO, sS, ATT_SLD, 63, T0, ATT_KLD, L, wm * 32, wn * 32, Kp, lane);
  if (a.ctx) store_acc<1>(accO, ctx, a.ld_ctx, wm * 32, wn * 32, L, d, lane);
  if (a.op.p) {                                  // T1 (K) was last read by the score product, two barriers ago
    acc_to_tile2(accO[0], mine, T1, wm * 32, wn * 32, L, lane);
    __syncthreads();
    tile2_to_planes(T1, L, d, a.op, (int64_t)b * L, h * d, a.op_fmt, a.status);
  }
}

__global__ void __launch_bounds__(ATT2_NT, 4) attn_bwd_mfma2_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  constexpr int NW = 8, ROWS = 64 / NW;
  __shared__ __attribute__((aligned(16))) float T0[ATT2_TR * ATT_KLD];   // dctx, then dV staging, then Q, then dK staging
  __shared__ __attribute__((aligned(16))) float T1[ATT2_TR * ATT_KLD];   // V, then K, then dQ staging
  __shared__ __attribute__((aligned(16))) float sS[ATT_MAXL * ATT_SLD];   // dP -> Pd -> dS/sqrt(d)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;
  const int L = a.L, d = a.d, Kp = (L + 7) & ~7;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;
  const int64_t cbase = (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  const int64_t dbase = (int64_t)b * L * a.ld_d + (int64_t)h * d;
  float4 r0[ATT2_NP], r1[ATT2_NP];
  tile2_load(r0, a.dctx + cbase, a.ld_ctx, L, d);
  tile2_load(r1, a.v + base, a.ld, L, d);
  float prow[ROWS];
#pragma unroll
  for (int g = 0; g < ROWS; ++g) {
    const int i = wave + NW * g;
    prow[g] = (i < L && lane < L) ? a.probs[(((int64_t)b * a.H + h) * L + i) * L + lane] : 0.f;
  }
  tile2_store(T0, r0, L, d);
  tile2_store(T1, r1, L, d);
  tile2_load(r1, a.k + base, a.ld, L, d);        // in flight under dP and the softmax backward
  tile2_load(r0, a.q + base, a.ld, L, d);        // ... and under dV
  __syncthreads();
  if (wave < 4) {
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 accP[1];
    zero_acc<1>(accP);
    lds_mma_z<true, true, 1>(accP, T0, ATT_KLD, L, T1, ATT_KLD, L, wm * 32, wn * 32, d, lane);   // dPd = dctx V^T
    const int hh = lane >> 5, r = lane & 31;
#pragma unroll
    for (int e = 0; e < 16; ++e)
      sS[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * ATT_SLD + wn * 32 + r] = accP[0][e];
  }
  __syncthreads();
  tile2_store(T1, r1, L, d);                     // K over V
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  float dsrow[ROWS];
#pragma unroll
  for (int g = 0; g < ROWS; ++g) {
    const int i = wave + NW * g;
    dsrow[g] = 0.f;
    if (i < L) {   // wave-uniform
      const float p = prow[g];
      float pd = 0.f, dp = 0.f;
      if (lane < L) {
        const int64_t pi = (((int64_t)b * a.H + h) * L + i) * L + lane;
        const bool keep = !drop || pxr_keep(a.seed, a.stream, (uint64_t)pi, a.drop_thr);
        const float kf = drop ? (keep ? inv_keep : 0.f) : 1.f;
        pd = p * kf;
        dp = sS[i * ATT_SLD + lane] * kf;
      }
      const float t = wave_sum(dp * p);
      dsrow[g] = (lane < L) ? p * (dp - t) / a.sqrt_d : 0.f;
      sS[i * ATT_SLD + lane] = pd;
    }
  }
  __syncthreads();
  const int wm = wave >> 2, n_base = (wave & 3) * 32;
  const bool mine = n_base < d;
  const float gsc = a.gexp ? ldexpf(1.0f, a.gexp[0]) : 0.f;      // != 0: stale-scale fp16 planes (tile2_to_planes)
  float gmax = 0.f;
  auto take = [&](const f32x16& acc) {
    const bool col_ok = n_base + (lane & 31) < d;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      if (col_ok && row < L) gmax = fmaxf(gmax, fabsf(acc[e]));
    }
  };
  f32x16 acc[1], acc2[1];
  zero_acc<1>(acc);
  if (mine) lds_mma_z<false, false, 1>(acc, sS, ATT_SLD, 63, T0, ATT_KLD, L, wm * 32, n_base, Kp, lane);   // dV = Pd^T dctx
  if (a.stat && mine) take(acc[0]);
  if (a.dv) store_acc<1>(acc, a.dv + dbase, a.ld_d, wm * 32, n_base, L, d, lane);
  __syncthreads();                               // dctx and Pd are consumed
#pragma unroll
  for (int g = 0; g < ROWS; ++g) {
    const int i = wave + NW * g;
    if (i < L) sS[i * ATT_SLD + lane] = dsrow[g];
  }
  if (a.op.p) {
    acc_to_tile2(acc[0], mine, T0, wm * 32, n_base, L, lane);
    __syncthreads();
    tile2_to_planes(T0, L, d, a.op, (int64_t)b * L, a.pcol[2] + h * d, PXR_PLANES_BF16X3, a.status, gsc);
    __syncthreads();
  }
  tile2_store(T0, r0, L, d);                     // Q over dctx (rewrites the zero row the staging never touched)
  __syncthreads();
  zero_acc<1>(acc);
  zero_acc<1>(acc2);
  if (mine) {
    lds_mma_z<true, false, 1>(acc, sS, ATT_SLD, 63, T1, ATT_KLD, L, wm * 32, n_base, Kp, lane);     // dQ = dS K
    lds_mma_z<false, false, 1>(acc2, sS, ATT_SLD, 63, T0, ATT_KLD, L, wm * 32, n_base, Kp, lane);   // dK = dS^T Q
  }
  if (a.stat && mine) { take(acc[0]); take(acc2[0]); }
  if (a.dq) store_acc<1>(acc, a.dq + dbase, a.ld_d, wm * 32, n_base, L, d, lane);
  if (a.dk) store_acc<1>(acc2, a.dk + dbase, a.ld_d, wm * 32, n_base, L, d, lane);
  if (a.op.p) {
    __syncthreads();
    acc_to_tile2(acc[0], mine, T1, wm * 32, n_base, L, lane);
    acc_to_tile2(acc2[0], mine, T0, wm * 32, n_base, L, lane);
    __syncthreads();
    tile2_to_planes(T1, L, d, a.op, (int64_t)b * L, a.pcol[0] + h * d, PXR_PLANES_BF16X3, a.status, gsc);
    tile2_to_planes(T0, L, d, a.op, (int64_t)b * L, a.pcol[1] + h * d, PXR_PLANES_BF16X3, a.status, gsc);
  }
  if (a.stat) {
    __shared__ float smax[NW];
    gmax = wave_max(gmax);
    if (lane == 0) smax[wave] = gmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = smax[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) m = fmaxf(m, smax[w]);
      float* slot = a.stat + (bh & (PXR_ATTN_STAT_SLOTS - 1));
      if (m > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(reinterpret_cast<int*>(slot), __float_as_int(m));
    }
  }
}

// ---- sequences of 65..128 positions ------------------------------------------------------------------------------
// Same algorithm as the d-chunked kernels above on 128-row tiles: 8 waves, the 128x128 score tile (4x4 blocks of 32x32,
// two per wave) stays in LDS, the row softmax gives every lane TWO keys (lane, lane + 64), Q/K/V/dO are staged in
// chunks of 64 head columns.  LDS: 67.6 KB (scores) + 2 x 34.8 KB (chunks) = 137 KB.
constexpr int ATTL_MAXL = 128;
constexpr int ATTL_SLD = ATTL_MAXL + 4;    // score / probability tile stride
constexpr int ATTL_DC = 64;                // head columns per staged chunk
constexpr int ATTL_KLD = ATTL_DC + 4;      // padded stride of a k-contiguous chunk tile
constexpr int ATTL_NT = 512;

// rows < L of a [L][w] global tile -> LDS with the given stride; rows L..127 zero-filled (w <= 64)
__device__ __forceinline__ void stage_tile_long(float* tile, int tstride, const float* src, int64_t ld, int L, int w) {
  const int q4 = w >> 2;
  const int total = ATTL_MAXL * q4;            // <= 128 * 16 = 2048 float4 => <= 4 per thread at 512 threads
  float4 v[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int f = threadIdx.x + p * ATTL_NT;
    v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < total) {
      const int row = f / q4, c = (f - row * q4) * 4;
      if (row < L) v[p] = *reinterpret_cast<const float4*>(src + (int64_t)row * ld + c);
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int f = threadIdx.x + p * ATTL_NT;
    if (f < total) {
      const int row = f / q4, c = (f - row * q4) * 4;
      *reinterpret_cast<float4*>(tile + row * tstride + c) = v[p];
    }
  }
}

// acc (two 32x32 blocks at rows wm*32, columns wn*64 + {0, 32}) -> score tile
__device__ __forceinline__ void acc2_to_tile(const f32x16 (&acc)[2], float* sS, int wm, int wn, int lane) {
  const int hh = lane >> 5, r = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      sS[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh) * ATTL_SLD + wn * 64 + j * 32 + r] = acc[j][e];
}

__global__ void __launch_bounds__(ATTL_NT) attn_fwd_long_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  __shared__ __attribute__((aligned(16))) float sA[ATTL_MAXL * ATTL_KLD];   // Q chunk (KC); later V chunk (XC, stride 64)
  __shared__ __attribute__((aligned(16))) float sB[ATTL_MAXL * ATTL_KLD];   // K chunk (KC)
  __shared__ __attribute__((aligned(16))) float sS[ATTL_MAXL * ATTL_SLD];   // scores -> dropped probabilities
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;   // one XCD: consecutive sequences
  const int L = a.L, d = a.d;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;

  f32x16 accS[2];
  zero_acc<2>(accS);
  for (int dc0 = 0; dc0 < d; dc0 += ATTL_DC) {
    const int w = min(ATTL_DC, d - dc0);
    stage_tile_long(sA, ATTL_KLD, a.q + base + dc0, a.ld, L, w);
    stage_tile_long(sB, ATTL_KLD, a.k + base + dc0, a.ld, L, w);
    __syncthreads();
    lds_mma<true, true, 2>(accS, sA, ATTL_KLD, sB, ATTL_KLD, wm * 32, wn * 64, w, lane);
    __syncthreads();
  }
  acc2_to_tile(accS, sS, wm, wn, lane);
  __syncthreads();

  const int k0 = lane, k1 = lane + 64;
  const bool real0 = (k0 < L) && (a.keymask[(int64_t)b * a.km_bstride + k0] != 0);
  const bool real1 = (k1 < L) && (a.keymask[(int64_t)b * a.km_bstride + k1] != 0);
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  for (int i = wave; i < L; i += ATTL_NT / 64) {
    float s0 = sS[i * ATTL_SLD + k0] / a.sqrt_d + ((real0 && k0 <= i) ? 0.0f : -1e9f);
    float s1 = sS[i * ATTL_SLD + k1] / a.sqrt_d + ((real1 && k1 <= i) ? 0.0f : -1e9f);
    if (k0 >= L) s0 = -INFINITY;
    if (k1 >= L) s1 = -INFINITY;
    const float m = wave_max(fmaxf(s0, s1));
    const float e0 = (k0 < L) ? expf(s0 - m) : 0.f;
    const float e1 = (k1 < L) ? expf(s1 - m) : 0.f;
    const float sum = wave_sum(e0 + e1);
    const float p0 = e0 / sum, p1 = e1 / sum;
    float pd0 = 0.f, pd1 = 0.f;
    const int64_t rowi = (((int64_t)b * a.H + h) * L + i) * L;
    if (k0 < L) {
      if (a.probs) a.probs[rowi + k0] = p0;
      pd0 = (!drop || pxr_keep(a.seed, a.stream, (uint64_t)(rowi + k0), a.drop_thr)) ? p0 * (drop ? inv_keep : 1.f) : 0.f;
    }
    if (k1 < L) {
      if (a.probs) a.probs[rowi + k1] = p1;
      pd1 = (!drop || pxr_keep(a.seed, a.stream, (uint64_t)(rowi + k1), a.drop_thr)) ? p1 * (drop ? inv_keep : 1.f) : 0.f;
    }
    sS[i * ATTL_SLD + k0] = pd0;   // keys >= L: exact zeros, so the padded V rows never contribute
    sS[i * ATTL_SLD + k1] = pd1;
  }
  __syncthreads();

  float* out = a.ctx + (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  for (int dc0 = 0; dc0 < d; dc0 += ATTL_DC) {
    const int w = min(ATTL_DC, d - dc0);
    stage_tile_long(sA, ATTL_DC, a.v + base + dc0, a.ld, L, w);
    __syncthreads();
    f32x16 accO[1];
    zero_acc<1>(accO);
    if (wn * 32 < w) lds_mma<true, false, 1>(accO, sS, ATTL_SLD, sA, ATTL_DC, wm * 32, wn * 32, ATTL_MAXL, lane);
    store_acc<1>(accO, out + dc0, a.ld_ctx, wm * 32, wn * 32, L, w, lane);
    __syncthreads();
  }
}

__global__ void __launch_bounds__(ATTL_NT) attn_bwd_long_kernel(AttnArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  __shared__ __attribute__((aligned(16))) float sA[ATTL_MAXL * ATTL_KLD];   // dO chunk (KC) / generic XC tile (stride 64)
  __shared__ __attribute__((aligned(16))) float sB[ATTL_MAXL * ATTL_KLD];   // V chunk (KC)
  __shared__ __attribute__((aligned(16))) float sS[ATTL_MAXL * ATTL_SLD];   // dPd -> Pd -> dS / sqrt(d)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bh = xcd_remap(blockIdx.x, gridDim.x), b = bh / a.H, h = bh - b * a.H;   // one XCD: consecutive sequences
  const int L = a.L, d = a.d;
  const int64_t base = (int64_t)b * L * a.ld + (int64_t)h * d;
  const int64_t cbase = (int64_t)b * L * a.ld_ctx + (int64_t)h * d;
  const int64_t dbase = (int64_t)b * L * a.ld_d + (int64_t)h * d;

  f32x16 accP[2];
  zero_acc<2>(accP);
  for (int dc0 = 0; dc0 < d; dc0 += ATTL_DC) {      // dPd[i][j] = sum_c dctx[i][c] * V[j][c]
    const int w = min(ATTL_DC, d - dc0);
    stage_tile_long(sA, ATTL_KLD, a.dctx + cbase + dc0, a.ld_ctx, L, w);
    stage_tile_long(sB, ATTL_KLD, a.v + base + dc0, a.ld, L, w);
    __syncthreads();
    lds_mma<true, true, 2>(accP, sA, ATTL_KLD, sB, ATTL_KLD, wm * 32, wn * 64, w, lane);
    __syncthreads();
  }
  acc2_to_tile(accP, sS, wm, wn, lane);
  __syncthreads();

  constexpr int NW = ATTL_NT / 64, ROWS = ATTL_MAXL / NW;   // 16 query rows per wave
  const int k0 = lane, k1 = lane + 64;
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  float ds0[ROWS], ds1[ROWS];
#pragma unroll
  for (int g = 0; g < ROWS; ++g) {
    const int i = wave + NW * g;
    ds0[g] = ds1[g] = 0.f;
    if (i < L) {   // wave-uniform
      const int64_t rowi = (((int64_t)b * a.H + h) * L + i) * L;
      float p0 = 0.f, p1 = 0.f, pd0 = 0.f, pd1 = 0.f, dp0 = 0.f, dp1 = 0.f;
      if (k0 < L) {
        p0 = a.probs[rowi + k0];
        const float kf = drop ? (pxr_keep(a.seed, a.stream, (uint64_t)(rowi + k0), a.drop_thr) ? inv_keep : 0.f) : 1.f;
        pd0 = p0 * kf;
        dp0 = sS[i * ATTL_SLD + k0] * kf;
      }
      if (k1 < L) {
        p1 = a.probs[rowi + k1];
        const float kf = drop ? (pxr_keep(a.seed, a.stream, (uint64_t)(rowi + k1), a.drop_thr) ? inv_keep : 0.f) : 1.f;
        pd1 = p1 * kf;
        dp1 = sS[i * ATTL_SLD + k1] * kf;
      }
      const float t = wave_sum(dp0 * p0 + dp1 * p1);
      ds0[g] = (k0 < L) ? p0 * (dp0 - t) / a.sqrt_d : 0.f;   // softmax backward, then the 1/sqrt(d) of layers.py:597
      ds1[g] = (k1 < L) ? p1 * (dp1 - t) / a.sqrt_d : 0.f;
      sS[i * ATTL_SLD + k0] = pd0;                             // tile now holds Pd (rows >= L stay 0: dctx rows are 0)
      sS[i * ATTL_SLD + k1] = pd1;
    }
  }
  __syncthreads();

  for (int dc0 = 0; dc0 < d; dc0 += ATTL_DC) {       // dV[j][c] = sum_i Pd[i][j] * dctx[i][c]
    const int w = min(ATTL_DC, d - dc0);
    stage_tile_long(sA, ATTL_DC, a.dctx + cbase + dc0, a.ld_ctx, L, w);
    __syncthreads();
    f32x16 acc[1];
    zero_acc<1>(acc);
    if (wn * 32 < w) lds_mma<false, false, 1>(acc, sS, ATTL_SLD, sA, ATTL_DC, wm * 32, wn * 32, ATTL_MAXL, lane);
    store_acc<1>(acc, a.dv + dbase + dc0, a.ld_d, wm * 32, wn * 32, L, w, lane);
    __syncthreads();
  }
#pragma unroll
  for (int g = 0; g < ROWS; ++g) {
    const int i = wave + NW * g;
    if (i < L) { sS[i * ATTL_SLD + k0] = ds0[g]; sS[i * ATTL_SLD + k1] = ds1[g]; }
  }
  __syncthreads();
  for (int dc0 = 0; dc0 < d; dc0 += ATTL_DC) {
    const int w = min(ATTL_DC, d - dc0);
    const bool mine = wn * 32 < w;
    f32x16 acc[1];
    stage_tile_long(sA, ATTL_DC, a.k + base + dc0, a.ld, L, w);          // dQ[i][c] = sum_j dS[i][j] * K[j][c]
    __syncthreads();
    zero_acc<1>(acc);
    if (mine) lds_mma<true, false, 1>(acc, sS, ATTL_SLD, sA, ATTL_DC, wm * 32, wn * 32, ATTL_MAXL, lane);
    store_acc<1>(acc, a.dq + dbase + dc0, a.ld_d, wm * 32, wn * 32, L, w, lane);
    __syncthreads();
    stage_tile_long(sA, ATTL_DC, a.q + base + dc0, a.ld, L, w);          // dK[j][c] = sum_i dS[i][j] * Q[i][c]
    __syncthreads();
    zero_acc<1>(acc);
    if (mine) lds_mma<false, false, 1>(acc, sS, ATTL_SLD, sA, ATTL_DC, wm * 32, wn * 32, ATTL_MAXL, lane);
    store_acc<1>(acc, a.dk + dbase + dc0, a.ld_d, wm * 32, wn * 32, L, w, lane);
    __syncthreads();
  }
}

}  // namespace pxr

using namespace pxr;

// waves per workgroup of the single-phase MFMA kernels: PXR_ATTN_MFMA_WAVES=4|8 (tuning knob)
static int attn_mfma_waves() {
  static const int nw = getenv("PXR_ATTN_MFMA_WAVES") ? atoi(getenv("PXR_ATTN_MFMA_WAVES")) : 8;
  return nw == 8 ? 8 : 4;
}

// waves per (batch, head) workgroup: 8 by default; PXR_ATTN_WAVES=4|8|16 overrides it (tuning knob)
static int attn_waves() {
  static int nw = 0;
  if (nw == 0) {
    const char* e = getenv("PXR_ATTN_WAVES");
    const int v = e ? atoi(e) : 8;
    nw = (v == 4 || v == 16) ? v : 8;
  }
  return nw;
}

// the two-per-CU kernels serve every grid they can (L <= 51, d <= 128, 8 waves): measured faster than the single-phase kernels
// even at one workgroup per CU (B*H = 256: 15.2 -> 13.6 us forward, 24.2 -> 22.7 backward -- shorter k loops, operands written to
// LDS as they are needed), 1.45 x at B*H = 8 192.  PXR_ATTN_TWO=0 selects the single-phase kernels (read per call: a test compares
// the two families in one process)
static bool attn_use_two(int L, int d) {
  const char* e = getenv("PXR_ATTN_TWO");
  return L < ATT2_TR && d <= ATT_DC && attn_mfma_waves() == 8 && !(e && atoi(e) == 0);
}

// MFMA attention is the default whenever the head size allows it; PXR_ATTN_MFMA=0 selects the VALU kernels
static bool attn_use_mfma(int d) {
  static int flag = -1;
  if (flag < 0) {
    const char* e = getenv("PXR_ATTN_MFMA");
    flag = (e && atoi(e) == 0) ? 0 : 1;
  }
  return flag == 1 && (d % 8) == 0;
}

static int attn_check(int B, int H, int L, int d, int64_t ld, const char* who) {
  PXR_REQUIRE(B >= 0 && H > 0 && L > 0 && d > 0, "%s: bad shape", who);
  PXR_REQUIRE(L <= ATTL_MAXL, "%s: L=%d > %d positions is not supported", who, L, ATTL_MAXL);
  PXR_REQUIRE(L <= ATT_MAXL || (attn_use_mfma(d)), "%s: L=%d > %d needs the MFMA kernels (head size %d %% 8 == 0)", who, L,
              ATT_MAXL, d);
  PXR_REQUIRE(d % 4 == 0 && ld % 4 == 0, "%s: d and ld must be multiples of 4", who);
  return PXR_OK;
}

// ctx[b,t,h*d:(h+1)*d] = softmax(q k^T / sqrt(d) + mask) v       (eval: p_drop = 0)
// 1 when the fused kernels that can write their outputs as planes serve this shape (pxr_attn_*_planes_f32)
extern "C" int pxr_attn_planes_supported(int L, int d) { return (L <= ATT_MAXL && attn_use_mfma(d) && d <= ATT_DC && d % 8 == 0) ? 1 : 0; }

extern "C" int pxr_attn_fwd_planes_f32(const float* q, const float* k, const float* v, int64_t ld, const int64_t* keymask,
                                       int64_t km_bstride, int B, int H, int L, int d, float* ctx, int64_t ld_ctx,
                                       float* probs, float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev,
                                       void* ctx_planes, int64_t ctx_plane_stride, int64_t ctx_panel_rows, void* stream);
extern "C" int pxr_attn_fwd_f32(const float* q, const float* k, const float* v, int64_t ld, const int64_t* keymask,
                                int64_t km_bstride, int B, int H, int L, int d, float* ctx, int64_t ld_ctx,
                                float* probs, float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev,
                                void* stream) {
  PXR_REQUIRE(ctx, "pxr_attn_fwd_f32: null pointer");
  return pxr_attn_fwd_planes_f32(q, k, v, ld, keymask, km_bstride, B, H, L, d, ctx, ld_ctx, probs, p_drop, seed, stream_id,
                                 step_dev, nullptr, 0, 0, stream);
}
// the same with ctx (the [B*L, H*d] matrix) additionally -- or, ctx == NULL, only -- written as bf16x3 planes: the operand
// format of the output projection that follows (layers.py:613).  Shapes: pxr_attn_planes_supported.
static int attn_fwd_planes_impl(const float* q, const float* k, const float* v, int64_t ld, const int64_t* keymask,
                                int64_t km_bstride, int B, int H, int L, int d, float* ctx, int64_t ld_ctx, float* probs, float p_drop,
                                uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* ctx_planes,
                                int64_t ctx_plane_stride, int64_t ctx_panel_rows, int c_fmt, void* stream);
extern "C" int pxr_attn_fwd_planes_f32(const float* q, const float* k, const float* v, int64_t ld, const int64_t* keymask,
                                       int64_t km_bstride, int B, int H, int L, int d, float* ctx, int64_t ld_ctx,
                                       float* probs, float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev,
                                       void* ctx_planes, int64_t ctx_plane_stride, int64_t ctx_panel_rows, void* stream) {
  return attn_fwd_planes_impl(q, k, v, ld, keymask, km_bstride, B, H, L, d, ctx, ld_ctx, probs, p_drop, seed, stream_id, step_dev,
                              ctx_planes, ctx_plane_stride, ctx_panel_rows, PXR_PLANES_BF16X3, stream);
}
// ... with the ctx planes in the two-plane fp16 format (planes.cuh "h2", unit scale): the operand of pxr_gemm_h2_f32
extern "C" int pxr_attn_fwd_h2_f32(const float* q, const float* k, const float* v, int64_t ld, const int64_t* keymask,
                                   int64_t km_bstride, int B, int H, int L, int d, float* ctx, int64_t ld_ctx, float* probs,
                                   float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* ctx_planes,
                                   int64_t ctx_plane_stride, int64_t ctx_panel_rows, void* stream) {
  PXR_REQUIRE(ctx_planes, "pxr_attn_fwd_h2_f32: no planes");
  return attn_fwd_planes_impl(q, k, v, ld, keymask, km_bstride, B, H, L, d, ctx, ld_ctx, probs, p_drop, seed, stream_id, step_dev,
                              ctx_planes, ctx_plane_stride, ctx_panel_rows, PXR_PLANES_H2, stream);
}
static int attn_fwd_planes_impl(const float* q, const float* k, const float* v, int64_t ld, const int64_t* keymask,
                                int64_t km_bstride, int B, int H, int L, int d, float* ctx, int64_t ld_ctx, float* probs, float p_drop,
                                uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* ctx_planes,
                                int64_t ctx_plane_stride, int64_t ctx_panel_rows, int c_fmt, void* stream) {
  PXR_REQUIRE(q && k && v && keymask && (ctx || ctx_planes), "pxr_attn_fwd_f32: null pointer");
  PXR_REQUIRE(!ctx_planes || (pxr_attn_planes_supported(L, d) && p3_mat_ok(ctx_planes, ctx_plane_stride, ctx_panel_rows, (int64_t)B * L, (int64_t)H * d)),
              "pxr_attn_fwd_planes_f32: planes are not available for this shape (L=%d, d=%d)", L, d);
  int rc = attn_check(B, H, L, d, ld, "pxr_attn_fwd_f32");
  if (rc) return rc;
  PXR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "pxr_attn_fwd_f32: bad dropout p");
  if (B == 0) return PXR_OK;
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.ld = ld; a.keymask = keymask; a.km_bstride = km_bstride;
  a.ctx = ctx; a.ld_ctx = ld_ctx; a.probs = probs; a.B = B; a.H = H; a.L = L; a.d = d;
  a.op = P3Mat{reinterpret_cast<__bf16*>(ctx_planes), ctx_plane_stride, ctx_panel_rows};
  a.op_fmt = c_fmt;
  a.status = pxr_status_word();
  a.sqrt_d = sqrtf((float)d);
  a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id; a.seed = seed;
  a.step_dev = step_dev;
  if (L > ATT_MAXL) {   // 65..128 positions: two keys per lane, 128-row tiles
    hipLaunchKernelGGL(attn_fwd_long_kernel, dim3(B * H), dim3(ATTL_NT), 0, (hipStream_t)stream, a);
    return pxr_check_launch("pxr_attn_fwd_f32(long)");
  }
  if (attn_use_mfma(d)) {
    if (d <= ATT_DC) {
      if (attn_use_two(L, d)) hipLaunchKernelGGL(attn_fwd_mfma2_kernel, dim3(B * H), dim3(ATT2_NT), 0, (hipStream_t)stream, a);
      else if (attn_mfma_waves() == 8) hipLaunchKernelGGL(attn_fwd_mfma1_kernel<8>, dim3(B * H), dim3(512), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL(attn_fwd_mfma1_kernel<4>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
    }
    else hipLaunchKernelGGL(attn_fwd_mfma_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
    return pxr_check_launch("pxr_attn_fwd_f32(mfma)");
  }
  switch (attn_waves()) {
    case 4: hipLaunchKernelGGL(attn_fwd_kernel<4>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a); break;
    case 16: hipLaunchKernelGGL(attn_fwd_kernel<16>, dim3(B * H), dim3(1024), 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(attn_fwd_kernel<8>, dim3(B * H), dim3(512), 0, (hipStream_t)stream, a); break;
  }
  return pxr_check_launch("pxr_attn_fwd_f32");
}

// Gradients w.r.t. q, k, v (written with row stride ld_d, head h at column h*d) from dctx and the saved probs.
extern "C" int pxr_attn_bwd_planes_f32(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v,
                                       int64_t ld, const float* probs, int B, int H, int L, int d, float* dq, float* dk,
                                       float* dv, int64_t ld_d, float p_drop, uint64_t seed, uint32_t stream_id,
                                       const int64_t* step_dev, void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows,
                                       int g_cols, int col_q, int col_k, int col_v, void* stream);
extern "C" int pxr_attn_bwd_f32(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v,
                                int64_t ld, const float* probs, int B, int H, int L, int d, float* dq, float* dk,
                                float* dv, int64_t ld_d, float p_drop, uint64_t seed, uint32_t stream_id,
                                const int64_t* step_dev, void* stream) {
  PXR_REQUIRE(dq && dk && dv, "pxr_attn_bwd_f32: null pointer");
  return pxr_attn_bwd_planes_f32(dctx, ld_ctx, q, k, v, ld, probs, B, H, L, d, dq, dk, dv, ld_d, p_drop, seed, stream_id,
                                 step_dev, nullptr, 0, 0, 0, 0, 0, 0, stream);
}
// the same with dq | dk | dv additionally -- or, all three NULL, only -- written as bf16x3 planes: column ranges starting at
// col_q / col_k / col_v of one [B*L, g_cols] planes matrix (the gradient of the fused QKV projection's output).
static int attn_bwd_impl(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v, int64_t ld,
                         const float* probs, int B, int H, int L, int d, float* dq, float* dk, float* dv, int64_t ld_d, float p_drop,
                         uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* g_planes, int64_t g_plane_stride,
                         int64_t g_panel_rows, int g_cols, int col_q, int col_k, int col_v, float* stat, void* stream,
                         const int* g_exp = nullptr);
extern "C" int pxr_attn_bwd_planes_f32(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v,
                                       int64_t ld, const float* probs, int B, int H, int L, int d, float* dq, float* dk,
                                       float* dv, int64_t ld_d, float p_drop, uint64_t seed, uint32_t stream_id,
                                       const int64_t* step_dev, void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows,
                                       int g_cols, int col_q, int col_k, int col_v, void* stream) {
  return attn_bwd_impl(dctx, ld_ctx, q, k, v, ld, probs, B, H, L, d, dq, dk, dv, ld_d, p_drop, seed, stream_id, step_dev, g_planes,
                       g_plane_stride, g_panel_rows, g_cols, col_q, col_k, col_v, nullptr, stream);
}
// pxr_attn_bwd_f32 that also leaves max(|dq|, |dk|, |dv|) in *stat by atomic maxima (the caller zeroes the slot): the statistics
// pxr_h2_split_auto_multi_f32(col_stats = 2) needs.  Only the shapes the fused MFMA kernel serves (pxr_attn_planes_supported).
extern "C" int pxr_attn_bwd_stat_f32(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v, int64_t ld,
                                     const float* probs, int B, int H, int L, int d, float* dq, float* dk, float* dv, int64_t ld_d,
                                     float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, float* stat,
                                     void* stream) {
  PXR_REQUIRE(stat && pxr_attn_planes_supported(L, d), "pxr_attn_bwd_stat_f32: no slot, or a shape the fused kernel does not serve (L=%d, d=%d)", L, d);
  return attn_bwd_impl(dctx, ld_ctx, q, k, v, ld, probs, B, H, L, d, dq, dk, dv, ld_d, p_drop, seed, stream_id, step_dev, nullptr, 0, 0, 0,
                       0, 0, 0, stat, stream);
}
// dq | dk | dv ONLY as two fp16 planes of gradient * 2^g_exp_dev[0] (column ranges of one [B*L, g_cols] matrix as in
// pxr_attn_bwd_planes_f32) under an exponent that exists before the launch (pxr_h2_sites_update: the previous step's maximum less the
// headroom), range-checked and saturated (PXR_STATUS_H2_STALE), + this step's partial maxima in the PXR_ATTN_STAT_SLOTS words of
// `stat` (zeroed by the caller / the LayerNorm launch in front).  Replaces pxr_attn_bwd_stat_f32 + pxr_h2_split_parts_f32.
extern "C" int pxr_attn_bwd_h2s_f32(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v, int64_t ld,
                                    const float* probs, int B, int H, int L, int d, float p_drop, uint64_t seed, uint32_t stream_id,
                                    const int64_t* step_dev, void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows, int g_cols,
                                    int col_q, int col_k, int col_v, const int* g_exp_dev, float* stat, void* stream) {
  PXR_REQUIRE(g_planes && g_exp_dev && stat && pxr_attn_planes_supported(L, d),
              "pxr_attn_bwd_h2s_f32: planes, exponent and slots are required; shapes of the fused kernel only (L=%d, d=%d)", L, d);
  return attn_bwd_impl(dctx, ld_ctx, q, k, v, ld, probs, B, H, L, d, nullptr, nullptr, nullptr, 4, p_drop, seed, stream_id, step_dev, g_planes,
                       g_plane_stride, g_panel_rows, g_cols, col_q, col_k, col_v, stat, stream, g_exp_dev);
}
static int attn_bwd_impl(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v, int64_t ld,
                         const float* probs, int B, int H, int L, int d, float* dq, float* dk, float* dv, int64_t ld_d, float p_drop,
                         uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* g_planes, int64_t g_plane_stride,
                         int64_t g_panel_rows, int g_cols, int col_q, int col_k, int col_v, float* stat, void* stream, const int* g_exp) {
  PXR_REQUIRE(dctx && q && k && v && probs && ((dq && dk && dv) || (g_planes && !dq && !dk && !dv)), "pxr_attn_bwd_f32: null pointer");
  PXR_REQUIRE(!g_exp || (g_planes && stat), "pxr_attn_bwd_h2s_f32: the stale-scale form needs planes and the statistics slots");
  PXR_REQUIRE(!g_planes || (pxr_attn_planes_supported(L, d) && p3_mat_ok(g_planes, g_plane_stride, g_panel_rows, (int64_t)B * L, g_cols) &&
                            col_q % 8 == 0 && col_k % 8 == 0 && col_v % 8 == 0 && col_q >= 0 && col_k >= 0 && col_v >= 0 &&
                            (int64_t)H * d + (col_q > col_k ? (col_q > col_v ? col_q : col_v) : (col_k > col_v ? col_k : col_v)) <= g_cols),
              "pxr_attn_bwd_planes_f32: planes are not available for this shape (L=%d, d=%d) or bad column ranges", L, d);
  int rc = attn_check(B, H, L, d, ld, "pxr_attn_bwd_f32");
  if (rc) return rc;
  PXR_REQUIRE(ld_d % 4 == 0 && ld_ctx % 4 == 0, "pxr_attn_bwd_f32: strides must be multiples of 4");
  if (B == 0) return PXR_OK;
  AttnArgs a{};
  a.q = q; a.k = k; a.v = v; a.ld = ld; a.ld_ctx = ld_ctx; a.probs = const_cast<float*>(probs);
  a.dctx = dctx; a.dq = dq; a.dk = dk; a.dv = dv; a.ld_d = ld_d; a.B = B; a.H = H; a.L = L; a.d = d;
  a.op = P3Mat{reinterpret_cast<__bf16*>(g_planes), g_plane_stride, g_panel_rows};
  a.pcol[0] = col_q; a.pcol[1] = col_k; a.pcol[2] = col_v;
  a.stat = stat;
  a.gexp = g_exp; a.status = pxr_status_word();
  a.sqrt_d = sqrtf((float)d);
  a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id; a.seed = seed;
  a.step_dev = step_dev;
  if (L > ATT_MAXL) {
    hipLaunchKernelGGL(attn_bwd_long_kernel, dim3(B * H), dim3(ATTL_NT), 0, (hipStream_t)stream, a);
    return pxr_check_launch("pxr_attn_bwd_f32(long)");
  }
  if (attn_use_mfma(d)) {
    if (d <= ATT_DC) {
      if (attn_use_two(L, d)) hipLaunchKernelGGL(attn_bwd_mfma2_kernel, dim3(B * H), dim3(ATT2_NT), 0, (hipStream_t)stream, a);
      else if (attn_mfma_waves() == 8) hipLaunchKernelGGL(attn_bwd_mfma1_kernel<8>, dim3(B * H), dim3(512), 0, (hipStream_t)stream, a);
      else hipLaunchKernelGGL(attn_bwd_mfma1_kernel<4>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
    }
    else hipLaunchKernelGGL(attn_bwd_mfma_kernel, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a);
    return pxr_check_launch("pxr_attn_bwd_f32(mfma)");
  }
  switch (attn_waves()) {
    case 4: hipLaunchKernelGGL(attn_bwd_kernel<4>, dim3(B * H), dim3(256), 0, (hipStream_t)stream, a); break;
    case 16: hipLaunchKernelGGL(attn_bwd_kernel<16>, dim3(B * H), dim3(1024), 0, (hipStream_t)stream, a); break;
    default: hipLaunchKernelGGL(attn_bwd_kernel<8>, dim3(B * H), dim3(512), 0, (hipStream_t)stream, a); break;
  }
  return pxr_check_launch("pxr_attn_bwd_f32");
}
