// layernorm.hip -- the three LayerNorm sites of the SASRec block, fused with their neighbours.
//
//   mode GATHER   : z = table[idx[b,t]] + pos_emb[t];  y = dropout(LN(z))          (sasrec.py:68,77-82 / :99-104)
//   mode RESIDUAL : z = dropout(x) + res;              y = LN(z)                   (layers.py:614-615, :670-671)
//
// LN is nn.LayerNorm: biased variance over the last dim, y = (z-mean)/sqrt(var+eps)*gamma + beta.
// One 64-lane wave owns one row and keeps it in registers (VEC float4 per lane, D <= VEC*256), two-pass
// mean / variance from registers (no E[x^2]-mean^2 cancellation), wave-shuffle reductions, 16-byte accesses.
// The forward saves xhat (the normalised row) and rstd for the backward; dropout masks are regenerated from
// the counter hash (pxr_common.h), never stored.
//
// Backward (autograd of the above):   a = dy*gamma;  dz = rstd*(a - mean(a) - xhat*mean(a*xhat))
//   dgamma = sum_rows dy*xhat, dbeta = sum_rows dy  -> per-block partials, summed in fixed order by
//   pxr_colsum-style stage 2 (deterministic, no float atomics).
#include "planes.cuh"

#include <cstdlib>

namespace pxr {

// The loss head fused into the LAST LayerNorm of the block (sasrec.py:86-92: the head reads exactly the rows this LayerNorm
// writes): forward = the two target-row dot products + the per-position loss term of bpr_loss.hip's bpr_fwd_kernel in the wave
// that holds the row; backward = bpr_bwd_kernel's d out row formed in registers instead of being written and read back.  Same
// formulas, same order of operations as bpr_loss.hip (bit-identical scores / coefficients).  items == null: no head.
struct BprHead {
  const float* table;      // [n_table, D]
  const int64_t* items;    // [B, 2, L+1]
  const int64_t* mask;     // [B, L]
  float* pos; float* neg;  // [B*L] scores (fwd: written; bwd: read)
  float* lossrow;          // fwd: [B*L] per-position loss term
  float* coef;             // bwd: [B*L] d loss / d(pos - neg)
  int64_t n_table;
  int B, L;
  float grad_scale;
  const float* grad_scale_dev;
};
__device__ __forceinline__ int64_t ln_clamp_id(int64_t r, int64_t n) { return r < 0 ? 0 : (r >= n ? n - 1 : r); }

struct LnFwdArgs {
  const float* x;          // RESIDUAL: [rows, D]
  const float* res;        // RESIDUAL: [rows, D] (may be null => no residual)
  const float* table;      // GATHER: [N, D]
  const int64_t* idx;      // GATHER: ids, element (b, t) at idx[b * idx_bstride + t]
  const float* pos;        // GATHER: [L, D]
  const float* gamma;
  const float* beta;
  float* y;                // [rows, D]
  float* xhat;             // [rows, D] or null (inference)
  float* rstd;             // [rows] or null
  int64_t idx_bstride;
  int64_t n_table;
  int rows, D, L;
  float eps;
  float p_drop;            // dropout probability (0 => off)
  uint32_t drop_thr;
  uint32_t stream;
  uint64_t seed;
  const int64_t* step_dev;  // optional device counter added to the seed (hipGraph replays advance it on the device)
  int32_t* status;          // GATHER: device status word (bad-index flag) or null
  P3Mat yp;                 // optional: y also as planes (the next GEMM's operand format, planes.cuh); p == null: none
  int yp_fmt;               // PXR_PLANES_BF16X3 | PXR_PLANES_H2 (two fp16 planes; a value beyond the fp16 range flags `status`)
  BprHead head;             // RESIDUAL only: the loss head's forward on the rows this launch writes (items == null: none)
  int nt;                   // streaming-store policy, set by launch_ln_fwd: bit 0 y, bit 1 the planes of y, bit 2 xhat, bit 3 table-row loads
};

// RPW = rows per wave.  RPW = 2 (large batches): both rows' ids and table rows are requested before either is reduced,
// so twice as many random 2 KB row fetches are in flight per CU (the kernel is a chain id -> row -> two wave
// reductions -> stores; at B >= 512 it is HBM-latency bound with one row per wave), and xhat -- written once, read
// only by the backward pass -- goes out with streaming stores so that it does not evict y, which the QKV GEMM reads next.
// STAGE (RPW = 1 only, big launches): the planes of y leave through LDS.  A row-per-wave store of a plane touches, per instruction,
// eight 64-byte row segments 64 * pr bytes apart -- HALF cache lines, and the memory side prices a half line like a whole one
// (measured at 102 400 rows of D = 512: the two fp16 planes, 210 MB, cost 98 us, as much as y + xhat together, 420 MB).  The four
// rows of a workgroup are adjacent in every panel, so the workgroup assembles the 256-byte run of each (plane, panel) in LDS
// (the memory image, padded to 320 bytes per run against bank conflicts) and writes it with 16-byte stores: whole lines.  The
// arithmetic and the bytes written are those of the unstaged kernel.
constexpr int LN_STAGE_RUN = 320;
template <int VEC, bool GATHER, int RPW = 1, bool STAGE = false>
__global__ void __launch_bounds__(256) ln_fwd_kernel(LnFwdArgs a) {
  static_assert(!STAGE || RPW == 1, "staging assumes one row per wave, four adjacent rows per workgroup");
  __shared__ __attribute__((aligned(16))) unsigned char stage_buf[STAGE ? 3 * VEC * 8 * LN_STAGE_RUN : 16];
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * RPW;   // rows of one XCD are contiguous (pxr_common.h)
  if (!STAGE && row0 >= a.rows) return;       // (a staging workgroup keeps its idle waves for the barrier)
  const int D = a.D;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  const bool drop = a.drop_thr != 0u;
  float4 v[RPW][VEC];
  float s[RPW];
  bool live[RPW];
#pragma unroll
  for (int w = 0; w < RPW; ++w) {
    const int row = row0 + w;
    live[w] = row < a.rows;
    s[w] = 0.f;
    const float* src;
    const float* add;
    if constexpr (GATHER) {
      const int rr = live[w] ? row : (STAGE ? a.rows - 1 : row0);     // (an idle wave of a staging workgroup re-reads the last row)
      const int b = rr / a.L, t = rr - b * a.L;
      int64_t r = a.idx[(int64_t)b * a.idx_bstride + t];
      if (r < 0 || r >= a.n_table) {   // an error in the reference (nn.Embedding raises): flag it, then clamp
        if (a.status && lane == 0) atomicOr(a.status, PXR_STATUS_BAD_INDEX);
        r = r < 0 ? 0 : a.n_table - 1;
      }
      src = a.table + r * D;
      add = a.pos + (int64_t)t * D;
    } else {
      const int rr = live[w] ? row : (STAGE ? a.rows - 1 : row0);
      src = a.x + (int64_t)rr * D;
      add = a.res ? a.res + (int64_t)rr * D : nullptr;
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < D) {
        float4 t4;
        if (GATHER && (a.nt & 8)) t4 = pxr_ld_stream(src + c);
        else t4 = *reinterpret_cast<const float4*>(src + c);
        if constexpr (!GATHER) {
          if (drop) {  // dropout on the sub-layer output BEFORE the residual add (layers.py:614, :670)
            const uint64_t e = (uint64_t)row * D + c;
            t4.x = pxr_keep(a.seed, a.stream, e + 0, a.drop_thr) ? t4.x * inv_keep : 0.f;
            t4.y = pxr_keep(a.seed, a.stream, e + 1, a.drop_thr) ? t4.y * inv_keep : 0.f;
            t4.z = pxr_keep(a.seed, a.stream, e + 2, a.drop_thr) ? t4.z * inv_keep : 0.f;
            t4.w = pxr_keep(a.seed, a.stream, e + 3, a.drop_thr) ? t4.w * inv_keep : 0.f;
          }
        }
        if (add) {
          const float4 r4 = *reinterpret_cast<const float4*>(add + c);
          t4.x += r4.x; t4.y += r4.y; t4.z += r4.z; t4.w += r4.w;
        }
        v[w][k] = t4;
        s[w] += (t4.x + t4.y) + (t4.z + t4.w);
      } else {
        v[w][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
#pragma unroll
  for (int w = 0; w < RPW; ++w) {
    const int row = row0 + w;
    if (!live[w]) continue;   // wave-uniform
    const float mean = wave_sum(s[w]) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < D) {
        const float dx = v[w][k].x - mean, dy = v[w][k].y - mean, dz = v[w][k].z - mean, dw = v[w][k].w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if (a.rstd && lane == 0) a.rstd[row] = rstd;
    const float* ep = nullptr;
    const float* en = nullptr;
    float hp = 0.f, hn = 0.f;
    if constexpr (!GATHER) {
      if (a.head.items) {   // wave-uniform
        const int b = row / a.head.L, t = row - b * a.head.L;
        const int64_t* it = a.head.items + (int64_t)b * 2 * (a.head.L + 1);
        ep = a.head.table + ln_clamp_id(it[t + 1], a.head.n_table) * D;
        en = a.head.table + ln_clamp_id(it[(a.head.L + 1) + t + 1], a.head.n_table) * D;
      }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < D) {
        const float4 g = *reinterpret_cast<const float4*>(a.gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(a.beta + c);
        float4 xh;
        xh.x = (v[w][k].x - mean) * rstd; xh.y = (v[w][k].y - mean) * rstd;
        xh.z = (v[w][k].z - mean) * rstd; xh.w = (v[w][k].w - mean) * rstd;
        if (a.xhat) {
          if (RPW > 1 || (a.nt & 4)) pxr_st_stream(a.xhat + (int64_t)row * D + c, xh);
          else *reinterpret_cast<float4*>(a.xhat + (int64_t)row * D + c) = xh;
        }
        float4 y;
        y.x = xh.x * g.x + be.x; y.y = xh.y * g.y + be.y; y.z = xh.z * g.z + be.z; y.w = xh.w * g.w + be.w;
        if constexpr (GATHER) {
          if (drop) {  // dropout AFTER the input LayerNorm (sasrec.py:82)
            const uint64_t e = (uint64_t)row * D + c;
            y.x = pxr_keep(a.seed, a.stream, e + 0, a.drop_thr) ? y.x * inv_keep : 0.f;
            y.y = pxr_keep(a.seed, a.stream, e + 1, a.drop_thr) ? y.y * inv_keep : 0.f;
            y.z = pxr_keep(a.seed, a.stream, e + 2, a.drop_thr) ? y.z * inv_keep : 0.f;
            y.w = pxr_keep(a.seed, a.stream, e + 3, a.drop_thr) ? y.w * inv_keep : 0.f;
          }
        }
        if (a.y) {
          if (a.nt & 1) pxr_st_stream(a.y + (int64_t)row * D + c, y);
          else *reinterpret_cast<float4*>(a.y + (int64_t)row * D + c) = y;
        }
        if constexpr (STAGE) {
          // the run of (plane q, panel cb) starts at (q * NCB + cb) * LN_STAGE_RUN; inside it the global panel image of the 4 rows
          const int cb = c >> 5;
          unsigned char* dst = stage_buf + cb * LN_STAGE_RUN + wave * 64 + ((((c >> 3) & 3) ^ ((row >> 2) & 3)) << 4) + ((c & 4) << 1);
          constexpr int PL = VEC * 8 * LN_STAGE_RUN;      // bytes of one plane's runs (NCB = D / 32 <= VEC * 8)
          if (a.yp_fmt == PXR_PLANES_H2) {
            const bool bad = !(fabsf(y.x) <= 65504.f) | !(fabsf(y.y) <= 65504.f) | !(fabsf(y.z) <= 65504.f) | !(fabsf(y.w) <= 65504.f);
            if (bad && a.status) atomicOr(a.status, PXR_STATUS_H2_RANGE);
            unsigned h0, l0, h1, l1;
            h2_split2(y.x, y.y, h0, l0);
            h2_split2(y.z, y.w, h1, l1);
            *reinterpret_cast<p3_u32x2*>(dst) = p3_u32x2{h0, h1};
            *reinterpret_cast<p3_u32x2*>(dst + PL) = p3_u32x2{l0, l1};
          } else {
            unsigned h0, m0, l0, h1, m1, l1;
            p3_split2(y.x, y.y, h0, m0, l0);
            p3_split2(y.z, y.w, h1, m1, l1);
            *reinterpret_cast<p3_u32x2*>(dst) = p3_u32x2{h0, h1};
            *reinterpret_cast<p3_u32x2*>(dst + PL) = p3_u32x2{m0, m1};
            *reinterpret_cast<p3_u32x2*>(dst + 2 * PL) = p3_u32x2{l0, l1};
          }
        } else {
          if (a.yp.p) px_store4s(a.yp, a.yp_fmt, a.status, row, c, y, (a.nt & 2) != 0);
        }
        if constexpr (!GATHER) {
          if (ep) {   // (same association as bpr_fwd_kernel: per float4, then across a lane's chunks in column order)
            const float4 pv = *reinterpret_cast<const float4*>(ep + c);
            const float4 nv = *reinterpret_cast<const float4*>(en + c);
            hp += (y.x * pv.x + y.y * pv.y) + (y.z * pv.z + y.w * pv.w);
            hn += (y.x * nv.x + y.y * nv.y) + (y.z * nv.z + y.w * nv.w);
          }
        }
      }
    }
    if constexpr (!GATHER) {
      if (ep) {
        hp = wave_sum(hp);
        hn = wave_sum(hn);
        if (lane == 0) {
          a.head.pos[row] = hp;
          a.head.neg[row] = hn;
          const float x = hp - hn;
          const float sg = 1.0f / (1.0f + expf(-x));
          a.head.lossrow[row] = -logf(sg + 1e-8f) * (float)a.head.mask[row];
        }
      }
    }
  }
  if constexpr (STAGE) {
    __syncthreads();
    const int r_blk = xcd_remap(blockIdx.x, gridDim.x) * 4;                 // first row of this workgroup (a multiple of 4)
    const int ncb = D >> 5, npl = a.yp_fmt == PXR_PLANES_H2 ? 2 : 3;
    constexpr int PL = VEC * 8 * LN_STAGE_RUN;
    unsigned char* base = reinterpret_cast<unsigned char*>(a.yp.p);
    for (int i = threadIdx.x; i < npl * ncb * 16; i += 256) {                // 16-byte pieces: 16 per 256-byte run
      const int piece = i & 15, run = i >> 4, q = run / ncb, cb = run - q * ncb;
      if (r_blk + (piece >> 2) >= a.rows) continue;                          // rows past the end hold nothing
      const p3_u32x4 v = *reinterpret_cast<const p3_u32x4*>(stage_buf + q * PL + cb * LN_STAGE_RUN + piece * 16);
      unsigned char* dst = base + ((int64_t)q * a.yp.ps + ((int64_t)cb * a.yp.pr + r_blk) * 32) * 2 + piece * 16;
      if (a.nt & 2) __builtin_nontemporal_store(v, reinterpret_cast<p3_u32x4*>(dst));
      else *reinterpret_cast<p3_u32x4*>(dst) = v;
    }
  }
}

struct LnBwdArgs {
  const float* dy;      // [rows, D] gradient w.r.t. the LN site's output (for GATHER: w.r.t. dropout(LN(z)))
  const float* xhat;    // [rows, D]
  const float* rstd;    // [rows]
  const float* gamma;
  float* dz;            // [rows, D] gradient w.r.t. z (= d residual; = d(table row + pos) in GATHER mode)
  float* dx;            // RESIDUAL with dropout: gradient w.r.t. x (= mask*dz/(1-p)); null => not needed
  float* part;          // [nblk, 2*D] per-block partial (dgamma | dbeta)
  int rows, D, rows_per_block;
  float p_drop;
  uint32_t drop_thr;
  uint32_t stream;
  uint64_t seed;
  const int64_t* step_dev;
  P3Mat gp;             // optional: planes of the gradient the next GEMMs read (dx when it is written, else dz)
  const int* gp_exp;    // null: three bf16 planes.  Else: two fp16 planes of gradient * 2^gp_exp[0] (a DEVICE exponent chosen before
                        // this launch -- the previous step's statistics, pxr_h2_sites_update -- range-checked and saturated)
  int dx_virtual;       // RESIDUAL with dropout, planes only: the site HAS a dx (the mask is applied) but its fp32 copy is not wanted
  int32_t* status;
  float* stat;          // optional: stat[blk] = max of |that gradient| over workgroup blk's rows -- one plain store per workgroup
                        // (pxr_ln_bwd_partial_rows(rows) of them, nothing to zero); pxr_h2_split_parts_f32 reduces them.  (Round 4
                        // raised ONE word with an atomic per wave: 3 200 same-address atomics, 30 of the launch's 40 us at B = 64.)
  float* zero;          // optional: zero_n floats this launch clears (the spread slots of the attention backward that follows it)
  int zero_n;
  BprHead head;         // RESIDUAL only: dy is not read but formed from the loss head's backward (dy == null then)
  const float* res;     // RESIDUAL without dropout, optional: [rows, D] added to dz before it is stored / measured / split -- the
                        // gradient that flows AROUND a pre-LN block's branch (dx = d_out + LN-backward(branch)): saves the add launch
};

// H2S: the planes are two fp16 planes under the device exponent a.gp_exp (stale scales), dx may be virtual (instantiated for D <= 1024)
template <int VEC, bool GATHER, bool H2S = false>
__global__ void __launch_bounds__(256) ln_bwd_kernel(LnBwdArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  __shared__ float red[3][2 * VEC * 256];  // waves 1..3 park their partial (dgamma | dbeta) here
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D = a.D;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  const bool drop = a.drop_thr != 0u;
  float4 accg[VEC], accb[VEC], gam[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    accg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    accb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (k * 64 + lane) * 4;
    gam[k] = (c < D) ? *reinterpret_cast<const float4*>(a.gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int blk = xcd_remap(blockIdx.x, gridDim.x);      // rows of one XCD are contiguous (pxr_common.h); partial index = blk
  const int r0 = blk * a.rows_per_block;
  const int r1 = min(a.rows, r0 + a.rows_per_block);
  float gmax = 0.f;        // max |gradient the next GEMMs read| over this thread's elements (a.stat)
  const bool has_dx = a.dx != nullptr || (H2S && a.dx_virtual != 0);
  float gsc = 1.0f;
  if constexpr (H2S) gsc = ldexpf(1.0f, a.gp_exp[0]);
  for (int row = r0 + wave; row < r1; row += 4) {
    float4 g4[VEC], xh[VEC];
    float s1 = 0.f, s2 = 0.f;
    const float* ep = nullptr;
    const float* en = nullptr;
    float cf = 0.f;
    if constexpr (!GATHER) {
      if (a.head.items) {   // wave-uniform; bpr_bwd_kernel's coefficient and rows
        const int b = row / a.head.L, t = row - b * a.head.L;
        const float x = a.head.pos[row] - a.head.neg[row];
        const float sg = 1.0f / (1.0f + expf(-x));
        cf = -((float)a.head.mask[row] / (float)a.head.B) * (sg * (1.0f - sg)) / (sg + 1e-8f) * a.head.grad_scale;
        if (a.head.grad_scale_dev) cf *= a.head.grad_scale_dev[0];
        if (lane == 0) a.head.coef[row] = cf;
        const int64_t* it = a.head.items + (int64_t)b * 2 * (a.head.L + 1);
        ep = a.head.table + ln_clamp_id(it[t + 1], a.head.n_table) * D;
        en = a.head.table + ln_clamp_id(it[(a.head.L + 1) + t + 1], a.head.n_table) * D;
      }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < D) {
        float4 d;
        if (ep) {
          d = make_float4(0.f, 0.f, 0.f, 0.f);
          if (cf != 0.f) {
            const float4 pv = *reinterpret_cast<const float4*>(ep + c);
            const float4 nv = *reinterpret_cast<const float4*>(en + c);
            d.x = cf * (pv.x - nv.x); d.y = cf * (pv.y - nv.y); d.z = cf * (pv.z - nv.z); d.w = cf * (pv.w - nv.w);
          }
        } else {
          d = *reinterpret_cast<const float4*>(a.dy + (int64_t)row * D + c);
        }
        if constexpr (GATHER) {
          if (drop) {
            const uint64_t e = (uint64_t)row * D + c;
            d.x = pxr_keep(a.seed, a.stream, e + 0, a.drop_thr) ? d.x * inv_keep : 0.f;
            d.y = pxr_keep(a.seed, a.stream, e + 1, a.drop_thr) ? d.y * inv_keep : 0.f;
            d.z = pxr_keep(a.seed, a.stream, e + 2, a.drop_thr) ? d.z * inv_keep : 0.f;
            d.w = pxr_keep(a.seed, a.stream, e + 3, a.drop_thr) ? d.w * inv_keep : 0.f;
          }
        }
        const float4 x = *reinterpret_cast<const float4*>(a.xhat + (int64_t)row * D + c);
        xh[k] = x;
        accg[k].x += d.x * x.x; accg[k].y += d.y * x.y; accg[k].z += d.z * x.z; accg[k].w += d.w * x.w;
        accb[k].x += d.x; accb[k].y += d.y; accb[k].z += d.z; accb[k].w += d.w;
        float4 t;
        t.x = d.x * gam[k].x; t.y = d.y * gam[k].y; t.z = d.z * gam[k].z; t.w = d.w * gam[k].w;
        g4[k] = t;
        s1 += (t.x + t.y) + (t.z + t.w);
        s2 += (t.x * x.x + t.y * x.y) + (t.z * x.z + t.w * x.w);
      } else {
        g4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        xh[k] = g4[k];
      }
    }
    const float c1 = wave_sum(s1) / (float)D;
    const float c2 = wave_sum(s2) / (float)D;
    const float rs = a.rstd[row];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < D) {
        float4 z;
        z.x = rs * (g4[k].x - c1 - xh[k].x * c2); z.y = rs * (g4[k].y - c1 - xh[k].y * c2);
        z.z = rs * (g4[k].z - c1 - xh[k].z * c2); z.w = rs * (g4[k].w - c1 - xh[k].w * c2);
        if constexpr (!GATHER) {
          if (a.res) {       // (res + z, the operand order of add_kernel(res, z): the same bits as the separate launch)
            const float4 r = *reinterpret_cast<const float4*>(a.res + (int64_t)row * D + c);
            z.x = r.x + z.x; z.y = r.y + z.y; z.z = r.z + z.z; z.w = r.w + z.w;
          }
        }
        *reinterpret_cast<float4*>(a.dz + (int64_t)row * D + c) = z;
        if constexpr (!GATHER) {
          if (!has_dx) gmax = fmaxf(gmax, fmaxf(fmaxf(fabsf(z.x), fabsf(z.y)), fmaxf(fabsf(z.z), fabsf(z.w))));
          if (a.gp.p && !has_dx) {
            if constexpr (H2S) px_store4_h2s(a.gp, a.status, row, c, z, gsc);
            else p3_store4(a.gp, row, c, z);
          }
          if (has_dx) {
            float4 o = z;
            if (drop) {
              const uint64_t e = (uint64_t)row * D + c;
              o.x = pxr_keep(a.seed, a.stream, e + 0, a.drop_thr) ? z.x * inv_keep : 0.f;
              o.y = pxr_keep(a.seed, a.stream, e + 1, a.drop_thr) ? z.y * inv_keep : 0.f;
              o.z = pxr_keep(a.seed, a.stream, e + 2, a.drop_thr) ? z.z * inv_keep : 0.f;
              o.w = pxr_keep(a.seed, a.stream, e + 3, a.drop_thr) ? z.w * inv_keep : 0.f;
            }
            if (a.dx) *reinterpret_cast<float4*>(a.dx + (int64_t)row * D + c) = o;
            gmax = fmaxf(gmax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
            if (a.gp.p) {
              if constexpr (H2S) px_store4_h2s(a.gp, a.status, row, c, o, gsc);
              else p3_store4(a.gp, row, c, o);
            }
          }
        }
      }
    }
  }
  __shared__ float smax[4];
  if constexpr (!GATHER) {
    if (a.stat) {            // (NaN: fmaxf drops it -- a NaN gradient still reaches the planes and their range flag)
      gmax = wave_max(gmax);
      if (lane == 0) smax[wave] = gmax;
    }
    if (a.zero && blockIdx.x == 0 && (int)threadIdx.x < a.zero_n) a.zero[threadIdx.x] = 0.f;
  }
  // cross-wave reduction of the per-lane partials in a fixed order (wave0 + wave1 + wave2 + wave3)
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (k * 64 + lane) * 4;
      *reinterpret_cast<float4*>(&red[wave - 1][c]) = accg[k];
      *reinterpret_cast<float4*>(&red[wave - 1][VEC * 256 + c]) = accb[k];
    }
  }
  __syncthreads();
  if constexpr (!GATHER) {
    if (a.stat && threadIdx.x == 0) a.stat[blk] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < D) {
        float4 g = accg[k], b = accb[k];
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const float4 og = *reinterpret_cast<const float4*>(&red[w][c]);
          const float4 ob = *reinterpret_cast<const float4*>(&red[w][VEC * 256 + c]);
          g.x += og.x; g.y += og.y; g.z += og.z; g.w += og.w;
          b.x += ob.x; b.y += ob.y; b.z += ob.z; b.w += ob.w;
        }
        *reinterpret_cast<float4*>(a.part + (int64_t)blk * 2 * D + c) = g;
        *reinterpret_cast<float4*>(a.part + (int64_t)blk * 2 * D + D + c) = b;
      }
    }
  }
}

static inline int ln_vec_for(int D) {
  int v = (D + 255) / 256;
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Store policy of a LayerNorm forward launch.  Its outputs are read by LATER launches (y and its planes by the next GEMM, xhat by
// the backward pass): once they are far larger than the L2s (4 MB per XCD) nothing of them survives there, and write-back
// allocation only evicts what the launch still needs (gamma / beta / pos, the ids).  From PXR_LN_NT_BYTES (default 48 MB) of fp32
// output per tensor the stores stream (non-temporal).  PXR_LN_NT overrides the bits (measuring knob).
static int ln_fwd_nt(const LnFwdArgs& a) {
  static const int env_nt = getenv("PXR_LN_NT") ? atoi(getenv("PXR_LN_NT")) : -1;
  static const int64_t nt_bytes = getenv("PXR_LN_NT_BYTES") ? atoll(getenv("PXR_LN_NT_BYTES")) : (48ll << 20);
  if (env_nt >= 0) return env_nt;
  return (int64_t)a.rows * a.D * 4 >= nt_bytes ? 7 : 0;
}

template <bool GATHER>
static int launch_ln_fwd(const LnFwdArgs& a_in, hipStream_t st) {
  LnFwdArgs a = a_in;
  a.nt = ln_fwd_nt(a);
  // large batches: two rows per wave (see ln_fwd_kernel); small ones keep one row per wave -- there the launch is a
  // few workgroups per CU and the shortest chain wins
  // (A/B knob, OFF by default: measured 3.75 vs 4.05 TB/s at B = 512 -- the row fetches are not what limits the kernel)
  static const int rpw2_rows = getenv("PXR_LN_RPW2_ROWS") ? atoi(getenv("PXR_LN_RPW2_ROWS")) : 0x7fffffff;
  if (a.rows >= rpw2_rows && a.D <= 1024) {
    const int blocks2 = (a.rows + 7) / 8;
    switch (ln_vec_for(a.D)) {
      case 1: hipLaunchKernelGGL((ln_fwd_kernel<1, GATHER, 2>), dim3(blocks2), dim3(256), 0, st, a); break;
      case 2: hipLaunchKernelGGL((ln_fwd_kernel<2, GATHER, 2>), dim3(blocks2), dim3(256), 0, st, a); break;
      default: hipLaunchKernelGGL((ln_fwd_kernel<4, GATHER, 2>), dim3(blocks2), dim3(256), 0, st, a); break;
    }
    return pxr_check_launch("pxr_ln_fwd");
  }
  const int blocks = (a.rows + 3) / 4;
  // planes through LDS (whole-line stores) once the launch streams to HBM: from PXR_LN_STAGE_ROWS rows when set, else from 96 MB
  // of fp32 y (measured at D = 512: 102 400 rows 234 -> 192 us with two fp16 planes, 281 -> 213 with three bf16 ones; at 25 600
  // rows, where the outputs still fit the 256 MB MALL, half lines cost nothing and the extra barrier does: 42 -> 50 us);
  // D a multiple of 32 in 256 .. 512 (PXR_LN_STAGE_ROWS set: up to 1024)
  static const int stage_rows = getenv("PXR_LN_STAGE_ROWS") ? atoi(getenv("PXR_LN_STAGE_ROWS")) : -1;
  const bool stage = stage_rows >= 0 ? a.rows >= stage_rows : (int64_t)a.rows * a.D * 4 >= (96ll << 20);
  // (D <= 512: at D = 768 -- the image tower, 69 344 rows -- the staged launch measured 104 us against 99.5: 30 KB of LDS per
  // workgroup there, and three quarters of a row's segments already share lines with their neighbours' in time)
  if (a.yp.p && stage && a.D % 32 == 0 && a.D <= (stage_rows >= 0 ? 1024 : 512) && a.D >= 256) {
    switch (ln_vec_for(a.D)) {
      case 1: hipLaunchKernelGGL((ln_fwd_kernel<1, GATHER, 1, true>), dim3(blocks), dim3(256), 0, st, a); break;
      case 2: hipLaunchKernelGGL((ln_fwd_kernel<2, GATHER, 1, true>), dim3(blocks), dim3(256), 0, st, a); break;
      default: hipLaunchKernelGGL((ln_fwd_kernel<4, GATHER, 1, true>), dim3(blocks), dim3(256), 0, st, a); break;
    }
    return pxr_check_launch("pxr_ln_fwd");
  }
  switch (ln_vec_for(a.D)) {
    case 1: hipLaunchKernelGGL((ln_fwd_kernel<1, GATHER>), dim3(blocks), dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((ln_fwd_kernel<2, GATHER>), dim3(blocks), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((ln_fwd_kernel<4, GATHER>), dim3(blocks), dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((ln_fwd_kernel<8, GATHER>), dim3(blocks), dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL((ln_fwd_kernel<16, GATHER>), dim3(blocks), dim3(256), 0, st, a); break;
    default: pxr_set_error("layernorm: D=%d > 4096 unsupported", a.D); return PXR_ERR_BAD_ARG;
  }
  return pxr_check_launch("pxr_ln_fwd");
}

template <bool GATHER>
static int launch_ln_bwd(const LnBwdArgs& a, int nblk, hipStream_t st) {
  if (a.gp_exp) {       // stale-scale fp16 planes: residual sites, D <= 1024
    if constexpr (!GATHER) {
      switch (ln_vec_for(a.D)) {
        case 1: hipLaunchKernelGGL((ln_bwd_kernel<1, false, true>), dim3(nblk), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL((ln_bwd_kernel<2, false, true>), dim3(nblk), dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL((ln_bwd_kernel<4, false, true>), dim3(nblk), dim3(256), 0, st, a); break;
        default: pxr_set_error("pxr_ln_bwd_h2s_f32: D=%d > 1024 unsupported", a.D); return PXR_ERR_BAD_ARG;
      }
      return pxr_check_launch("pxr_ln_bwd_h2s");
    } else {
      pxr_set_error("pxr_ln_bwd_h2s_f32: residual sites only");
      return PXR_ERR_BAD_ARG;
    }
  }
  switch (ln_vec_for(a.D)) {
    case 1: hipLaunchKernelGGL((ln_bwd_kernel<1, GATHER>), dim3(nblk), dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((ln_bwd_kernel<2, GATHER>), dim3(nblk), dim3(256), 0, st, a); break;
    case 4: hipLaunchKernelGGL((ln_bwd_kernel<4, GATHER>), dim3(nblk), dim3(256), 0, st, a); break;
    case 8: hipLaunchKernelGGL((ln_bwd_kernel<8, GATHER>), dim3(nblk), dim3(256), 0, st, a); break;
    case 16: hipLaunchKernelGGL((ln_bwd_kernel<16, GATHER>), dim3(nblk), dim3(256), 0, st, a); break;
    default: pxr_set_error("layernorm backward: D=%d > 4096 unsupported", a.D); return PXR_ERR_BAD_ARG;
  }
  return pxr_check_launch("pxr_ln_bwd");
}

static inline int ln_bwd_blocks(int rows, int* rows_per_block) {
  // <= 1024 blocks (4 per CU: a lone workgroup per CU walks its rows at load latency) => <= 1024 partial rows for
  // the stage-2 reduction, which sums them with 8 independent loads in flight per lane
  int rpb = (rows + 1023) / 1024;
  if (rpb < 4) rpb = 4;
  *rows_per_block = rpb;
  return (rows + rpb - 1) / rpb;
}

}  // namespace pxr

using namespace pxr;

// y = dropout(LN(table[idx] + pos))   rows = B*L; idx element (b,t) at idx[b*idx_bstride + t].
// xhat / rstd may be null (inference).  (sasrec.py:68,77-82 train; :99-104 predict)
extern "C" int pxr_input_ln_fwd_planes_f32(const float* table, int64_t n_table, const int64_t* idx, int64_t idx_bstride,
                                           const float* pos, const float* gamma, const float* beta, float eps, int B, int L,
                                           int D, float* y, float* xhat, float* rstd, float p_drop, uint64_t seed,
                                           uint32_t stream_id, const int64_t* step_dev, void* y_planes,
                                           int64_t y_plane_stride, int64_t y_panel_rows, void* stream);
extern "C" int pxr_input_ln_fwd_f32(const float* table, int64_t n_table, const int64_t* idx, int64_t idx_bstride,
                                    const float* pos, const float* gamma, const float* beta, float eps, int B, int L,
                                    int D, float* y, float* xhat, float* rstd, float p_drop, uint64_t seed,
                                    uint32_t stream_id, const int64_t* step_dev, void* stream) {
  return pxr_input_ln_fwd_planes_f32(table, n_table, idx, idx_bstride, pos, gamma, beta, eps, B, L, D, y, xhat, rstd, p_drop,
                                     seed, stream_id, step_dev, nullptr, 0, 0, stream);
}
// the same, y additionally written as bf16x3 planes (y_planes may be NULL): the QKV GEMM's operand
extern "C" int pxr_input_ln_fwd_planes_f32(const float* table, int64_t n_table, const int64_t* idx, int64_t idx_bstride,
                                           const float* pos, const float* gamma, const float* beta, float eps, int B, int L,
                                           int D, float* y, float* xhat, float* rstd, float p_drop, uint64_t seed,
                                           uint32_t stream_id, const int64_t* step_dev, void* y_planes,
                                           int64_t y_plane_stride, int64_t y_panel_rows, void* stream) {
  PXR_REQUIRE(table && idx && pos && gamma && beta && y, "pxr_input_ln_fwd_f32: null pointer");
  PXR_REQUIRE(p3_mat_ok(y_planes, y_plane_stride, y_panel_rows, (int64_t)B * L, D), "pxr_input_ln_fwd_planes_f32: bad planes");
  PXR_REQUIRE(D > 0 && D % 4 == 0 && B >= 0 && L > 0, "pxr_input_ln_fwd_f32: bad shape");
  PXR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "pxr_input_ln_fwd_f32: bad dropout p");
  if (B == 0) return PXR_OK;
  LnFwdArgs a{};
  a.table = table; a.idx = idx; a.pos = pos; a.gamma = gamma; a.beta = beta; a.y = y; a.xhat = xhat; a.rstd = rstd;
  a.idx_bstride = idx_bstride; a.n_table = n_table; a.rows = B * L; a.D = D; a.L = L; a.eps = eps;
  a.status = pxr_status_word();
  a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id; a.seed = seed;
  a.step_dev = step_dev;
  a.yp = P3Mat{reinterpret_cast<__bf16*>(y_planes), y_plane_stride, y_panel_rows};
  return launch_ln_fwd<true>(a, (hipStream_t)stream);
}

// y = LN(dropout(x) + res)   (layers.py:614-615, :670-671).  res may be null.
extern "C" int pxr_ln_residual_fwd_planes_f32(const float* x, const float* res, const float* gamma, const float* beta,
                                              float eps, int rows, int D, float* y, float* xhat, float* rstd, float p_drop,
                                              uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* y_planes,
                                              int64_t y_plane_stride, int64_t y_panel_rows, void* stream);
extern "C" int pxr_ln_residual_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta,
                                       float eps, int rows, int D, float* y, float* xhat, float* rstd, float p_drop,
                                       uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* stream) {
  return pxr_ln_residual_fwd_planes_f32(x, res, gamma, beta, eps, rows, D, y, xhat, rstd, p_drop, seed, stream_id, step_dev,
                                        nullptr, 0, 0, stream);
}
extern "C" int pxr_ln_residual_fwd_planes_f32(const float* x, const float* res, const float* gamma, const float* beta,
                                              float eps, int rows, int D, float* y, float* xhat, float* rstd, float p_drop,
                                              uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* y_planes,
                                              int64_t y_plane_stride, int64_t y_panel_rows, void* stream) {
  PXR_REQUIRE(x && gamma && beta && (y || y_planes), "pxr_ln_residual_fwd_f32: null pointer");   // y optional next to planes
  PXR_REQUIRE(p3_mat_ok(y_planes, y_plane_stride, y_panel_rows, rows, D), "pxr_ln_residual_fwd_planes_f32: bad planes");
  PXR_REQUIRE(D > 0 && D % 4 == 0 && rows >= 0, "pxr_ln_residual_fwd_f32: bad shape");
  PXR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "pxr_ln_residual_fwd_f32: bad dropout p");
  if (rows == 0) return PXR_OK;
  LnFwdArgs a{};
  a.x = x; a.res = res; a.gamma = gamma; a.beta = beta; a.y = y; a.xhat = xhat; a.rstd = rstd;
  a.rows = rows; a.D = D; a.L = 1; a.eps = eps;
  a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id; a.seed = seed;
  a.step_dev = step_dev;
  a.yp = P3Mat{reinterpret_cast<__bf16*>(y_planes), y_plane_stride, y_panel_rows};
  return launch_ln_fwd<false>(a, (hipStream_t)stream);
}
// The block's LAST LayerNorm with the loss head's forward fused in (BprHead): y = LN(dropout(x) + res) over rows = B*L, and
// for every row the two target-row scores + the per-position loss term; then the fixed-order loss reduction (bpr_loss.hip).
// Replaces pxr_ln_residual_fwd_f32 + pxr_bpr_loss_fwd_f32 (reference layers.py:670-671 + sasrec.py:86-92): one launch and one
// pass over `y` less; bit-identical outputs.
extern "C" int pxr_bpr_loss_reduce_f32(const float* lossrow, int B, int L, float* loss, void* stream);
extern "C" int pxr_ln_residual_bpr_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta, float eps, int B,
                                           int L, int D, float* y, float* xhat, float* rstd, float p_drop, uint64_t seed,
                                           uint32_t stream_id, const int64_t* step_dev, const float* table, int64_t n_table,
                                           const int64_t* items, const int64_t* masked_index, float* pos_score, float* neg_score,
                                           float* lossrow, float* loss, void* stream) {
  PXR_REQUIRE(x && gamma && beta && y && table && items && masked_index && pos_score && neg_score && lossrow && loss,
              "pxr_ln_residual_bpr_fwd_f32: null pointer");
  PXR_REQUIRE(D > 0 && D % 4 == 0 && B > 0 && L > 0 && n_table > 0, "pxr_ln_residual_bpr_fwd_f32: bad shape");
  PXR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "pxr_ln_residual_bpr_fwd_f32: bad dropout p");
  LnFwdArgs a{};
  a.x = x; a.res = res; a.gamma = gamma; a.beta = beta; a.y = y; a.xhat = xhat; a.rstd = rstd;
  a.rows = B * L; a.D = D; a.L = 1; a.eps = eps;
  a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id; a.seed = seed;
  a.step_dev = step_dev;
  a.head.table = table; a.head.items = items; a.head.mask = masked_index; a.head.pos = pos_score; a.head.neg = neg_score;
  a.head.lossrow = lossrow; a.head.n_table = n_table; a.head.B = B; a.head.L = L;
  const int rc = launch_ln_fwd<false>(a, (hipStream_t)stream);
  if (rc) return rc;
  return pxr_bpr_loss_reduce_f32(lossrow, B, L, loss, stream);
}
// the same with y as TWO fp16 planes (planes.cuh "h2", unit scale): the operand of pxr_gemm_h2_f32 -- the image tower, the sequence
// block of large batches.  A LayerNorm output beyond the fp16 range sets PXR_STATUS_H2_RANGE in the registered status word.
extern "C" int pxr_ln_residual_fwd_h2_f32(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                                          int rows, int D, float* y, float* xhat, float* rstd, float p_drop, uint64_t seed,
                                          uint32_t stream_id, const int64_t* step_dev, void* y_planes, int64_t y_plane_stride,
                                          int64_t y_panel_rows, void* stream) {
  PXR_REQUIRE(x && gamma && beta && y_planes, "pxr_ln_residual_fwd_h2_f32: null pointer");
  PXR_REQUIRE(p3_mat_ok(y_planes, y_plane_stride, y_panel_rows, rows, D), "pxr_ln_residual_fwd_h2_f32: bad planes");
  PXR_REQUIRE(D > 0 && D % 4 == 0 && rows >= 0, "pxr_ln_residual_fwd_h2_f32: bad shape");
  PXR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "pxr_ln_residual_fwd_h2_f32: bad dropout p");
  if (rows == 0) return PXR_OK;
  LnFwdArgs a{};
  a.x = x; a.res = res; a.gamma = gamma; a.beta = beta; a.y = y; a.xhat = xhat; a.rstd = rstd;
  a.rows = rows; a.D = D; a.L = 1; a.eps = eps;
  a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id; a.seed = seed;
  a.step_dev = step_dev;
  a.status = pxr_status_word();
  a.yp = P3Mat{reinterpret_cast<__bf16*>(y_planes), y_plane_stride, y_panel_rows};
  a.yp_fmt = PXR_PLANES_H2;
  return launch_ln_fwd<false>(a, (hipStream_t)stream);
}
extern "C" int pxr_input_ln_fwd_h2_f32(const float* table, int64_t n_table, const int64_t* idx, int64_t idx_bstride,
                                       const float* pos, const float* gamma, const float* beta, float eps, int B, int L, int D,
                                       float* y, float* xhat, float* rstd, float p_drop, uint64_t seed, uint32_t stream_id,
                                       const int64_t* step_dev, void* y_planes, int64_t y_plane_stride, int64_t y_panel_rows,
                                       void* stream) {
  PXR_REQUIRE(table && idx && pos && gamma && beta && y && y_planes, "pxr_input_ln_fwd_h2_f32: null pointer");
  PXR_REQUIRE(p3_mat_ok(y_planes, y_plane_stride, y_panel_rows, (int64_t)B * L, D), "pxr_input_ln_fwd_h2_f32: bad planes");
  PXR_REQUIRE(D > 0 && D % 4 == 0 && B >= 0 && L > 0, "pxr_input_ln_fwd_h2_f32: bad shape");
  PXR_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "pxr_input_ln_fwd_h2_f32: bad dropout p");
  if (B == 0) return PXR_OK;
  LnFwdArgs a{};
  a.table = table; a.idx = idx; a.pos = pos; a.gamma = gamma; a.beta = beta; a.y = y; a.xhat = xhat; a.rstd = rstd;
  a.idx_bstride = idx_bstride; a.n_table = n_table; a.rows = B * L; a.D = D; a.L = L; a.eps = eps;
  a.status = pxr_status_word();
  a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id; a.seed = seed;
  a.step_dev = step_dev;
  a.yp = P3Mat{reinterpret_cast<__bf16*>(y_planes), y_plane_stride, y_panel_rows};
  a.yp_fmt = PXR_PLANES_H2;
  return launch_ln_fwd<true>(a, (hipStream_t)stream);
}

extern "C" int pxr_ln_bwd_partial_rows(int rows) {
  int rpb;
  return ln_bwd_blocks(rows, &rpb);
}

extern "C" int64_t pxr_ln_bwd_ws_bytes(int rows, int D) {
  int rpb;
  const int nblk = ln_bwd_blocks(rows, &rpb);
  return (int64_t)nblk * 2 * D * (int64_t)sizeof(float);
}

// Backward of either LN site.  gather_mode=1: dy is w.r.t. dropout(LN(z)) and the mask is re-applied to dy;
// gather_mode=0: dx (optional) = dropout-mask(dz)/(1-p) is the gradient w.r.t. the sub-layer output x and dz
// the gradient w.r.t. the residual.  dgamma/dbeta are OVERWRITTEN (not accumulated).
extern "C" int pxr_ln_bwd_planes_f32(int gather_mode, const float* dy, const float* xhat, const float* rstd,
                                     const float* gamma, int rows, int D, float* dz, float* dx, float* dgamma, float* dbeta,
                                     float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* ws,
                                     int64_t ws_bytes, void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows,
                                     void* stream);
extern "C" int pxr_ln_bwd_f32(int gather_mode, const float* dy, const float* xhat, const float* rstd,
                              const float* gamma, int rows, int D, float* dz, float* dx, float* dgamma, float* dbeta,
                              float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* ws,
                              int64_t ws_bytes, void* stream) {
  return pxr_ln_bwd_planes_f32(gather_mode, dy, xhat, rstd, gamma, rows, D, dz, dx, dgamma, dbeta, p_drop, seed, stream_id,
                               step_dev, ws, ws_bytes, nullptr, 0, 0, stream);
}
// the same (gather_mode = 0), the gradient the next GEMMs read (dx when given, else dz) additionally written as planes
static int ln_bwd_impl(int gather_mode, const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int D,
                       float* dz, float* dx, float* dgamma, float* dbeta, float p_drop, uint64_t seed, uint32_t stream_id,
                       const int64_t* step_dev, void* ws, int64_t ws_bytes, void* g_planes, int64_t g_plane_stride,
                       int64_t g_panel_rows, float* stat, void* stream, const BprHead* head = nullptr, float* zero = nullptr,
                       int zero_n = 0, const int* g_exp = nullptr, int dx_virtual = 0, const float* res = nullptr);
extern "C" int pxr_ln_bwd_planes_f32(int gather_mode, const float* dy, const float* xhat, const float* rstd,
                                     const float* gamma, int rows, int D, float* dz, float* dx, float* dgamma, float* dbeta,
                                     float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* ws,
                                     int64_t ws_bytes, void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows,
                                     void* stream) {
  return ln_bwd_impl(gather_mode, dy, xhat, rstd, gamma, rows, D, dz, dx, dgamma, dbeta, p_drop, seed, stream_id, step_dev, ws, ws_bytes,
                     g_planes, g_plane_stride, g_panel_rows, nullptr, stream);
}
// pxr_ln_bwd_f32 (residual sites) that also leaves max |gradient the next GEMMs read| (dx when given, else dz) in *stat by atomic
// maxima -- the caller zeroes the slot; pxr_h2_split_auto_multi_f32(col_stats = 2) then needs no statistics pass of its own
extern "C" int pxr_ln_bwd_stat_f32(const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int D,
                                   float* dz, float* dx, float* dgamma, float* dbeta, float p_drop, uint64_t seed, uint32_t stream_id,
                                   const int64_t* step_dev, void* ws, int64_t ws_bytes, float* stat, float* zero, int zero_n,
                                   void* stream) {
  PXR_REQUIRE(stat, "pxr_ln_bwd_stat_f32: null statistics buffer");
  return ln_bwd_impl(0, dy, xhat, rstd, gamma, rows, D, dz, dx, dgamma, dbeta, p_drop, seed, stream_id, step_dev, ws, ws_bytes, nullptr, 0, 0,
                     stat, stream, nullptr, zero, zero_n);
}
// A residual site's backward in a PRE-LN block (the image tower): dz_out = res + LN-backward(dy) -- the gradient of the block's input
// is the gradient of its output plus what comes back through the branch's LayerNorm -- in one launch instead of LayerNorm backward +
// add; stat (optional): this launch's partial maxima of |dz_out| (pxr_ln_bwd_partial_rows(rows) words) for pxr_h2_split_parts_f32.
// No dropout at these sites (CLIPEncoderLayer has none: HF modeling_clip.py, reached from the reference's load.py:90-120).
extern "C" int pxr_ln_bwd_res_f32(const float* dy, const float* xhat, const float* rstd, const float* gamma, const float* res, int rows,
                                  int D, float* dz, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, float* stat, void* stream) {
  PXR_REQUIRE(res, "pxr_ln_bwd_res_f32: null residual gradient");
  return ln_bwd_impl(0, dy, xhat, rstd, gamma, rows, D, dz, nullptr, dgamma, dbeta, 0.f, 0, 0, nullptr, ws, ws_bytes, nullptr, 0, 0, stat,
                     stream, nullptr, nullptr, 0, nullptr, 0, res);
}
// The backward of the block's LAST LayerNorm with the loss head's backward fused in: dy is not read but formed per row from the
// saved scores (bpr_loss.hip: coef * (E[pos] - E[neg])); coef [B*L] is written for the table-gradient segment sums.  Replaces
// pxr_bpr_loss_bwd_f32 + pxr_ln_bwd_planes_f32 / pxr_ln_bwd_stat_f32 (gather_mode 0); g_planes and stat are both optional.
extern "C" int pxr_bpr_ln_bwd_f32(const float* pos_score, const float* neg_score, const float* table, int64_t n_table,
                                  const int64_t* items, const int64_t* masked_index, int B, int L, float grad_scale,
                                  const float* grad_scale_dev, float* coef, const float* xhat, const float* rstd,
                                  const float* gamma, int D, float* dz, float* dx, float* dgamma, float* dbeta, float p_drop,
                                  uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* ws, int64_t ws_bytes,
                                  void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows, float* stat, void* stream) {
  PXR_REQUIRE(pos_score && neg_score && table && items && masked_index && coef && B > 0 && L > 0 && n_table > 0,
              "pxr_bpr_ln_bwd_f32: null pointer / bad shape");
  BprHead h{};
  h.table = table; h.items = items; h.mask = masked_index; h.pos = const_cast<float*>(pos_score); h.neg = const_cast<float*>(neg_score);
  h.coef = coef; h.n_table = n_table; h.B = B; h.L = L; h.grad_scale = grad_scale; h.grad_scale_dev = grad_scale_dev;
  return ln_bwd_impl(0, nullptr, xhat, rstd, gamma, B * L, D, dz, dx, dgamma, dbeta, p_drop, seed, stream_id, step_dev, ws, ws_bytes,
                     g_planes, g_plane_stride, g_panel_rows, stat, stream, &h);
}
// A residual site's backward whose GEMM-facing gradient (dropout applied when p_drop > 0; no fp32 copy of it is written) leaves ONLY as
// two fp16 planes of gradient * 2^g_exp_dev[0] -- an exponent that exists BEFORE the launch (the previous step's maximum of the same
// gradient less PXR headroom binades: pxr_h2_sites_update) -- range-checked and saturated (PXR_STATUS_H2_STALE), together with this
// step's partial maxima in stat[pxr_ln_bwd_partial_rows(rows)] for the next update.  pos_score != NULL: the loss head's backward is
// fused in as in pxr_bpr_ln_bwd_f32 (dy unused).  Replaces pxr_ln_bwd_stat_f32 / pxr_bpr_ln_bwd_f32 + pxr_h2_split_parts_f32.
extern "C" int pxr_ln_bwd_h2s_f32(const float* pos_score, const float* neg_score, const float* table, int64_t n_table, const int64_t* items,
                                  const int64_t* masked_index, int B, int L, float grad_scale, const float* grad_scale_dev, float* coef,
                                  const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int D, float* dz,
                                  float* dgamma, float* dbeta, float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev,
                                  void* ws, int64_t ws_bytes, void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows,
                                  const int* g_exp_dev, float* stat, float* zero, int zero_n, void* stream) {
  PXR_REQUIRE(g_planes && g_exp_dev && stat, "pxr_ln_bwd_h2s_f32: planes, their device exponent and the statistics buffer are required");
  BprHead h{};
  if (pos_score) {
    PXR_REQUIRE(neg_score && table && items && masked_index && coef && B > 0 && L > 0 && n_table > 0 && rows == B * L,
                "pxr_ln_bwd_h2s_f32: null pointer / bad shape of the fused loss head");
    h.table = table; h.items = items; h.mask = masked_index; h.pos = const_cast<float*>(pos_score); h.neg = const_cast<float*>(neg_score);
    h.coef = coef; h.n_table = n_table; h.B = B; h.L = L; h.grad_scale = grad_scale; h.grad_scale_dev = grad_scale_dev;
  }
  return ln_bwd_impl(0, pos_score ? nullptr : dy, xhat, rstd, gamma, rows, D, dz, nullptr, dgamma, dbeta, p_drop, seed, stream_id, step_dev, ws,
                     ws_bytes, g_planes, g_plane_stride, g_panel_rows, stat, stream, pos_score ? &h : nullptr, zero, zero_n, g_exp_dev,
                     p_drop > 0.f ? 1 : 0);
}
static int ln_bwd_impl(int gather_mode, const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int D,
                       float* dz, float* dx, float* dgamma, float* dbeta, float p_drop, uint64_t seed, uint32_t stream_id,
                       const int64_t* step_dev, void* ws, int64_t ws_bytes, void* g_planes, int64_t g_plane_stride,
                       int64_t g_panel_rows, float* stat, void* stream, const BprHead* head, float* zero, int zero_n, const int* g_exp,
                       int dx_virtual, const float* res) {
  PXR_REQUIRE((dy || head) && xhat && rstd && gamma && dz && ws, "pxr_ln_bwd_f32: null pointer");
  PXR_REQUIRE(!res || (!gather_mode && !dx && !dx_virtual && p_drop == 0.f), "pxr_ln_bwd_res_f32: residual sites without dropout only");
  PXR_REQUIRE(!g_exp || g_planes, "pxr_ln_bwd_h2s_f32: a plane exponent without planes");
  PXR_REQUIRE(!dx_virtual || (g_planes && !dx && !gather_mode), "pxr_ln_bwd_h2s_f32: a planes-only dx needs planes (residual sites)");
  PXR_REQUIRE(zero_n >= 0 && zero_n <= 256 && (zero || zero_n == 0), "pxr_ln_bwd_f32: at most 256 floats to clear");
  PXR_REQUIRE(p3_mat_ok(g_planes, g_plane_stride, g_panel_rows, rows, D) && !(g_planes && gather_mode),
              "pxr_ln_bwd_planes_f32: bad planes (residual sites only)");
  PXR_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "pxr_ln_bwd_f32: dgamma and dbeta must both be given or both NULL");
  PXR_REQUIRE(D > 0 && D % 4 == 0 && rows > 0, "pxr_ln_bwd_f32: bad shape");
  LnBwdArgs a{};
  a.dy = dy; a.xhat = xhat; a.rstd = rstd; a.gamma = gamma; a.dz = dz; a.dx = dx; a.part = (float*)ws;
  a.rows = rows; a.D = D;
  const int nblk = ln_bwd_blocks(rows, &a.rows_per_block);
  if ((int64_t)nblk * 2 * D * 4 > ws_bytes) {
    pxr_set_error("pxr_ln_bwd_f32: workspace too small");
    return PXR_ERR_WORKSPACE;
  }
  a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id; a.seed = seed;
  a.step_dev = step_dev;
  a.gp = P3Mat{reinterpret_cast<__bf16*>(g_planes), g_plane_stride, g_panel_rows};
  a.gp_exp = g_exp; a.dx_virtual = dx_virtual; a.status = pxr_status_word();
  a.stat = stat;
  a.zero = zero; a.zero_n = zero_n;
  a.res = res;
  if (head) a.head = *head;
  hipStream_t st = (hipStream_t)stream;
  int rc = gather_mode ? launch_ln_bwd<true>(a, nblk, st) : launch_ln_bwd<false>(a, nblk, st);
  if (rc) return rc;
  if (!dgamma) return PXR_OK;  // deferred: the caller reduces ws ([pxr_ln_bwd_partial_rows, 2*D]) later, e.g. with
                               // pxr_reduce_partials_multi_f32 together with the other sites of the backward pass
  hipLaunchKernelGGL(pxr_reduce_partials_kernel, dim3((2 * D + PXR_RED_CX - 1) / PXR_RED_CX), dim3(256), 0, st, (const float*)ws, nblk,
                     2 * D, dgamma, dbeta, D);
  return pxr_check_launch("pxr_ln_bwd_f32(reduce)");
}

// ---- several partial reductions in one launch ------------------------------------------------------------------
struct MultiReduce {
  const float* part[16]; float* out_a[16]; float* out_b[16];
  int P[16], N[16], split[16], blk_begin[17];
  int n;
  int64_t* bump;   // optional device counter incremented by one (e.g. the dropout step counter: this is the last
                   // kernel of a backward pass that could read it, so the separate 1-thread launch is saved)
};
__global__ void __launch_bounds__(256) reduce_partials_multi_kernel(MultiReduce m) {
  __shared__ float red[PXR_RED_CY][PXR_RED_CX + 1];
  int pi = 0;
  for (int i = 1; i < m.n; ++i)
    if ((int)blockIdx.x >= m.blk_begin[i]) pi = i;
  const float* part = m.part[pi];
  const int P = m.P[pi], N = m.N[pi];
  const int tx = threadIdx.x % PXR_RED_CX, ty = threadIdx.x / PXR_RED_CX;
  const int col = ((int)blockIdx.x - m.blk_begin[pi]) * PXR_RED_CX + tx;
  const float s = (col < N) ? pxr_strided_column_sum(part, P, N, col, ty) : 0.f;
  const float v = pxr_combine_row_lanes(s, red, tx, ty);
  if (ty == 0 && col < N) {
    if (col < m.split[pi]) m.out_a[pi][col] = v;
    else m.out_b[pi][col - m.split[pi]] = v;
  }
  if (m.bump && blockIdx.x == 0 && threadIdx.x == 0) m.bump[0] += 1;
}

// out_a[i][c] (c < split[i]) / out_b[i][c - split[i]] = sum_p part[i][p][c], c < N[i], for up to 16 independent
// partial buffers in ONE launch (same fixed-order tree as the single form => same bits).
extern "C" int pxr_reduce_partials_multi_f32(int n, const float* const* part, const int* P, const int* N,
                                             float* const* out_a, float* const* out_b, const int* split,
                                             int64_t* bump_counter, void* stream) {
  PXR_REQUIRE(n >= 1 && n <= 16 && part && P && N && out_a && out_b && split, "pxr_reduce_partials_multi_f32: bad args");
  MultiReduce m{};
  m.n = n;
  m.bump = bump_counter;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    PXR_REQUIRE(part[i] && out_a[i] && P[i] > 0 && N[i] > 0, "pxr_reduce_partials_multi_f32: problem %d is bad", i);
    m.part[i] = part[i]; m.out_a[i] = out_a[i]; m.out_b[i] = out_b[i] ? out_b[i] : out_a[i];
    m.P[i] = P[i]; m.N[i] = N[i]; m.split[i] = out_b[i] ? split[i] : N[i];
    m.blk_begin[i] = blocks;
    blocks += (N[i] + PXR_RED_CX - 1) / PXR_RED_CX;
  }
  m.blk_begin[n] = blocks;
  hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, m);
  return pxr_check_launch("pxr_reduce_partials_multi_f32");
}
