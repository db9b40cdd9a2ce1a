// adamw.hip -- fused AdamW (K16): torch.optim.AdamW semantics (decoupled weight decay), called at trainer.py:125
// on the optimizer built at trainer.py:102 (betas (0.9,0.999), eps 1e-8 = torch defaults, lr / weight_decay
// from the YAML `optim_args`).
//
// Per element, mirroring torch's single-tensor update (torch/optim/adamw.py):
//     p *= 1 - lr*wd;  m += (g - m)(1-b1);  v = v*b2 + (1-b2) g^2;
//     p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
//
// Two forms:
//   * pxr_adamw_flat_f32   : all non-table parameters live in ONE flat buffer (17 MB at D=512) -> one launch.
//   * pxr_adamw_table_f32  : the item-embedding table.  weight_decay applies to the table too
//     (overall/ID.yaml:20-23) and rows touched earlier keep moving by their moments, so the update is a DENSE
//     sweep of p, m, v (SURVEY.md §7 hard part 2) -- but the GRADIENT is sparse: the sweep looks up each row's
//     gradient through a row -> slot map (slot = index into the compact uniq_rows of embed_grad.hip, -1 = no
//     gradient this step).  HBM traffic per step = 6 x 4 B x N x D (read+write p, m, v) instead of the
//     reference's 7 x 4 B plus the 2 x dense-gradient zero-fill/scatter; no 819 MB dense gradient exists.
//     The kernel clears the slots it consumes, so the map is all -1 again afterwards.
#include <string.h>

#include "planes.cuh"

#include <cstdlib>

namespace pxr {

struct AdamHyper {
  float decay;        // 1 - lr*wd
  float one_m_b1;     // 1 - beta1
  float b2;           // beta2
  float one_m_b2;     // 1 - beta2
  float step_size;    // lr / (1 - beta1^t)
  float inv_sqrt_bc2; // 1 / sqrt(1 - beta2^t)
  float eps;
};

static AdamHyper make_hyper(double lr, double b1, double b2, double eps, double wd, int64_t step) {
  AdamHyper h;
  h.decay = (float)(1.0 - lr * wd);
  h.one_m_b1 = (float)(1.0 - b1);
  h.b2 = (float)b2;
  h.one_m_b2 = (float)(1.0 - b2);
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  h.step_size = (float)(lr / bc1);
  h.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  h.eps = (float)eps;
  return h;
}

// The operation order is PINNED with explicit fmaf / __fmul_rn so that every kernel that updates a row (dense sweep,
// lazy replay, lazy apply) produces bit-identical results regardless of how the compiler would contract a*b+c.
// sqrt and the division are the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 (2 quarter-rate instructions) instead of the
// correctly rounded sequences (~25 VALU instructions): a lazily updated row REPLAYS this body once per missed step, so
// its cost is what the lazy schedule pays per step (bench.py `roofline_adamw_rows`).  The update term is <= lr in
// magnitude, so a 1-ulp relative difference in it is ~1e-11 absolute against the +-1e-5 parity bar on parameters.
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, const AdamHyper& h) {
  p = __fmul_rn(p, h.decay);
  m = fmaf(g - m, h.one_m_b1, m);
  v = fmaf(v, h.b2, __fmul_rn(__fmul_rn(h.one_m_b2, g), g));
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), h.inv_sqrt_bc2, h.eps);
  p = fmaf(-h.step_size, __fmul_rn(m, __builtin_amdgcn_rcpf(denom)), p);
}
// zero-gradient form of adam_elem (same values bit for bit: fmaf(v, b2, +0) == v * b2, g - m == -m)
__device__ __forceinline__ void adam_elem0(float& p, float& m, float& v, const AdamHyper& h) {
  p = __fmul_rn(p, h.decay);
  m = fmaf(-m, h.one_m_b1, m);
  v = __fmul_rn(v, h.b2);
  const float denom = fmaf(__builtin_amdgcn_sqrtf(v), h.inv_sqrt_bc2, h.eps);
  p = fmaf(-h.step_size, __fmul_rn(m, __builtin_amdgcn_rcpf(denom)), p);
}

__global__ void __launch_bounds__(256) adamw_flat_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                         float4* __restrict__ m, float4* __restrict__ v, int64_t n4,
                                                         AdamHyper h) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = p[i], mm = m[i], vv = v[i];
    const float4 gg = g[i];
    adam_elem(pp.x, mm.x, vv.x, gg.x, h); adam_elem(pp.y, mm.y, vv.y, gg.y, h);
    adam_elem(pp.z, mm.z, vv.z, gg.z, h); adam_elem(pp.w, mm.w, vv.w, gg.w, h);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

// slot[uniq_idx[u]] = u  for u < *n_uniq
__global__ void __launch_bounds__(256) slot_set_kernel(const int64_t* __restrict__ uniq_idx,
                                                       const int* __restrict__ n_uniq, int* __restrict__ slot,
                                                       int64_t n_table) {
  const int nu = *n_uniq;
  for (int u = blockIdx.x * 256 + threadIdx.x; u < nu; u += gridDim.x * 256) {
    const int64_t r = uniq_idx[u];
    if (r > 0 && r < n_table) slot[r] = u;   // id 0 marks an empty slot of a merged (non-compacted) list
  }
}

// One wave sweeps whole rows (grid-stride): coalesced float4 streams of p, m, v; gradient rows come from the
// compact buffer when slot >= 0.  ROWS_PER_ITER rows are in flight per wave to keep enough loads outstanding.
template <int VEC, int NT>
__global__ void __launch_bounds__(256) adamw_table_kernel(float* __restrict__ p, float* __restrict__ m,
                                                          float* __restrict__ v, int64_t n_rows, int D,
                                                          int* __restrict__ slot, const float* __restrict__ grows,
                                                          AdamHyper h) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t row = wave_id; row < n_rows; row += n_waves) {
    const int s = slot[row];
    if (s >= 0 && lane == 0) slot[row] = -1;
    const float* gr = (s >= 0) ? grows + (int64_t)s * D : nullptr;
    float4 pp[VEC], mm[VEC], vv[VEC], gg[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < D) {
        const int64_t o = row * D + c;
        if constexpr (NT & 1) {
          pp[k] = pxr_ld_stream(p + o); mm[k] = pxr_ld_stream(m + o); vv[k] = pxr_ld_stream(v + o);
        } else {
          pp[k] = *reinterpret_cast<const float4*>(p + o);
          mm[k] = *reinterpret_cast<const float4*>(m + o);
          vv[k] = *reinterpret_cast<const float4*>(v + o);
        }
        gg[k] = gr ? *reinterpret_cast<const float4*>(gr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int c = (k * 64 + lane) * 4;
      if (c < D) {
        const int64_t o = row * D + c;
        adam_elem(pp[k].x, mm[k].x, vv[k].x, gg[k].x, h); adam_elem(pp[k].y, mm[k].y, vv[k].y, gg[k].y, h);
        adam_elem(pp[k].z, mm[k].z, vv[k].z, gg[k].z, h); adam_elem(pp[k].w, mm[k].w, vv[k].w, gg[k].w, h);
        if constexpr (NT & 2) {
          pxr_st_stream(p + o, pp[k]); pxr_st_stream(m + o, mm[k]); pxr_st_stream(v + o, vv[k]);
        } else {
          *reinterpret_cast<float4*>(p + o) = pp[k];
          *reinterpret_cast<float4*>(m + o) = mm[k];
          *reinterpret_cast<float4*>(v + o) = vv[k];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ lazy table update
// Dense AdamW semantics WITHOUT the dense sweep.  A row whose gradient is zero at step s evolves by a closed
// recurrence that depends only on its own (p, m, v) and the step's scalars, so it can be replayed later: every row
// carries last[r] = the step through which it is up to date; before a row is read (forward) or updated (apply) the
// missed zero-gradient steps last[r]+1 .. t are replayed with the SAME per-element arithmetic as the dense kernel
// (adam_elem with g = 0  =>  bit-identical to the sweep for gaps <= PXR_LAZY_EXACT).  Beyond PXR_LAZY_EXACT replayed
// steps the Adam term is provably below fp32 resolution (|m|/sqrt(v) <= 31.6 * (beta1/sqrt(beta2))^j, so the skipped
// updates sum to < 1e-13 for j > 256 at lr 1e-4) and only the weight decay acts: p *= prod(decay_s) in closed form
// from a double-precision cumulative-log table, m *= beta1^j, v *= beta2^j.
// hyper[s] = {decay_s, step_size_s, inv_sqrt_bc2_s, 0} and cumlog[s] = sum_{i<=s} log(decay_i) are appended by
// pxr_adamw_hyper_append at every optimizer step (index 0 = identity), so a changing lr is replayed correctly.
constexpr int PXR_LAZY_EXACT = 256;

struct RowsArgs {
  float* p; float* m; float* v; int* last;
  const int64_t* rows; const int* n_rows; int64_t n_fixed;   // rows == null: all rows [0, n_fixed)
  const float* grows;                                        // [n_rows, D] gradient rows (apply) or null
  const float4* hyper; const double* cumlog;
  int t_prev, t_apply, D;
  float one_m_b1, b2, one_m_b2, eps, b1;
  const int64_t* step_dev;   // optional: t_prev = *step_dev (completed steps), t_apply = t_prev+1 if t_apply != 0
  int t_prev_bias;           // added to *step_dev (1 = "through the step being applied right now": prefetched rows)
  // claim mode (catch-up only, t_apply == 0): `rows` is a RAW id list of n_list entries -- duplicates, id 0 and out-of-range
  // ids allowed -- e.g. the batch's item tensor as it is.  The group that raises last[row] (atomicMax) replays the row; the
  // groups of its duplicates see it current and leave.  No sorted unique list is needed before the forward pass.
  int claim; int64_t n_list; int64_t n_table;
  int row_len, row_stride;   // claim mode, 2-D list: entry i is rows[(i / row_len) * row_stride + i % row_len] (0: a flat list)
  float4* cur_out;           // optional: one thread copies the scalars of the step ABOUT to run (entry t_prev + 1) here -- the flat
                             // update at the end of the step reads them from this slot and may then close the step itself
  // fast replay (default; PXR_LAZY_REPLAY=exact selects the bit-identical one): sqrt(v) and 1/denominator are carried from
  // step to step (see adamw_rows_kernel) -- 8 VALU issue slots per element-step instead of 14
  int fast; float sqrt_b2, log2_b2;
  int window;                // replayed steps per row before the closed form takes over (PXR_LAZY_EXACT; PXR_LAZY_WINDOW: a measuring knob)
  // series replay (fast mode, round 6; see adamw_rows_kernel): a gap of >= series_min steps is summed in closed form -- the lanes of a
  // wave share out the STEPS of the gap (six moments of the per-step scalars, wave-reduced), every element then takes one polynomial
  int series, series_min; float log2_sb;
  double log2_b1;            // log2 of the factor the exact step multiplies m by: 1 - float(1 - beta1), not float(beta1)
};

// One row is spread over LPR = ceil(D/EPL / 64) * 64 lanes (EPL = 2 or 4 elements per lane), so a row with a long gap is
// a chain of EPL elements per lane instead of D/64: the kernel's duration is set by the longest replays in the batch (a
// Zipf-tail or uniformly drawn negative id returns after hundreds of steps; measured on the aged bench stream: mean 80
// replayed steps per row, 10 % of the rows at the 256-step cap), not by the average.  All lanes of a row sit in ONE
// workgroup (block = max(256, LPR) threads = RPB rows): last[row] is read by every wave of the row at the top and
// written by one lane after a barrier.  The replayed step number is wave-uniform (readfirstlane), so the per-step
// scalars come through the scalar cache, four steps per request.
// EPL consecutive floats of a row (EPL = 2: one 8-byte access, 4 / 8: one / two 16-byte accesses; the offsets are multiples of EPL)
template <int EPL>
__device__ __forceinline__ void row_ld(const float* __restrict__ src, float (&d)[EPL]) {
  if constexpr (EPL == 2) {
    const float2 t = *reinterpret_cast<const float2*>(src);
    d[0] = t.x; d[1] = t.y;
  } else {
#pragma unroll
    for (int q = 0; q < EPL / 4; ++q) {
      const float4 t = reinterpret_cast<const float4*>(src)[q];
      d[4 * q] = t.x; d[4 * q + 1] = t.y; d[4 * q + 2] = t.z; d[4 * q + 3] = t.w;
    }
  }
}
template <int EPL>
__device__ __forceinline__ void row_st(float* __restrict__ dst, const float (&d)[EPL]) {
  if constexpr (EPL == 2) {
    *reinterpret_cast<float2*>(dst) = make_float2(d[0], d[1]);
  } else {
#pragma unroll
    for (int q = 0; q < EPL / 4; ++q) reinterpret_cast<float4*>(dst)[q] = make_float4(d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]);
  }
}

template <int LPR, int EPL>
__global__ void __launch_bounds__((LPR > 256 ? LPR : 256)) adamw_rows_kernel(RowsArgs a,
                                                                               const float4* __restrict__ hyper) {
  constexpr int NT = LPR > 256 ? LPR : 256;
  constexpr int RPB = NT / LPR;
  if (a.step_dev) {
    a.t_prev = (int)a.step_dev[0] + a.t_prev_bias;
    if (a.t_apply) a.t_apply = a.t_prev + 1;
  }
  if (a.cur_out && blockIdx.x == 0 && threadIdx.x == 0) a.cur_out[0] = hyper[a.t_prev + 1];
  const int rib = threadIdx.x / LPR;            // row of this block's group
  const int c = (threadIdx.x % LPR) * EPL;      // first column of this lane's elements
  const bool col_ok = c < a.D;
  const int64_t n = a.rows ? (a.claim ? a.n_list : (int64_t)(*a.n_rows)) : a.n_fixed;
  const int64_t n_groups = (n + RPB - 1) / RPB;
  __shared__ int s_claim[RPB];
  for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t i = grp * RPB + rib;
    int64_t row = 0;
    int k0 = 0;
    bool work = i < n;
    if (work) {
      row = a.rows ? a.rows[a.row_len ? (i / a.row_len) * a.row_stride + i % a.row_len : i] : i;
      if (a.rows && row <= 0) work = false;   // id 0 = empty slot of a merged (non-compacted) row list
    }
    // p / m / v are requested together with last[row] (both need only the row id): one memory round trip less on the
    // dependent chain  row id -> last -> values  that every group walks before it can compute (rows that turn out to be
    // current -- few -- have read 6 KB for nothing)
    const bool in_table = work && (!a.claim || row < a.n_table);
    const int64_t o = row * a.D + c;
    float pe[EPL], me[EPL], ve[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) pe[e] = me[e] = ve[e] = 0.f;
    if (in_table && col_ok) {
      row_ld<EPL>(a.p + o, pe); row_ld<EPL>(a.m + o, me); row_ld<EPL>(a.v + o, ve);
    }
    if (a.claim) {
      if (work && row >= a.n_table) work = false;
      if ((threadIdx.x % LPR) == 0) s_claim[rib] = work ? atomicMax(a.last + row, a.t_prev) : a.t_prev;
      __syncthreads();
      k0 = s_claim[rib];
      if (k0 >= a.t_prev) work = false;
      __syncthreads();                        // s_claim is rewritten by the next group
    } else if (work) {
      k0 = a.last[row];
      if (k0 >= a.t_prev && a.t_apply == 0) work = false;
    }
    // wave-uniform by construction (a wave never straddles two rows: LPR is a multiple of 64)
    work = __builtin_amdgcn_readfirstlane((int)work) != 0;
    k0 = __builtin_amdgcn_readfirstlane(k0);
    if (work) {
      AdamHyper h;
      h.one_m_b1 = a.one_m_b1; h.b2 = a.b2; h.one_m_b2 = a.one_m_b2; h.eps = a.eps;
      int s = k0 + 1;
      const int exact_end = min(a.t_prev, k0 + a.window);
      auto replay = [&](const float4 hs) {
        h.decay = hs.x; h.step_size = hs.y; h.inv_sqrt_bc2 = hs.z;
#pragma unroll
        for (int e = 0; e < EPL; ++e) adam_elem0(pe[e], me[e], ve[e], h);
      };
      // (the series costs the same for any gap: it covers PXR_LAZY_EXACT steps, the exact mode's window, where the loop stops at 128)
      const int series_end = min(a.t_prev, k0 + PXR_LAZY_EXACT);
      if (a.series && series_end - s + 1 >= a.series_min) {
        // SERIES replay.  With zero gradient, step j = 1..n of the gap (table entry s + j - 1) adds
        //     - ss_j b1^j m_0 / (sqrt(v_0) c_j + eps),   c_j = sqrt(b2)^j / sqrt(bc2_j),
        // to p, later shrunk by the weight decay of the steps behind it (P_j = prod_{i > j} decay_i).  The denominator moves slowly:
        // D_j = D_1 (1 + w sig_j) with D_1 = sqrt(v_0) c_1 + eps, w = sqrt(v_0) c_1 / D_1 in [0, 1) (the ELEMENT's share) and
        // sig_j = c_j / c_1 - 1 (the STEP's share: 0.05 % per step from sqrt(b2), plus what the bias correction still moves).  So
        //     p_n = p_0 P_0 - (m_0 / D_1) sum_q (-w)^q T_q,      T_q = sum_j ss_j b1^j P_j sig_j^q,
        // and the T_q depend on the row's gap only: lane l of every wave of the row computes the terms of steps l + 1, l + 65, ... and
        // the wave sums them (DPP) -- ~150 VALU instructions per row instead of 8 per element-STEP (a mean of 80 steps per row on the
        // aged bench stream: this loop was VALU-issue bound).  Truncated after q = 5; the remainder is BOUNDED per row
        // (sum_j |a_j| |sig_j|^6 / (1 - |sig_j|), |w| < 1) and a row whose bound exceeds 2e-7 of T_0 -- the first ~200 optimizer
        // steps, where the bias correction moves the denominator by percents per step -- takes the step-by-step loop below.
        // Differences to the dense sweep: as the carried-product loop's (< 1e-6 relative on the sum of update terms; tests:
        // tests/test_gpu_lazy_adamw.py fast_replay / series tests).
        const int n = series_end - s + 1;
        const int ln = threadIdx.x & 63;
        const double cl_end = a.cumlog[series_end];
        const float c1 = __builtin_amdgcn_exp2f(a.log2_sb) * hyper[s].z;
        const float rc1 = __builtin_amdgcn_rcpf(c1);
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f, t4 = 0.f, t5 = 0.f, rr = 0.f;
        // b1^j with the exponent formed in double (its rounding in fp32 is a relative 1e-8 j, the same sign in every catch-up)
        auto pow_b1 = [&](const int j) {
          const double ad = (double)j * a.log2_b1;
          const float ah = (float)ad, al = (float)(ad - (double)ah);
          return __builtin_amdgcn_exp2f(ah) * fmaf(al, 0.69314718f, 1.0f);
        };
        for (int j0 = 0; j0 < n; j0 += 64) {
          const int j = j0 + ln + 1;
          if (j <= n) {
            const float4 hs = hyper[s + j - 1];
            const float pj = __expf((float)(cl_end - a.cumlog[s + j - 1]));
            const float fj = (float)j;
            const float aj = hs.y * pow_b1(j) * pj;
            const float sg = fmaf(__builtin_amdgcn_exp2f(fj * a.log2_sb) * hs.z, rc1, -1.0f);
            const float s2 = sg * sg, s3 = s2 * sg, s4 = s2 * s2, s5 = s4 * sg;
            t0 += aj; t1 = fmaf(aj, sg, t1); t2 = fmaf(aj, s2, t2); t3 = fmaf(aj, s3, t3); t4 = fmaf(aj, s4, t4); t5 = fmaf(aj, s5, t5);
            rr = fmaf(fabsf(aj), s4 * s2 * __builtin_amdgcn_rcpf(fmaxf(1.0f - fabsf(sg), 1e-3f)), rr);
          }
        }
        t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2); t3 = wave_sum(t3); t4 = wave_sum(t4); t5 = wave_sum(t5);
        rr = wave_sum(rr);
        if (rr <= 2e-7f * fabsf(t0)) {                 // wave-uniform (wave_sum hands every lane lane 63's total)
          const float p0f = __expf((float)(cl_end - a.cumlog[s - 1]));
          const float fn = (float)n;
          const float fm = pow_b1(n), fv = __builtin_amdgcn_exp2f(fn * a.log2_b2);
#pragma unroll
          for (int e = 0; e < EPL; ++e) {
            const float sq = __builtin_amdgcn_sqrtf(ve[e]);
            const float d1 = fmaf(sq, c1, h.eps);
            float r1 = __builtin_amdgcn_rcpf(d1);
            r1 = r1 * fmaf(-d1, r1, 2.0f);
            const float w = sq * c1 * r1;
            const float poly = fmaf(-w, fmaf(-w, fmaf(-w, fmaf(-w, fmaf(-w, t5, t4), t3), t2), t1), t0);
            pe[e] = fmaf(-(me[e] * r1), poly, pe[e] * p0f);
            me[e] *= fm;
            ve[e] = __fmul_rn(ve[e], fv);
          }
          s = series_end + 1;
        }
      }
      if (a.fast && s <= exact_end) {
        // The zero-gradient recurrence without its two quarter-rate instructions per element-step: sqrt(v_t) =
        // sqrt(v_0) sqrt(b2)^t is carried as a product, and 1/denom_t comes from 1/denom_{t-1} by Newton steps (the
        // denominator moves by 0.05 % per step through sqrt(b2), plus what the settling bias correction adds: 1 % at optimizer
        // step 64, 0.03 % at step 1000).  Differences to the dense sweep: < 1e-6 relative on update terms of at most a few lr
        // each, i.e. ~1e-9 absolute on p over a 256-step gap (measured: tests/test_gpu_lazy_adamw.py, fast_replay tests);
        // m exact to rounding, v = v_0 b2^gap in one rounding.
        // Two elements per instruction: the recurrence is written on float2 values so that it compiles to the packed fp32
        // instructions of the CDNA VALU (v_pk_mul_f32 / v_pk_fma_f32) -- this loop is VALU-issue bound (4464 rows x 80
        // replayed steps x 512 elements per training step), so that halves its time.
        constexpr int NP = EPL / 2;
        p3_f32x2 pv[NP], mv[NP], sv[NP], rv[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          pv[q] = p3_f32x2{pe[2 * q], pe[2 * q + 1]};
          mv[q] = p3_f32x2{me[2 * q], me[2 * q + 1]};
          sv[q] = p3_f32x2{__builtin_amdgcn_sqrtf(ve[2 * q]), __builtin_amdgcn_sqrtf(ve[2 * q + 1])};
          rv[q] = p3_f32x2{0.f, 0.f};
        }
        const int s0 = s;
        const int rcp_until = max(s0, 64);
        const p3_f32x2 b1v = {a.b1, a.b1}, sbv = {a.sqrt_b2, a.sqrt_b2}, epsv = {h.eps, h.eps}, two = {2.0f, 2.0f};
        // MODE 0: hardware reciprocal; 1: one Newton step; 2: two (optimizer steps < 1024, where the bias correction still
        // moves the denominator by up to 1 % per step: one step would leave 6e-5 relative there, two leave it at rounding;
        // switching to one step at 256 already was measured: no faster, and outside the 1e-7 the tests hold the replay to)
        auto step_fast = [&](const float4 hs, const int mode) {
          const p3_f32x2 dec = {hs.x, hs.x}, nss = {-hs.y, -hs.y}, isb = {hs.z, hs.z};
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            pv[q] = pv[q] * dec;
            mv[q] = mv[q] * b1v;
            sv[q] = sv[q] * sbv;
            const p3_f32x2 denom = __builtin_elementwise_fma(sv[q], isb, epsv);
            if (mode == 0) {
              rv[q] = p3_f32x2{__builtin_amdgcn_rcpf(denom[0]), __builtin_amdgcn_rcpf(denom[1])};
            } else {
              rv[q] = rv[q] * __builtin_elementwise_fma(-denom, rv[q], two);
              if (mode == 2) rv[q] = rv[q] * __builtin_elementwise_fma(-denom, rv[q], two);
            }
            pv[q] = __builtin_elementwise_fma(nss, mv[q] * rv[q], pv[q]);
          }
        };
        for (; s <= exact_end && s <= rcp_until; ++s) step_fast(hyper[s], 0);
        const int early_end = min(exact_end, 1023);
        for (; s + 3 <= early_end; s += 4) {
          const float4 h0 = hyper[s], h1 = hyper[s + 1], h2 = hyper[s + 2], h3 = hyper[s + 3];
          step_fast(h0, 2); step_fast(h1, 2); step_fast(h2, 2); step_fast(h3, 2);
        }
        for (; s <= early_end; ++s) step_fast(hyper[s], 2);
        for (; s + 3 <= exact_end; s += 4) {
          const float4 h0 = hyper[s], h1 = hyper[s + 1], h2 = hyper[s + 2], h3 = hyper[s + 3];
          step_fast(h0, 1); step_fast(h1, 1); step_fast(h2, 1); step_fast(h3, 1);
        }
        for (; s <= exact_end; ++s) step_fast(hyper[s], 1);
        const float fv = __builtin_amdgcn_exp2f((float)(s - s0) * a.log2_b2);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          pe[2 * q] = pv[q][0]; pe[2 * q + 1] = pv[q][1];
          me[2 * q] = mv[q][0]; me[2 * q + 1] = mv[q][1];
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) ve[e] = __fmul_rn(ve[e], fv);
      }
      // four steps per trip: their scalars are ONE scalar-cache line (64 B), waited for once
      for (; s + 3 <= exact_end; s += 4) {
        const float4 h0 = hyper[s], h1 = hyper[s + 1], h2 = hyper[s + 2], h3 = hyper[s + 3];
        replay(h0); replay(h1); replay(h2); replay(h3);
      }
      for (; s <= exact_end; ++s) replay(hyper[s]);
      if (s <= a.t_prev) {  // closed-form tail: only the weight decay still moves p
        const int rem = a.t_prev - s + 1;
        const float fp = (float)exp(a.cumlog[a.t_prev] - a.cumlog[s - 1]);
        const float fm = (float)pow((double)a.b1, (double)rem);
        const float fv = (float)pow((double)a.b2, (double)rem);
#pragma unroll
        for (int e = 0; e < EPL; ++e) { pe[e] *= fp; me[e] *= fm; ve[e] *= fv; }
      }
      if (a.t_apply) {
        const float4 hs = hyper[a.t_apply];
        h.decay = hs.x; h.step_size = hs.y; h.inv_sqrt_bc2 = hs.z;
        float ge[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) ge[e] = 0.f;
        if (a.grows && col_ok) row_ld<EPL>(a.grows + i * a.D + c, ge);
#pragma unroll
        for (int e = 0; e < EPL; ++e) adam_elem(pe[e], me[e], ve[e], ge[e], h);
      }
      if (col_ok) {
        row_st<EPL>(a.p + o, pe); row_st<EPL>(a.m + o, me); row_st<EPL>(a.v + o, ve);
      }
    }
    if constexpr (LPR > 64) __syncthreads();   // every wave of the row has read last[row]
    if (work && !a.claim && (threadIdx.x % LPR) == 0) a.last[row] = a.t_apply ? a.t_apply : a.t_prev;
  }
}

// One thread computes the scalars of optimizer step `step` in double precision (same formulas as make_hyper) and
// appends them.  step_dev != null: step = *step_dev + 1 (device counter => valid under hipGraph replay).
// advance != 0 (needs step_dev): FIRST count the step that just finished (*step_dev += 1), then append the entry of the
// next one -- the end-of-step form: one launch closes step t and prepares step t+1.
__device__ __forceinline__ void hyper_append_body(float4* hyper, double* cumlog, int64_t capacity, int64_t step,
                                                  int64_t* step_dev, double lr, double b1, double b2, double wd, int advance) {
  if (step_dev && advance) step_dev[0] += 1;
  if (step_dev) step = step_dev[0] + 1;
  if (step < 1 || step >= capacity) return;
  if (step == 1) { hyper[0] = make_float4(1.f, 0.f, 1.f, 0.f); cumlog[0] = 0.0; }
  const float decay = (float)(1.0 - lr * wd);
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  hyper[step] = make_float4(decay, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), 0.f);
  cumlog[step] = cumlog[step - 1] + log((double)decay);
}

__global__ void hyper_append_kernel(float4* hyper, double* cumlog, int64_t capacity, int64_t step,
                                    int64_t* step_dev, double lr, double b1, double b2, double wd, int advance) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  hyper_append_body(hyper, cumlog, capacity, step, step_dev, lr, b1, b2, wd, advance);
}

// Weight matrices inside the flat buffer whose updated values are ALSO written as bf16x3 planes (the operand format of the
// planes GEMMs, planes.cuh): the next forward then needs no split launch.  Segment i covers flat elements
// [off, off + rows * cols) as a row-major [rows][cols] matrix; offsets are multiples of 4 and cols of 32, so a thread's
// float4 never straddles a row or a 8-column chunk half.
struct FlatPlanes {
  int64_t off[16];
  int rows[16], cols[16];
  P3Mat out[16];
  int n;
  int fmt;                 // PXR_PLANES_BF16X3 | PXR_PLANES_H2 (all segments)
  const int* exps;         // h2: segment i is written as w 2^exps[i] (device ints: the exponents the segment's planes were last
                           // split with -- the weights move by ~lr per step, the exponent leaves 2 binades of headroom and is
                           // re-derived from the values whenever anything but the optimizer rewrites them; a weight that outgrows
                           // the range in between raises PXR_STATUS_H2_RANGE)
  int32_t* status;
};
__device__ __forceinline__ void flat_planes_store(const FlatPlanes& fp, int64_t e, const float4& pp) {
#pragma unroll 1
  for (int i = 0; i < fp.n; ++i) {
    const int64_t r = e - fp.off[i];
    if (r >= 0 && r < (int64_t)fp.rows[i] * fp.cols[i]) {
      const int row = (int)(r / fp.cols[i]);
      const int col = (int)(r - (int64_t)row * fp.cols[i]);
      if (fp.fmt == PXR_PLANES_H2) {
        const float sc = ldexpf(1.0f, fp.exps[i]);
        px_store4(fp.out[i], PXR_PLANES_H2, fp.status, row, col, make_float4(pp.x * sc, pp.y * sc, pp.z * sc, pp.w * sc));
      } else {
        p3_store4(fp.out[i], row, col, pp);
      }
      return;
    }
  }
}

// flat AdamW reading the step's scalars from the hyper table (graph-replayable form of adamw_flat_kernel)
// `close` (optional): the launch also CLOSES the optimizer step -- what pxr_adamw_hyper_append(advance = 1) does in a launch of
// its own: count the step on the device, append the next step's scalars.  Only legal when this step's scalars come from
// `close.cur` (a slot an EARLIER kernel of the step filled, RowsArgs::cur_out): no workgroup of this launch then reads the
// counter the closing thread advances, however late it starts.
struct FlatClose {
  const float4* cur;       // this step's scalars (null: read hyper[step] as before, no closing)
  float4* hyper; double* cumlog; int64_t capacity; int64_t* step_dev;
  double lr, b1, b2, wd;
};
template <bool PLANES>
__global__ void __launch_bounds__(256) adamw_flat_tab_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                             float4* __restrict__ m, float4* __restrict__ v,
                                                             int64_t n4, const float4* __restrict__ hyper, int64_t step,
                                                             const int64_t* step_dev, float one_m_b1, float b2,
                                                             float one_m_b2, float eps, const FlatPlanes fp, const FlatClose cl) {
  if (step_dev && !cl.cur) step = step_dev[0] + 1;
  const float4 hs = cl.cur ? cl.cur[0] : hyper[step];
  if (cl.cur && blockIdx.x == 0 && threadIdx.x == 0)
    hyper_append_body(cl.hyper, cl.cumlog, cl.capacity, 0, cl.step_dev, cl.lr, cl.b1, cl.b2, cl.wd, 1);
  AdamHyper h;
  h.decay = hs.x; h.step_size = hs.y; h.inv_sqrt_bc2 = hs.z;
  h.one_m_b1 = one_m_b1; h.b2 = b2; h.one_m_b2 = one_m_b2; h.eps = eps;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = p[i], mm = m[i], vv = v[i];
    const float4 gg = g[i];
    adam_elem(pp.x, mm.x, vv.x, gg.x, h); adam_elem(pp.y, mm.y, vv.y, gg.y, h);
    adam_elem(pp.z, mm.z, vv.z, gg.z, h); adam_elem(pp.w, mm.w, vv.w, gg.w, h);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if constexpr (PLANES) flat_planes_store(fp, i * 4, pp);
  }
}

__global__ void counter_add_kernel(int64_t* c, int64_t delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += delta;
}

__global__ void __launch_bounds__(256) fill_i32_kernel(int* __restrict__ x, int64_t n, int val) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] = val;
}

}  // namespace pxr

using namespace pxr;

extern "C" int pxr_adamw_flat_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                                  double beta2, double eps, double weight_decay, int64_t step, void* stream) {
  PXR_REQUIRE(p && g && m && v, "pxr_adamw_flat_f32: null pointer");
  PXR_REQUIRE(n >= 0 && n % 4 == 0 && step >= 1, "pxr_adamw_flat_f32: n must be a multiple of 4 and step >= 1");
  if (n == 0) return PXR_OK;
  const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, step);
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)p,
                     (const float4*)g, (float4*)m, (float4*)v, n4, h);
  return pxr_check_launch("pxr_adamw_flat_f32");
}

// slot map helpers: the map is an int32 [N] array owned by the caller, all -1 between steps.
extern "C" int pxr_slot_fill_i32(int32_t* slot, int64_t n, int32_t value, void* stream) {
  PXR_REQUIRE(slot && n > 0, "pxr_slot_fill_i32: bad args");
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, slot, n, value);
  return pxr_check_launch("pxr_slot_fill_i32");
}

// Dense-semantics AdamW over the table with the step's gradient given sparsely as (uniq_idx, uniq_rows, *n_uniq).
// max_uniq bounds the launch of the slot-set pass (the true count is read on the device).
extern "C" int pxr_adamw_table_f32(float* table, float* m, float* v, int64_t n_rows, int D, int32_t* slot,
                                   const int64_t* uniq_idx, const float* uniq_rows, const int32_t* n_uniq_dev,
                                   int64_t max_uniq, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, int64_t step, void* stream) {
  PXR_REQUIRE(table && m && v && slot, "pxr_adamw_table_f32: null pointer");
  PXR_REQUIRE(n_rows > 0 && D > 0 && D % 4 == 0 && D <= 4096 && step >= 1, "pxr_adamw_table_f32: bad shape");
  hipStream_t st = (hipStream_t)stream;
  if (uniq_idx && max_uniq > 0) {
    PXR_REQUIRE(uniq_rows && n_uniq_dev, "pxr_adamw_table_f32: sparse gradient needs rows and count");
    int64_t blocks = (max_uniq + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(slot_set_kernel, dim3((unsigned)blocks), dim3(256), 0, st, uniq_idx, n_uniq_dev, slot, n_rows);
  }
  const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, step);
  int64_t blocks = (n_rows + 3) / 4;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const int vec = (D + 255) / 256;
  static const int nt = getenv("PXR_ADAMW_NT") ? atoi(getenv("PXR_ADAMW_NT")) : 3;   // A/B knob: 0 none, 1 loads, 2 stores, 3 both
#define PXR_TABLE_LAUNCH(V, T) \
  hipLaunchKernelGGL((adamw_table_kernel<V, T>), dim3((unsigned)blocks), dim3(256), 0, st, table, m, v, n_rows, D, slot, uniq_rows, h)
#define PXR_TABLE_CASE(V)                                  \
  do {                                                     \
    if (nt == 0) PXR_TABLE_LAUNCH(V, 0);                   \
    else if (nt == 1) PXR_TABLE_LAUNCH(V, 1);              \
    else if (nt == 2) PXR_TABLE_LAUNCH(V, 2);              \
    else PXR_TABLE_LAUNCH(V, 3);                           \
  } while (0)
  if (vec <= 1) PXR_TABLE_CASE(1);
  else if (vec <= 2) PXR_TABLE_CASE(2);
  else if (vec <= 4) PXR_TABLE_CASE(4);
  else if (vec <= 8) PXR_TABLE_CASE(8);
  else PXR_TABLE_CASE(16);
#undef PXR_TABLE_CASE
#undef PXR_TABLE_LAUNCH
  return pxr_check_launch("pxr_adamw_table_f32");
}

// ---- lazy (exact catch-up) table AdamW ----------------------------------------------------------------------
// hyper: float4[capacity], cumlog: double[capacity]; appends the scalars of optimizer step `step` (1-based), or of
// step *step_dev + 1 when a device counter is given (hipGraph-replayable).
extern "C" int pxr_adamw_hyper_append(void* hyper, void* cumlog, int64_t capacity, int64_t step, int64_t* step_dev,
                                      double lr, double beta1, double beta2, double eps, double weight_decay,
                                      int advance, void* stream) {
  (void)eps;
  PXR_REQUIRE(hyper && cumlog && (step_dev || (step >= 1 && step < capacity)),
              "pxr_adamw_hyper_append: bad args (step %lld, capacity %lld)", (long long)step, (long long)capacity);
  PXR_REQUIRE(!advance || step_dev, "pxr_adamw_hyper_append: advance needs the device step counter");
  hipLaunchKernelGGL(hyper_append_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (float4*)hyper, (double*)cumlog,
                     capacity, step, step_dev, lr, beta1, beta2, weight_decay, advance);
  return pxr_check_launch("pxr_adamw_hyper_append");
}

// Flat AdamW with the step's scalars taken from the hyper table entry `step` (or *step_dev + 1).
extern "C" int pxr_adamw_flat_tab_planes_f32(float* p, const float* g, float* m, float* v, int64_t n, const void* hyper,
                                             int64_t step, const int64_t* step_dev, double beta1, double beta2, double eps,
                                             int n_seg, const int64_t* seg_off, const int64_t* seg_rows, const int64_t* seg_cols,
                                             void* const* seg_planes, const int64_t* seg_plane_stride,
                                             const int64_t* seg_panel_rows, void* stream);
extern "C" int pxr_adamw_flat_tab_f32(float* p, const float* g, float* m, float* v, int64_t n, const void* hyper,
                                      int64_t step, const int64_t* step_dev, double beta1, double beta2, double eps,
                                      void* stream) {
  return pxr_adamw_flat_tab_planes_f32(p, g, m, v, n, hyper, step, step_dev, beta1, beta2, eps, 0, nullptr, nullptr, nullptr,
                                       nullptr, nullptr, nullptr, stream);
}
// the same; the updated values of n_seg (<= 16) weight matrices inside the flat buffer ([seg_rows, seg_cols] row-major at
// element seg_off) are additionally written as bf16x3 planes (pxr.h: planes) -- the operands of the next step's GEMMs
static int flat_tab_impl(float* p, const float* g, float* m, float* v, int64_t n, const void* hyper, int64_t step,
                         const int64_t* step_dev, double beta1, double beta2, double eps, int n_seg, const int64_t* seg_off,
                         const int64_t* seg_rows, const int64_t* seg_cols, void* const* seg_planes, const int64_t* seg_plane_stride,
                         const int64_t* seg_panel_rows, const FlatClose& cl, void* stream, int seg_fmt = 0, const int* seg_exps = nullptr);
extern "C" int pxr_adamw_flat_tab_planes_f32(float* p, const float* g, float* m, float* v, int64_t n, const void* hyper,
                                             int64_t step, const int64_t* step_dev, double beta1, double beta2, double eps,
                                             int n_seg, const int64_t* seg_off, const int64_t* seg_rows, const int64_t* seg_cols,
                                             void* const* seg_planes, const int64_t* seg_plane_stride,
                                             const int64_t* seg_panel_rows, void* stream) {
  return flat_tab_impl(p, g, m, v, n, hyper, step, step_dev, beta1, beta2, eps, n_seg, seg_off, seg_rows, seg_cols, seg_planes,
                       seg_plane_stride, seg_panel_rows, FlatClose{}, stream);
}
// pxr_adamw_flat_tab_planes_f32 with two options.  (a) seg_fmt = PXR_PLANES_H2: the weight segments are written as fp16 two-plane
// operands, segment i scaled by 2^seg_exps[i] (device ints, see FlatPlanes).  (b) cur_hyper != NULL: the launch also CLOSES the
// optimizer step (what pxr_adamw_hyper_append(advance = 1) does in a launch of its own): this step's scalars are read from
// `cur_hyper` (float4, filled at the head of the step by pxr_adamw_rows_ids2d_f32's cur_hyper_out), one thread counts the step in
// *step_dev and appends the next step's entry to hyper / cumlog.  cur_hyper == NULL: scalars from hyper[step | *step_dev + 1].
extern "C" int pxr_adamw_flat_tab_ex_f32(float* p, const float* g, float* m, float* v, int64_t n, void* hyper, void* cumlog,
                                         int64_t capacity, int64_t step, int64_t* step_dev, const void* cur_hyper, double lr,
                                         double beta1, double beta2, double eps, double weight_decay, int n_seg,
                                         const int64_t* seg_off, const int64_t* seg_rows, const int64_t* seg_cols,
                                         void* const* seg_planes, const int64_t* seg_plane_stride, const int64_t* seg_panel_rows,
                                         int seg_fmt, const int* seg_exps, void* stream) {
  PXR_REQUIRE(n > 0, "pxr_adamw_flat_tab_ex_f32: empty buffer");
  FlatClose cl{};
  if (cur_hyper) {
    PXR_REQUIRE(cumlog && step_dev && capacity > 2, "pxr_adamw_flat_tab_ex_f32: closing the step needs cumlog, the device counter and the table's capacity");
    cl.cur = (const float4*)cur_hyper; cl.hyper = (float4*)hyper; cl.cumlog = (double*)cumlog; cl.capacity = capacity;
    cl.step_dev = step_dev; cl.lr = lr; cl.b1 = beta1; cl.b2 = beta2; cl.wd = weight_decay;
  }
  return flat_tab_impl(p, g, m, v, n, hyper, step, step_dev, beta1, beta2, eps, n_seg, seg_off, seg_rows, seg_cols, seg_planes,
                       seg_plane_stride, seg_panel_rows, cl, stream, seg_fmt, seg_exps);
}
static int flat_tab_impl(float* p, const float* g, float* m, float* v, int64_t n, const void* hyper, int64_t step,
                         const int64_t* step_dev, double beta1, double beta2, double eps, int n_seg, const int64_t* seg_off,
                         const int64_t* seg_rows, const int64_t* seg_cols, void* const* seg_planes, const int64_t* seg_plane_stride,
                         const int64_t* seg_panel_rows, const FlatClose& cl, void* stream, int seg_fmt, const int* seg_exps) {
  PXR_REQUIRE(p && g && m && v && hyper, "pxr_adamw_flat_tab_f32: null pointer");
  PXR_REQUIRE(seg_fmt == PXR_PLANES_BF16X3 || (seg_fmt == PXR_PLANES_H2 && (seg_exps || n_seg == 0)),
              "pxr_adamw_flat_tab_f32: h2 segments need their device exponents");
  PXR_REQUIRE(n >= 0 && n % 4 == 0 && (step_dev || step >= 1), "pxr_adamw_flat_tab_f32: bad n / step");
  PXR_REQUIRE(n_seg >= 0 && n_seg <= 16 && (n_seg == 0 || (seg_off && seg_rows && seg_cols && seg_planes && seg_plane_stride && seg_panel_rows)),
              "pxr_adamw_flat_tab_planes_f32: bad segment table");
  if (n == 0) return PXR_OK;
  FlatPlanes fp{};
  fp.n = n_seg; fp.fmt = seg_fmt; fp.exps = seg_exps; fp.status = pxr_status_word();
  for (int i = 0; i < n_seg; ++i) {
    PXR_REQUIRE(seg_off[i] >= 0 && seg_off[i] % 4 == 0 && seg_rows[i] > 0 && seg_rows[i] < (1ll << 31) && seg_cols[i] > 0 &&
                    seg_off[i] + seg_rows[i] * seg_cols[i] <= n && seg_planes[i] &&
                    p3_mat_ok(seg_planes[i], seg_plane_stride[i], seg_panel_rows[i], seg_rows[i], seg_cols[i]),
                "pxr_adamw_flat_tab_planes_f32: segment %d is bad", i);
    fp.off[i] = seg_off[i]; fp.rows[i] = (int)seg_rows[i]; fp.cols[i] = (int)seg_cols[i];
    fp.out[i] = P3Mat{reinterpret_cast<__bf16*>(seg_planes[i]), seg_plane_stride[i], seg_panel_rows[i]};
  }
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (n_seg > 0)
    hipLaunchKernelGGL(adamw_flat_tab_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)p,
                       (const float4*)g, (float4*)m, (float4*)v, n4, (const float4*)hyper, step, step_dev,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, fp, cl);
  else
    hipLaunchKernelGGL(adamw_flat_tab_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)p,
                       (const float4*)g, (float4*)m, (float4*)v, n4, (const float4*)hyper, step, step_dev,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, fp, cl);
  return pxr_check_launch("pxr_adamw_flat_tab_f32");
}

// *counter += delta on the device (step counters that hipGraph replays must advance without the host).
extern "C" int pxr_counter_add_i64(int64_t* counter, int64_t delta, void* stream) {
  PXR_REQUIRE(counter, "pxr_counter_add_i64: null pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, delta);
  return pxr_check_launch("pxr_counter_add_i64");
}

// Brings rows up to date through step t_prev (replaying their missed zero-gradient steps) and, if t_apply != 0,
// applies step t_apply with gradient rows grows[i,:] (row i of the list).  rows == NULL: every row of the table
// (flush; n_rows_dev ignored).  last: int32[N] "up to date through" step per row.  step_dev != NULL: t_prev is read
// from the device counter, t_prev = *step_dev + step_dev_bias (and t_apply = t_prev + 1 when t_apply != 0), which makes
// the call hipGraph-replayable.  step_dev_bias = 1 brings rows current through the step that is being applied right
// now (its scalars are in the table): used for the rows of the NEXT batch, so that its forward finds them current.
// max_blocks > 0 caps the grid (grid-stride loop inside): a thin launch for side-stream work.
// PXR_LAZY_REPLAY=exact: replay missed steps with the dense sweep's own arithmetic (lazy == dense bit for bit for gaps <= 256)
static bool lazy_replay_fast() {      // read per launch ON PURPOSE: the test suite switches modes inside one process (a getenv is
  const char* e = getenv("PXR_LAZY_REPLAY");   // ~50 ns against a 3 us launch; under hipGraph replay it is not called at all)
  return !(e && strcmp(e, "exact") == 0);
}
static int adamw_rows_launch(const RowsArgs& a_in, int64_t work, int64_t max_blocks, int D, void* stream, const char* who) {
  if (work <= 0) return PXR_OK;
  RowsArgs a = a_in;
  a.fast = lazy_replay_fast() ? 1 : 0;
  a.sqrt_b2 = sqrtf(a.b2);
  a.log2_b2 = log2f(a.b2);
  a.log2_b1 = log2(1.0 - (double)a.one_m_b1);
  a.log2_sb = (float)(0.5 * log2((double)a.b2));
  // series replay (adamw_rows_kernel): fast mode only; its truncation bound is checked per row on the device, the moments need the
  // update terms to die out inside the window (rho = b1 / sqrt(b2) <= 0.95: the defaults give 0.90045).  PXR_LAZY_SERIES=0: the
  // carried-product loop for every gap (A/B; read per launch like PXR_LAZY_REPLAY)
  {
    const char* e = getenv("PXR_LAZY_SERIES");
    a.series = (a.fast && !(e && atoi(e) == 0) && a.b1 > 0.f && a.b2 > 0.f && a.b1 / a.sqrt_b2 <= 0.95f) ? 1 : 0;
    static const int env_min = getenv("PXR_LAZY_SERIES_MIN") ? atoi(getenv("PXR_LAZY_SERIES_MIN")) : 6;
    a.series_min = env_min < 2 ? 2 : env_min;
  }
  // Replayed steps per row before the closed form takes over.  Exact mode: 256 (the bit-identity window of the tests).  Fast
  // mode: 128 -- the Adam terms dropped beyond it sum to < 31.6 lr rho^128 / (1 - rho) = 4.7e-4 lr (rho = b1 / sqrt(b2) = 0.90045;
  // 4.7e-8 at lr 1e-4, against the 1e-5 parity budget), and the rows at the cap are the ones that set the kernel's duration.
  static const int env_window = getenv("PXR_LAZY_WINDOW") ? atoi(getenv("PXR_LAZY_WINDOW")) : -1;
  a.window = env_window >= 0 ? env_window : (a.fast ? PXR_LAZY_EXACT / 2 : PXR_LAZY_EXACT);
  // (0 would turn the whole replay into the closed form, which drops every Adam term: at least one replayed step)
  if (a.window < 1) a.window = 1;
  if (a.window > PXR_LAZY_EXACT) a.window = PXR_LAZY_EXACT;
  hipStream_t st = (hipStream_t)stream;
  // Elements per lane.  Step-by-step replays (exact mode, or betas that do not arm the series): 2 up to D = 2048 (a 512-wide row = 4
  // waves: the loop is VALU-issue bound and its chain per lane is what a long gap costs), 4 beyond (block size caps at 1024 threads).
  // Series replay: the kernel is a chain of memory round trips per row (id -> values | claim -> scalars -> stores); wider lanes mean
  // fewer waves per row, more rows resident at once and the gap's moments computed by fewer waves -- measured at D = 512, B = 64:
  // step 0.760 -> 0.752 ms from 2 to 4 (PXR_ROWS_EPL = 2 | 4 | 8: measuring knob)
  int epl = D <= 2048 ? 2 : 4;
  if (a.series && D % 4 == 0) epl = 4;
  static const int env_epl = getenv("PXR_ROWS_EPL") ? atoi(getenv("PXR_ROWS_EPL")) : 0;
  if ((env_epl == 4 || env_epl == 8) && D % env_epl == 0 && (D / env_epl + 63) / 64 * 64 <= 1024) epl = env_epl;
  if (env_epl == 2 && D <= 2048) epl = 2;
  const int lpr = ((D / epl + 63) / 64) * 64;          // lanes per row
  const int tl = lpr <= 64 ? 64 : (lpr <= 128 ? 128 : (lpr <= 256 ? 256 : (lpr <= 512 ? 512 : 1024)));
  const int rpb = tl >= 256 ? 1 : 256 / tl;          // == the kernel's RPB
  int64_t blocks = (work + rpb - 1) / rpb;
  if (blocks > 256 * 64) blocks = 256 * 64;
  // max_blocks > 0: a deliberately THIN grid (e.g. one workgroup per CU) for work that runs beside the step's GEMMs on
  // a second stream: it then takes one wave slot per SIMD instead of flooding every CU ahead of the GEMM workgroups
  if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
#define PXR_ROWS_CASE(L_, E_) hipLaunchKernelGGL((adamw_rows_kernel<L_, E_>), dim3((unsigned)blocks), dim3(L_ > 256 ? L_ : 256), 0, st, a, a.hyper)
  if (epl == 2) {
    switch (tl) {
      case 64: PXR_ROWS_CASE(64, 2); break;
      case 128: PXR_ROWS_CASE(128, 2); break;
      case 256: PXR_ROWS_CASE(256, 2); break;
      case 512: PXR_ROWS_CASE(512, 2); break;
      default: PXR_ROWS_CASE(1024, 2); break;
    }
  } else if (epl == 4) {
    switch (tl) {
      case 64: PXR_ROWS_CASE(64, 4); break;
      case 128: PXR_ROWS_CASE(128, 4); break;
      case 256: PXR_ROWS_CASE(256, 4); break;
      case 512: PXR_ROWS_CASE(512, 4); break;
      default: PXR_ROWS_CASE(1024, 4); break;
    }
  } else {
    switch (tl) {
      case 64: PXR_ROWS_CASE(64, 8); break;
      case 128: PXR_ROWS_CASE(128, 8); break;
      case 256: PXR_ROWS_CASE(256, 8); break;
      default: PXR_ROWS_CASE(512, 8); break;
    }
  }
#undef PXR_ROWS_CASE
  return pxr_check_launch(who);
}
extern "C" int pxr_adamw_rows_f32(float* table, float* m, float* v, int32_t* last, int64_t n_table, int D,
                                  const int64_t* rows, const int32_t* n_rows_dev, int64_t max_rows, const float* grows,
                                  const void* hyper, const void* cumlog, int64_t t_prev, int64_t t_apply,
                                  const int64_t* step_dev, int64_t step_dev_bias, int64_t max_blocks, double beta1,
                                  double beta2, double eps, void* stream) {
  PXR_REQUIRE(table && m && v && last && hyper && cumlog, "pxr_adamw_rows_f32: null pointer");
  PXR_REQUIRE(n_table > 0 && D > 0 && D % 4 == 0 && D <= 4096, "pxr_adamw_rows_f32: bad shape (D <= 4096)");
  PXR_REQUIRE(!rows || n_rows_dev, "pxr_adamw_rows_f32: row list needs its device count");
  PXR_REQUIRE(t_prev >= 0 && (t_apply == 0 || t_apply == t_prev + 1), "pxr_adamw_rows_f32: t_apply must be t_prev+1 or 0");
  RowsArgs a{};
  a.p = table; a.m = m; a.v = v; a.last = last; a.rows = rows; a.n_rows = n_rows_dev; a.n_fixed = n_table;
  a.grows = grows; a.hyper = (const float4*)hyper; a.cumlog = (const double*)cumlog;
  a.t_prev = (int)t_prev; a.t_apply = (int)t_apply; a.D = D;
  a.one_m_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_m_b2 = (float)(1.0 - beta2); a.eps = (float)eps;
  a.b1 = (float)beta1;
  a.step_dev = step_dev;
  a.t_prev_bias = (int)step_dev_bias;
  return adamw_rows_launch(a, rows ? max_rows : n_table, max_blocks, D, stream, "pxr_adamw_rows_f32");
}

// Catch-up of the rows named by a RAW id list (claim mode, see RowsArgs): ids[n_ids] may hold duplicates, 0 and
// out-of-range values (both skipped).  Same replay arithmetic as pxr_adamw_rows_f32 with t_apply = 0.
extern "C" int pxr_adamw_rows_ids_f32(float* table, float* m, float* v, int32_t* last, int64_t n_table, int D,
                                      const int64_t* ids, int64_t n_ids, const void* hyper, const void* cumlog,
                                      int64_t t_prev, const int64_t* step_dev, double beta1, double beta2, double eps,
                                      void* stream) {
  PXR_REQUIRE(table && m && v && last && hyper && cumlog && ids, "pxr_adamw_rows_ids_f32: null pointer");
  PXR_REQUIRE(n_table > 0 && D > 0 && D % 4 == 0 && D <= 4096, "pxr_adamw_rows_ids_f32: bad shape (D <= 4096)");
  PXR_REQUIRE(t_prev >= 0 && n_ids >= 0, "pxr_adamw_rows_ids_f32: bad step / count");
  RowsArgs a{};
  a.p = table; a.m = m; a.v = v; a.last = last; a.rows = ids; a.n_rows = nullptr; a.n_fixed = n_table;
  a.hyper = (const float4*)hyper; a.cumlog = (const double*)cumlog;
  a.t_prev = (int)t_prev; a.t_apply = 0; a.D = D;
  a.one_m_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_m_b2 = (float)(1.0 - beta2); a.eps = (float)eps;
  a.b1 = (float)beta1;
  a.step_dev = step_dev;
  a.claim = 1; a.n_list = n_ids; a.n_table = n_table;
  return adamw_rows_launch(a, n_ids, 0, D, stream, "pxr_adamw_rows_ids_f32");
}

// The same over a 2-D window of an id tensor: n_lists rows of row_len ids, row r starting at ids[r * row_stride] -- e.g. the
// INPUT ids items[:, 0, 0:L] of a SASRec batch [B, 2, L+1] (n_lists = B, row_len = L, row_stride = 2 (L+1)): the rows the
// forward pass reads first, caught up without a gather of the window.
extern "C" int pxr_adamw_rows_ids2d_f32(float* table, float* m, float* v, int32_t* last, int64_t n_table, int D,
                                        const int64_t* ids, int64_t n_lists, int64_t row_len, int64_t row_stride,
                                        const void* hyper, const void* cumlog, int64_t t_prev, const int64_t* step_dev,
                                        double beta1, double beta2, double eps, void* cur_hyper_out, void* stream) {
  PXR_REQUIRE(table && m && v && last && hyper && cumlog && ids, "pxr_adamw_rows_ids2d_f32: null pointer");
  PXR_REQUIRE(n_table > 0 && D > 0 && D % 4 == 0 && D <= 4096, "pxr_adamw_rows_ids2d_f32: bad shape (D <= 4096)");
  PXR_REQUIRE(t_prev >= 0 && n_lists >= 0 && row_len > 0 && row_stride >= row_len && row_len < (1ll << 30) && row_stride < (1ll << 30) &&
                  n_lists * row_len < (1ll << 31), "pxr_adamw_rows_ids2d_f32: bad step / window");
  RowsArgs a{};
  a.p = table; a.m = m; a.v = v; a.last = last; a.rows = ids; a.n_rows = nullptr; a.n_fixed = n_table;
  a.hyper = (const float4*)hyper; a.cumlog = (const double*)cumlog;
  a.t_prev = (int)t_prev; a.t_apply = 0; a.D = D;
  a.one_m_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_m_b2 = (float)(1.0 - beta2); a.eps = (float)eps;
  a.b1 = (float)beta1;
  a.step_dev = step_dev;
  a.claim = 1; a.n_list = n_lists * row_len; a.n_table = n_table;
  a.row_len = (int)row_len; a.row_stride = (int)row_stride;
  a.cur_out = (float4*)cur_hyper_out;      // optional: the scalars of the step about to run, for pxr_adamw_flat_tab_ex_f32
  return adamw_rows_launch(a, a.n_list, 0, D, stream, "pxr_adamw_rows_ids2d_f32");
}
