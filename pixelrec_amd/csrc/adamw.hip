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
#include "pxr_common.h"

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

__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, const AdamHyper& h) {
  p *= h.decay;
  m += (g - m) * h.one_m_b1;
  v = v * h.b2 + h.one_m_b2 * g * g;
  const float denom = sqrtf(v) * h.inv_sqrt_bc2 + h.eps;
  p -= h.step_size * (m / denom);
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
    if (r >= 0 && r < n_table) slot[r] = u;
  }
}

// One wave sweeps whole rows (grid-stride): coalesced float4 streams of p, m, v; gradient rows come from the
// compact buffer when slot >= 0.  ROWS_PER_ITER rows are in flight per wave to keep enough loads outstanding.
template <int VEC>
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
        pp[k] = *reinterpret_cast<const float4*>(p + o);
        mm[k] = *reinterpret_cast<const float4*>(m + o);
        vv[k] = *reinterpret_cast<const float4*>(v + o);
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
        *reinterpret_cast<float4*>(p + o) = pp[k];
        *reinterpret_cast<float4*>(m + o) = mm[k];
        *reinterpret_cast<float4*>(v + o) = vv[k];
      }
    }
  }
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
#define PXR_TABLE_CASE(V) \
  hipLaunchKernelGGL((adamw_table_kernel<V>), dim3((unsigned)blocks), dim3(256), 0, st, table, m, v, n_rows, D, slot, uniq_rows, h)
  if (vec <= 1) PXR_TABLE_CASE(1);
  else if (vec <= 2) PXR_TABLE_CASE(2);
  else if (vec <= 4) PXR_TABLE_CASE(4);
  else if (vec <= 8) PXR_TABLE_CASE(8);
  else PXR_TABLE_CASE(16);
#undef PXR_TABLE_CASE
  return pxr_check_launch("pxr_adamw_table_f32");
}
