// gru.hip -- the gate arithmetic of one GRU time step, forward and backward (GRU4Rec: code/REC/model/IDNet/gru4rec.py:27-33,
// torch.nn.GRU with bias=False, batch_first=True).  The two matrix products of a step (x_t W_ih^T for all t at once,
// h_{t-1} W_hh^T per step) are the library's GEMMs; what is left is elementwise over [B, H]:
//     r = sigmoid(gi_r + gh_r),  z = sigmoid(gi_z + gh_z),  n = tanh(gi_n + r * gh_n),  h_t = (1 - z) * n + z * h_{t-1}
// (gate order r | z | n in the 3H columns, as torch lays weight_ih / weight_hh out).  The forward saves r, z, n and gh_n per
// step; the backward turns d h_t into d gi_t, d gh_t and the direct part of d h_{t-1}:
//     dn = dh (1 - z),  dz = dh (h_{t-1} - n),  da_n = dn (1 - n^2),  da_z = dz z (1 - z),  da_r = da_n gh_n r (1 - r)
//     d gi_t = [da_r | da_z | da_n],   d gh_t = [da_r | da_z | da_n * r],   d h_{t-1} (direct) = dh z
// (the other part of d h_{t-1} is d gh_t W_hh: a GEMM with this as its additive epilogue operand).
#include "pxr_common.h"

namespace pxr {

__global__ void __launch_bounds__(256) gru_gates_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                            const float* __restrict__ h_prev, float* __restrict__ h_out,
                                                            float* __restrict__ save, int64_t n, int H) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / H;
    const int c = (int)(i - b * H);
    const float* gir = gi + b * 3 * H;
    const float* ghr = gh + b * 3 * H;
    const float r = 1.0f / (1.0f + expf(-(gir[c] + ghr[c])));
    const float z = 1.0f / (1.0f + expf(-(gir[H + c] + ghr[H + c])));
    const float ghn = ghr[2 * H + c];
    const float nn = tanhf(gir[2 * H + c] + r * ghn);
    const float hp = h_prev ? h_prev[i] : 0.f;
    h_out[i] = (1.0f - z) * nn + z * hp;
    if (save) {
      float* s = save + b * 4 * H;
      s[c] = r; s[H + c] = z; s[2 * H + c] = nn; s[3 * H + c] = ghn;
    }
  }
}

__global__ void __launch_bounds__(256) gru_gates_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ save,
                                                            const float* __restrict__ h_prev, float* __restrict__ dgi,
                                                            float* __restrict__ dgh, float* __restrict__ dh_prev, int64_t n,
                                                            int H) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / H;
    const int c = (int)(i - b * H);
    const float* s = save + b * 4 * H;
    const float r = s[c], z = s[H + c], nn = s[2 * H + c], ghn = s[3 * H + c];
    const float g = dh[i];
    const float hp = h_prev ? h_prev[i] : 0.f;
    const float dan = g * (1.0f - z) * (1.0f - nn * nn);
    const float daz = g * (hp - nn) * z * (1.0f - z);
    const float dar = dan * ghn * r * (1.0f - r);
    float* a = dgi + b * 3 * H;
    float* e = dgh + b * 3 * H;
    a[c] = dar; a[H + c] = daz; a[2 * H + c] = dan;
    e[c] = dar; e[H + c] = daz; e[2 * H + c] = dan * r;
    dh_prev[i] = g * z;
  }
}

}  // namespace pxr

using namespace pxr;

static inline unsigned gru_grid(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (unsigned)(b > 4096 ? 4096 : b);
}

// One GRU step's gates.  gi, gh: [B, 3H] (r | z | n), h_prev: [B, H] or NULL (= zeros, the first step), h_out: [B, H];
// save (optional): [B, 4H] = r | z | n | gh_n for the backward.  All contiguous.
extern "C" int pxr_gru_gates_fwd_f32(const float* gi, const float* gh, const float* h_prev, float* h_out, float* save, int64_t B,
                                     int H, void* stream) {
  PXR_REQUIRE(gi && gh && h_out && B >= 0 && H > 0, "pxr_gru_gates_fwd_f32: bad args");
  if (B == 0) return PXR_OK;
  const int64_t n = B * H;
  hipLaunchKernelGGL(gru_gates_fwd_kernel, dim3(gru_grid(n)), dim3(256), 0, (hipStream_t)stream, gi, gh, h_prev, h_out, save, n, H);
  return pxr_check_launch("pxr_gru_gates_fwd_f32");
}

// Backward of the above: dh [B, H] = d loss / d h_t (everything that reaches h_t), save from the forward ->
// dgi, dgh [B, 3H], dh_prev [B, H] = the direct path dh * z (add dgh W_hh to it for the full d h_{t-1}).
extern "C" int pxr_gru_gates_bwd_f32(const float* dh, const float* save, const float* h_prev, float* dgi, float* dgh,
                                     float* dh_prev, int64_t B, int H, void* stream) {
  PXR_REQUIRE(dh && save && dgi && dgh && dh_prev && B >= 0 && H > 0, "pxr_gru_gates_bwd_f32: bad args");
  if (B == 0) return PXR_OK;
  const int64_t n = B * H;
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(gru_grid(n)), dim3(256), 0, (hipStream_t)stream, dh, save, h_prev, dgi, dgh,
                     dh_prev, n, H);
  return pxr_check_launch("pxr_gru_gates_bwd_f32");
}
