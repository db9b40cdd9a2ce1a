// vit.hip -- the non-GEMM pieces of the PixelNet image encoder (HF CLIPVisionModel as the reference builds it in
// code/REC/model/load.py:90-120, wrapped by MeanItemEncoder, code/REC/model/layers.py:121-128).
//
// The ViT blocks run on the fp32-MFMA GEMM kernels (gemm_f32.hip: projections with the bias / residual / quick-GELU /
// ReLU epilogues, and BATCHED launches for the attention contractions S = Q K^T, O = P V and their gradients).  What is
// left is row-wise or elementwise and HBM-bound:
//   * pxr_softmax_rows_f32 / _bwd_f32   softmax over the keys of one (image, head, query) row, in place on the score
//                                       matrix [rows, ld] (ld = T rounded up to 4; the pad columns are written as zeros so
//                                       the matrix can be a k-contiguous GEMM operand);  no mask, no dropout (CLIP has
//                                       neither; the SASRec attention has its own fused kernels in attention.hip);
//   * pxr_vit_embed_f32                 [class token | patch projections] + position embedding -> token matrix;
//   * pxr_token_mean_f32 / pxr_token_mean_relu_bwd_f32   mean over the T tokens of relu(rec_fc(.)) and its gradient;
//   * pxr_add_f32                       out = a + b (the two branches of a residual gradient);
//   * pxr_dropout_f32                   y = dropout(x) with the library's counter-hash mask (GRU4Rec's emb_dropout; its own backward).
#include "pxr_common.h"

namespace pxr {

// One wave per row; lanes stride over the T columns.  p = exp(scale * (s - max)) / sum  (HF: softmax(q*scale . k)).
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ S, int64_t rows, int T, int ld,
                                                           float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* s = S + row * ld;
  constexpr int MAXV = 8;                       // T <= 512 in registers; longer rows re-read
  float v[MAXV];
  float m = -3.0e38f;
  const int nv = (T + 63) / 64;
  if (nv <= MAXV) {
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = k * 64 + lane;
      v[k] = (k < nv && c < T) ? s[c] * scale : -3.0e38f;
      m = fmaxf(m, v[k]);
    }
    m = wave_max(m);
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = k * 64 + lane;
      v[k] = (k < nv && c < T) ? __expf(v[k] - m) : 0.f;
      z += v[k];
    }
    const float inv = 1.0f / wave_sum(z);
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int c = k * 64 + lane;
      if (k < nv && c < ld) s[c] = c < T ? v[k] * inv : 0.f;
    }
  } else {
    for (int c = lane; c < T; c += 64) m = fmaxf(m, s[c] * scale);
    m = wave_max(m);
    float z = 0.f;
    for (int c = lane; c < T; c += 64) z += __expf(s[c] * scale - m);
    const float inv = 1.0f / wave_sum(z);
    for (int c = lane; c < ld; c += 64) s[c] = c < T ? __expf(s[c] * scale - m) * inv : 0.f;
  }
}

// dS = scale * P o (dP - sum_k dP o P), in place on dP (pad columns -> 0)
__global__ void __launch_bounds__(256) softmax_rows_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP,
                                                               int64_t rows, int T, int ld, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = P + row * ld;
  float* d = dP + row * ld;
  float dot = 0.f;
  for (int c = lane; c < T; c += 64) dot += p[c] * d[c];
  dot = wave_sum(dot);
  for (int c = lane; c < ld; c += 64) d[c] = c < T ? scale * p[c] * (d[c] - dot) : 0.f;
}

// out[n, t, :] = (t == 0 ? cls : patches[n, t-1, :]) + pos[t, :]       (HF CLIPVisionEmbeddings.forward)
__global__ void __launch_bounds__(256) vit_embed_kernel(const float4* __restrict__ patches, const float4* __restrict__ cls,
                                                        const float4* __restrict__ pos, float4* __restrict__ out,
                                                        int64_t n, int T, int H4) {
  const int64_t total = n * T * H4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % H4);
    const int64_t r = i / H4;
    const int t = (int)(r % T);
    const int64_t img = r / T;
    const float4 a = t == 0 ? cls[c] : patches[(img * (T - 1) + (t - 1)) * H4 + c];
    const float4 p = pos[(int64_t)t * H4 + c];
    out[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  }
}

// out[n, :] = mean_t x[n, t, :]      (MeanItemEncoder: torch.mean(rec_fc(x), dim=1), layers.py:128)
__global__ void __launch_bounds__(256) token_mean_kernel(const float4* __restrict__ x, float4* __restrict__ out, int64_t n,
                                                         int T, int D4) {
  const int64_t total = n * D4;
  const float inv = 1.0f / (float)T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % D4);
    const int64_t img = i / D4;
    const float4* p = x + img * T * D4 + c;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < T; ++t) {      // fixed order => deterministic
      const float4 v = p[(int64_t)t * D4];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[i] = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}

// dact[n, t, :] = act[n, t, :] > 0 ? dout[n, :] / T : 0     (through the token mean and the ReLU of rec_fc)
__global__ void __launch_bounds__(256) token_mean_relu_bwd_kernel(const float4* __restrict__ dout,
                                                                  const float4* __restrict__ act,
                                                                  float4* __restrict__ dact, int64_t n, int T, int D4) {
  const int64_t total = n * T * D4;
  const float inv = 1.0f / (float)T;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % D4);
    const int64_t img = i / ((int64_t)T * D4);
    const float4 a = act[i];
    const float4 g = dout[img * D4 + c];
    dact[i] = make_float4(a.x > 0.f ? g.x * inv : 0.f, a.y > 0.f ? g.y * inv : 0.f, a.z > 0.f ? g.z * inv : 0.f,
                          a.w > 0.f ? g.w * inv : 0.f);
  }
}

__global__ void __launch_bounds__(256) add_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                  float4* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = a[i], y = b[i];
    out[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

// ---- SASRec attention rows for sequences longer than the fused kernels of attention.hip hold (L > 128) ---------------
// The same arithmetic as there (reference layers.py:595-608 + sasrec.py:119-126): s = S / sqrt(d) + (key real and j <= i
// ? 0 : -1e9) -- ADDITIVE, so a fully masked (left-padded) query row becomes uniform over all L keys --, p = softmax(s)
// saved in place, pd = dropout(p) with the counter-hash mask of element ((b*H + h)*L + i)*L + j written to PD.
struct AttnRowsArgs {
  float* S; float* PD;                 // [B*H, L, ld]; PD may be null when p_drop == 0
  const float* dPD_in;                 // backward: gradient w.r.t. the dropped probabilities (in place -> dS)
  const int64_t* keymask; int64_t km_bstride;
  int B, H, L, ld;
  float sqrt_d, p_drop; uint32_t drop_thr, stream; uint64_t seed;
  const int64_t* step_dev;
};

__global__ void __launch_bounds__(256) attn_rows_fwd_kernel(AttnRowsArgs a) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);       // (b*H + h)*L + i
  if (row >= (int64_t)a.B * a.H * a.L) return;
  const int i = (int)(row % a.L);
  const int b = (int)(row / ((int64_t)a.H * a.L));
  float* s = a.S + row * a.ld;
  const int64_t* km = a.keymask + (int64_t)b * a.km_bstride;
  float m = -3.0e38f;
  for (int j = lane; j < a.L; j += 64) {
    const float v = s[j] / a.sqrt_d + ((km[j] != 0 && j <= i) ? 0.0f : -1e9f);
    s[j] = v;
    m = fmaxf(m, v);
  }
  m = wave_max(m);
  float z = 0.f;
  for (int j = lane; j < a.L; j += 64) z += __expf(s[j] - m);
  const float inv = 1.0f / wave_sum(z);
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  for (int j = lane; j < a.ld; j += 64) {
    const float p = j < a.L ? __expf(s[j] - m) * inv : 0.f;
    s[j] = p;
    if (a.PD) {
      const uint64_t pi = (uint64_t)row * a.L + j;
      a.PD[row * a.ld + j] = (j < a.L && (!drop || pxr_keep(a.seed, a.stream, pi, a.drop_thr))) ? (drop ? p * inv_keep : p) : 0.f;
    }
  }
}

// dS = P o (dP - rowsum(dP o P)) / sqrt(d)  with  dP = dPD * keep / (1 - p_drop); in place on dPD
__global__ void __launch_bounds__(256) attn_rows_bwd_kernel(AttnRowsArgs a, float* dPD) {
  if (a.step_dev) a.seed += (uint64_t)a.step_dev[0];
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)a.B * a.H * a.L) return;
  const float* p = a.S + row * a.ld;
  float* d = dPD + row * a.ld;
  const bool drop = a.drop_thr != 0u;
  const float inv_keep = 1.0f / (1.0f - a.p_drop);
  float dot = 0.f;
  for (int j = lane; j < a.L; j += 64) {
    float g = d[j];
    if (drop) g = pxr_keep(a.seed, a.stream, (uint64_t)row * a.L + j, a.drop_thr) ? g * inv_keep : 0.f;
    d[j] = g;
    dot += g * p[j];
  }
  dot = wave_sum(dot);
  for (int j = lane; j < a.ld; j += 64) d[j] = j < a.L ? p[j] * (d[j] - dot) / a.sqrt_d : 0.f;
}

static inline unsigned grid_for(int64_t work_items) {
  int64_t b = (work_items + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace pxr

using namespace pxr;

// In place: S[row, :T] <- softmax(scale * S[row, :T]), S[row, T:ld] <- 0.   rows = images * heads * T.
extern "C" int pxr_softmax_rows_f32(float* S, int64_t rows, int T, int ld, float scale, void* stream) {
  PXR_REQUIRE(S && rows >= 0 && T > 0 && ld >= T, "pxr_softmax_rows_f32: bad args");
  if (rows == 0) return PXR_OK;
  PXR_REQUIRE((rows + 3) / 4 < (1ll << 31), "pxr_softmax_rows_f32: too many rows");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, rows, T,
                     ld, scale);
  return pxr_check_launch("pxr_softmax_rows_f32");
}

// In place on dP: dS = scale * P o (dP - rowsum(dP o P))   (gradient w.r.t. the UNscaled scores)
extern "C" int pxr_softmax_rows_bwd_f32(const float* P, float* dP, int64_t rows, int T, int ld, float scale,
                                        void* stream) {
  PXR_REQUIRE(P && dP && rows >= 0 && T > 0 && ld >= T, "pxr_softmax_rows_bwd_f32: bad args");
  if (rows == 0) return PXR_OK;
  PXR_REQUIRE((rows + 3) / 4 < (1ll << 31), "pxr_softmax_rows_bwd_f32: too many rows");
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP,
                     rows, T, ld, scale);
  return pxr_check_launch("pxr_softmax_rows_bwd_f32");
}

extern "C" int pxr_vit_embed_f32(const float* patches, const float* cls, const float* pos, float* out, int64_t n, int T,
                                 int H, void* stream) {
  PXR_REQUIRE(patches && cls && pos && out && n >= 0 && T >= 2 && H > 0 && H % 4 == 0, "pxr_vit_embed_f32: bad args");
  if (n == 0) return PXR_OK;
  hipLaunchKernelGGL(vit_embed_kernel, dim3(grid_for(n * T * (H / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)patches, (const float4*)cls, (const float4*)pos, (float4*)out, n, T, H / 4);
  return pxr_check_launch("pxr_vit_embed_f32");
}

extern "C" int pxr_token_mean_f32(const float* x, float* out, int64_t n, int T, int D, void* stream) {
  PXR_REQUIRE(x && out && n >= 0 && T > 0 && D > 0 && D % 4 == 0, "pxr_token_mean_f32: bad args");
  if (n == 0) return PXR_OK;
  hipLaunchKernelGGL(token_mean_kernel, dim3(grid_for(n * (D / 4))), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                     (float4*)out, n, T, D / 4);
  return pxr_check_launch("pxr_token_mean_f32");
}

extern "C" int pxr_token_mean_relu_bwd_f32(const float* dout, const float* act, float* dact, int64_t n, int T, int D,
                                           void* stream) {
  PXR_REQUIRE(dout && act && dact && n >= 0 && T > 0 && D > 0 && D % 4 == 0, "pxr_token_mean_relu_bwd_f32: bad args");
  if (n == 0) return PXR_OK;
  hipLaunchKernelGGL(token_mean_relu_bwd_kernel, dim3(grid_for(n * T * (D / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)dout, (const float4*)act, (float4*)dact, n, T, D / 4);
  return pxr_check_launch("pxr_token_mean_relu_bwd_f32");
}

// y[i] = keep(i) ? x[i] / (1 - p) : 0, keep = the counter hash of (seed + *step_dev, stream_id, i): calling it on the upstream
// gradient regenerates the forward's mask (nothing is stored).  n % 4 == 0.
namespace pxr {
__global__ void __launch_bounds__(256) dropout_kernel(const float4* __restrict__ x, float4* __restrict__ y, int64_t n4, float inv_keep,
                                                      uint32_t thr, uint64_t seed, uint32_t stream_id, const int64_t* step_dev) {
  if (step_dev) seed += (uint64_t)step_dev[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 v = x[i];
    const uint64_t e = (uint64_t)i * 4;
    v.x = pxr_keep(seed, stream_id, e, thr) ? v.x * inv_keep : 0.f;
    v.y = pxr_keep(seed, stream_id, e + 1, thr) ? v.y * inv_keep : 0.f;
    v.z = pxr_keep(seed, stream_id, e + 2, thr) ? v.z * inv_keep : 0.f;
    v.w = pxr_keep(seed, stream_id, e + 3, thr) ? v.w * inv_keep : 0.f;
    y[i] = v;
  }
}
}  // namespace pxr
extern "C" int pxr_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, uint32_t stream_id, const int64_t* step_dev,
                               void* stream) {
  PXR_REQUIRE(x && y && n >= 0 && n % 4 == 0, "pxr_dropout_f32: bad args (n must be a multiple of 4)");
  PXR_REQUIRE(p >= 0.f && p < 1.f, "pxr_dropout_f32: p must be in [0, 1)");
  PXR_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "pxr_dropout_f32: operands must be 16-byte aligned");
  if (n == 0) return PXR_OK;
  hipLaunchKernelGGL(pxr::dropout_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, (float4*)y, n / 4,
                     1.0f / (1.0f - p), pxr_drop_threshold(p), seed, stream_id, step_dev);
  return pxr_check_launch("pxr_dropout_f32");
}

extern "C" int pxr_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
  PXR_REQUIRE(a && b && out && n >= 0 && n % 4 == 0, "pxr_add_f32: bad args (n must be a multiple of 4)");
  if (n == 0) return PXR_OK;
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)a,
                     (const float4*)b, (float4*)out, n / 4);
  return pxr_check_launch("pxr_add_f32");
}

// S [B*H, L, ld] (= Q K^T, unscaled) -> in place the softmax probabilities of the SASRec attention (additive -1e9
// causal + key mask, reference layers.py:595-604, sasrec.py:119-126); PD (may be NULL when p_drop == 0) receives the
// dropped probabilities (layers.py:608).  For sequences beyond the fused kernels of attention.hip (L > 128).
extern "C" int pxr_attn_rows_fwd_f32(float* S, float* PD, const int64_t* keymask, int64_t km_bstride, int B, int H, int L,
                                     int ld, float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev,
                                     int d, void* stream) {
  PXR_REQUIRE(S && keymask && B >= 0 && H > 0 && L > 0 && ld >= L && d > 0, "pxr_attn_rows_fwd_f32: bad args");
  PXR_REQUIRE(p_drop >= 0.f && p_drop < 1.f && (p_drop == 0.f || PD), "pxr_attn_rows_fwd_f32: dropout needs PD");
  if (B == 0) return PXR_OK;
  AttnRowsArgs a{};
  a.S = S; a.PD = PD; a.keymask = keymask; a.km_bstride = km_bstride; a.B = B; a.H = H; a.L = L; a.ld = ld;
  a.sqrt_d = sqrtf((float)d); a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id;
  a.seed = seed; a.step_dev = step_dev;
  const int64_t rows = (int64_t)B * H * L;
  hipLaunchKernelGGL(attn_rows_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return pxr_check_launch("pxr_attn_rows_fwd_f32");
}

// dPD [B*H, L, ld] (gradient w.r.t. the dropped probabilities) -> in place the gradient w.r.t. the unscaled scores S
extern "C" int pxr_attn_rows_bwd_f32(const float* P, float* dPD, int B, int H, int L, int ld, float p_drop, uint64_t seed,
                                     uint32_t stream_id, const int64_t* step_dev, int d, void* stream) {
  PXR_REQUIRE(P && dPD && B >= 0 && H > 0 && L > 0 && ld >= L && d > 0, "pxr_attn_rows_bwd_f32: bad args");
  if (B == 0) return PXR_OK;
  AttnRowsArgs a{};
  a.S = const_cast<float*>(P); a.B = B; a.H = H; a.L = L; a.ld = ld;
  a.sqrt_d = sqrtf((float)d); a.p_drop = p_drop; a.drop_thr = pxr_drop_threshold(p_drop); a.stream = stream_id;
  a.seed = seed; a.step_dev = step_dev;
  const int64_t rows = (int64_t)B * H * L;
  hipLaunchKernelGGL(attn_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, dPD);
  return pxr_check_launch("pxr_attn_rows_bwd_f32");
}
