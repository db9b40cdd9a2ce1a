// pixel.hip -- the two PixelNet-specific device pieces around the shared sequence block.
//
// (1) pxr_mosasrec_emb_grad_f32: gradient w.r.t. the visual encoder's output rows.  MOSASRec.forward views the
//     encoder output as [B, L+1, 2, D] (pos_t, neg_t interleaved, mosasrec.py:69-74): row (b,t,0) is used as INPUT at
//     position t (t < L) and as positive TARGET of position t-1 (t >= 1); row (b,t,1) only as negative target of
//     position t-1.  So, unlike the ID model, no de-duplication is needed: every row gets at most two terms,
//         d_emb[b,t,0] = [t<L] dx0[b,t] + [t>=1] coef[b,t-1] * out[b,t-1]
//         d_emb[b,t,1] =               - [t>=1] coef[b,t-1] * out[b,t-1]
//     which is what autograd produces for the slices/views of mosasrec.py:69-74 and the scores of :88-89.
//
// (2) pxr_image_u8_to_f32: the reference's per-item image transform (trainset.py:85-90: Resize(224) is the identity
//     on the 224x224 LMDB images of generate_lmdb.py, ToTensor = /255 and HWC->CHW, Normalize(mean=std=0.5)) done on
//     the GPU for a whole batch of uint8 HWC images gathered by item id; id 0 (padding) yields the all-zero image
//     (`self.pad_image = torch.zeros((3,224,224))`, trainset.py:96).
#include "pxr_common.h"

namespace pxr {

__global__ void __launch_bounds__(256) mosasrec_emb_grad_kernel(const float4* __restrict__ dx0,
                                                                const float4* __restrict__ out,
                                                                const float* __restrict__ coef, int B, int L, int dv,
                                                                float4* __restrict__ demb) {
  // one thread per float4 of d_emb [B, L+1, 2, D]
  const int64_t total = (int64_t)B * (L + 1) * 2 * dv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % dv);
    const int64_t row = i / dv;          // (b*(L+1) + t)*2 + s
    const int sgn = (int)(row & 1);
    const int64_t bt = row >> 1;
    const int b = (int)(bt / (L + 1)), t = (int)(bt - (int64_t)b * (L + 1));
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sgn == 0 && t < L) g = dx0[((int64_t)b * L + t) * dv + c];
    if (t >= 1) {
      const int64_t r = (int64_t)b * L + (t - 1);
      const float cf = sgn == 0 ? coef[r] : -coef[r];
      const float4 o = out[r * dv + c];
      g.x += cf * o.x; g.y += cf * o.y; g.z += cf * o.z; g.w += cf * o.w;
    }
    demb[i] = g;
  }
}

// out[n, c, y, x] = (img[ids[n], y, x, c] / 255 - 0.5) / 0.5   (0 for ids[n] == 0)
__global__ void __launch_bounds__(256) image_u8_to_f32_kernel(const uint8_t* __restrict__ store,
                                                              const int64_t* __restrict__ ids, int n, int64_t n_store,
                                                              int H, int W, float* __restrict__ out) {
  const int64_t hw = (int64_t)H * W;
  const int64_t per = hw * 3;
  const int64_t total = (int64_t)n * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int img = (int)(i / per);
    const int64_t rem = i - (int64_t)img * per;       // c*hw + p   (output is CHW, written coalesced)
    const int c = (int)(rem / hw);
    const int64_t p = rem - (int64_t)c * hw;
    const int64_t id = ids[img];
    float v = 0.f;
    if (id > 0 && id < n_store) v = ((float)store[id * per + p * 3 + c] / 255.0f - 0.5f) / 0.5f;
    out[i] = v;
  }
}

}  // namespace pxr

using namespace pxr;

extern "C" int pxr_mosasrec_emb_grad_f32(const float* dx0, const float* out, const float* coef, int B, int L, int D,
                                         float* d_emb, void* stream) {
  PXR_REQUIRE(dx0 && out && coef && d_emb, "pxr_mosasrec_emb_grad_f32: null pointer");
  PXR_REQUIRE(B > 0 && L > 0 && D > 0 && D % 4 == 0, "pxr_mosasrec_emb_grad_f32: bad shape");
  const int64_t total = (int64_t)B * (L + 1) * 2 * (D / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mosasrec_emb_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)dx0, (const float4*)out, coef, B, L, D / 4, (float4*)d_emb);
  return pxr_check_launch("pxr_mosasrec_emb_grad_f32");
}

// store: uint8 [n_store, H, W, 3] (row 0 unused = padding); ids int64 [n]; out fp32 [n, 3, H, W]
extern "C" int pxr_image_u8_to_f32(const uint8_t* store, int64_t n_store, int H, int W, const int64_t* ids, int n,
                                   float* out, void* stream) {
  PXR_REQUIRE(store && ids && out, "pxr_image_u8_to_f32: null pointer");
  PXR_REQUIRE(n_store > 0 && H > 0 && W > 0 && n >= 0, "pxr_image_u8_to_f32: bad shape");
  if (n == 0) return PXR_OK;
  const int64_t total = (int64_t)n * H * W * 3;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(image_u8_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, store, ids, n,
                     n_store, H, W, out);
  return pxr_check_launch("pxr_image_u8_to_f32");
}
