// conv1d.hip -- the causal dilated convolution of NextItNet's residual blocks (code/REC/model/IDNet/nextitnet.py:160-194:
// ZeroPad2d((k-1) d, 0) on the left + nn.Conv2d(C, C, kernel_size=(1, k), dilation=d)) as a GEMM:
//     out[b, t, o] = bias[o] + sum_c sum_j W[o, c, 0, j] * x[b, t - (k-1-j) d, c]            (x = 0 for negative positions)
// With the im2col matrix  xcol[b L + t, c k + j] = x[b, t - (k-1-j) d, c]  the product is  xcol . W.view(C_out, C_in k)^T --
// the column order c k + j is exactly the memory order of the reference's Conv2d weight [C_out, C_in, 1, k], so the
// parameter is used in place (no permuted copy) by the library's GEMM entry points.  This file is the two data movers:
//   pxr_causal_im2col_f32   x [B, L, C]          -> xcol [B, L, C k]
//   pxr_causal_col2im_f32   dxcol [B, L, C k]    -> dx [B, L, C]     dx[b, t, c] = sum_j dxcol[b, t + (k-1-j) d, c k + j]
// (the transpose map: the gradient of the above; positions beyond L contribute nothing).
#include "pxr_common.h"

namespace pxr {

__global__ void __launch_bounds__(256) causal_im2col_kernel(const float* __restrict__ x, float* __restrict__ xcol, int64_t n,
                                                            int L, int C, int k, int d) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % k);
    const int64_t q = i / k;
    const int c = (int)(q % C);
    const int64_t bt = q / C;
    const int t = (int)(bt % L);
    const int ts = t - (k - 1 - j) * d;
    xcol[i] = ts >= 0 ? x[(bt - t + ts) * C + c] : 0.f;
  }
}

__global__ void __launch_bounds__(256) causal_col2im_kernel(const float* __restrict__ dxcol, float* __restrict__ dx, int64_t n,
                                                            int L, int C, int k, int d) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t bt = i / C;
    const int t = (int)(bt % L);
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
      const int td = t + (k - 1 - j) * d;
      if (td < L) acc += dxcol[((bt - t + td) * C + c) * k + j];
    }
    dx[i] = acc;
  }
}

}  // namespace pxr

using namespace pxr;

static inline unsigned c1d_grid(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (unsigned)(b > 8192 ? 8192 : b);
}

extern "C" int pxr_causal_im2col_f32(const float* x, float* xcol, int64_t B, int L, int C, int k, int dilation, void* stream) {
  PXR_REQUIRE(B >= 0 && L > 0 && C > 0 && k > 0 && dilation > 0, "pxr_causal_im2col_f32: bad shape");
  if (B == 0) return PXR_OK;
  PXR_REQUIRE(x && xcol, "pxr_causal_im2col_f32: null pointer");
  const int64_t n = B * L * C * k;
  hipLaunchKernelGGL(causal_im2col_kernel, dim3(c1d_grid(n)), dim3(256), 0, (hipStream_t)stream, x, xcol, n, L, C, k, dilation);
  return pxr_check_launch("pxr_causal_im2col_f32");
}

extern "C" int pxr_causal_col2im_f32(const float* dxcol, float* dx, int64_t B, int L, int C, int k, int dilation, void* stream) {
  PXR_REQUIRE(B >= 0 && L > 0 && C > 0 && k > 0 && dilation > 0, "pxr_causal_col2im_f32: bad shape");
  if (B == 0) return PXR_OK;
  PXR_REQUIRE(dxcol && dx, "pxr_causal_col2im_f32: null pointer");
  const int64_t n = B * L * C;
  hipLaunchKernelGGL(causal_col2im_kernel, dim3(c1d_grid(n)), dim3(256), 0, (hipStream_t)stream, dxcol, dx, n, L, C, k, dilation);
  return pxr_check_launch("pxr_causal_col2im_f32");
}
