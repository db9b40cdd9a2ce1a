// embed_gather.hip -- item-ID embedding row gather (K1 of SURVEY.md §2.1).
//
// Replaces `self.item_embedding(items)` (reference code/REC/model/IDNet/sasrec.py:68,101) and the
// `item_feature[item_seq]` lookup of code/REC/model/PixelNet/mosasrec.py:102.
//
// HBM-bound: every output row is D*4 bytes read from a random table row + D*4 bytes written.
// Layout: table [N, D] row-major fp32, idx int64 [n], out [n, D].  One lane moves one 16-byte chunk
// (float4) so a wave moves 1 KiB per instruction; a block owns U*256 *consecutive* chunks and issues all
// U loads before the first store so that U row fetches per lane are in flight (random rows miss L2 and the
// 256 MiB Infinity Cache once the table is larger than it: 400 K x 512 x 4 B = 819 MB).
#include "pxr_common.h"

#include <cstdlib>

typedef float gvec4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ void __launch_bounds__(256) embed_gather_kernel(const gvec4* __restrict__ table,
                                                           const int64_t* __restrict__ idx,
                                                           gvec4* __restrict__ out, int64_t total_chunks,
                                                           int dv, int64_t n_rows_table, int32_t* status) {
  const int64_t base = ((int64_t)blockIdx.x * U) * 256 + threadIdx.x;
  gvec4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t c = base + (int64_t)u * 256;
    if (c < total_chunks) {
      const int64_t row = c / dv;
      const int col = (int)(c - row * dv);
      int64_t r = idx[row];
      // an out-of-range id is an ERROR (nn.Embedding raises, sasrec.py:68): flag it in the device status word -- the
      // host raises IndexError at its next check (ops.raise_on_bad_indices) -- and clamp so the access stays in bounds
      if (r < 0 || r >= n_rows_table) {
        if (status && col == 0) atomicOr(status, PXR_STATUS_BAD_INDEX);
        r = r < 0 ? 0 : n_rows_table - 1;
      }
      if constexpr (NT & 1) v[u] = __builtin_nontemporal_load(&table[r * dv + col]);
      else v[u] = table[r * dv + col];
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t c = base + (int64_t)u * 256;
    if (c < total_chunks) {
      if constexpr (NT & 2) __builtin_nontemporal_store(v[u], &out[c]);
      else out[c] = v[u];
    }
  }
}

extern "C" int pxr_embed_gather_f32(const float* table, int64_t N, int D, const int64_t* idx, int64_t n,
                                    float* out, void* stream) {
  PXR_REQUIRE(n >= 0, "pxr_embed_gather_f32: negative n");
  if (n == 0) return PXR_OK;   // an empty index list is legal (and its tensors have null data pointers)
  PXR_REQUIRE(table && idx && out, "pxr_embed_gather_f32: null pointer");
  PXR_REQUIRE(N > 0 && D > 0 && (D % 4) == 0, "pxr_embed_gather_f32: need N>0 and D %% 4 == 0 (D=%d)", D);
  const int dv = D / 4;
  const int64_t total = n * dv;
  // A/B knob (bit 0: non-temporal loads, bit 1: non-temporal stores, +4: 2 chunks per lane instead of 4).  Measured
  // on 208 896 uniform rows of the 819 MB table: plain 5.7 TB/s, NT stores 6.8 TB/s (default), NT both 5.9 TB/s.
  static const int variant = getenv("PXR_GATHER_VARIANT") ? atoi(getenv("PXR_GATHER_VARIANT")) : 2;
  const int U = variant >= 4 ? 2 : 4;
  const int64_t blocks = (total + (int64_t)U * 256 - 1) / ((int64_t)U * 256);
  PXR_REQUIRE(blocks < (1ll << 31), "pxr_embed_gather_f32: too many rows");
#define PXR_GATHER_LAUNCH(UU, NN)                                                                                  \
  hipLaunchKernelGGL((embed_gather_kernel<UU, NN>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,     \
                     (const gvec4*)table, idx, (gvec4*)out, total, dv, N, pxr_status_word())
  switch (variant) {
    case 0: PXR_GATHER_LAUNCH(4, 0); break;
    case 1: PXR_GATHER_LAUNCH(4, 1); break;
    case 3: PXR_GATHER_LAUNCH(4, 3); break;
    case 4: PXR_GATHER_LAUNCH(2, 0); break;
    case 6: PXR_GATHER_LAUNCH(2, 2); break;
    default: PXR_GATHER_LAUNCH(4, 2); break;
  }
#undef PXR_GATHER_LAUNCH
  return pxr_check_launch("pxr_embed_gather_f32");
}
