// sampler.hip -- on-device construction of a training batch from the positive windows.
//
// Reference code/REC/data/dataset/trainset.py:40-63: for a window of `len` items (left-padded to W = L+1) one negative is
// drawn per target position -- `random.randint(1, item_num - 1)` redrawn while it is one of the window's own items --
// giving len-1 negatives, left-padded to W; masked_index = len-1 ones, left-padded to L.  The reference does this in
// 10 DataLoader worker processes per rank, one Python loop per sample; at ~1 ms per step that is the bottleneck
// (SURVEY.md §8 f1).  Here one thread per (sequence, position) draws from a stateless counter hash
// (seed, batch counter, position, attempt) -- same distribution (uniform over [1, item_num-1] minus the window's own
// items), not the same stream as Python's Mersenne twister.  The host only gathers the windows.
#include "pxr_common.h"

namespace pxr {

constexpr int SAMPLER_MAX_TRIES = 64;   // P(64 straight hits on <= W own items out of item_num) is nil; then keep the last draw

__global__ void __launch_bounds__(256) sample_negatives_kernel(const int64_t* __restrict__ pos, int B, int W,
                                                               int64_t n_items, uint64_t seed, uint64_t batch_counter,
                                                               int64_t* __restrict__ items,
                                                               int64_t* __restrict__ masked_index) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * W) return;
  const int b = i / W, t = i - b * W;
  const int64_t* row = pos + (int64_t)b * W;
  int len = 0;                                   // ids are >= 1 and left-padded with 0: len = number of non-zeros
  for (int c = 0; c < W; ++c) len += row[c] != 0 ? 1 : 0;
  const bool target = t >= W - len + 1;          // the last len-1 columns carry a negative (and mask 1)
  int64_t neg = 0;
  if (target) {
    const bool reject = n_items > 2 * (int64_t)W;   // tiny catalogues: every id may be in the window -> accept anything
    for (int a = 0; a < SAMPLER_MAX_TRIES; ++a) {
      const uint32_t h = pxr_hash32(seed + batch_counter * 0x9E3779B97F4A7C15ull, (uint32_t)a, (uint64_t)i);
      // unbiased enough: 32 random bits scaled to [0, n_items-1) (bias < (n_items)/2^32)
      neg = 1 + (int64_t)(((uint64_t)h * (uint64_t)(n_items - 1)) >> 32);
      if (!reject) break;
      bool hit = false;
      for (int c = 0; c < W; ++c) hit |= row[c] == neg;
      if (!hit) break;
    }
  }
  int64_t* out = items + (int64_t)b * 2 * W;
  out[t] = row[t];
  out[W + t] = neg;
  if (t >= 1) masked_index[(int64_t)b * (W - 1) + (t - 1)] = target ? 1 : 0;
}

}  // namespace pxr

using namespace pxr;

// pos int64 [B, W] left-padded positive windows (W = L+1) -> items int64 [B, 2, W] (row 0 = pos, row 1 = negatives),
// masked_index int64 [B, W-1].  (seed, batch_counter) select the random stream: the same pair reproduces the batch.
extern "C" int pxr_sample_negatives_i64(const int64_t* pos, int B, int W, int64_t n_items, uint64_t seed,
                                        uint64_t batch_counter, int64_t* items, int64_t* masked_index, void* stream) {
  PXR_REQUIRE(B >= 0 && W >= 2 && n_items >= 2, "pxr_sample_negatives_i64: bad shape");
  if (B == 0) return PXR_OK;
  PXR_REQUIRE(pos && items && masked_index, "pxr_sample_negatives_i64: null pointer");
  const int n = B * W;
  hipLaunchKernelGGL(sample_negatives_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, pos, B, W,
                     n_items, seed, batch_counter, items, masked_index);
  return pxr_check_launch("pxr_sample_negatives_i64");
}
