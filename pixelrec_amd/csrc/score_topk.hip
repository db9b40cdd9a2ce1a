// score_topk.hip -- full-catalog scoring fused with the full-sort evaluation epilogue (K19-K21).
//
// Reference: scores = seq_output @ item_feature^T   [B_e, N]  (sasrec.py:112)            1.64 GB at 1024 x 400K
//            scores[:, 0] = -inf; scores[(history_u, history_i)] = -inf                   (trainer.py:333-336)
//            torch.topk(scores, max(topk)) + [B_e, N] int positive matrix + gather         (collector.py:131-139)
// Here the scores never leave the chip: every workgroup owns 128 users x one contiguous range of item tiles,
// runs the fp32-MFMA main loop of gemm_f32.cuh per 128x128 tile, drops the tile into LDS (re-using the GEMM staging
// buffers), applies the -inf masks there (column 0, the users' histories, the ragged last tile) and lets each
// thread keep a private sorted top-K list in registers for (one user, half of the tile's columns).  A second
// tiny kernel merges the 2 x n_split partial lists per user.  HBM traffic = the table once (819 MB) instead of
// table + 2 x 1.64 GB of scores + the int matrix.
//
// Ties: scores are continuous fp32 dot products, so ties only occur among -inf entries, which can reach the
// list only when fewer than K unmasked items exist.
#include "gemm_b3.cuh"
#include "gemm_p4.cuh"

#include <cstdlib>

extern "C" int pxr_get_gemm_mode(void);

namespace pxr {

constexpr int ST_BM = 128, ST_BN = 128;
constexpr int ST_LD = ST_BN + 1;  // score-tile row stride in LDS: lane=row scans are conflict-free
using StCfg = GemmCfg<ST_BM, ST_BN, true, true>;
constexpr int ST_SMEM_FLOATS = (2 * StCfg::STAGE > ST_BM * ST_LD) ? 2 * StCfg::STAGE : ST_BM * ST_LD;

template <int KT>
struct TopList {
  float v[KT];
  int i[KT];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int k = 0; k < KT; ++k) { v[k] = -INFINITY; i[k] = -1; }
  }
  // sorted descending; v[KT-1] is the admission threshold
  __device__ __forceinline__ void insert(float x, int id) {
    if (!(x > v[KT - 1])) return;
    v[KT - 1] = x; i[KT - 1] = id;
#pragma unroll
    for (int k = KT - 1; k > 0; --k) {
      if (v[k] > v[k - 1]) {
        const float tv = v[k]; v[k] = v[k - 1]; v[k - 1] = tv;
        const int ti = i[k]; i[k] = i[k - 1]; i[k - 1] = ti;
      }
    }
  }
};

struct ScoreTopkArgs {
  const float* users; int64_t ld_users;   // [B, D] rows at stride ld_users
  const float* table;                     // [N, D]
  const int* hist_ptr;                    // [B+1] CSR offsets into hist_items (may be null)
  const int64_t* hist_items;              // item ids to mask per user
  float* part_val; int* part_idx;         // [B, n_split*2, KT]
  int B, N, D, tiles_n, n_split, row_blocks;
  int tile_stride;                        // 1 = every item tile; s > 1 = a SAMPLE (every s-th tile of each split's range)
  // threshold variant (score_thresh_kernel): per-user admission threshold and append buffers
  const float* tau; int* cand_cnt; float* cand_val; int* cand_idx; int cand_cap;
  unsigned long long* clk;                // measurement hook (pxr_score_topk_clock_out; null: off): see score_clock_start
};

// The clock a main-pass kernel ACTUALLY runs at.  MI355X lowers its shader clock under sustained MFMA load (1.55-1.9 GHz instead of
// 2.4 on the scoring passes: the part sits at its power limit), so a roofline fraction against the nominal peak mixes kernel
// quality with power management.  When a buffer is registered, thread 0 of workgroup 0 samples the shader-clock counter (s_memtime)
// and the constant 100 MHz reference (s_memrealtime) at its first and last instruction and ADDS both differences to clk[0] / clk[1]:
// clock [GHz] = clk[0] / clk[1] * 0.1 over every main-pass launch since the buffer was zeroed.  Measured INSIDE the kernel it prices
// (round 5's one-wave probe on a second stream read 2.4 GHz beside the same kernels: its wave evidently ran in the gaps).
// (nothing is kept live across the kernel -- every register of the main passes is taken: the start samples are SUBTRACTED from the
// sums at the first instruction, the end samples added at the last)
__device__ __forceinline__ void score_clock_start(unsigned long long* clk) {
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
    atomicAdd(clk, 0ull - (unsigned long long)__builtin_readcyclecounter());
    atomicAdd(clk + 1, 0ull - (unsigned long long)wall_clock64());
  }
}
__device__ __forceinline__ void score_clock_stop(unsigned long long* clk) {
  if (clk && blockIdx.x == 0 && threadIdx.x == 0) {
    atomicAdd(clk, (unsigned long long)__builtin_readcyclecounter());
    atomicAdd(clk + 1, (unsigned long long)wall_clock64());
  }
}

template <int KT>
__global__ void __launch_bounds__(GEMM_THREADS) score_topk_kernel(ScoreTopkArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[ST_SMEM_FLOATS];
  const int tid = threadIdx.x;
  const int t = xcd_remap(blockIdx.x, a.row_blocks * a.n_split);
  const int rb = t % a.row_blocks, sp = t / a.row_blocks;   // row-block fastest: neighbours share the item tiles
  const int m0 = rb * ST_BM;
  const int per = (a.tiles_n + a.n_split - 1) / a.n_split;
  const int tn0 = sp * per, tn1 = min(a.tiles_n, tn0 + per);

  const int my_row = tid & 127, my_half = tid >> 7;
  const int urow = m0 + my_row;
  TopList<KT> top;
  top.init();

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, r = lane & 31;
  int hb = 0, he = 0;
  if (a.hist_ptr) {
    hb = a.hist_ptr[m0];
    he = a.hist_ptr[min(a.B, m0 + ST_BM)];
  }

  for (int tn = tn0; tn < tn1; tn += a.tile_stride) {
    const int n0 = tn * ST_BN;
    typename StCfg::Acc accs;
    gemm_mainloop<ST_BM, ST_BN, true, true, false, 1, 1>(accs, a.users, a.ld_users, a.table, (int64_t)a.D, a.B, a.N, 0, a.D, m0,
                                             n0, smem);
    // (the main loop ends with a barrier: the staging buffers are free) -> score tile in LDS, masks applied
#pragma unroll
    for (int j = 0; j < StCfg::TN; ++j) {
      const int cl = wn * StCfg::WN + j * 32 + r;
      const int col = n0 + cl;
      const bool dead = (col >= a.N) || (col == 0);       // ragged edge; padding item 0 (trainer.py:334)
#pragma unroll
      for (int i = 0; i < StCfg::TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = wm * StCfg::WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          smem[rl * ST_LD + cl] = dead ? -INFINITY : accs.v[i][j][e];
        }
    }
    __syncthreads();
    // history mask (trainer.py:335-336): every (user, item) pair of this row block that falls in this tile
    for (int p = hb + tid; p < he; p += GEMM_THREADS) {
      const int64_t it = a.hist_items[p];
      if (it >= n0 && it < n0 + ST_BN) {
        // owner row of pair p: the user u with hist_ptr[u] <= p < hist_ptr[u+1]; binary search inside the block
        int lo = m0, hi = min(a.B, m0 + ST_BM) - 1;
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (a.hist_ptr[mid] <= p) lo = mid; else hi = mid - 1;
        }
        smem[(lo - m0) * ST_LD + (int)(it - n0)] = -INFINITY;
      }
    }
    __syncthreads();
    if (urow < a.B) {
      const float* rowp = smem + my_row * ST_LD + my_half * 64;
      const int cbase = n0 + my_half * 64;
#pragma unroll 8
      for (int c = 0; c < 64; ++c) top.insert(rowp[c], cbase + c);
    }
    __syncthreads();
  }
  if (urow < a.B) {
    const int64_t o = ((int64_t)urow * (a.n_split * 2) + (sp * 2 + my_half)) * KT;
#pragma unroll
    for (int k = 0; k < KT; ++k) { a.part_val[o + k] = top.v[k]; a.part_idx[o + k] = top.i[k]; }
  }
}

// ---- variant 2: selection in the accumulator registers --------------------------------------------------------------
// The product is taken TRANSPOSED (A = the 128-item table tile, B = the 128 users), so in the 32x32 MFMA result layout
// a lane holds ONE user (column = lane & 31) and 16 items of it in registers: the top-K list of (user, item sub-block) is
// private to the lane and is fed straight from the accumulators -- no score tile in LDS, no scan.  8 waves per workgroup
// (each owns a 64-item x 32-user block: FINE = 2) at <= 128 VGPRs, i.e. two workgroups per CU, so that one workgroup's
// selection (VALU) runs under the other's MFMAs.  (16 waves of 32x32 -- the fastest cut of the plain GEMM -- would need
// <= 64 VGPRs for two workgroups per CU: the 20 list registers then spill.)
// History pairs become a [128 users] x [128 items] bitmap in LDS (2 KB) per tile; item 0 and the ragged edge are id tests.
// Per user and workgroup there are 4 partial lists (2 item sub-blocks x 2 half-waves); the merge kernel is unchanged.
// FINE = 1 (PXR_TOPK_VARIANT=3): 16 waves of 32x32, one workgroup per CU (the lists do not fit in the 64 VGPRs that two
// 1024-thread workgroups per CU would leave).
constexpr int ST2_BITMAP_WORDS = ST_BN * (ST_BM / 32);        // [user][4 words of 32 items]

template <int KT, int FINE>
__global__ void __launch_bounds__((FINE == 1 ? 1024 : 512), (FINE == 1 ? 4 : 4)) score_topk2_kernel(ScoreTopkArgs a) {
  using St2Cfg = GemmCfg<ST_BM, ST_BN, true, true, 1, FINE>;
  constexpr int ST2_THREADS = St2Cfg::NT;
  constexpr int ST2_LISTS = St2Cfg::WGM * 2;                  // item sub-blocks of the wave grid x 2 half-waves
  constexpr int ST2_SMEM_FLOATS = 2 * St2Cfg::STAGE + ST2_BITMAP_WORDS;
  __shared__ __attribute__((aligned(16))) float smem[ST2_SMEM_FLOATS];
  unsigned* bitmap = reinterpret_cast<unsigned*>(smem + 2 * St2Cfg::STAGE);
  const int tid = threadIdx.x;
  const int t = xcd_remap(blockIdx.x, a.row_blocks * a.n_split);
  const int rb = t % a.row_blocks, sp = t / a.row_blocks;   // row-block fastest: neighbours share the item tiles
  const int u0 = rb * ST_BN;                                // first user of this workgroup
  const int per = (a.tiles_n + a.n_split - 1) / a.n_split;
  const int tn0 = sp * per, tn1 = min(a.tiles_n, tn0 + per);
  const int lane = tid & 63, wave = tid >> 6;
  const int wi = wave / St2Cfg::WGN, wu = wave % St2Cfg::WGN;   // item sub-block / user sub-block of this wave
  const int h = lane >> 5, r = lane & 31;
  const int ul = wu * 32 + r;                                // user (local) of this lane
  TopList<KT> top;
  top.init();
  int hb = 0, he = 0;
  if (a.hist_ptr) {
    hb = a.hist_ptr[u0];
    he = a.hist_ptr[min(a.B, u0 + ST_BN)];
  }
  for (int tn = tn0; tn < tn1; ++tn) {
    const int i0 = tn * ST_BM;                               // first item of the tile
    static_assert(ST2_BITMAP_WORDS <= ST2_THREADS, "one thread per bitmap word");
    if (tid < ST2_BITMAP_WORDS) bitmap[tid] = 0u;
    typename St2Cfg::Acc accs;
    // rows (M) = items of the table tile, columns (N) = users
    gemm_mainloop<ST_BM, ST_BN, true, true, false, 1, (FINE == 1 ? 2 : 1), 2, FINE>(accs, a.table, (int64_t)a.D, a.users, a.ld_users, a.N, a.B,
                                                               0, a.D, i0, u0, smem);
    // (the main loop starts and ends with barriers: the zeroed bitmap is visible, the staging buffers are not touched here)
    for (int p = hb + tid; p < he; p += ST2_THREADS) {
      const int64_t it = a.hist_items[p];
      if (it >= i0 && it < i0 + ST_BM) {
        int lo = u0, hi = min(a.B, u0 + ST_BN) - 1;          // owner of pair p: hist_ptr[u] <= p < hist_ptr[u+1]
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (a.hist_ptr[mid] <= p) lo = mid; else hi = mid - 1;
        }
        const int il = (int)(it - i0);
        atomicOr(&bitmap[(lo - u0) * (ST_BM / 32) + (il >> 5)], 1u << (il & 31));
      }
    }
    __syncthreads();
    const bool user_ok = (u0 + ul) < a.B;
#pragma unroll
    for (int bi = 0; bi < St2Cfg::TM; ++bi) {
      const unsigned bits = bitmap[ul * (ST_BM / 32) + wi * St2Cfg::TM + bi];   // this lane's user x 32 items
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int il = (e & 3) + 8 * (e >> 2) + 4 * h;       // item within the 32-item sub-block
        const int item = i0 + wi * St2Cfg::WM + bi * 32 + il;
        const float x = accs.v[bi][0][e];
        const bool cand = user_ok && item != 0 && item < a.N && !((bits >> il) & 1u) && (x > top.v[KT - 1]);
        if (__any(cand)) {
          if (cand) top.insert(x, item);
        }
      }
    }
    __syncthreads();                                         // bitmap is re-zeroed at the top of the next tile
  }
  if ((u0 + ul) < a.B) {
    const int64_t o = ((int64_t)(u0 + ul) * (a.n_split * ST2_LISTS) + (sp * ST2_LISTS + wi * 2 + h)) * KT;
#pragma unroll
    for (int k = 0; k < KT; ++k) { a.part_val[o + k] = top.v[k]; a.part_idx[o + k] = top.i[k]; }
  }
}

// ---- variant 4: two passes around a per-user threshold ------------------------------------------------------------
// Keeping a top-K list per lane costs 2K registers, which the fastest cut of the GEMM (16 waves of 32x32 at <= 64 VGPRs,
// two workgroups per CU) does not have.  So: (1) the list kernel above scores a SAMPLE of the item tiles (every 64th)
// and its merged K-th best value becomes tau[user] -- the K-th best of a subset is a LOWER bound of the K-th best of the
// catalogue, so no true top-K item scores below it; (2) this kernel scores EVERY tile at full GEMM speed and appends the
// few survivors (score >= tau, not masked: ~64 K per user) to a per-user candidate buffer with one atomic each;
// (3) the merge kernel picks the K best candidates (ties: lower item id first, so the result does not depend on the
// order of the appends).
constexpr int ST4_HIST_CAP = 2048;   // history pairs of the workgroup's (128 users x its item range) kept in LDS

// B3 = 1: the product runs on the bf16 matrix pipe through the exact 3 x bf16 split (gemm_b3.cuh; GEMM mode bf16x3): 96 KB
// of staging LDS, one 16-wave workgroup per CU at 128 VGPRs instead of two at 64.
template <int B3>
__global__ void __launch_bounds__(1024, (B3 ? 4 : 8)) score_thresh_kernel(ScoreTopkArgs a) {
  using Cfg = GemmCfg<ST_BM, ST_BN, true, true, 1, 1>;
  constexpr int MAIN_FLOATS = B3 ? B3Cfg<ST_BM, ST_BN, 1>::LDS_BYTES / 4 : 2 * Cfg::STAGE;
  constexpr int SMEM_FLOATS = MAIN_FLOATS + ST2_BITMAP_WORDS + ST4_HIST_CAP + 4;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  unsigned* bitmap = reinterpret_cast<unsigned*>(smem + MAIN_FLOATS);
  unsigned* hlist = bitmap + ST2_BITMAP_WORDS;             // (user_local << 20) | (item - first item of the range)
  int* hcount = reinterpret_cast<int*>(hlist + ST4_HIST_CAP);
  const int tid = threadIdx.x;
  const int t = xcd_remap(blockIdx.x, a.row_blocks * a.n_split);
  const int rb = t % a.row_blocks, sp = t / a.row_blocks;
  const int u0 = rb * ST_BN;
  const int per = (a.tiles_n + a.n_split - 1) / a.n_split;
  const int tn0 = sp * per, tn1 = min(a.tiles_n, tn0 + per);
  int hb = 0, he = 0;
  if (a.hist_ptr) {
    hb = a.hist_ptr[u0];
    he = a.hist_ptr[min(a.B, u0 + ST_BN)];
  }
  // The workgroup's users have a few thousand history pairs, of which only those inside ITS item range (a 1/n_split
  // slice of the catalogue) can ever hit one of its tiles: collect them once (owner resolved here), so that the per-tile
  // mask pass walks a short LDS list instead of re-reading every pair from global memory for each of its ~50 tiles.
  if (tid == 0) *hcount = 0;
  __syncthreads();
  const int64_t r_lo = (int64_t)tn0 * ST_BM, r_hi = (int64_t)tn1 * ST_BM;
  for (int p = hb + tid; p < he; p += 1024) {
    const int64_t it = a.hist_items[p];
    if (it >= r_lo && it < r_hi) {
      int lo = u0, hi = min(a.B, u0 + ST_BN) - 1;          // owner of pair p: hist_ptr[u] <= p < hist_ptr[u+1]
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.hist_ptr[mid] <= p) lo = mid; else hi = mid - 1;
      }
      const int pos = atomicAdd(hcount, 1);
      if (pos < ST4_HIST_CAP) hlist[pos] = ((unsigned)(lo - u0) << 20) | (unsigned)(it - r_lo);
    }
  }
  __syncthreads();
  const int n_hist = *hcount;
  // (else: walk the global pairs per tile, as before.  The packed pair keeps 20 bits for the item offset inside the split's
  // range: a wider range -- few splits over a huge catalogue -- would run into the user field)
  const bool list_ok = n_hist <= ST4_HIST_CAP && (r_hi - r_lo) <= (1ll << 20);
  for (int tn = tn0; tn < tn1; ++tn) {
    const int i0 = tn * ST_BM;
    if (tid < ST2_BITMAP_WORDS) bitmap[tid] = 0u;
    typename Cfg::Acc accs;
    if constexpr (B3)
      gemm_b3_mainloop<ST_BM, ST_BN, true, true, 1>(accs, a.table, (int64_t)a.D, a.users, a.ld_users, a.N, a.B, 0, a.D, i0, u0,
                                                    reinterpret_cast<char*>(smem));
    else
      gemm_mainloop<ST_BM, ST_BN, true, true, false, 1, 2, 2, 1>(accs, a.table, (int64_t)a.D, a.users, a.ld_users, a.N, a.B,
                                                                 0, a.D, i0, u0, smem);
    if (list_ok) {
      const unsigned off0 = (unsigned)(i0 - (int)r_lo);
      for (int q = tid; q < n_hist; q += 1024) {
        const unsigned e = hlist[q], off = e & 0xFFFFFu;
        if (off >= off0 && off < off0 + ST_BM) {
          const int il = (int)(off - off0);
          atomicOr(&bitmap[(e >> 20) * (ST_BM / 32) + (il >> 5)], 1u << (il & 31));
        }
      }
    } else {
      for (int p = hb + tid; p < he; p += 1024) {
        const int64_t it = a.hist_items[p];
        if (it >= i0 && it < i0 + ST_BM) {
          int lo = u0, hi = min(a.B, u0 + ST_BN) - 1;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.hist_ptr[mid] <= p) lo = mid; else hi = mid - 1;
          }
          const int il = (int)(it - i0);
          atomicOr(&bitmap[(lo - u0) * (ST_BM / 32) + (il >> 5)], 1u << (il & 31));
        }
      }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const int wi = wave / Cfg::WGN, wu = wave % Cfg::WGN;
    const int h = lane >> 5, r = lane & 31;
    const int ul = wu * 32 + r;
    const int user = u0 + ul;
    const bool user_ok = user < a.B;
    const float thr = user_ok ? a.tau[user] : INFINITY;
    const unsigned bits = bitmap[ul * (ST_BM / 32) + wi];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int il = (e & 3) + 8 * (e >> 2) + 4 * h;
      const int item = i0 + wi * 32 + il;
      const float x = accs.v[0][0][e];
      const bool cand = user_ok && (x >= thr) && item != 0 && item < a.N && !((bits >> il) & 1u);
      if (__any(cand)) {
        if (cand) {
          const int pos = atomicAdd(&a.cand_cnt[user], 1);
          if (pos < a.cand_cap) {
            a.cand_val[(int64_t)user * a.cand_cap + pos] = x;
            a.cand_idx[(int64_t)user * a.cand_cap + pos] = item;
          }
        }
      }
    }
    __syncthreads();
  }
}

// tau[u] = K-th best value of the sample pass (-inf when the sample held fewer than K unmasked items); cnt[u] = 0
__global__ void __launch_bounds__(256) topk_tau_kernel(const float* __restrict__ sample_val, int B, int K, float* __restrict__ tau,
                                                       int* __restrict__ cnt) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= B) return;
  // the sample pass and the main pass may compute a score with different kernels (f32-input MFMA vs the bf16x3 split): a
  // score equal to the K-th best up to summation order must still pass `x >= tau`, so tau is lowered by 2^-18 |tau|
  // (thousands of ulps: it only admits a few more candidates, the merge below decides)
  const float t = sample_val[(int64_t)u * K + (K - 1)];
  // (a non-finite K-th sample value -- +inf scores, NaN embeddings -- must not become a NaN threshold that rejects everything:
  // -inf admits every item, the candidate buffer then overflows and the host raises THAT)
  tau[u] = isfinite(t) ? t - fabsf(t) * 3.814697265625e-06f : -INFINITY;
  cnt[u] = 0;
}

// one wave per user: the K best of its cnt[u] appended candidates, descending by value, ties by ascending item id
// (independent of the append order).  A count above the capacity sets the status word: candidates were dropped.
// hist_ptr / hist_items (may be null): candidates whose item is in the user's history list are dropped here (value -inf) when
// the pass that appended them did not mask the history itself (score_thresh_p4_kernel).
__global__ void __launch_bounds__(256) topk_cand_merge_kernel(float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                              const int* __restrict__ cnt, int cap, int B, int K,
                                                              int64_t* __restrict__ out_idx, float* __restrict__ out_val,
                                                              int32_t* status, const int* __restrict__ hist_ptr,
                                                              const int64_t* __restrict__ hist_items) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x * 4 + wave;
  if (u >= B) return;
  int n = cnt[u];
  if (n > cap || n < K) {    // dropped candidates, or fewer than K survivors (idx -1 in the output): two different errors the host raises
    if (status && lane == 0) atomicOr(status, n > cap ? PXR_STATUS_TOPK_OVERFLOW : PXR_STATUS_TOPK_UNDERFLOW);
    n = min(n, cap);
  }
  float* pv = cand_val + (int64_t)u * cap;
  const int* pi = cand_idx + (int64_t)u * cap;
  if (hist_ptr != nullptr) {
    const int hb = hist_ptr[u], he = hist_ptr[u + 1];
    if (he > hb) {
      for (int c = lane; c < n; c += 64) {
        const int64_t id = pi[c];
        bool seen = false;
        for (int p = hb; p < he; ++p) seen |= (hist_items[p] == id);
        if (seen) pv[c] = -INFINITY;        // (this lane re-reads pv[c] below: same thread, program order)
      }
    }
  }
  float last_v = INFINITY;
  int last_id = -1;
  for (int k = 0; k < K; ++k) {
    float bv = -INFINITY;
    int bid = 0x7fffffff;
    for (int c = lane; c < n; c += 64) {
      const float v = pv[c];
      const int id = pi[c];
      const bool remaining = (v < last_v) || (v == last_v && id > last_id);
      if (remaining && (v > bv || (v == bv && id < bid))) { bv = v; bid = id; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int oi = __shfl_xor(bid, off, 64);
      if (ov > bv || (ov == bv && oi < bid)) { bv = ov; bid = oi; }
    }
    if (lane == 0) {
      const bool ok = bid != 0x7fffffff;
      out_val[(int64_t)u * K + k] = ok ? bv : -INFINITY;
      out_idx[(int64_t)u * K + k] = ok ? (int64_t)bid : (int64_t)-1;
    }
    last_v = bv;
    last_id = bid;
  }
}

// one wave per user: pick the K best of its n_cand partial candidates, descending
__global__ void __launch_bounds__(256) topk_merge_kernel(const float* __restrict__ part_val,
                                                         const int* __restrict__ part_idx, int B, int n_cand, int K,
                                                         int64_t* __restrict__ out_idx, float* __restrict__ out_val) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x * 4 + wave;
  if (u >= B) return;
  const float* pv = part_val + (int64_t)u * n_cand;
  const int* pi = part_idx + (int64_t)u * n_cand;
  float last_v = INFINITY;
  int last_pos = -1;
  for (int k = 0; k < K; ++k) {
    // best remaining candidate = max over (value, -position) strictly "after" the previously emitted one
    float bv = -INFINITY;
    int bp = 0x7fffffff;
    for (int c = lane; c < n_cand; c += 64) {
      const float v = pv[c];
      const bool remaining = (v < last_v) || (v == last_v && c > last_pos);
      if (remaining && (v > bv || (v == bv && c < bp))) { bv = v; bp = c; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(bv, off, 64);
      const int op = __shfl_xor(bp, off, 64);
      if (ov > bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
    }
    if (lane == 0) {
      const bool ok = bp != 0x7fffffff;
      out_val[(int64_t)u * K + k] = ok ? bv : -INFINITY;
      out_idx[(int64_t)u * K + k] = ok ? (int64_t)pi[bp] : (int64_t)-1;
    }
    last_v = bv;
    last_pos = bp;
  }
}

static int pick_kt(int K) { return K <= 10 ? 10 : (K <= 16 ? 16 : (K <= 32 ? 32 : 0)); }
// selection variant: 2 / 3 = in the accumulator registers (8 / 16 waves per workgroup), 1 = score tile through LDS
// (0 / unset: variant 4 -- two passes around a per-user threshold -- on catalogues of >= 65 536 items, else variant 2)
static int topk_variant() {
  static const int v = getenv("PXR_TOPK_VARIANT") ? atoi(getenv("PXR_TOPK_VARIANT")) : 2;
  return (v == 1 || v == 3) ? v : 2;
}
static int lists_per_split() { return topk_variant() == 2 ? 4 : (topk_variant() == 3 ? 8 : 2); }
static int pick_split(int B, int N) {
  const int row_blocks = (B + ST_BM - 1) / ST_BM;
  const int tiles_n = (N + ST_BN - 1) / ST_BN;
  int s = (512 + row_blocks - 1) / row_blocks;  // ~2 resident workgroups per CU
  if (s > tiles_n) s = tiles_n;
  if (s < 1) s = 1;
  return s;
}

}  // namespace pxr

namespace pxr {
// ---- the threshold pass on PRE-SPLIT operands (planes, gemm_p3.cuh): 256 items x 128 users per tile, 8 waves, and ONE
// LDS-DMA stream over all item tiles of the workgroup (gemm_p3_stream): the ring does not drain between tiles, tile t's
// comparison with tau runs while tile t+1's operands arrive.  Same candidates as score_thresh_kernel (the products are
// those of GEMM mode bf16x3, bit for bit).
constexpr int SP3_BM = 256, SP3_BN = 128;
using Sp3Cfg = P3Cfg<SP3_BM, SP3_BN, 4, 2, 2>;
constexpr int SP3_BITMAP_WORDS = SP3_BN * (SP3_BM / 32);
constexpr int SP3_LDS = Sp3Cfg::LDS_BYTES + 4 * SP3_BITMAP_WORDS + 4 * ST4_HIST_CAP + 16;
static_assert(SP3_LDS <= 160 * 1024, "LDS");

__global__ void __launch_bounds__(Sp3Cfg::NT) score_thresh_p3_kernel(ScoreTopkArgs a, P3Mat table_p, P3Mat users_p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned* bitmap = reinterpret_cast<unsigned*>(smem + Sp3Cfg::LDS_BYTES);
  unsigned* hlist = bitmap + SP3_BITMAP_WORDS;             // (user_local << 20) | (item - first item of the range)
  int* hcount = reinterpret_cast<int*>(hlist + ST4_HIST_CAP);
  constexpr int NT = Sp3Cfg::NT;
  const int tid = threadIdx.x;
  const int t = xcd_remap(blockIdx.x, a.row_blocks * a.n_split);
  const int rb = t % a.row_blocks, sp = t / a.row_blocks;
  const int u0 = rb * SP3_BN;
  const int tiles = (a.N + SP3_BM - 1) / SP3_BM;
  const int per = (tiles + a.n_split - 1) / a.n_split;
  const int tn0 = sp * per, tn1 = min(tiles, tn0 + per);
  if (tn0 >= tn1) return;
  int hb = 0, he = 0;
  if (a.hist_ptr) {
    hb = a.hist_ptr[u0];
    he = a.hist_ptr[min(a.B, u0 + SP3_BN)];
  }
  if (tid == 0) *hcount = 0;
  __syncthreads();
  const int64_t r_lo = (int64_t)tn0 * SP3_BM, r_hi = (int64_t)tn1 * SP3_BM;
  for (int p = hb + tid; p < he; p += NT) {
    const int64_t it = a.hist_items[p];
    if (it >= r_lo && it < r_hi) {
      int lo = u0, hi = min(a.B, u0 + SP3_BN) - 1;         // owner of pair p: hist_ptr[u] <= p < hist_ptr[u+1]
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.hist_ptr[mid] <= p) lo = mid; else hi = mid - 1;
      }
      const int pos = atomicAdd(hcount, 1);
      if (pos < ST4_HIST_CAP) hlist[pos] = ((unsigned)(lo - u0) << 20) | (unsigned)(it - r_lo);
    }
  }
  __syncthreads();
  const int n_hist = *hcount;
  const bool list_ok = n_hist <= ST4_HIST_CAP && (r_hi - r_lo) <= (1ll << 20);
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Sp3Cfg::WGN, wn = wave % Sp3Cfg::WGN;
  const int h = lane >> 5, r = lane & 31;
  float thr[Sp3Cfg::TN];
  bool uok[Sp3Cfg::TN];
#pragma unroll
  for (int j = 0; j < Sp3Cfg::TN; ++j) {
    const int user = u0 + wn * Sp3Cfg::WN + j * 32 + r;
    uok[j] = user < a.B;
    thr[j] = uok[j] ? a.tau[user] : INFINITY;
  }
  gemm_p3_stream<Sp3Cfg>(table_p, users_p, a.D, tn0 * SP3_BM, u0, tn1 - tn0, smem, [&](int tile, const typename Sp3Cfg::Acc& accs) {
    const int i0 = (tn0 + tile) * SP3_BM;
    for (int w = tid; w < SP3_BITMAP_WORDS; w += NT) bitmap[w] = 0u;
    p3_lds_barrier();
    if (list_ok) {
      const unsigned off0 = (unsigned)(i0 - (int)r_lo);
      for (int q = tid; q < n_hist; q += NT) {
        const unsigned e = hlist[q], off = e & 0xFFFFFu;
        if (off >= off0 && off < off0 + SP3_BM) {
          const int il = (int)(off - off0);
          atomicOr(&bitmap[(e >> 20) * (SP3_BM / 32) + (il >> 5)], 1u << (il & 31));
        }
      }
    } else {
      for (int p = hb + tid; p < he; p += NT) {
        const int64_t it = a.hist_items[p];
        if (it >= i0 && it < i0 + SP3_BM) {
          int lo = u0, hi = min(a.B, u0 + SP3_BN) - 1;
          while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (a.hist_ptr[mid] <= p) lo = mid; else hi = mid - 1;
          }
          const int il = (int)(it - i0);
          atomicOr(&bitmap[(lo - u0) * (SP3_BM / 32) + (il >> 5)], 1u << (il & 31));
        }
      }
    }
    p3_lds_barrier();
#pragma unroll
    for (int j = 0; j < Sp3Cfg::TN; ++j) {
      const int ul = wn * Sp3Cfg::WN + j * 32 + r;
      const int user = u0 + ul;
#pragma unroll
      for (int i = 0; i < Sp3Cfg::TM; ++i) {
        const unsigned bits = bitmap[ul * (SP3_BM / 32) + wm * (Sp3Cfg::WM / 32) + i];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int il = (e & 3) + 8 * (e >> 2) + 4 * h;
          const int item = i0 + wm * Sp3Cfg::WM + i * 32 + il;
          const float x = accs.v[i][j][e];
          const bool cand = uok[j] && (x >= thr[j]) && item != 0 && item < a.N && !((bits >> il) & 1u);
          if (__any(cand)) {
            if (cand) {
              const int pos = atomicAdd(&a.cand_cnt[user], 1);
              if (pos < a.cand_cap) {
                a.cand_val[(int64_t)user * a.cand_cap + pos] = x;
                a.cand_idx[(int64_t)user * a.cand_cap + pos] = item;
              }
            }
          }
        }
      }
    }
    p3_lds_barrier();                                      // the bitmap is re-zeroed at the top of the next tile
  });
}

// ---- the same pass on the ping-pong main loop (gemm_p4.cuh; round 4): one k-block stream over all item tiles of the workgroup,
// the comparison with tau runs wave by wave at the head of the next tile's first L segment -- no workgroup barrier, no LDS: the
// (user, item) pairs of the users' HISTORY are no longer masked here (the bitmap needed two barriers per tile) but in the
// candidate merge (topk_cand_merge_kernel drops candidates found in the user's history list): at most `history length` extra
// candidates per user.  Products, k order and accumulator sets are gemm_p3's: the same candidate VALUES as before, bit for bit.
using Sp4Cfg = P4Cfg<SP3_BM, SP3_BN, 4, 2, 3, 3>;
__global__ void __launch_bounds__(Sp4Cfg::NT) score_thresh_p4_kernel(ScoreTopkArgs a, P3Mat table_p, P3Mat users_p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  score_clock_start(a.clk);
  const int t = xcd_remap(blockIdx.x, a.row_blocks * a.n_split);
  const int rb = t % a.row_blocks, sp = t / a.row_blocks;
  const int u0 = rb * SP3_BN;
  const int tiles = (a.N + SP3_BM - 1) / SP3_BM;
  const int per = (tiles + a.n_split - 1) / a.n_split;
  const int tn0 = sp * per, tn1 = min(tiles, tn0 + per);
  if (tn0 >= tn1) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / Sp4Cfg::WGN, wn = wave % Sp4Cfg::WGN;
  gemm_p4_stream<Sp4Cfg>(table_p, users_p, a.D, tn0 * SP3_BM, u0, tn1 - tn0, smem, [&](int tile, const typename Sp4Cfg::Acc& accs) {
    // every per-lane value of the comparison is derived HERE from an opaque lane id: anything hoisted out of the k-block stream
    // would have to live beside 192 accumulator + 48 fragment registers and comes back as scratch traffic inside the K loop
    // (a VMEM wait in front of the DMA ring).  tau is re-read per tile (L2-resident, once per 32 k blocks).
    unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(lane));
    const int h = (int)(lane >> 5), r = (int)(lane & 31u);
    const int i0 = (tn0 + tile) * SP3_BM + wm * Sp4Cfg::WM + 4 * h;
#pragma unroll
    for (int j = 0; j < Sp4Cfg::TN; ++j) {
      const int user = u0 + wn * Sp4Cfg::WN + j * 32 + r;
      const float thr_j = user < a.B ? a.tau[user] : INFINITY;          // users past B never pass
#pragma unroll
      for (int i = 0; i < Sp4Cfg::TM; ++i) {
        // most 32x32 blocks hold no candidate at all: one wave-wide test per block, the per-element work only behind it
        float mx = accs.v[i][j][0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, accs.v[i][j][e]);
        if (!__any(mx >= thr_j)) continue;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int item = i0 + i * 32 + (e & 3) + 8 * (e >> 2);
          const float x = accs.v[i][j][e];
          if (user < a.B && x >= thr_j && item != 0 && item < a.N) {   // (user < B: a +inf score would pass thr = +inf)
            const int pos = atomicAdd(&a.cand_cnt[user], 1);
            if (pos < a.cand_cap) {
              a.cand_val[(int64_t)user * a.cand_cap + pos] = x;
              a.cand_idx[(int64_t)user * a.cand_cap + pos] = item;
            }
          }
        }
      }
    }
  });
  score_clock_stop(a.clk);
}

// ---- "fewer products, exact results" (round 4): the threshold pass only has to DECIDE score >= tau, so it runs on a subset of
// the six bf16 products -- NPL = 2 planes: hi*hi + mid*hi + hi*mid (error <= 2^-13.7 sum|u||v| at K = 512), NPL = 1: hi*hi alone
// (<= 1.03 * 2^-7 sum|u||v|; both derived at spf_margin below) -- with tau lowered by a rigorous per-user bound delta_u = c ||u||_2 max_i ||v_i||_2 (Cauchy-Schwarz; the table's
// largest row norm is computed once per evaluation, pxr_row_norm_max_f32).  The survivors carry APPROXIMATE values; per user the
// few that can still reach the top K (approximate value within 2 delta_u of the K-th best approximate value) are re-scored
// with all six products through the SAME MFMA sequence as the full pass -- same k order, same three accumulator sets, same
// final fold -- so ids AND values are bit-identical to the six-product schedule (tests/test_gpu_configs.py).
// Tile: 256 items x 256 users, one accumulator set (128 registers), ping-pong k-block stream (gemm_p4_stream).
constexpr int SPF_BM = 256, SPF_BN = 256;
// The margin constants: |six-product value - reduced-product value| <= c * sum_k |u_k| |v_k| <= c ||u||_2 ||v||_2.
// With x = hi + mid + lo (bf16 RNE: |x - hi| <= 2^-8 |x|, so |mid| <= 2^-8 (1 + 2^-8) |x|, |lo| <= 2^-16 |x|, |hi| <= (1 + 2^-8) |x|):
//   one product:    dropped mid*hi + hi*mid + hi*lo + lo*hi + mid*mid <= (2 * 2^-8 (1 + 2^-8)^2 + 3.03 * 2^-16) |u||v| = 1.013 * 2^-7 |u||v|
//   three products: dropped hi*lo + lo*hi + mid*mid <= 3.03 * 2^-16 |u||v| = 2^-14.4 |u||v|
// plus the two sums' own fp32 accumulation error, <= (MFMA accumulations x <= 4 internal roundings each) * 2^-24 sum|u||v|: 2^-16.9
// for the three-set six-product sum, 2^-15.4 for 96 accumulations into one set (K = 512; both grow with K / 16, hence the K factor
// below).  c = the sum with a 1.5x margin, evaluated for the reduction length at hand.
__host__ __device__ inline float spf_margin(int products, int D) {
  const float acc = (D / 16.0f) * (4.0f * 4.0f + 4.0f * 3.0f) * 5.9604645e-08f;      // (six-product + three-product chains) * 2^-24
  const float dropped = products == 3 ? 3.03f * 1.52587890625e-05f : 1.013f * 7.8125e-03f + 3.03f * 1.52587890625e-05f;
  return 1.5f * (dropped + acc);
}
template <int NPL, int NS>
using SpfCfg = P4Cfg<SPF_BM, SPF_BN, 4, 2, NS, 1, 0, NPL>;

// NS = ring slots: the fewer products a k block carries, the shorter its slot-times and the more k blocks must be in flight to
// cover the HBM latency of the table stream (NPL = 2: 32 KB per slot; NPL = 1: 16 KB)
template <int NPL, int NS>
__global__ void __launch_bounds__(512) score_thresh_fast_kernel(ScoreTopkArgs a, P3Mat table_p, P3Mat users_p) {
  using Cfg = SpfCfg<NPL, NS>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  score_clock_start(a.clk);
  const int t = xcd_remap(blockIdx.x, a.row_blocks * a.n_split);
  const int rb = t % a.row_blocks, sp = t / a.row_blocks;
  const int u0 = rb * SPF_BN;
  const int tiles = (a.N + SPF_BM - 1) / SPF_BM;
  const int per = (tiles + a.n_split - 1) / a.n_split;
  const int tn0 = sp * per, tn1 = min(tiles, tn0 + per);
  if (tn0 >= tn1) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  // one accumulator set leaves registers for the lane's thresholds (a global load per tile in front of the comparison would be a
  // round trip on the critical path of BOTH wave groups: the other group waits at the barrier meanwhile)
  float thr[Cfg::TN];
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const int uj = u0 + wn * Cfg::WN + j * 32 + (int)(threadIdx.x & 31u);
    thr[j] = uj < a.B ? a.tau[uj] : INFINITY;
  }
  // Survivors of a tile are STASHED (up to two per lane and column block) and their slots reserved with one returning atomic per
  // (lane, block) whose result is not consumed before the NEXT tile's comparison, a whole K loop later: the atomic's round trip
  // (~2 us, which both wave groups would otherwise spend at the barrier) disappears behind the stream.  The atomic is issued from
  // inline asm so that the compiler neither waits for it (it would drain the DMA ring with a vmcnt(0)) nor counts it: by the
  // time it is used, at least one counted wait of the stream (all but the youngest `keep` operations retired) lies behind it --
  // for reductions shorter than the ring that is forced below.  A lane with more than two hits in a block (a 1e-5 event on the
  // north-star shape) sends its wave through the blocking path for that block.
  int pn[Cfg::TN], pbase[Cfg::TN], pit[Cfg::TN][2];
  float pval[Cfg::TN][2];
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) { pn[j] = 0; pbase[j] = 0; pit[j][0] = pit[j][1] = 0; pval[j][0] = pval[j][1] = 0.f; }
  const bool short_k = (a.D / 16) <= NS + 1;
  auto flush = [&]() {
    if (short_k) p3_wait_vm<0>();
    const int r = (int)(threadIdx.x & 31u);
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      if (pn[j] > 0) {
        const int user = u0 + wn * Cfg::WN + j * 32 + r;
#pragma unroll
        for (int k = 0; k < 2; ++k)
          if (k < pn[j] && pbase[j] + k < a.cand_cap) {
            a.cand_val[(int64_t)user * a.cand_cap + pbase[j] + k] = pval[j][k];
            a.cand_idx[(int64_t)user * a.cand_cap + pbase[j] + k] = pit[j][k];
          }
      }
      pn[j] = 0;
    }
  };
  gemm_p4_stream<Cfg>(table_p, users_p, a.D, tn0 * SPF_BM, u0, tn1 - tn0, smem, [&](int tile, const typename Cfg::Acc& accs) {
    flush();                                              // the previous tile's survivors: their slots were reserved a K loop ago
    const int lane = (int)(threadIdx.x & 63u);
    const int h = lane >> 5, r = lane & 31;
    const int i0 = (tn0 + tile) * SPF_BM + wm * Cfg::WM + 4 * h;
    int user[Cfg::TN], nh[Cfg::TN];
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) user[j] = u0 + wn * Cfg::WN + j * 32 + r;
    unsigned hot = 0;                                     // bit (j * TM + i): block (i, j) holds a hit in SOME lane (wave-uniform)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      nh[j] = 0;
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        float mx = accs.v[i][j][0];
#pragma unroll
        for (int e = 1; e < 16; ++e) mx = fmaxf(mx, accs.v[i][j][e]);
        if (!__any(mx >= thr[j])) continue;
        hot |= 1u << (j * Cfg::TM + i);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int item = i0 + i * 32 + (e & 3) + 8 * (e >> 2);
          const float x = accs.v[i][j][e];
          if (user[j] < a.B && x >= thr[j] && item != 0 && item < a.N) {
            if (nh[j] == 0) { pval[j][0] = x; pit[j][0] = item; }
            else if (nh[j] == 1) { pval[j][1] = x; pit[j][1] = item; }
            ++nh[j];
          }
        }
      }
    }
    if (hot == 0) return;
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      if (!__any(nh[j] > 0)) continue;
      if (!__any(nh[j] > 2)) {
        if (nh[j] > 0) {
          int* cp = &a.cand_cnt[user[j]];
          asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(pbase[j]) : "v"(cp), "v"(nh[j]) : "memory");
        }
        pn[j] = nh[j];
      } else {
        // blocking path: reserve, then walk the block column again and write
        int base = nh[j] > 0 ? atomicAdd(&a.cand_cnt[user[j]], nh[j]) : 0;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
          if (!((hot >> (j * Cfg::TM + i)) & 1u)) continue;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int item = i0 + i * 32 + (e & 3) + 8 * (e >> 2);
            const float x = accs.v[i][j][e];
            if (user[j] < a.B && x >= thr[j] && item != 0 && item < a.N) {
              if (base < a.cand_cap) {
                a.cand_val[(int64_t)user[j] * a.cand_cap + base] = x;
                a.cand_idx[(int64_t)user[j] * a.cand_cap + base] = item;
              }
              ++base;
            }
          }
        }
        pn[j] = 0;
      }
    }
  });
  p3_wait_vm<0>();                                        // the last tile's reservations
  flush();
  score_clock_stop(a.clk);
}

// tau'[u] = tau[u] - delta[u]: tau as topk_tau_kernel, delta[u] = c * ||users[u]||_2 * vmax[0]; cnt[u] = 0
__global__ void __launch_bounds__(256) topk_tau_fast_kernel(const float* __restrict__ sample_val, int B, int K, const float* __restrict__ users,
                                                            int64_t ld_users, int D, const float* __restrict__ vmax, float c,
                                                            float* __restrict__ tau, float* __restrict__ delta, int* __restrict__ cnt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x * 4 + wave;
  if (u >= B) return;
  float ss = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float x = users[(int64_t)u * ld_users + k];
    ss += x * x;
  }
  ss = wave_sum(ss);
  if (lane == 0) {
    const float d = c * sqrtf(ss) * vmax[0];
    const float t = sample_val[(int64_t)u * K + (K - 1)];
    tau[u] = (isfinite(t) && isfinite(d)) ? (t - fabsf(t) * 3.814697265625e-06f) - d : -INFINITY;
    delta[u] = d;
    cnt[u] = 0;
  }
}

// largest L2 norm of a row of x[rows, cols] -> out[0] (must be zeroed before the launch); one wave per row
__global__ void __launch_bounds__(256) row_norm_max_kernel(const float* __restrict__ x, int64_t rows, int cols, int64_t ldx, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float best = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    float ss = 0.f;
    for (int k = lane * 4; k < cols; k += 256) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + k);
      ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    best = fmaxf(best, sqrtf(wave_sum(ss)));
  }
  if (lane == 0) atomicMax(reinterpret_cast<int*>(out), __float_as_int(best));     // non-negative floats order like their bits
}

// One wave per user: drop the history, find the K-th best APPROXIMATE value, re-score every candidate within 2 delta of it with the
// six-product MFMA sequence of the full pass (32 candidates per group: A = their table rows gathered from the planes, B = the
// user's row in every column), emit the K best by (exact value, ascending id).  The candidates live in registers (RS_CPL per lane;
// a user with more than 64 * RS_CPL of them takes the same steps through global memory), the user's planes in LDS.
constexpr int RS_MAX = 256;                 // shortlist slots per user
constexpr int RS_CPL = 32;                  // candidates per lane held in registers
constexpr int RS_DMAX = 1024;               // largest D whose user row is staged in LDS (3 planes x 2 B x D x 4 users = 24 KB)

// best remaining (value, id) of a wave's candidates after (last_v, last_id), descending by value, ascending by id
__device__ __forceinline__ void rs_wave_best(float& bv, int& bid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(bv, off, 64);
    const int oi = __shfl_xor(bid, off, 64);
    if (ov > bv || (ov == bv && oi < bid)) { bv = ov; bid = oi; }
  }
}

__global__ void __launch_bounds__(256) topk_rescore_kernel(float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                           const int* __restrict__ cnt, int cap, int B, int K,
                                                           const float* __restrict__ delta, P3Mat table_p, P3Mat users_p, int D,
                                                           int64_t* __restrict__ out_idx, float* __restrict__ out_val, int32_t* status,
                                                           const int* __restrict__ hist_ptr, const int64_t* __restrict__ hist_items) {
  __shared__ int s_id[4][RS_MAX];
  __shared__ float s_val[4][RS_MAX];
  __shared__ __attribute__((aligned(16))) __bf16 s_user[4][3][RS_DMAX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x * 4 + wave;
  if (u >= B) return;                      // (wave-uniform; no workgroup barrier below)
  int n = cnt[u];
  bool bad = n > cap;
  if (n < K && status && lane == 0) atomicOr(status, PXR_STATUS_TOPK_UNDERFLOW);
  n = min(n, cap);
  float* pv = cand_val + (int64_t)u * cap;
  const int* pi = cand_idx + (int64_t)u * cap;
  const bool lds_user = D <= RS_DMAX;
  // the user's planes, k-contiguous per plane: chunk c (8 values) of the row at s_user[.][p][8 c]
  if (lds_user) {
    const int uswz = (u >> 2) & 3;
    for (int c = lane; c < (D >> 3) * 3; c += 64) {
      const int p = c / (D >> 3), ck = c - p * (D >> 3);            // chunk ck = 8 consecutive k of plane p
      const int kt = ck >> 2, ch = ck & 3;
      const p3_bf16x8 v = *reinterpret_cast<const p3_bf16x8*>(users_p.p + p * users_p.ps + ((int64_t)kt * users_p.pr + u) * 32 + ((ch ^ uswz) << 3));
      *reinterpret_cast<p3_bf16x8*>(&s_user[wave][p][ck << 3]) = v;
    }
  }
  const bool inreg = n <= 64 * RS_CPL;     // wave-uniform
  float cv[RS_CPL];
  int ci[RS_CPL];
  const int hb = hist_ptr ? hist_ptr[u] : 0, he = hist_ptr ? hist_ptr[u + 1] : 0;
  if (inreg) {
#pragma unroll
    for (int q = 0; q < RS_CPL; ++q) {
      const int c = q * 64 + lane;
      const bool ok = c < n;
      cv[q] = ok ? pv[c] : -INFINITY;
      ci[q] = ok ? pi[c] : 0x7fffffff;
    }
    for (int p = hb; p < he; ++p) {
      const int hid = (int)hist_items[p];
#pragma unroll
      for (int q = 0; q < RS_CPL; ++q)
        if (ci[q] == hid) cv[q] = -INFINITY;
    }
  } else if (he > hb) {
    for (int c = lane; c < n; c += 64) {
      const int64_t id = pi[c];
      bool seen = false;
      for (int p = hb; p < he; ++p) seen |= (hist_items[p] == id);
      if (seen) pv[c] = -INFINITY;
    }
  }
  // K-th best approximate value (ties by id, as the final selection)
  float last_v = INFINITY;
  int last_id = -1;
  for (int k = 0; k < K; ++k) {
    float bv = -INFINITY;
    int bid = 0x7fffffff;
    if (inreg) {
#pragma unroll
      for (int q = 0; q < RS_CPL; ++q) {
        const float v = cv[q];
        const int id = ci[q];
        const bool remaining = (v < last_v) || (v == last_v && id > last_id);
        if (remaining && (v > bv || (v == bv && id < bid))) { bv = v; bid = id; }
      }
    } else {
      for (int c = lane; c < n; c += 64) {
        const float v = pv[c];
        const int id = pi[c];
        const bool remaining = (v < last_v) || (v == last_v && id > last_id);
        if (remaining && (v > bv || (v == bv && id < bid))) { bv = v; bid = id; }
      }
    }
    rs_wave_best(bv, bid);
    last_v = bv;
    last_id = bid;
  }
  const float cut = last_v - 2.0f * delta[u];          // (K-th best approx = -inf when fewer than K candidates: everything stays)
  if (bad && status && lane == 0) atomicOr(status, PXR_STATUS_TOPK_OVERFLOW);      // (only a full CANDIDATE buffer loses items)
  // The shortlist -- candidates whose approximate value is within 2 delta of the K-th best -- is re-scored exactly in CHUNKS of
  // RS_MAX slots: a catalogue with a crowd of (near-)ties at a user's K-th score (duplicated items, identical images) puts
  // thousands of candidates on the shortlist; round 4 raised the overflow error beyond 256 of them (found by
  // tests/test_gpu_stress.py).  Chunk c holds the K winners so far in its first slots and the next RS_MAX - K shortlist entries
  // behind them; everything in the chunk is re-scored (the winners' values come out the same bits again) and the K best by
  // (exact value, ascending id) are kept.  One chunk in all but pathological catalogues.
  const int r = lane & 31, h = lane >> 5;
  const __bf16* ub = users_p.p + ((int64_t)u << 5);
  const int uswz = (u >> 2) & 3;
  const int nkt = D >> 5;
  __shared__ int w_id[4][32];
  int carry = 0, base = 0, total = 0;
  do {
    const int room = RS_MAX - carry;
    int run = 0;                                         // running index of the shortlist entry (candidate order)
    if (inreg) {
#pragma unroll
      for (int q = 0; q < RS_CPL; ++q) {
        const bool in = cv[q] >= cut && cv[q] > -INFINITY;
        const unsigned long long m = __ballot(in);
        if (in) {
          const int pos = run + __popcll(m & ((1ull << lane) - 1ull));
          if (pos >= base && pos < base + room) s_id[wave][carry + pos - base] = ci[q];
        }
        run += __popcll(m);
      }
    } else {
      for (int c0 = 0; c0 < n; c0 += 64) {
        const int c = c0 + lane;
        const bool in = c < n && pv[c] >= cut && pv[c] > -INFINITY;
        const unsigned long long m = __ballot(in);
        if (in) {
          const int pos = run + __popcll(m & ((1ull << lane) - 1ull));
          if (pos >= base && pos < base + room) s_id[wave][carry + pos - base] = pi[c];
        }
        run += __popcll(m);
      }
    }
    total = run;
    const int ns = carry + min(room, total - base);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // exact re-scoring, 32 shortlist entries per group
    for (int g0 = 0; g0 < ns; g0 += 32) {
      const int item = (g0 + r < ns) ? s_id[wave][g0 + r] : 1;
      const __bf16* tb = table_p.p + ((int64_t)item << 5);
      const int iswz = (item >> 2) & 3;
      f32x16 accs, accm, accl;
#pragma unroll
      for (int e = 0; e < 16; ++e) { accs[e] = 0.f; accm[e] = 0.f; accl[e] = 0.f; }
#pragma unroll 4
      for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          p3_bf16x8 a[3], b[3];
          const int64_t ao = (int64_t)kt * table_p.pr * 32 + ((((kb << 1) | h) ^ iswz) << 3);
#pragma unroll
          for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const p3_bf16x8*>(tb + p * table_p.ps + ao);
          if (lds_user) {
#pragma unroll
            for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const p3_bf16x8*>(&s_user[wave][p][(kt << 5) + (((kb << 1) | h) << 3)]);
          } else {
            const int64_t bo = (int64_t)kt * users_p.pr * 32 + ((((kb << 1) | h) ^ uswz) << 3);
#pragma unroll
            for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const p3_bf16x8*>(ub + p * users_p.ps + bo);
          }
          accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], accl, 0, 0, 0);
          accm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], accm, 0, 0, 0);
          accs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], accs, 0, 0, 0);
          accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], accl, 0, 0, 0);
          accm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], accm, 0, 0, 0);
          accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], accl, 0, 0, 0);
        }
      }
      if (r == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          if (g0 + row < ns) s_val[wave][g0 + row] = accs[e] + (accm[e] + accl[e]);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the K best of the chunk by exact value
    last_v = INFINITY;
    last_id = -1;
    int kept = 0;
    for (int k = 0; k < K; ++k) {
      float bv = -INFINITY;
      int bid = 0x7fffffff;
      for (int c = lane; c < ns; c += 64) {
        const float v = s_val[wave][c];
        const int id = s_id[wave][c];
        const bool remaining = (v < last_v) || (v == last_v && id > last_id);
        if (remaining && (v > bv || (v == bv && id < bid))) { bv = v; bid = id; }
      }
      rs_wave_best(bv, bid);
      const bool ok = bid != 0x7fffffff;
      if (lane == 0) {
        out_val[(int64_t)u * K + k] = ok ? bv : -INFINITY;
        out_idx[(int64_t)u * K + k] = ok ? (int64_t)bid : (int64_t)-1;
        if (ok) w_id[wave][k] = bid;
      }
      kept += ok ? 1 : 0;
      last_v = bv;
      last_id = bid;
    }
    base += room;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (base < total) {                                  // more shortlist entries: the winners so far lead the next chunk
      if (lane < kept) s_id[wave][lane] = w_id[wave][lane];
      carry = kept;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } while (base < total);
}
}  // namespace pxr

using namespace pxr;

// variant 4 (two passes around a threshold) pays off on big catalogues; small ones keep the register lists
constexpr int ST4_MIN_TILES = 512;       // >= 65 536 items
constexpr int ST4_CAP = 4096;            // candidates per user
// sample = every s-th item tile; expected candidates per user ~ s * K (640 / 1 024 / 1 024 of the 4 096 slots; s = 128 measured slower: more appends)
static int st4_stride(int kt) { return kt <= 16 ? 64 : 32; }
static bool use_thresh(int N) {
  static const int env = getenv("PXR_TOPK_VARIANT") ? atoi(getenv("PXR_TOPK_VARIANT")) : 0;
  const int tiles_n = (N + ST_BN - 1) / ST_BN;
  return (env == 0 || env == 4) && tiles_n >= ST4_MIN_TILES;
}
static int64_t a256(int64_t x) { return (x + 255) & ~(int64_t)255; }
static int sample_splits(int N, int kt) { return (((N + ST_BN - 1) / ST_BN) + st4_stride(kt) - 1) / st4_stride(kt); }
static int64_t lists_bytes(int B, int N, int kt, int lists) { return a256((int64_t)B * sample_splits(N, kt) * lists * kt * 4) * 2; }

static unsigned long long* g_score_clk = nullptr;
// Measurement hook: register two uint64 in DEVICE memory (NULL unregisters).  Every later main-pass launch of the fused scoring
// (score_thresh_fast / score_thresh_p4 kernels; the six-product score_thresh_p3_kernel has no register to spare for the hook: it ran 18 % slower with it) adds the shader-clock cycles and the 100 MHz reference ticks its workgroup 0 lived through
// (score_clock_start / _stop above): sustained clock [GHz] = clk[0] / clk[1] * 0.1.  The caller zeroes the buffer.  Process-wide, not stream-ordered.
extern "C" int pxr_score_topk_clock_out(uint64_t* clk2) {
  g_score_clk = reinterpret_cast<unsigned long long*>(clk2);
  return PXR_OK;
}

extern "C" int64_t pxr_score_topk_ws_bytes(int B, int N, int K) {
  const int kt = pick_kt(K);
  if (kt == 0) return -1;
  if (use_thresh(N))   // sample lists (variant-1 layout) | sample top-K (idx, val) | tau | cnt | candidate values | ids
    return lists_bytes(B, N, kt, 2) + a256((int64_t)B * K * 8) + a256((int64_t)B * K * 4) + 3 * a256((int64_t)B * 4) +
           2 * a256((int64_t)B * ST4_CAP * 4) + 256;                     // (+ delta, the reduced-product schedule's margin)
  const int64_t cand = (int64_t)pick_split(B, N) * lists_per_split() * kt;
  return (int64_t)B * cand * 8 + 256;
}

static int score_topk_thresh(ScoreTopkArgs a, int K, int kt, int64_t* topk_idx, float* topk_val, void* ws, hipStream_t st,
                             const P3Mat* table_p = nullptr, const P3Mat* users_p = nullptr, int products = 6,
                             const float* table_norm_max = nullptr) {
  char* w = (char*)ws;
  const int64_t lb = lists_bytes(a.B, a.N, kt, 2) / 2;
  a.part_val = (float*)w;              a.part_idx = (int*)(w + lb);         w += 2 * lb;
  int64_t* s_idx = (int64_t*)w;        w += a256((int64_t)a.B * K * 8);
  float* s_val = (float*)w;            w += a256((int64_t)a.B * K * 4);
  float* tau = (float*)w;              w += a256((int64_t)a.B * 4);
  int* cnt = (int*)w;                  w += a256((int64_t)a.B * 4);
  float* cval = (float*)w;             w += a256((int64_t)a.B * ST4_CAP * 4);
  int* cidx = (int*)w;                 w += a256((int64_t)a.B * ST4_CAP * 4);
  float* delta = (float*)w;
  const bool fast = table_p != nullptr && products != 6 && table_norm_max != nullptr;
  const dim3 grid(a.row_blocks * a.n_split);
  // pass 1: lists over a SAMPLE of the tiles (one workgroup per sampled tile and row block) -> K-th best value per user
  const int n_split_full = a.n_split;
  a.n_split = sample_splits(a.N, kt);
  a.tile_stride = (a.tiles_n + a.n_split - 1) / a.n_split;     // == the kernel's tiles-per-split: exactly one tile each
  const dim3 grid_s(a.row_blocks * a.n_split);
  switch (kt) {
    case 10: hipLaunchKernelGGL(score_topk_kernel<10>, grid_s, dim3(GEMM_THREADS), 0, st, a); break;
    case 16: hipLaunchKernelGGL(score_topk_kernel<16>, grid_s, dim3(GEMM_THREADS), 0, st, a); break;
    default: hipLaunchKernelGGL(score_topk_kernel<32>, grid_s, dim3(GEMM_THREADS), 0, st, a); break;
  }
  const int64_t cand = (int64_t)a.n_split * 2 * kt;
  hipLaunchKernelGGL(topk_merge_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, (const float*)a.part_val,
                     (const int*)a.part_idx, a.B, (int)cand, K, s_idx, s_val);
  if (fast)      // tau lowered by the rigorous margin of the reduced-product pass (see score_thresh_fast_kernel)
    hipLaunchKernelGGL(topk_tau_fast_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, (const float*)s_val, a.B, K, a.users, a.ld_users, a.D,
                       table_norm_max, spf_margin(products, a.D), tau, delta, cnt);
  else
    hipLaunchKernelGGL(topk_tau_kernel, dim3((a.B + 255) / 256), dim3(256), 0, st, (const float*)s_val, a.B, K, tau, cnt);
  // pass 2: every tile at full GEMM speed, survivors appended
  a.n_split = n_split_full;
  a.tile_stride = 1; a.tau = tau; a.cand_cnt = cnt; a.cand_val = cval; a.cand_idx = cidx; a.cand_cap = ST4_CAP;
  bool hist_in_merge = false;
  if (fast) {
    // ring depth: measuring knob PXR_TOPK_NS (defaults = the measured best, tools/eval_bench.py)
    const int env_ns = getenv("PXR_TOPK_NS") ? atoi(getenv("PXR_TOPK_NS")) : 0;
    const int tiles256 = (a.N + SPF_BM - 1) / SPF_BM;
    a.row_blocks = (a.B + SPF_BN - 1) / SPF_BN;
    int nsp = (256 + a.row_blocks - 1) / a.row_blocks;
    if (nsp > tiles256) nsp = tiles256;
    if (nsp < 1) nsp = 1;
    a.n_split = nsp;
    const dim3 fgrid(a.row_blocks * a.n_split);
#define PXR_FAST(NPL_, NS_)                                                                                                   \
  do {                                                                                                                        \
    auto kern = score_thresh_fast_kernel<NPL_, NS_>;                                                                          \
    constexpr int lds = SpfCfg<NPL_, NS_>::RING_BYTES;                                                                        \
    static bool attr = false;                                                                                                 \
    if (!attr) {                                                                                                              \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { \
        (void)hipGetLastError();                                                                                              \
        pxr_set_error("pxr_score_topk_fast_f32: cannot reserve %d bytes of LDS", lds);                                        \
        return PXR_ERR_LAUNCH;                                                                                                \
      }                                                                                                                       \
      attr = true;                                                                                                            \
    }                                                                                                                         \
    hipLaunchKernelGGL(kern, fgrid, dim3(512), lds, st, a, *table_p, *users_p);                                               \
  } while (0)
    if (products == 3) {
      if (env_ns == 3) PXR_FAST(2, 3);
      else if (env_ns == 5) PXR_FAST(2, 5);
      else PXR_FAST(2, 4);
    } else {
      if (env_ns == 3) PXR_FAST(1, 3);
      else if (env_ns == 4) PXR_FAST(1, 4);
      else if (env_ns == 6) PXR_FAST(1, 6);
      else PXR_FAST(1, 8);
    }
#undef PXR_FAST
    int rc = pxr_check_launch("pxr_score_topk_fast_f32(threshold pass)");
    if (rc) return rc;
    hipLaunchKernelGGL(topk_rescore_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, cval, (const int*)cidx, (const int*)cnt, ST4_CAP, a.B, K,
                       (const float*)delta, *table_p, *users_p, a.D, topk_idx, topk_val, pxr_status_word(), a.hist_ptr, a.hist_items);
    return pxr_check_launch("pxr_score_topk_fast_f32(re-scoring)");
  }
  if (table_p != nullptr) {
    // planes: 256-item tiles, one workgroup per CU, every CU the same number of tiles.  Default: round 3's lockstep stream with
    // the history bitmap; PXR_SCORE_P4=1 (read per call: a measuring / test knob): the ping-pong stream (score_thresh_p4_kernel),
    // measured equal on the 1024 x 400 001 x 512 evaluation (2.146 vs 2.137 ms: 32 k blocks per tile are too few to amortise
    // the per-tile comparison, and the history scan moves into the merge: + 50 us)
    const int env_p4 = getenv("PXR_SCORE_P4") ? atoi(getenv("PXR_SCORE_P4")) : 0;
    const void* kern = env_p4 ? reinterpret_cast<const void*>(score_thresh_p4_kernel) : reinterpret_cast<const void*>(score_thresh_p3_kernel);
    const int lds = env_p4 ? Sp4Cfg::LDS_BYTES : SP3_LDS;
    static bool attr_set[2] = {false, false};
    if (!attr_set[env_p4 ? 1 : 0]) {
      if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        (void)hipGetLastError();
        pxr_set_error("pxr_score_topk_planes_f32: cannot reserve %d bytes of LDS", lds);
        return PXR_ERR_LAUNCH;
      }
      attr_set[env_p4 ? 1 : 0] = true;
    }
    const int tiles256 = (a.N + SP3_BM - 1) / SP3_BM;
    a.row_blocks = (a.B + SP3_BN - 1) / SP3_BN;
    int ns = (256 + a.row_blocks - 1) / a.row_blocks;
    if (ns > tiles256) ns = tiles256;
    if (ns < 1) ns = 1;
    a.n_split = ns;
    if (env_p4) {
      hist_in_merge = a.hist_ptr != nullptr;
      hipLaunchKernelGGL(score_thresh_p4_kernel, dim3(a.row_blocks * a.n_split), dim3(Sp4Cfg::NT), lds, st, a, *table_p, *users_p);
    } else {
      hipLaunchKernelGGL(score_thresh_p3_kernel, dim3(a.row_blocks * a.n_split), dim3(Sp3Cfg::NT), lds, st, a, *table_p, *users_p);
    }
  } else if (pxr_get_gemm_mode() && a.D % 4 == 0)
    hipLaunchKernelGGL(score_thresh_kernel<1>, grid, dim3(1024), 0, st, a);
  else
    hipLaunchKernelGGL(score_thresh_kernel<0>, grid, dim3(1024), 0, st, a);
  int rc = pxr_check_launch("pxr_score_topk_f32(threshold pass)");
  if (rc) return rc;
  // pass 3: K best candidates per user
  hipLaunchKernelGGL(topk_cand_merge_kernel, dim3((a.B + 3) / 4), dim3(256), 0, st, cval, (const int*)cidx,
                     (const int*)cnt, ST4_CAP, a.B, K, topk_idx, topk_val, pxr_status_word(),
                     hist_in_merge ? a.hist_ptr : (const int*)nullptr, hist_in_merge ? a.hist_items : (const int64_t*)nullptr);
  return pxr_check_launch("pxr_score_topk_f32(candidate merge)");
}

extern "C" int pxr_score_topk_planes_f32(const float* users, int64_t ld_users, int B, const float* table, int N, int D,
                                         const void* users_planes, int64_t users_plane_stride, int64_t users_panel_rows,
                                         const void* table_planes, int64_t table_plane_stride, int64_t table_panel_rows,
                                         const int32_t* hist_ptr, const int64_t* hist_items, int K, int64_t* topk_idx,
                                         float* topk_val, void* ws, int64_t ws_bytes, void* stream);

// Top-K item ids / scores per user of  users[B,D] x table[N,D]^T  with item 0 and each user's history masked.
// hist_ptr int32 [B+1] / hist_items int64: CSR of the (history_u, history_i) pairs of seq_eval_collate (may be NULL).
extern "C" int pxr_score_topk_f32(const float* users, int64_t ld_users, int B, const float* table, int N, int D,
                                  const int32_t* hist_ptr, const int64_t* hist_items, int K, int64_t* topk_idx,
                                  float* topk_val, void* ws, int64_t ws_bytes, void* stream) {
  return pxr_score_topk_planes_f32(users, ld_users, B, table, N, D, nullptr, 0, 0, nullptr, 0, 0, hist_ptr, hist_items, K, topk_idx,
                                   topk_val, ws, ws_bytes, stream);
}
extern "C" int pxr_score_topk_fast_f32(const float* users, int64_t ld_users, int B, const float* table, int N, int D,
                                       const void* users_planes, int64_t users_plane_stride, int64_t users_panel_rows,
                                       const void* table_planes, int64_t table_plane_stride, int64_t table_panel_rows,
                                       const float* table_row_norm_max, int products, const int32_t* hist_ptr,
                                       const int64_t* hist_items, int K, int64_t* topk_idx, float* topk_val, void* ws, int64_t ws_bytes,
                                       void* stream);

// The same with the operands ALSO given as planes (include/pxr.h "pre-split operands"; both NULL: the plain function): on
// catalogues that take the threshold schedule the main pass -- every item tile -- runs on the planes (gemm_p3_stream); the
// sample pass keeps the fp32 operands.  The table's planes are made once per evaluation (pxr_split_planes_f32).
extern "C" int pxr_score_topk_planes_f32(const float* users, int64_t ld_users, int B, const float* table, int N, int D,
                                         const void* users_planes, int64_t users_plane_stride, int64_t users_panel_rows,
                                         const void* table_planes, int64_t table_plane_stride, int64_t table_panel_rows,
                                         const int32_t* hist_ptr, const int64_t* hist_items, int K, int64_t* topk_idx,
                                         float* topk_val, void* ws, int64_t ws_bytes, void* stream) {
  return pxr_score_topk_fast_f32(users, ld_users, B, table, N, D, users_planes, users_plane_stride, users_panel_rows, table_planes,
                                 table_plane_stride, table_panel_rows, nullptr, 6, hist_ptr, hist_items, K, topk_idx, topk_val, ws,
                                 ws_bytes, stream);
}
// ... and with the threshold pass on `products` = 3 or 1 of the six bf16 products (6: the function above): needs the planes and
// table_row_norm_max (device pointer to max_i ||table[i]||_2, pxr_row_norm_max_f32, once per evaluation).  Results are the
// six-product schedule's, bit for bit (the survivors are re-scored with all six products).
extern "C" int pxr_score_topk_fast_f32(const float* users, int64_t ld_users, int B, const float* table, int N, int D,
                                       const void* users_planes, int64_t users_plane_stride, int64_t users_panel_rows,
                                       const void* table_planes, int64_t table_plane_stride, int64_t table_panel_rows,
                                       const float* table_row_norm_max, int products, const int32_t* hist_ptr,
                                       const int64_t* hist_items, int K, int64_t* topk_idx, float* topk_val, void* ws, int64_t ws_bytes,
                                       void* stream) {
  PXR_REQUIRE(products == 6 || products == 3 || products == 1, "pxr_score_topk_fast_f32: products must be 6, 3 or 1");
  PXR_REQUIRE(products == 6 || (table_row_norm_max && users_planes && table_planes),
              "pxr_score_topk_fast_f32: the reduced-product pass needs both planes and the table's largest row norm");
  PXR_REQUIRE(users && table && topk_idx && topk_val && ws, "pxr_score_topk_f32: null pointer");
  PXR_REQUIRE((users_planes == nullptr) == (table_planes == nullptr), "pxr_score_topk_planes_f32: give both planes or neither");
  PXR_REQUIRE(!users_planes || (D % 32 == 0 && p3_mat_ok(users_planes, users_plane_stride, users_panel_rows, B, D) &&
                                p3_mat_ok(table_planes, table_plane_stride, table_panel_rows, N, D) &&
                                table_plane_stride * 6 < 0x7FFFFFF0ll),
              "pxr_score_topk_planes_f32: bad planes (D %% 32 == 0, three planes < 2 GiB)");
  PXR_REQUIRE(B > 0 && N > 0 && D > 0 && D % 4 == 0 && ld_users % 4 == 0, "pxr_score_topk_f32: bad shape");
  PXR_REQUIRE(!hist_ptr || hist_items, "pxr_score_topk_f32: hist_ptr without hist_items");
  const int kt = pick_kt(K);
  PXR_REQUIRE(K >= 1 && kt != 0, "pxr_score_topk_f32: K must be in [1, 32]");
  ScoreTopkArgs a{};
  a.users = users; a.ld_users = ld_users; a.table = table; a.hist_ptr = hist_ptr; a.hist_items = hist_items;
  a.B = B; a.N = N; a.D = D;
  a.clk = g_score_clk;
  a.row_blocks = (B + ST_BM - 1) / ST_BM;
  a.tiles_n = (N + ST_BN - 1) / ST_BN;
  a.n_split = pick_split(B, N);
  a.tile_stride = 1;
  if (pxr_score_topk_ws_bytes(B, N, K) > ws_bytes) { pxr_set_error("pxr_score_topk_f32: workspace too small"); return PXR_ERR_WORKSPACE; }
  if (use_thresh(N)) {
    if (users_planes && pxr_get_gemm_mode()) {
      const P3Mat tp{reinterpret_cast<__bf16*>(const_cast<void*>(table_planes)), table_plane_stride, table_panel_rows};
      const P3Mat up{reinterpret_cast<__bf16*>(const_cast<void*>(users_planes)), users_plane_stride, users_panel_rows};
      return score_topk_thresh(a, K, kt, topk_idx, topk_val, ws, (hipStream_t)stream, &tp, &up, products, table_row_norm_max);
    }
    return score_topk_thresh(a, K, kt, topk_idx, topk_val, ws, (hipStream_t)stream);
  }
  const int64_t cand = (int64_t)a.n_split * lists_per_split() * kt;
  if ((int64_t)B * cand * 8 + 256 > ws_bytes) { pxr_set_error("pxr_score_topk_f32: workspace too small"); return PXR_ERR_WORKSPACE; }
  a.part_val = (float*)ws;
  a.part_idx = (int*)((char*)ws + (((int64_t)B * cand * 4 + 255) & ~(int64_t)255));
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(a.row_blocks * a.n_split);
  if (topk_variant() == 2) {
    switch (kt) {
      case 10: hipLaunchKernelGGL((score_topk2_kernel<10, 2>), grid, dim3(512), 0, st, a); break;
      case 16: hipLaunchKernelGGL((score_topk2_kernel<16, 2>), grid, dim3(512), 0, st, a); break;
      default: hipLaunchKernelGGL((score_topk2_kernel<32, 2>), grid, dim3(512), 0, st, a); break;
    }
  } else if (topk_variant() == 3) {
    switch (kt) {
      case 10: hipLaunchKernelGGL((score_topk2_kernel<10, 1>), grid, dim3(1024), 0, st, a); break;
      case 16: hipLaunchKernelGGL((score_topk2_kernel<16, 1>), grid, dim3(1024), 0, st, a); break;
      default: hipLaunchKernelGGL((score_topk2_kernel<32, 1>), grid, dim3(1024), 0, st, a); break;
    }
  } else {
    switch (kt) {
      case 10: hipLaunchKernelGGL(score_topk_kernel<10>, grid, dim3(GEMM_THREADS), 0, st, a); break;
      case 16: hipLaunchKernelGGL(score_topk_kernel<16>, grid, dim3(GEMM_THREADS), 0, st, a); break;
      default: hipLaunchKernelGGL(score_topk_kernel<32>, grid, dim3(GEMM_THREADS), 0, st, a); break;
    }
  }
  int rc = pxr_check_launch("pxr_score_topk_f32");
  if (rc) return rc;
  hipLaunchKernelGGL(topk_merge_kernel, dim3((B + 3) / 4), dim3(256), 0, st, (const float*)a.part_val,
                     (const int*)a.part_idx, B, (int)cand, K, topk_idx, topk_val);
  return pxr_check_launch("pxr_score_topk_f32(merge)");
}

// out[0] = max_i ||x[i, :]||_2 (x [rows, cols] fp32, row stride ldx, cols % 4 == 0): the table statistic behind the margin of the
// reduced-product top-k pass.  out is a device float (zeroed here, on the stream).
extern "C" int pxr_row_norm_max_f32(const float* x, int64_t rows, int64_t cols, int64_t ldx, float* out, void* stream) {
  PXR_REQUIRE(x && out && rows >= 0 && cols > 0 && cols % 4 == 0 && ldx % 4 == 0 && cols < (1ll << 31), "pxr_row_norm_max_f32: bad args");
  PXR_REQUIRE((((uintptr_t)x) & 15) == 0, "pxr_row_norm_max_f32: x must be 16-byte aligned");
  if (hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return pxr_check_launch("pxr_row_norm_max_f32(memset)");
  if (rows == 0) return PXR_OK;
  const int64_t blocks = rows < 4096 * 4 ? (rows + 3) / 4 : 4096;
  hipLaunchKernelGGL(row_norm_max_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, rows, (int)cols, ldx, out);
  return pxr_check_launch("pxr_row_norm_max_f32");
}
