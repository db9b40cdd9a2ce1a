// gemm_b3.cuh -- fp32 GEMM main loop on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16
// terms while it is staged into LDS and six of the nine cross products are accumulated in fp32 ("3 x bf16 split",
// SURVEY.md §7 hard part 1).  Same operand flavours, tiles, accumulator layout and epilogues as gemm_f32.cuh, so it is
// a drop-in for that main loop.
//
// Why: gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the f32-input MFMA (2.5 PFLOP/s against 157 TFLOP/s);
// six bf16 products cost 6/16 of one fp32 product.
//
// Arithmetic.  x = hi + mid + lo with hi = bf16_rne(x), mid = bf16_rne(x - hi), lo = x - hi - mid.  Both remainders
// are exact in fp32 and lo has at most 8 significant bits, so the three terms carry all 24 bits of x (exponents of
// normal fp32 values are preserved; values below 2^-110 lose low bits, irrelevant here; an infinite operand -- or one
// above bf16's largest finite value, 3.39e38 -- turns into NaN through inf - inf in the remainder, where the f32-input
// kernel would propagate the infinity: non-finite activations are an error state in this path either way, the trainer
// raises on a NaN loss like the reference's trainer.py:192-194).  bf16 x bf16 products are
// exact in the fp32 accumulator.  Kept: hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi.  Dropped: mid*lo, lo*mid, lo*lo
// <= (2^-9 * 2^-17) * 2 + 2^-34 ~ 2^-25 |a||b| per product -- half an fp32 ulp of the product, the same order as the
// rounding of an fp32 multiply, random in sign.  The small terms are accumulated apart from hi*hi and added at the end,
// so they are not swamped.  tests/test_gpu_gemm_b3.py holds the results against fp64 with the SAME tolerance as the
// f32-input MFMA kernel and next to that kernel's own error.
//
// LDS per K tile (32 k), per operand and plane: [row][32 bf16] = 64-byte rows, 16-byte chunks (8 consecutive k)
// XOR-swizzled by (row >> 2) & 3 -- conflict-free for ds_write_b128 (8-lane groups) and ds_read_b128 (its 16-lane
// groups {0-3,12-15,20-27}...).  A lane of v_mfma_f32_32x32x16_bf16 holds row (lane & 31), k = 8 * (lane >> 5) .. +7 of a
// 16-wide k block on BOTH operands.  Staging item = (row, chunk): a k-contiguous operand ("KC", [X][K] row-major) loads
// its 8 values as two float4; an x-contiguous operand ("XC", [K][X] row-major) loads them as 8 dwords from 8
// consecutive k rows (lanes run along x: coalesced) -- the transpose happens in the choice of who loads what, the LDS
// image is the same.
#pragma once
#include "gemm_f32.cuh"

namespace pxr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// (a, b) -> packed bf16 pairs of the three terms
__device__ __forceinline__ void b3_split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const f32x2 v = {a, b};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
  const bf16x2 m = __builtin_convertvector(r1, bf16x2);
  const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
  const bf16x2 l = __builtin_convertvector(r2, bf16x2);
  hi = __builtin_bit_cast(unsigned, h);
  mid = __builtin_bit_cast(unsigned, m);
  lo = __builtin_bit_cast(unsigned, l);
}
// 8 consecutive k of one row -> one 16-byte chunk per plane
__device__ __forceinline__ void b3_split8(const float (&x)[8], u32x4v (&p)[3]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned h, m, l;
    b3_split2(x[2 * j], x[2 * j + 1], h, m, l);
    p[0][j] = h; p[1][j] = m; p[2][j] = l;
  }
}

// STAGES = 3: a third LDS stage, so that the first fragments of the NEXT K tile are read before the barrier that ends the
// current one (they were written an iteration earlier) -- the matrix pipe does not wait for an LDS round trip after
// every barrier.
template <int BM, int BN, int FINE = 0, int STAGES = 2>
struct B3Cfg {
  using F = GemmCfg<BM, BN, true, true, 1, FINE>;     // same wave grid, accumulators and epilogues as the f32 kernels
  static constexpr int NT = F::NT;
  static constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;       // bytes per plane per stage
  static constexpr int STAGE = 3 * (A_PLANE + B_PLANE);            // bytes
  static constexpr int LDS_BYTES = STAGES * STAGE;
  static constexpr int A_ITEMS = BM * 4, B_ITEMS = BN * 4;         // (row, 8-k chunk) staging items per K tile
  // every thread stages A_PT items of A and B_PT of B, or (16-wave tiles) the first half of the workgroup stages A
  // and the second half B
  static constexpr bool ROLES = (A_ITEMS < NT);
  static constexpr int A_PT = ROLES ? 1 : A_ITEMS / NT, B_PT = ROLES ? 1 : B_ITEMS / NT;
  static_assert(ROLES ? (A_ITEMS + B_ITEMS == NT) : (A_ITEMS % NT == 0 && B_ITEMS % NT == 0), "staging map");
  static_assert(F::TM == 1 && F::TN == 1, "one 32x32 block per wave");
};

__device__ __forceinline__ int b3_chunk_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

// One operand's staging: which (row, chunk) an item is and how its 8 values are fetched.
template <int BX, bool KC>
struct B3Operand {
  bufrsrc rs;            // KC: the tile's rows for the whole K range (XC: re-based per K tile from `base`)
  const float* base;     // XC: matrix base + x0
  int64_t ld;
  int x_left;            // XC: columns of the matrix from the tile origin on
  __device__ __forceinline__ void init(const float* P, int64_t ld_, int X, int x0) {
    ld = ld_;
    x_left = X - x0;
    base = P + x0;
    // KC: rows past X lie outside the descriptor (the hardware returns zeros: no per-lane row test, which the compiler
    // would turn into a divergent branch around the loads)
    rs = KC ? make_rsrc(P + (int64_t)x0 * ld_, (int64_t)(X - x0) * ld_ * 4) : make_rsrc(P, 0);
  }
  static __device__ __forceinline__ int row_of(int it) { return KC ? (it >> 2) : (it % BX); }
  static __device__ __forceinline__ int c_of(int it) { return KC ? (it & 3) : (it / BX); }
  __device__ __forceinline__ void fetch(float (&v)[8], int it, int k0, int kend) const {
    const int row = row_of(it), c = c_of(it);
    if constexpr (KC) {
      const int gk = k0 + c * 8;
      // K is a multiple of 4 for KC operands, so each float4 lies entirely inside or outside [.., kend)
      const float4 a = buf_ld16(rs, gk < kend ? (unsigned)(row * (int)ld + gk) * 4u : BUF_OOB);
      const float4 b = buf_ld16(rs, gk + 4 < kend ? (unsigned)(row * (int)ld + gk + 4) * 4u : BUF_OOB);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      // descriptor = rows k0 .. kend-1 from column x0 on: rows past kend are outside it -> zeros; a column outside the
      // matrix pushes the offset out of range (arithmetic, not a select: a loop-invariant per-lane select around loads
      // becomes a divergent branch)
      const bufrsrc r = make_rsrc(base + (int64_t)k0 * ld, (int64_t)(kend - k0) * ld * 4);
      const unsigned ox = (row < x_left) ? 0u : 0x80000000u;
      // one per-lane offset (row 8c of the chunk); the 8 k rows are reached through the instruction's SCALAR offset, which
      // the raw-buffer range check includes -- no per-row offset registers (they spilled in the 16-wave kernels)
      const unsigned vo = ((unsigned)(c * 8 * (int)ld + row) * 4u) | ox;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, vo, (unsigned)(j * (int)ld) * 4u, 0));
    }
  }
};

// acc tile (m0, n0) = A_op x B_op over k in [kbeg, kend); operand flavours as gemm_f32.cuh (KC: [X][K], XC: [K][X]).
// CS (XC A operand only): cs[q] += the sum over k of the values this thread staged for its q-th A item (a column of the
// stored matrix; 16-wave tiles: threads 0 .. 4 BM - 1 stage A, one item each); the caller adds the 4 chunk owners of each
// column (items x, x + BM, x + 2 BM, x + 3 BM).
#ifndef PXR_B3_PD
#define PXR_B3_PD 2      // register-ring depth of the global prefetch (tiles in flight); 4 was measured, see DESIGN.md
#endif
template <int BM, int BN, bool A_KC, bool B_KC, int FINE = 0, bool CS = false, int STAGES = 2, int PD = (STAGES == 2 ? PXR_B3_PD : 2)>
__device__ __forceinline__ void gemm_b3_mainloop(typename B3Cfg<BM, BN, FINE, STAGES>::F::Acc& accs, const float* __restrict__ A,
                                                 int64_t lda, const float* __restrict__ B, int64_t ldb, int M, int N,
                                                 int kbeg, int kend, int m0, int n0, char* smem, float* cs = nullptr) {
  using Cfg = B3Cfg<BM, BN, FINE, STAGES>;
  using F = typename Cfg::F;
  constexpr int NT = Cfg::NT;
  static_assert(PD == 2 || (PD == 4 && STAGES == 2), "register ring: two tiles deep (four: 2-stage loop only)");
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / F::WGN, wn = wave % F::WGN;
  const int h = lane >> 5, r = lane & 31;
  auto& acc = accs.v[0][0];
  f32x16 accm, accl;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc[e] = 0.f; accm[e] = 0.f; accl[e] = 0.f; }
  const int nk = (kend - kbeg + 31) / 32;
  if (nk <= 0) return;

  B3Operand<BM, A_KC> opA;
  B3Operand<BN, B_KC> opB;
  opA.init(A, lda, M, m0);
  opB.init(B, ldb, N, n0);

  const bool stage_a = !Cfg::ROLES || tid < Cfg::A_ITEMS;
  const bool stage_b = !Cfg::ROLES || tid >= Cfg::A_ITEMS;
  const int tb = Cfg::ROLES ? tid - Cfg::A_ITEMS : tid;
  // 16-wave tiles: a thread stages A or B, never both -- one set of registers serves either
  constexpr int SLOTS = Cfg::ROLES ? 1 : Cfg::A_PT + Cfg::B_PT;
  struct Ring {
    float v[SLOTS][8];
  };
  Ring ring[PD];
  auto fetch = [&](int kt, Ring& g) {
    const int k0 = kbeg + kt * 32;
    if (stage_a) {
#pragma unroll
      for (int q = 0; q < Cfg::A_PT; ++q) opA.fetch(g.v[q], tid + q * NT, k0, kend);
    }
    if (stage_b) {
#pragma unroll
      for (int q = 0; q < Cfg::B_PT; ++q) opB.fetch(g.v[Cfg::ROLES ? 0 : Cfg::A_PT + q], tb + q * NT, k0, kend);
    }
  };
  // 16-wave tiles have 128 VGPRs per lane: the split terms are produced right where they are written instead of being
  // carried across the MFMAs
  constexpr bool LATE_SPLIT = (NT == 1024);
  struct Staged {
    u32x4v p[LATE_SPLIT ? 1 : SLOTS][3];
  };
  auto split = [&](const Ring& g, Staged& st) {      // VALU only
    if constexpr (!LATE_SPLIT) {
#pragma unroll
      for (int q = 0; q < SLOTS; ++q) b3_split8(g.v[q], st.p[q]);
    }
    if constexpr (CS) {
      static_assert(!A_KC, "column sums are taken over an x-contiguous A operand");
      if (stage_a) {
#pragma unroll
        for (int q = 0; q < Cfg::A_PT; ++q) {
          const float (&v)[8] = g.v[Cfg::ROLES ? 0 : q];
          cs[q] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
      }
    }
  };
  auto put = [&](int buf, const Staged& st, const Ring& g) {   // LDS writes only
    char* sa = smem + buf * Cfg::STAGE;
    char* sb = sa + 3 * Cfg::A_PLANE;
    if (stage_a) {
#pragma unroll
      for (int q = 0; q < Cfg::A_PT; ++q) {
        const int it = tid + q * NT;
        const int off = b3_chunk_off(B3Operand<BM, A_KC>::row_of(it), B3Operand<BM, A_KC>::c_of(it));
        if constexpr (LATE_SPLIT) {
          u32x4v t3[3];
          b3_split8(g.v[0], t3);
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4v*>(sa + p * Cfg::A_PLANE + off) = t3[p];
        } else {
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4v*>(sa + p * Cfg::A_PLANE + off) = st.p[q][p];
        }
      }
    }
    if (stage_b) {
#pragma unroll
      for (int q = 0; q < Cfg::B_PT; ++q) {
        const int it = tb + q * NT;
        const int off = b3_chunk_off(B3Operand<BN, B_KC>::row_of(it), B3Operand<BN, B_KC>::c_of(it));
        if constexpr (LATE_SPLIT) {
          u32x4v t3[3];
          b3_split8(g.v[0], t3);
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4v*>(sb + p * Cfg::B_PLANE + off) = t3[p];
        } else {
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4v*>(sb + p * Cfg::B_PLANE + off) = st.p[Cfg::A_PT + q][p];
        }
      }
    }
  };
  struct Frag {
    bf16x8 a[3], b[3];
  };
  auto read_frag = [&](Frag& f, int buf, int kb) {
    const char* sa = smem + buf * Cfg::STAGE;
    const char* sb = sa + 3 * Cfg::A_PLANE;
    const int c = kb * 2 + h;
    const int offa = b3_chunk_off(wm * F::WM + r, c), offb = b3_chunk_off(wn * F::WN + r, c);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      f.a[p] = *reinterpret_cast<const bf16x8*>(sa + p * Cfg::A_PLANE + offa);
      f.b[p] = *reinterpret_cast<const bf16x8*>(sb + p * Cfg::B_PLANE + offb);
    }
  };
  // six products per 16-wide k block; neighbouring MFMAs never share an accumulator
  auto mfma = [&](const Frag& f) {
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[2], f.b[0], accl, 0, 0, 0);   // lo  * hi
    accm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], f.b[0], accm, 0, 0, 0);   // mid * hi
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[0], acc, 0, 0, 0);     // hi  * hi
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[2], accl, 0, 0, 0);   // hi  * lo
    accm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0], f.b[1], accm, 0, 0, 0);   // hi  * mid
    accl = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1], f.b[1], accl, 0, 0, 0);   // mid * mid
  };

  Staged st;
  Frag fr[2];
  auto hints = [&]() {
    if constexpr (!LATE_SPLIT) {
      // issue order: the split's VALU work and the LDS writes in the shadow of the MFMAs (a wave issues in order;
      // 32 cycles of matrix pipe per MFMA cover ~7 other instructions)
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 7 * SLOTS / 2 + 1, 0);
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
  };
  if constexpr (STAGES == 2) {
#pragma unroll
    for (int s = 0; s < PD; ++s) fetch(s, ring[s]);
    split(ring[0], st);
    put(0, st, ring[0]);
    __syncthreads();
    read_frag(fr[0], 0, 0);
    for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
      for (int s = 0; s < PD; ++s) {
        const int kt = kt0 + s;
        if (kt < nk) {   // block-uniform
          const int buf = kt & 1, nxt = buf ^ 1;
          fetch(kt + PD, ring[s]);          // slot s held tile kt, which already sits in LDS
          read_frag(fr[1], buf, 1);
          split(ring[(s + 1) % PD], st);    // (zeros after the last tile)
          mfma(fr[0]);
          put(nxt, st, ring[(s + 1) % PD]);
          mfma(fr[1]);
          hints();
          __syncthreads();
          read_frag(fr[0], nxt, 0);
        }
      }
    }
  } else {
    // tiles 0 and 1 into stages 0 and 1, tiles 2 and 3 into the ring; iteration kt multiplies tile kt, writes tile kt + 2
    // into the stage tile kt - 1 left, fetches tile kt + 4 and -- before its closing barrier -- reads the first fragments
    // of tile kt + 1 (complete since the previous barrier)
    static_assert(STAGES == 3, "2 or 3 LDS stages");
    fetch(0, ring[0]);
    fetch(1, ring[1]);
    split(ring[0], st);
    put(0, st, ring[0]);
    fetch(2, ring[0]);
    split(ring[1], st);
    put(1, st, ring[1]);
    fetch(3, ring[1]);
    __syncthreads();
    read_frag(fr[0], 0, 0);
    int cur = 0;                            // stage of tile kt
    for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kt = kt0 + s;
        if (kt < nk) {   // block-uniform
          const int nx1 = cur == 2 ? 0 : cur + 1, nx2 = nx1 == 2 ? 0 : nx1 + 1;
          read_frag(fr[1], cur, 1);
          split(ring[s], st);               // tile kt + 2 (zeros after the last tile)
          mfma(fr[0]);
          read_frag(fr[0], nx1, 0);         // tile kt + 1
          put(nx2, st, ring[s]);
          fetch(kt + 4, ring[s]);
          mfma(fr[1]);
          hints();
          __syncthreads();
          cur = nx1;
        }
      }
    }
  }
  __syncthreads();   // the staging LDS is reused by the callers' epilogues
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] += (accm[e] + accl[e]);
}

}  // namespace pxr
