// gemm_b3.cuh -- fp32 GEMM main loop on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16
// terms and six of the nine cross products are accumulated in fp32 ("3 x bf16 split", SURVEY.md §7 hard part 1).
//
// Why: gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the f32-input MFMA the rest of this library uses
// (2.5 PFLOP/s against 157 TFLOP/s); six bf16 products cost 6/16 of one fp32 product.
//
// Arithmetic.  x = hi + mid + lo with hi = bf16_rne(x), mid = bf16_rne(x - hi), lo = x - hi - mid.  Both remainders
// are exact in fp32 and lo has at most 8 significant bits, so the three terms carry all 24 bits of x (exponents of
// normal fp32 values are preserved; values below 2^-110 lose low bits, irrelevant here).  bf16 x bf16 products are
// exact in the fp32 accumulator.  Kept: hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi.  Dropped: mid*lo, lo*mid, lo*lo
// <= (2^-9 * 2^-17) * 2 + 2^-34 ~ 2^-25 |a||b| per product -- half an fp32 ulp of the product, the same order as the
// rounding of an fp32 multiply, random in sign.  The small terms are accumulated apart from hi*hi and added at the end,
// so they are not swamped.  tests/test_gpu_gemm_b3.py holds the result against fp64 with the SAME tolerance as the
// fp32-MFMA kernel and reports both errors.
//
// Operands: A is fp32 [M][K] row-major (activations; split while it is staged into LDS);  B comes PRE-SPLIT as three
// bf16 planes [3][N][K] (weights / the item table: split once per step / per evaluation by split_planes_kernel, also
// transposed there when the consumer is an input-gradient GEMM).  LDS per K tile (32 k): per plane [row][32 bf16] =
// 64 B rows, 16-byte chunks XOR-swizzled by (row >> 2) & 3 -- conflict-free for ds_write_b128 (8-lane groups) and
// ds_read_b128 (its 16-lane groups {0-3,12-15,20-27}...).  A lane of v_mfma_f32_32x32x16_bf16 holds row (lane & 31),
// k = 8 * (lane >> 5) .. +7 of a 16-wide k block on BOTH operands (any k permutation common to A and B is harmless).
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
__device__ __forceinline__ void b3_split8(const float4& x0, const float4& x1, u32x4v& hi, u32x4v& mid, u32x4v& lo) {
  unsigned h, m, l;
  b3_split2(x0.x, x0.y, h, m, l); hi[0] = h; mid[0] = m; lo[0] = l;
  b3_split2(x0.z, x0.w, h, m, l); hi[1] = h; mid[1] = m; lo[1] = l;
  b3_split2(x1.x, x1.y, h, m, l); hi[2] = h; mid[2] = m; lo[2] = l;
  b3_split2(x1.z, x1.w, h, m, l); hi[3] = h; mid[3] = m; lo[3] = l;
}

// KW = 2: two wave groups per output tile, group g multiplies the g-th 16-wide k block of every K tile (twice the waves
// per SIMD for the same tiles: at M = 3200 tokens a CU holds only 1-2 workgroups, and a wave's K-tile iteration -- loads,
// split, LDS writes, barrier, LDS reads, MFMAs -- is a latency chain that only other waves can cover).
template <int BM, int BN, int FINE = 0, int KW = 1>
struct B3Cfg {
  using F = GemmCfg<BM, BN, true, true, KW, FINE>;    // same wave grid, accumulators and epilogues as the fp32 kernel
  static constexpr int NT = F::NT;
  static constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;       // bytes per plane per stage
  static constexpr int STAGE = 3 * (A_PLANE + B_PLANE);            // bytes
  static constexpr int LDS_BYTES = 2 * STAGE;
  static constexpr int A_ITEMS = BM * 4, B_ITEMS = BN * 4;         // (row, 8-k chunk) staging items per K tile
  // every thread stages A_PT items of A and B_PT of B, or (16-wave tiles) the first half of the workgroup stages A
  // and the second half B
  static constexpr bool ROLES = (A_ITEMS < NT);
  static constexpr int A_PT = ROLES ? 1 : A_ITEMS / NT, B_PT = ROLES ? 1 : B_ITEMS / NT;
  static_assert(ROLES ? (A_ITEMS + B_ITEMS == NT) : (A_ITEMS % NT == 0 && B_ITEMS % NT == 0), "staging map");
};

__device__ __forceinline__ int b3_chunk_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }

// C[m0.., n0..] tile += A[M][K] (fp32, lda) x planes Bp[3][N][K] (bf16, row stride ldb elements, plane stride
// bplane elements), k in [kbeg, kend) (multiples of 8 apart from the ragged end, which reads zeros).
template <int BM, int BN, int FINE = 0, int KW = 1, int HINT = 0, int PD = 2>
__device__ __forceinline__ void gemm_b3_mainloop(typename B3Cfg<BM, BN, FINE, KW>::F::Acc& accs, const float* __restrict__ A,
                                                 int64_t lda, const __bf16* __restrict__ Bp, int64_t ldb, int64_t bplane,
                                                 int M, int N, int kbeg, int kend, int m0, int n0, char* smem) {
  using Cfg = B3Cfg<BM, BN, FINE, KW>;
  using F = typename Cfg::F;
  constexpr int NT = Cfg::NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wk = wave / F::G, w4 = wave % F::G;
  const int wm = w4 / F::WGN, wn = w4 % F::WGN;
  const int h = lane >> 5, r = lane & 31;
  auto& acc = accs.v;
  f32x16 accm[F::TM][F::TN], accl[F::TM][F::TN];
#pragma unroll
  for (int i = 0; i < F::TM; ++i)
#pragma unroll
    for (int j = 0; j < F::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accm[i][j][e] = 0.f; accl[i][j][e] = 0.f; }
  const int nk = (kend - kbeg + 31) / 32;
  if (nk <= 0) return;

  const bufrsrc rsA = make_rsrc(A + (int64_t)m0 * lda, (int64_t)(M - m0) * lda * 4);
  bufrsrc rsB[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) rsB[p] = make_rsrc(reinterpret_cast<const float*>(Bp + p * bplane + (int64_t)n0 * ldb), (int64_t)(N - n0) * ldb * 2);

  const bool stage_a = !Cfg::ROLES || tid < Cfg::A_ITEMS;
  const bool stage_b = !Cfg::ROLES || tid >= Cfg::A_ITEMS;
  const int tb = Cfg::ROLES ? tid - Cfg::A_ITEMS : tid;
  struct Ring {     // (16-wave tiles: a thread stages A or B, never both -- A's two float4 share the registers of b[0][0..1])
    float4 a[Cfg::ROLES ? 0 : Cfg::A_PT][2];
    u32x4v b[Cfg::B_PT][3];
  };
  Ring ring[PD];
  auto fetch = [&](int kt, Ring& g) {
    const int k0 = kbeg + kt * 32;
    if (stage_a) {
#pragma unroll
      for (int q = 0; q < Cfg::A_PT; ++q) {
        const int it = tid + q * NT, row = it >> 2, c = it & 3;
        const int gk = k0 + c * 8;
        // rows past M lie outside the descriptor (the hardware returns zeros: no per-lane row test, which the compiler
        // would turn into a divergent branch around the loads); K is a multiple of 4, so each float4 is entirely
        // inside or outside [kbeg, kend)
        const float4 x0 = buf_ld16(rsA, gk < kend ? (unsigned)(row * (int)lda + gk) * 4u : BUF_OOB);
        const float4 x1 = buf_ld16(rsA, gk + 4 < kend ? (unsigned)(row * (int)lda + gk + 4) * 4u : BUF_OOB);
        if constexpr (Cfg::ROLES) {
          g.b[0][0] = __builtin_bit_cast(u32x4v, x0);
          g.b[0][1] = __builtin_bit_cast(u32x4v, x1);
        } else {
          g.a[q][0] = x0;
          g.a[q][1] = x1;
        }
      }
    }
    if (stage_b) {
#pragma unroll
      for (int q = 0; q < Cfg::B_PT; ++q) {
        const int it = tb + q * NT, row = it >> 2, c = it & 3;
        const int gk = k0 + c * 8;
        const unsigned off = gk < kend ? (unsigned)(row * (int)ldb + gk) * 2u : BUF_OOB;   // planes are padded to K % 8 == 0
#pragma unroll
        for (int p = 0; p < 3; ++p) g.b[q][p] = __builtin_amdgcn_raw_buffer_load_b128(rsB[p], off, 0, 0);
      }
    }
  };
  // 16-wave tiles have 128 VGPRs per lane: the split terms are produced right where they are written instead of being
  // carried across the MFMAs
  constexpr bool LATE_SPLIT = (NT == 1024);
  struct Staged {
    u32x4v p[LATE_SPLIT ? 0 : (Cfg::ROLES ? 1 : Cfg::A_PT)][3];
  };
  auto split = [&](const Ring& g, Staged& st) {      // VALU only
    if (!LATE_SPLIT && stage_a) {
#pragma unroll
      for (int q = 0; q < Cfg::A_PT; ++q) {
        if constexpr (Cfg::ROLES) b3_split8(__builtin_bit_cast(float4, g.b[0][0]), __builtin_bit_cast(float4, g.b[0][1]), st.p[q][0], st.p[q][1], st.p[q][2]);
        else b3_split8(g.a[q][0], g.a[q][1], st.p[q][0], st.p[q][1], st.p[q][2]);
      }
    }
  };
  auto put = [&](int buf, const Staged& st, const Ring& g) {   // LDS writes only
    char* sa = smem + buf * Cfg::STAGE;
    char* sb = sa + 3 * Cfg::A_PLANE;
    if (stage_a) {
#pragma unroll
      for (int q = 0; q < Cfg::A_PT; ++q) {
        const int it = tid + q * NT, row = it >> 2, c = it & 3;
        const int off = b3_chunk_off(row, c);
        if constexpr (LATE_SPLIT) {
          u32x4v t3[3];
          b3_split8(__builtin_bit_cast(float4, g.b[0][0]), __builtin_bit_cast(float4, g.b[0][1]), t3[0], t3[1], t3[2]);
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
        const int it = tb + q * NT, row = it >> 2, c = it & 3;
        const int off = b3_chunk_off(row, c);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4v*>(sb + p * Cfg::B_PLANE + off) = g.b[q][p];
      }
    }
  };
  struct Frag {
    bf16x8 a[3][F::TM], b[3][F::TN];
  };
  auto read_frag = [&](Frag& f, int buf, int kb) {
    const char* sa = smem + buf * Cfg::STAGE;
    const char* sb = sa + 3 * Cfg::A_PLANE;
    const int c = kb * 2 + h;
#pragma unroll
    for (int i = 0; i < F::TM; ++i) {
      const int off = b3_chunk_off(wm * F::WM + i * 32 + r, c);
#pragma unroll
      for (int p = 0; p < 3; ++p) f.a[p][i] = *reinterpret_cast<const bf16x8*>(sa + p * Cfg::A_PLANE + off);
    }
#pragma unroll
    for (int j = 0; j < F::TN; ++j) {
      const int off = b3_chunk_off(wn * F::WN + j * 32 + r, c);
#pragma unroll
      for (int p = 0; p < 3; ++p) f.b[p][j] = *reinterpret_cast<const bf16x8*>(sb + p * Cfg::B_PLANE + off);
    }
  };
  // six products per 16-wide k block; neighbouring MFMAs never share an accumulator
  auto mfma = [&](const Frag& f) {
#pragma unroll
    for (int i = 0; i < F::TM; ++i)
#pragma unroll
      for (int j = 0; j < F::TN; ++j) {
        accl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[2][i], f.b[0][j], accl[i][j], 0, 0, 0);   // lo  * hi
        accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1][i], f.b[0][j], accm[i][j], 0, 0, 0);   // mid * hi
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][i], f.b[0][j], acc[i][j], 0, 0, 0);     // hi  * hi
        accl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][i], f.b[2][j], accl[i][j], 0, 0, 0);   // hi  * lo
        accm[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[0][i], f.b[1][j], accm[i][j], 0, 0, 0);   // hi  * mid
        accl[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[1][i], f.b[1][j], accl[i][j], 0, 0, 0);   // mid * mid
      }
  };

#pragma unroll
  for (int s = 0; s < PD; ++s) fetch(s, ring[s]);
  Staged st;
  split(ring[0], st);
  put(0, st, ring[0]);
  __syncthreads();
  Frag fr[KW == 1 ? 2 : 1];
  read_frag(fr[0], 0, KW == 1 ? 0 : wk);
  // HINT: pin the issue order so that the split's VALU work and the LDS writes sit in the shadow of the MFMAs (a wave
  // issues in order; 32 cycles of matrix pipe per MFMA cover ~6 other instructions)
  for (int kt0 = 0; kt0 < nk; kt0 += PD) {
#pragma unroll
    for (int s = 0; s < PD; ++s) {
      const int kt = kt0 + s;
      if (kt < nk) {   // block-uniform
        const int buf = kt & 1, nxt = buf ^ 1;
        fetch(kt + PD, ring[s]);          // slot s held tile kt, which already sits in LDS
        if constexpr (KW == 1) {
          read_frag(fr[1], buf, 1);
          split(ring[(s + 1) % PD], st);  // (zeros after the last tile)
          mfma(fr[0]);
          put(nxt, st, ring[(s + 1) % PD]);
          mfma(fr[1]);
          if constexpr (HINT) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
          }
          __syncthreads();
          read_frag(fr[0], nxt, 0);
        } else {
          split(ring[(s + 1) % PD], st);
          mfma(fr[0]);
          put(nxt, st, ring[(s + 1) % PD]);
          if constexpr (HINT) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            }
          }
          __syncthreads();
          read_frag(fr[0], nxt, wk);
        }
      }
    }
  }
  __syncthreads();   // the staging LDS is reused by the callers' epilogues
#pragma unroll
  for (int i = 0; i < F::TM; ++i)
#pragma unroll
    for (int j = 0; j < F::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] += (accm[i][j][e] + accl[i][j][e]);
  if constexpr (KW > 1) {
    // add the partial sums of wave group 1 into wave group 0 through LDS: layout [w4][element][lane]
    static_assert(KW == 2 && F::TM == 1 && F::TN == 1, "k-split is built for 2 wave groups of one 32x32 block each");
    static_assert(F::G * 16 * 64 * 4 <= Cfg::STAGE, "accumulator exchange does not fit in the staging LDS");
    float* ex = reinterpret_cast<float*>(smem) + (w4 * 16) * 64 + lane;
    if (wk == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) ex[e * 64] = acc[0][0][e];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][0][e] += ex[e * 64];
    }
  }
}

}  // namespace pxr
