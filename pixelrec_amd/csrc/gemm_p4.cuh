// gemm_p4.cuh -- the planes GEMM main loop for BIG tiles (ViT tower, full-catalogue scoring, wide weight gradients):
// eight waves in two groups that run half a k block apart ("ping-pong"), a ring of 16-wide k blocks in LDS.
//
// Why a second main loop.  gemm_p3.cuh's 256x128 tile keeps all eight waves in lockstep: per 32-wide K tile every wave
// waits at the barrier, issues its 9 LDS-DMA pieces, reads 24 fragments and only then multiplies -- the two waves of a SIMD
// do their memory work at the SAME time and then contend for the SAME matrix pipe (MI355X_MICROARCH.md "Two waves per SIMD").
// Measured (profiles/r03): the pipe is busy 57 % of the cycles the clock actually delivers.  Here the waves 0-3 (group 0) and
// 4-7 (group 1; wave w and w + 4 share a SIMD) alternate roles every barrier: while one group issues its 24 MFMAs of a
// 16-wide k block (C segment), the other one reads its 12 fragments of the next k block and issues its share of the DMA
// (L segment).  Same arithmetic, same order of products and of k as gemm_p3.cuh with NACC = 3: bit-identical results.
//
// LDS ring.  A slot holds ONE 16-wide k block of the tile: for each operand, plane and 32-row block one 1 KiB piece that IS
// the MFMA fragment image -- lane l = (h = l >> 5, r = l & 31) finds its 8 k values (k = 8 h .. 8 h + 7 of the block, row r) at
// byte 16 l: a fragment read is one ds_read_b128 at lane * 16 + an immediate, conflict-free by construction.  The panel layout in
// memory (planes.cuh) is unchanged: an LDS-DMA lane may fetch its 16 bytes from anywhere, only the LDS side is linear, so lane l
// of a KC piece reads chunk (2 kb + h) ^ ((r >> 2) & 3) of row r's 64-byte segment.  (The two half pieces of a K tile are
// issued one L segment apart: the second one hits the lines the first one pulled into L1 / L2.)  XC operands (k along the
// panel rows: dX, dW) keep gemm_p3's image -- a 1 KiB run of 16 k rows x 32 x, read with ds_read_b64_tr_b16 -- which is
// already one k block per piece.
//
// Synchronisation (slot-time t = the interval between workgroup barriers t and t + 1; group 1 passes one extra barrier
// first, so it runs one slot-time behind):   group 0:  L_j at t = 2 j, C_j at 2 j + 1;   group 1:  L_j at 2 j + 1, C_j at 2 j + 2.
//   * RAW.  A wave ends L_j with s_waitcnt vmcnt(its pieces of k blocks > j + 1 may fly): its pieces of k block j + 1 have
//     landed; the barrier that opens slot-time 2 j + 2 has been passed by every wave after that wait, and k block j + 1 is
//     first read at 2 j + 2 (cdna_hip_programming.md "read a staged buffer one phase AFTER the wait that retires it").
//   * WAR.  k block j + NS - 1 is issued in L_j into the ring slot of k block j - 1, whose last reader is group 1's L_{j-1}
//     at slot-time 2 j - 1; every L segment ends with lgkmcnt(0) before its barrier, and the issue happens at t >= 2 j.
//   * Prefetch distance: a piece is needed 2 (NS - 1) - 1 slot-times (~ 4 000 cycles at NS = 4) after it is issued.
#pragma once
#include "gemm_p3.cuh"

namespace pxr {

// DMA_: where a wave issues its pieces of k block j + NS - 1:  0 = in L_j after the fragment reads;  1 = the A pieces in L_j, the B
// pieces between the MFMAs of C_j;  2 = all of them between the MFMAs of C_j (one piece per product group);  3 = in L_j BEFORE the reads
// NPL_: planes of each operand the loop stages and multiplies -- 3: all of them, the six products of gemm_p3 (the GEMMs); 2 (hi,
// mid: hi*hi + mid*hi + hi*mid) and 1 (hi*hi) are the CHEAP passes of the top-k threshold search, whose survivors are re-scored
// with all six products (score_topk.hip); they run on one accumulator set.
// HALF_: the operands are TWO fp16 planes (planes.cuh "h2": 22 significant bits) multiplied with three products (lo*hi, hi*lo, hi*hi on
// v_mfma_f32_32x32x16_f16) -- half the matrix-pipe work of the six bf16 products; NACC 1 (one set, small terms first) or 2 (hi*hi apart).
template <int BM_, int BN_, int WGM_, int WGN_, int NS_, int NACC_, int DMA_ = 0, int NPL_ = 3, bool HALF_ = false>
struct P4Cfg {
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, NS = NS_, NACC = NACC_, DMA = DMA_, NPL = NPL_;
  static constexpr bool HALF = HALF_;
  static_assert(NPL == 3 || NACC == 1 || (HALF && NACC == 2), "the reduced-product passes use one accumulator set");
  static_assert(!HALF || NPL == 2, "the fp16 format has two planes");
  static constexpr int G = WGM * WGN, NT = 64 * G, BK = 16;
  static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  static constexpr int A_FR = BM / 32, B_FR = BN / 32;                      // 32-row fragment blocks per operand
  static constexpr int A_PIECES = NPL * A_FR, B_PIECES = NPL * B_FR;        // 1 KiB pieces per k block
  static constexpr int SLOT = (A_PIECES + B_PIECES) * 1024;
  static constexpr int RING_BYTES = NS * SLOT;
  // the epilogue stages the fp32 tile in LDS in passes of EPI_COLS columns ([BM][EPI_COLS + 4] floats, p4_row_epilogue)
  static constexpr int EPI_COLS = BN > 128 ? 128 : BN, EPI_LD = EPI_COLS + 4, EPI_BYTES = BM * EPI_LD * 4;
  static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
  static constexpr bool PINGPONG = true;
  static_assert(G == 8, "two groups of four waves");
  static_assert(A_PIECES % 2 == 0 && B_PIECES % 2 == 0, "each group issues half of an operand's pieces");
  static_assert(WM % 32 == 0 && WN % 32 == 0 && TM >= 1 && TN >= 1, "wave tile = 32x32 blocks");
  static_assert(NS >= 3 && NS <= 8, "ring slots");
  static_assert(NACC == 1 || NACC == 2 || NACC == 3, "accumulator sets");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static constexpr bool EPI_OK = BN % EPI_COLS == 0 && (64 * G) % (EPI_COLS / 8) == 0 && BM % (64 * G / (EPI_COLS / 8)) == 0;   // p4_row_epilogue's chunk map
  struct Acc {
    f32x16 v[TM][TN];
  };
};

// wait until at most c of this wave's LDS-DMA pieces are still in flight (c wave-uniform: a scalar compare chain)
__device__ __forceinline__ void p4_wait_pieces(int c) {
  if (c >= 18) p3_wait_vm<18>();
  else if (c >= 15) p3_wait_vm<15>();
  else if (c >= 12) p3_wait_vm<12>();
  else if (c >= 10) p3_wait_vm<10>();
  else if (c >= 9) p3_wait_vm<9>();
  else if (c >= 8) p3_wait_vm<8>();
  else if (c >= 7) p3_wait_vm<7>();
  else if (c >= 6) p3_wait_vm<6>();
  else if (c >= 5) p3_wait_vm<5>();
  else if (c >= 4) p3_wait_vm<4>();
  else if (c >= 3) p3_wait_vm<3>();
  else if (c >= 2) p3_wait_vm<2>();
  else if (c >= 1) p3_wait_vm<1>();
  else p3_wait_vm<0>();
}

// One operand's share of a wave's DMA batch.  FR = 32-row blocks of the tile along the operand's x, KC = flavour.
template <int FR, bool KC, int NPL = 3>
struct P4Operand {
  static constexpr int PG = NPL * FR / 2;        // pieces per wave group and k block
  static constexpr int MAXP = (PG + 3) / 4;      // ... per wave (waves with wi + 4 t >= PG skip piece t)
  bufrsrc rs;
  unsigned scal[MAXP];     // loop-invariant byte offset of each piece (plane, block, tile origin)
  unsigned dst[MAXP];      // byte offset of each piece inside a ring slot
  int n;                   // pieces this wave issues per k block
  unsigned kstep;          // KC: bytes per 32-wide K tile (= 2 k blocks);  XC: 1024 per k block
  // m: the matrix; x0: tile origin along x; piece0: index of this operand's first piece inside a slot
  __device__ __forceinline__ void init(const P3Mat& m, int x0, int grp, int wi, int piece0) {
    rs = make_rsrc(reinterpret_cast<const float*>(m.p), m.ps * NPL * 2);      // the planes this loop reads, nothing behind them
#pragma unroll
    for (int t = 0; t < MAXP; ++t) {
      const int q = min(PG * grp + wi + 4 * t, NPL * FR - 1);
      const int pl = q / FR, blk = q % FR;
      unsigned o;
      if constexpr (KC) o = (unsigned)(pl * m.ps * 2) + (unsigned)((x0 + blk * 32) * 64);
      else o = (unsigned)(pl * m.ps * 2) + (unsigned)(((int64_t)(x0 / 32 + blk) * m.pr) * 64);
      scal[t] = __builtin_amdgcn_readfirstlane(o);
      dst[t] = __builtin_amdgcn_readfirstlane((unsigned)((piece0 + q) * 1024));
    }
    n = __builtin_amdgcn_readfirstlane(min(MAXP, (PG - wi + 3) / 4));
    kstep = KC ? (unsigned)(m.pr * 64) : 1024u;
  }
  // per-lane source offset of a piece of k block j (recomputed from the execution mask instead of being kept: with three
  // accumulator sets every VGPR is taken, and a spilled offset would come back as a scratch LOAD inside the K loop -- a VMEM
  // operation in front of which the compiler drains the whole DMA ring), and the wave-uniform offset of the k block
  __device__ __forceinline__ unsigned lane_off(int j) const {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if constexpr (KC) {
      const unsigned r = lane & 31u, h = lane >> 5;
      return (r << 6) + (((((unsigned)(j & 1) << 1) | h) ^ ((r >> 2) & 3u)) << 4);
    } else {
      return lane << 4;
    }
  }
  __device__ __forceinline__ unsigned k_off(int j) const { return KC ? (unsigned)(j >> 1) * kstep : (unsigned)j * 1024u; }
  // piece t of k block j into the ring slot at LDS byte address slot_base
  __device__ __forceinline__ void issue_one(int t, int j, unsigned slot_base) const {
    if (t < n) p3_dma16(rs, lane_off(j), __builtin_amdgcn_readfirstlane(k_off(j) + scal[t]), __builtin_amdgcn_readfirstlane(slot_base + dst[t]));
  }
  // all of this wave's pieces of k block j, `extra` bytes further into the operand (a later tile of a stream)
  __device__ __forceinline__ void issue_at(int j, unsigned slot_base, unsigned extra) const {
    const unsigned voff = lane_off(j), koff = k_off(j) + extra;
#pragma unroll
    for (int t = 0; t < MAXP; ++t)
      if (t < n) p3_dma16(rs, voff, __builtin_amdgcn_readfirstlane(koff + scal[t]), __builtin_amdgcn_readfirstlane(slot_base + dst[t]));
  }
  // all of this wave's pieces of k block j
  __device__ __forceinline__ void issue(int j, unsigned slot_base) const {
    const unsigned voff = lane_off(j), koff = k_off(j);
#pragma unroll
    for (int t = 0; t < MAXP; ++t)
      if (t < n) p3_dma16(rs, voff, __builtin_amdgcn_readfirstlane(koff + scal[t]), __builtin_amdgcn_readfirstlane(slot_base + dst[t]));
  }
};

// per-lane byte offsets of the fragment reads inside a piece
template <bool KC>
struct P4FragOff {
  int off[2];
  __device__ __forceinline__ void init(int lane) {
    if constexpr (KC) {
      off[0] = lane * 16;
      off[1] = 0;
    } else {
      // P3Frag<false> for a piece that holds the 16 k rows of ONE k block of a panel run: 16-lane group g = x block (g & 1) of 16
      // columns, k half h = g >> 1; lane li of the group addresses k row 8 h + (li >> 2) (+ 4 for the second read)
      const int h = lane >> 5, li = lane & 15, x16 = (lane >> 4) & 1;
      const int chunk = 2 * x16 + ((li & 3) >> 1), inner = (li & 1) * 8;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = 8 * h + 4 * t + (li >> 2);
        off[t] = row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4) + inner;
      }
    }
  }
  __device__ __forceinline__ p3_bf16x8 read(const char* piece) const {
    if constexpr (KC) {
      return *reinterpret_cast<const p3_bf16x8*>(piece + off[0]);
    } else {
      typedef __attribute__((address_space(3))) p3_bf16x4 lds_v4;
      const p3_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(piece + off[0]));
      const p3_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(piece + off[1]));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
};

// the per-wave state of a tile product: DMA plans of both operands, fragment addresses, the wave's place in the tile
template <class Cfg, bool A_KC, bool B_KC>
struct P4Loop {
  static constexpr int TM = Cfg::TM, TN = Cfg::TN, NS = Cfg::NS, PF = Cfg::NS - 1;
  static constexpr int MAXA = P4Operand<Cfg::A_FR, A_KC, Cfg::NPL>::MAXP, MAXB = P4Operand<Cfg::B_FR, B_KC, Cfg::NPL>::MAXP;
  // pieces of earlier-needed k blocks that may still be in flight when a wave certifies k block j + 1 at the end of L_j
  // (gemm_p4.cuh header, "RAW"): whole batches of the k blocks j + 2 .. issued so far, per operand
  static constexpr int KEEP_A = (Cfg::DMA == 2) ? PF - 2 : PF - 1;
  static constexpr int KEEP_B = (Cfg::DMA == 1 || Cfg::DMA == 2) ? PF - 2 : PF - 1;
  static_assert(KEEP_A >= 0 && KEEP_B >= 0, "pieces issued in the C segments need one more ring slot");
  struct Frag {
    p3_bf16x8 a[TM][Cfg::NPL], b[TN][Cfg::NPL];
  };
  P4Operand<Cfg::A_FR, A_KC, Cfg::NPL> opA;
  P4Operand<Cfg::B_FR, B_KC, Cfg::NPL> opB;
  P4FragOff<A_KC> foA;
  P4FragOff<B_KC> foB;
  char* smem;
  unsigned smem_base;
  int grp, wm, wn, keep;        // keep = KEEP_A * (A pieces per batch) + KEEP_B * (B pieces per batch) of this wave

  __device__ __forceinline__ void init(const P3Mat& A, const P3Mat& B, int m0, int n0, char* smem_) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    grp = wave >> 2;
    wm = wave / Cfg::WGN;
    wn = wave % Cfg::WGN;
    smem = smem_;
    smem_base = (unsigned)(size_t)smem_;
    opA.init(A, m0, grp, wave & 3, 0);
    opB.init(B, n0, grp, wave & 3, Cfg::A_PIECES);
    keep = KEEP_A * opA.n + KEEP_B * opB.n;
    foA.init(lane);
    foB.init(lane);
  }
  __device__ __forceinline__ unsigned slot_addr(int slot) const { return smem_base + (unsigned)slot * (unsigned)Cfg::SLOT; }
  __device__ __forceinline__ void issue(int j, int slot) const {
    opA.issue(j, slot_addr(slot));
    opB.issue(j, slot_addr(slot));
  }
  __device__ __forceinline__ void issue_a(int j, int slot) const { opA.issue(j, slot_addr(slot)); }
  // piece number u of the wave's batch (the A pieces first)
  __device__ __forceinline__ void issue_nth(int u, int j, int slot) const {
    if (u < MAXA) opA.issue_one(u, j, slot_addr(slot));
    else if (u < MAXA + MAXB) opB.issue_one(u - MAXA, j, slot_addr(slot));
  }
  // steady state: everything but `keep` pieces has landed
  __device__ __forceinline__ void wait_steady() const {
    constexpr int C_HI = KEEP_A * MAXA + KEEP_B * MAXB, C_LO = C_HI - KEEP_B, C_LO2 = C_HI - KEEP_A;
    if (keep == C_HI) p3_wait_vm<C_HI>();
    else if (keep == C_LO) p3_wait_vm<C_LO>();
    else if (keep == C_LO2) p3_wait_vm<C_LO2>();
    else p4_wait_pieces(keep);
  }
  __device__ __forceinline__ void read_frag(Frag& f, int slot) const {
    const char* s = smem + slot * Cfg::SLOT;
#pragma unroll
    for (int p = 0; p < Cfg::NPL; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f.a[i][p] = foA.read(s + (p * Cfg::A_FR + wm * TM + i) * 1024);
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b[j][p] = foB.read(s + (Cfg::A_PIECES + p * Cfg::B_FR + wn * TN + j) * 1024);
    }
  }
};

// six products per (block, 16-wide k block); consecutive MFMAs never share an accumulator.  NACC = 3: gemm_p3's three sets
// (hi*hi | the 2^-8 terms | the 2^-16 terms) -- bit-identical to it; 2: hi*hi | everything else; 1: one set, small terms first.
// `between(g)` runs after product group g = 0 .. 4, pinned there (DMA pieces issued among the MFMAs).
struct P4NoHook {
  __device__ __forceinline__ void operator()(int) const {}
};
template <class Cfg, bool HOOK, class Frag, class Hook>
__device__ __forceinline__ void p4_mfma(const Frag& f, f32x16 (&accs)[Cfg::TM][Cfg::TN], f32x16 (&accm)[Cfg::NACC >= 2 ? Cfg::TM : 1][Cfg::TN],
                                        f32x16 (&accl)[Cfg::NACC >= 3 ? Cfg::TM : 1][Cfg::TN], Hook&& between) {
  constexpr int TM = Cfg::TM, TN = Cfg::TN;
#define PXR_P4_PROD(G, ACC, PA, PB)                                                                             \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA], f.b[j][PB], ACC[i][j], 0, 0, 0);           \
  if constexpr (HOOK && G < 5) {                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    between(G);                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  }
#define PXR_P4_PRODH(G, ACC, PA, PB)                                                                            \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(p3_f16x8, f.a[i][PA]),                \
                                                         __builtin_bit_cast(p3_f16x8, f.b[j][PB]), ACC[i][j], 0, 0, 0);
  if constexpr (Cfg::HALF) {
    if constexpr (Cfg::NACC == 2) {
      PXR_P4_PRODH(0, accm, 1, 0)       // lo * hi
      PXR_P4_PRODH(1, accs, 0, 0)       // hi * hi
      PXR_P4_PRODH(2, accm, 0, 1)       // hi * lo
    } else {
      PXR_P4_PRODH(0, accs, 1, 0)
      PXR_P4_PRODH(1, accs, 0, 1)
      PXR_P4_PRODH(2, accs, 0, 0)
    }
  } else if constexpr (Cfg::NPL == 1) {
    PXR_P4_PROD(5, accs, 0, 0)
  } else if constexpr (Cfg::NPL == 2) {
    PXR_P4_PROD(5, accs, 1, 0)
    PXR_P4_PROD(5, accs, 0, 1)
    PXR_P4_PROD(5, accs, 0, 0)
  } else if constexpr (Cfg::NACC == 3) {
    PXR_P4_PROD(0, accl, 2, 0)        // lo  * hi
    PXR_P4_PROD(1, accm, 1, 0)        // mid * hi
    PXR_P4_PROD(2, accs, 0, 0)        // hi  * hi
    PXR_P4_PROD(3, accl, 0, 2)        // hi  * lo
    PXR_P4_PROD(4, accm, 0, 1)        // hi  * mid
    PXR_P4_PROD(5, accl, 1, 1)        // mid * mid
  } else if constexpr (Cfg::NACC == 2) {
    PXR_P4_PROD(0, accm, 2, 0)
    PXR_P4_PROD(1, accs, 0, 0)
    PXR_P4_PROD(2, accm, 0, 2)
    PXR_P4_PROD(3, accm, 1, 1)
    PXR_P4_PROD(4, accm, 1, 0)
    PXR_P4_PROD(5, accm, 0, 1)
  } else {
    PXR_P4_PROD(0, accs, 2, 0)
    PXR_P4_PROD(1, accs, 0, 2)
    PXR_P4_PROD(2, accs, 1, 1)
    PXR_P4_PROD(3, accs, 1, 0)
    PXR_P4_PROD(4, accs, 0, 1)
    PXR_P4_PROD(5, accs, 0, 0)
  }
#undef PXR_P4_PROD
#undef PXR_P4_PRODH
}

// acc tile (m0, n0) = A_op x B_op over k in [0, K) (K % 32 == 0), operands as planes.
// ONES (dW only): additionally accumulate in `ones_acc[i]` the products of the A fragments with an all-ones B fragment.
// dbg (compile-time; timing experiments only, results are wrong): 1 = no group stagger, 2 = no DMA, 4 = no MFMAs, 8 = no s_setprio,
// 16 = no fragment reads
template <class Cfg, bool A_KC, bool B_KC, bool ONES = false, int dbg = 0>
__device__ __forceinline__ void gemm_p4_mainloop(typename Cfg::Acc& acc_out, const P3Mat& A, const P3Mat& B, int K, int m0, int n0,
                                                 char* smem, f32x16* ones_acc = nullptr) {
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NS = Cfg::NS, PF = NS - 1, NACC = Cfg::NACC;
  using Loop = P4Loop<Cfg, A_KC, B_KC>;
  f32x16 accm[NACC >= 2 ? TM : 1][TN], accl[NACC >= 3 ? TM : 1][TN];
  auto& accs = acc_out.v;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        accs[i][j][e] = 0.f;
        if constexpr (NACC >= 2) accm[i][j][e] = 0.f;
        if constexpr (NACC >= 3) accl[i][j][e] = 0.f;
      }
  if constexpr (ONES) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) ones_acc[i][e] = 0.f;
  }
  const int nkb = K / 16;
  if (nkb <= 0) return;
  Loop L;
  L.init(A, B, m0, n0, smem);
  p3_bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  // ---- prologue: k blocks 0 .. PF-1 in flight, k block 0 landed ----------------------------------------------------------
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (s < nkb && !(dbg & 2)) L.issue(s, s);
  p4_wait_pieces(min(PF - 1, nkb - 1) * (L.opA.n + L.opB.n));
  __builtin_amdgcn_s_barrier();
  if (L.grp == 1 && !(dbg & 1)) __builtin_amdgcn_s_barrier();          // group 1 runs one slot-time behind
  typename Loop::Frag f;
  int slot = 0, islot = PF % NS;
  for (int j = 0; j < nkb; ++j) {
    const bool more = (j + PF < nkb) && !(dbg & 2);                      // k block j + PF is issued during this iteration
    // ---- L segment ----------------------------------------------------------------------------------------------------------
    if constexpr (Cfg::DMA == 3) {
      if (more) L.issue(j + PF, islot);
    }
    if (!(dbg & 16)) L.read_frag(f, slot);
    if constexpr (Cfg::DMA == 0) {
      if (more) L.issue(j + PF, islot);
    } else if constexpr (Cfg::DMA == 1) {
      if (more) L.issue_a(j + PF, islot);
    }
    // this wave's pieces of k block j + 1 have landed (the tail simply waits for everything: nothing newer is in flight that a
    // later k block could not wait for as well)
    if (j + PF < nkb) L.wait_steady();
    else p3_wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- C segment ----------------------------------------------------------------------------------------------------------
    if (!(dbg & 4)) {
      if (!(dbg & 8)) __builtin_amdgcn_s_setprio(1);
      if constexpr (Cfg::DMA == 1) {
        p4_mfma<Cfg, true>(f, accs, accm, accl, [&](int g) {
          if (more && g < Loop::MAXB) L.issue_nth(Loop::MAXA + g, j + PF, islot);
        });
      } else if constexpr (Cfg::DMA == 2) {
        p4_mfma<Cfg, true>(f, accs, accm, accl, [&](int g) {
          if (more) L.issue_nth(g, j + PF, islot);
        });
      } else {
        p4_mfma<Cfg, false>(f, accs, accm, accl, P4NoHook());
      }
      if constexpr (ONES && Cfg::HALF) {
        p3_f16x8 ones_h;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones_h[e] = (_Float16)1.0f;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            ones_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(p3_f16x8, f.a[i][p]), ones_h, ones_acc[i], 0, 0, 0);
      } else if constexpr (ONES) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < TM; ++i) ones_acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][p], ones, ones_acc[i], 0, 0, 0);
      }
      if (!(dbg & 8)) __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot = (slot + 1 == NS) ? 0 : slot + 1;
    if (j + PF < nkb) islot = (islot + 1 == NS) ? 0 : islot + 1;
  }
  if (L.grp == 0 && !(dbg & 1)) __builtin_amdgcn_s_barrier();
  __syncthreads();   // the staging LDS is reused by the epilogues
  if constexpr (NACC >= 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if constexpr (NACC == 3) accs[i][j][e] += (accm[i][j][e] + accl[i][j][e]);
          else accs[i][j][e] += accm[i][j][e];
        }
  }
}

// ---- the same product for a SEQUENCE of A tiles against one B tile as ONE k-block stream (full-catalogue scoring: item tiles
// m_first, m_first + BM, ... x one block of users; KC x KC): the ring never drains between tiles.  After the last k block of tile t
// every wave calls `done(t, accs)` on ITS accumulators (all sets folded) at the head of its next L segment -- while the other wave
// group multiplies -- and clears them.  `done` must not touch the LDS ring and must not contain workgroup barriers (the two groups
// reach it one slot-time apart); global atomics / stores are fine.
template <class Cfg, class TileFn>
__device__ __forceinline__ void gemm_p4_stream(const P3Mat& A, const P3Mat& B, int K, int m_first, int n0, int n_tiles, char* smem,
                                               TileFn&& done) {
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NS = Cfg::NS, PF = NS - 1, NACC = Cfg::NACC;
  static_assert(Cfg::DMA == 0, "the stream issues its pieces in the L segments");
  using Loop = P4Loop<Cfg, true, true>;
  const int nkb = K / 16;
  const int total = n_tiles * nkb;
  if (total <= 0) return;
  typename Cfg::Acc acc;
  auto& accs = acc.v;
  f32x16 accm[NACC >= 2 ? TM : 1][TN], accl[NACC >= 3 ? TM : 1][TN];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          accs[i][j][e] = 0.f;
          if constexpr (NACC >= 2) accm[i][j][e] = 0.f;
          if constexpr (NACC >= 3) accl[i][j][e] = 0.f;
        }
  };
  zero();
  Loop L;
  L.init(A, B, m_first, n0, smem);
  // the issue cursor: k block i_j of item tile i_tile; the A pieces of a later tile start i_tile * BM rows further down the panels
  int i_j = 0;
  unsigned i_tile_off = 0;
  auto issue_next = [&](int slot) {
    const unsigned sb = L.slot_addr(slot);
    L.opA.issue_at(i_j, sb, i_tile_off);
    L.opB.issue(i_j, sb);
    if (++i_j == nkb) { i_j = 0; i_tile_off += (unsigned)(Cfg::BM * 64); }
  };
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (s < total) issue_next(s);
  p4_wait_pieces(min(PF - 1, total - 1) * (L.opA.n + L.opB.n));
  __builtin_amdgcn_s_barrier();
  if (L.grp == 1) __builtin_amdgcn_s_barrier();                         // group 1 runs one slot-time behind
  typename Loop::Frag f;
  int slot = 0, islot = PF % NS, j = 0, tile = 0;
  bool pending = false;                                                  // tile `tile - 1` is complete in the accumulators
  for (int it = 0; it < total; ++it) {
    if (pending) {
      if constexpr (NACC >= 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              if constexpr (NACC == 3) accs[i][jj][e] += (accm[i][jj][e] + accl[i][jj][e]);
              else accs[i][jj][e] += accm[i][jj][e];
            }
      }
      done(tile - 1, acc);
      zero();
      pending = false;
    }
    // ---- L segment
    L.read_frag(f, slot);
    const bool more = it + PF < total;
    if (more) {
      issue_next(islot);
      L.wait_steady();
    } else {
      p3_wait_vm<0>();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- C segment
    __builtin_amdgcn_s_setprio(1);
    p4_mfma<Cfg, false>(f, accs, accm, accl, P4NoHook());
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot = (slot + 1 == NS) ? 0 : slot + 1;
    if (more) islot = (islot + 1 == NS) ? 0 : islot + 1;
    if (++j == nkb) { j = 0; ++tile; pending = true; }
  }
  if (L.grp == 0) __builtin_amdgcn_s_barrier();
  if constexpr (NACC >= 2) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if constexpr (NACC == 3) accs[i][jj][e] += (accm[i][jj][e] + accl[i][jj][e]);
          else accs[i][jj][e] += accm[i][jj][e];
        }
  }
  done(tile - 1, acc);
}

// ---- row epilogue (p3_row_epilogue for tiles whose fp32 image exceeds the LDS: EPI_COLS columns per pass): every thread owns 8
// CONSECUTIVE columns of a row.  Per pass: `pre(it, row, col, nv)` for each of the thread's chunks FIRST (the loads of what the
// epilogue reads -- bias, residual, saved derivative -- go out before the accumulators are staged, so their latency hides behind the
// LDS round trip instead of serialising with the stores chunk by chunk; the accumulator sets folded after the K loop left the
// registers for it), then `fn(it, row, col, nv, v)` once per chunk.  Chunks outside the matrix are skipped in both. --------------
template <class Cfg>
struct P4ChunkMap {
  static constexpr int EC = Cfg::EPI_COLS, NP = Cfg::BN / EC, CPR = EC / 8, RPI = Cfg::NT / CPR, CPT = Cfg::BM / RPI;
};
template <class Cfg, class Pre, class Fn>
__device__ __forceinline__ void p4_row_epilogue(const typename Cfg::Acc& accs, char* smem, int M, int N, int m0, int n0, Pre&& pre, Fn&& fn) {
  static_assert(Cfg::EPI_OK, "epilogue chunk map");
  float* t = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN, h = lane >> 5, r = lane & 31;
  using Map = P4ChunkMap<Cfg>;
  constexpr int EC = Map::EC, NP = Map::NP, CPR = Map::CPR, RPI = Map::RPI, CPT = Map::CPT;
  const int c8 = (threadIdx.x % CPR) * 8;
#pragma unroll
  for (int pass = 0; pass < NP; ++pass) {
    const int col = n0 + pass * EC + c8;
#pragma unroll
    for (int it = 0; it < CPT; ++it) {
      const int row = it * RPI + threadIdx.x / CPR;
      if (m0 + row < M && col < N) pre(it, m0 + row, col, min(8, N - col));
    }
    if (pass) __syncthreads();                         // everybody has read the previous pass
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col0 = wn * Cfg::WN + j * 32 - pass * EC;
      if (col0 >= 0 && col0 < EC) {                    // wave-uniform
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = wm * Cfg::WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            t[row * Cfg::EPI_LD + col0 + r] = accs.v[i][j][e];
          }
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < CPT; ++it) {
      const int row = it * RPI + threadIdx.x / CPR;
      if (m0 + row >= M || col >= N) continue;
      const float4 a = *reinterpret_cast<const float4*>(t + row * Cfg::EPI_LD + c8);
      const float4 b = *reinterpret_cast<const float4*>(t + row * Cfg::EPI_LD + c8 + 4);
      float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      fn(it, m0 + row, col, min(8, N - col), v);
    }
  }
}

}  // namespace pxr
