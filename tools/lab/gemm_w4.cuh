// gemm_w4.cuh -- LAB ONLY (tools/lab/p4_lab.hip; not part of libpxr.so): two more main-loop structures for the 256 x 256 planes tile,
// built in round 6 to find out whether the ping-pong loop of gemm_p4.cuh is what keeps the big GEMMs / the scoring pass at ~0.46 of
// the nominal MFMA peak.  It is not: on the scoring shape all three structures finish within 5 % of each other AT DIFFERENT CLOCKS
// (ping-pong 1 104 us at 1.68 GHz, one wave per SIMD 1 152 us at 1.76 GHz, eight free-running waves 1 148-1 180 us at 1.69-1.77 GHz;
// profiles/r06/lab/w4_f8_lab_scoring.log): the part is at its power limit, a structure that keeps the pipe busier is clocked
// lower.  Kept as the record of that experiment.
//
// The planes GEMM main loop for the BIGGEST tiles (full-catalogue scoring, the 256-wide GEMMs of large batches and of
// the image tower): FOUR waves per workgroup, one per SIMD, each owning a 128 x 128 quarter of a 256 x 256 tile -- 256 accumulator
// registers per lane (the AGPR half of the unified register file) -- with its memory work threaded BETWEEN its own MFMAs.
//
// Why a third main loop.  gemm_p4.cuh puts two waves on every SIMD and alternates them between a compute and a load segment.  Its
// lab (tools/lab/p4_lab.hip, profiles/r06/lab) shows where the matrix pipe's time goes on the 256 x 256 one-set tile, two planes per
// operand (the top-k threshold pass / every fp16 two-plane GEMM):
//     MFMAs + barriers only (no DMA, no fragment reads)      77 % of the pipe at 2.33 GHz -- the hand-over between the two waves of
//                                                            a SIMD costs ~230 cycles per 768-cycle compute segment;
//     everything, real operands                              72 % of the pipe at the 1.66 GHz the part sustains under this load
//     per 24 MFMAs a wave also issues 12 ds_read_b128 + 4 LDS-DMA pieces (+ 73 SALU, 60 VALU: rocprofv3, profiles/r04/pmc).
// One wave per SIMD has nobody to hand over to: its MFMAs issue back to back (32 cycles each), and each leaves ~7 issue slots in
// its shadow for the loads of the NEXT k block.  A 128 x 128 wave tile also halves the fragment reads per MFMA: per 16-wide k block
// a wave reads (4 + 4) x NPL fragments for 16 x {3, 6} MFMAs -- 0.33 reads per MFMA instead of 0.5 -- and issues the same
// 1 piece per 6 (NPL = 2) / 8 (NPL = 3) MFMAs.
//
// LDS ring, piece format and operand flavours are gemm_p4.cuh's (one 16-wide k block per slot, a 1 KiB piece = the fragment image of
// a 32-row block; KC / XC via P4Operand / P4FragOff).  Products, their order inside a k block and the k order are those of gemm_p4
// with ONE accumulator set (NACC = 1): bit-identical results to P4Cfg<256, 256, 4, 2, NS, 1, 0, NPL, HALF>.
//
// Schedule (block = 16-wide k block; PF = NS - 1 blocks ahead; F[2] = two fragment register sets).  Iteration j:
//     MFMAs of block j from F[j & 1], in units of 4 (one row of the wave tile, one product); after each unit a slice of the side work:
//       the fragment reads of block j + 1 into F[(j + 1) & 1] and this wave's pieces of block j + PF into ring slot (j + PF) % NS;
//     then  s_waitcnt vmcnt(own pieces of blocks > j + 2 may fly), lgkmcnt(0);  s_barrier.
//   RAW  block j + 1 is read in iteration j: every wave certified ITS pieces of it at the end of iteration j - 1, before the barrier.
//   WAR  slot (j + PF) % NS held block j - 1, last read in iteration j - 2; every wave's reads were retired (lgkmcnt(0)) before the
//        barrier that ended that iteration.
//   A piece is needed NS - 2 iterations (>= 1 536 / 3 072 cycles at NPL = 2 / 3) after it is issued.
#pragma once
#include "../../pixelrec_amd/csrc/gemm_p4.cuh"

namespace pxr {

// WGM_ x WGN_ waves: 2 x 2 (one wave per SIMD, 128 x 128 wave tiles, 256 accumulators in the AGPRs) or 4 x 2 (two free-running waves
// per SIMD, 64 x 128 wave tiles: each wave's loads hide behind its PARTNER's MFMAs as well as its own -- no roles, no hand-over)
template <int NS_, int NPL_ = 3, bool HALF_ = false, int WGM_ = 2, int WGN_ = 2>
struct W4Cfg {
  static constexpr int BM = 256, BN = 256, WGM = WGM_, WGN = WGN_, NS = NS_, NACC = 1, DMA = 0, NPL = NPL_;
  static constexpr bool HALF = HALF_;
  static_assert(!HALF || NPL == 2, "the fp16 format has two planes");
  static constexpr int G = WGM * WGN, NT = 64 * G, BK = 16;
  static_assert(G == 4 || G == 8, "four or eight waves");
  static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  static constexpr int A_FR = BM / 32, B_FR = BN / 32;
  static constexpr int A_PIECES = NPL * A_FR, B_PIECES = NPL * B_FR;
  static constexpr int SLOT = (A_PIECES + B_PIECES) * 1024;
  static constexpr int RING_BYTES = NS * SLOT;
  static constexpr int EPI_COLS = 128, EPI_LD = EPI_COLS + 4, EPI_BYTES = BM * EPI_LD * 4;
  static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
  static constexpr bool PINGPONG = false, W4 = true;
  static constexpr int PPW = (A_PIECES + B_PIECES) / G;      // pieces a wave issues per k block
  static constexpr int NPROD = HALF ? 3 : (NPL == 3 ? 6 : (NPL == 2 ? 3 : 1));
  static constexpr int READS = (TM + TN) * NPL;              // fragment reads per wave and k block
  static_assert(NS >= 3 && NS <= 5, "ring slots");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static constexpr bool EPI_OK = BN % EPI_COLS == 0 && NT % (EPI_COLS / 8) == 0 && BM % (NT / (EPI_COLS / 8)) == 0;
  struct Acc {
    f32x16 v[TM][TN];
  };
};

// P4Operand + the issue of ONE piece at a tile offset (this loop threads single pieces between its MFMAs; every wave owns all MAXP
// strides of its list half, so there is no `t < n` test)
template <int FR, bool KC, int NPL>
struct W4Operand : P4Operand<FR, KC, NPL> {
  __device__ __forceinline__ void issue_one_at(int t, int j, unsigned slot_base, unsigned extra) const {
    p3_dma16(this->rs, this->lane_off(j), __builtin_amdgcn_readfirstlane(this->k_off(j) + extra + this->scal[t]),
             __builtin_amdgcn_readfirstlane(slot_base + this->dst[t]));
  }
};

template <class Cfg, bool A_KC, bool B_KC>
struct W4Loop {
  static constexpr int TM = Cfg::TM, TN = Cfg::TN, NS = Cfg::NS, PF = Cfg::NS - 1, NPL = Cfg::NPL;
  using OpA = W4Operand<Cfg::A_FR, A_KC, NPL>;
  using OpB = W4Operand<Cfg::B_FR, B_KC, NPL>;
  static constexpr int PA = OpA::MAXP, PB = OpB::MAXP;       // pieces per half list
  static constexpr int H = Cfg::G == 4 ? 2 : 1;              // list halves a wave owns
  static_assert(H * (PA + PB) == Cfg::PPW && OpA::PG % 4 == 0 && OpB::PG % 4 == 0, "each wave issues its share of a slot, every stride of it");
  struct Frag {
    p3_bf16x8 a[TM][NPL], b[TN][NPL];
  };
  // P4Operand splits an operand's piece list in two halves ("groups" of the ping-pong loop), each strided over four waves: with
  // four waves a wave owns its stride of BOTH halves (a0 | a1), with eight the stride of the half wave / 4 (a0 only)
  OpA a0, a1;
  OpB b0, b1;
  P4FragOff<A_KC> foA;
  P4FragOff<B_KC> foB;
  char* smem;
  unsigned smem_base;
  int wm, wn;

  __device__ __forceinline__ void init(const P3Mat& A, const P3Mat& B, int m0, int n0, char* smem_) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    wm = wave / Cfg::WGN;
    wn = wave % Cfg::WGN;
    smem = smem_;
    smem_base = (unsigned)(size_t)smem_;
    if constexpr (H == 2) {
      a0.init(A, m0, 0, wave, 0);
      a1.init(A, m0, 1, wave, 0);
      b0.init(B, n0, 0, wave, Cfg::A_PIECES);
      b1.init(B, n0, 1, wave, Cfg::A_PIECES);
    } else {
      a0.init(A, m0, wave >> 2, wave & 3, 0);
      b0.init(B, n0, wave >> 2, wave & 3, Cfg::A_PIECES);
    }
    foA.init(lane);
    foB.init(lane);
  }
  __device__ __forceinline__ unsigned slot_addr(int slot) const { return smem_base + (unsigned)slot * (unsigned)Cfg::SLOT; }
  // piece u (0 .. PPW - 1) of this wave's batch for k block j; a_extra: byte offset of a later A tile (streams)
  template <int U>
  __device__ __forceinline__ void issue_nth(int j, unsigned sb, unsigned a_extra) const {
    if constexpr (H == 2) {
      if constexpr (U < PA) a0.issue_one_at(U, j, sb, a_extra);
      else if constexpr (U < 2 * PA) a1.issue_one_at(U - PA, j, sb, a_extra);
      else if constexpr (U < 2 * PA + PB) b0.issue_one_at(U - 2 * PA, j, sb, 0u);
      else b1.issue_one_at(U - 2 * PA - PB, j, sb, 0u);
    } else {
      if constexpr (U < PA) a0.issue_one_at(U, j, sb, a_extra);
      else b0.issue_one_at(U - PA, j, sb, 0u);
    }
  }
  __device__ __forceinline__ void issue_all(int j, int slot, unsigned a_extra) const {
    const unsigned sb = slot_addr(slot);
    a0.issue_at(j, sb, a_extra);
    b0.issue(j, sb);
    if constexpr (H == 2) {
      a1.issue_at(j, sb, a_extra);
      b1.issue(j, sb);
    }
  }
  // fragment read r (0 .. READS - 1) of a slot: plane-major, the A blocks of the wave tile first
  template <int R>
  __device__ __forceinline__ void read_nth(Frag& f, const char* s) const {
    constexpr int p = R / (TM + TN), q = R % (TM + TN);
    if constexpr (q < TM) f.a[q][p] = foA.read(s + (p * Cfg::A_FR + wm * TM + q) * 1024);
    else f.b[q - TM][p] = foB.read(s + (Cfg::A_PIECES + p * Cfg::B_FR + wn * TN + (q - TM)) * 1024);
  }
  template <int R = 0>
  __device__ __forceinline__ void read_all(Frag& f, int slot) const {
    if constexpr (R < Cfg::READS) {
      read_nth<R>(f, smem + slot * Cfg::SLOT);
      read_all<R + 1>(f, slot);
    }
  }
};

// product Q of a k block, accumulator row I: TN MFMAs.  Plane pairs in gemm_p4's one-set order (small terms first).
template <class Cfg, int Q, int I, class Frag>
__device__ __forceinline__ void w4_mfma_row(const Frag& f, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  constexpr int NPL = Cfg::NPL;
  constexpr int PA = Cfg::HALF ? (Q == 0 ? 1 : 0) : NPL == 1 ? 0 : NPL == 2 ? (Q == 0 ? 1 : 0)
                                                                            : (Q == 0 ? 2 : Q == 1 ? 0 : Q == 2 ? 1 : Q == 3 ? 1 : 0);
  constexpr int PB = Cfg::HALF ? (Q == 1 ? 1 : 0) : NPL == 1 ? 0 : NPL == 2 ? (Q == 1 ? 1 : 0)
                                                                            : (Q == 0 ? 0 : Q == 1 ? 2 : Q == 2 ? 1 : Q == 3 ? 0 : Q == 4 ? 1 : 0);
#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    if constexpr (Cfg::HALF)
      acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(p3_f16x8, f.a[I][PA]), __builtin_bit_cast(p3_f16x8, f.b[j][PB]),
                                                         acc[I][j], 0, 0, 0);
    else
      acc[I][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[I][PA], f.b[j][PB], acc[I][j], 0, 0, 0);
  }
}

// side operations S .. S1 - 1 of a k block.  The PPW DMA pieces are spread evenly among the READS fragment reads: operation S is a
// DMA piece when the running count floor((S + 1) PPW / SIDE) steps, else the next read.
template <class Cfg, class Loop, bool DO_READ, bool DO_DMA, int S, int S1>
__device__ __forceinline__ void w4_side(const Loop& L, typename Loop::Frag& nxt, const char* rd_slot, int jd, unsigned sb, unsigned a_extra) {
  constexpr int SIDE = Cfg::READS + Cfg::PPW;
  if constexpr (S < S1) {
    constexpr int d0 = S * Cfg::PPW / SIDE, d1 = (S + 1) * Cfg::PPW / SIDE;      // DMA pieces before / through operation S
    if constexpr (d1 > d0) {
      if constexpr (DO_DMA) L.template issue_nth<d0>(jd, sb, a_extra);
    } else {
      if constexpr (DO_READ) L.template read_nth<S - d0>(nxt, rd_slot);
    }
    w4_side<Cfg, Loop, DO_READ, DO_DMA, S + 1, S1>(L, nxt, rd_slot, jd, sb, a_extra);
  }
}

// One k block: the MFMAs from `cur`, with the side work -- READS fragment reads of the next block into `nxt` (DO_READ) and PPW DMA
// pieces of block jd into the ring slot at `sb` (DO_DMA) -- spread evenly behind the units.  Everything is pinned in program order.
template <class Cfg, class Loop, bool DO_READ, bool DO_DMA, int UNIT = 0>
__device__ __forceinline__ void w4_block(const Loop& L, const typename Loop::Frag& cur, typename Loop::Frag& nxt, f32x16 (&acc)[Cfg::TM][Cfg::TN],
                                         const char* rd_slot, int jd, unsigned sb, unsigned a_extra) {
  constexpr int UNITS = Cfg::NPROD * Cfg::TM;                 // units of TN MFMAs
  constexpr int SIDE = Cfg::READS + Cfg::PPW;
  if constexpr (UNIT < UNITS) {
    w4_mfma_row<Cfg, UNIT / Cfg::TM, UNIT % Cfg::TM>(cur, acc);
    __builtin_amdgcn_sched_barrier(0);
    w4_side<Cfg, Loop, DO_READ, DO_DMA, UNIT * SIDE / UNITS, (UNIT + 1) * SIDE / UNITS>(L, nxt, rd_slot, jd, sb, a_extra);
    __builtin_amdgcn_sched_barrier(0);
    w4_block<Cfg, Loop, DO_READ, DO_DMA, UNIT + 1>(L, cur, nxt, acc, rd_slot, jd, sb, a_extra);
  }
}
// acc tile (m0, n0) = A_op x B_op over k in [0, K) (K % 32 == 0), operands as planes.
// The loop body is branch-free: every iteration reads "the next block" and issues "block j + PF" -- past the end of K these are a
// re-read of a ring slot nobody needs and a re-fetch of the last k block into a slot nobody reads (PF junk blocks per tile, L2 hits)
// -- so that 256 accumulators never cross a control-flow merge inside the K loop and the vmcnt bookkeeping is one constant.
template <class Cfg, bool A_KC, bool B_KC>
__device__ __forceinline__ void gemm_w4_mainloop(typename Cfg::Acc& acc_out, const P3Mat& A, const P3Mat& B, int K, int m0, int n0, char* smem) {
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NS = Cfg::NS, PF = NS - 1;
  using Loop = W4Loop<Cfg, A_KC, B_KC>;
  auto& acc = acc_out.v;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nkb = K / 16;          // even
  if (nkb <= 0) return;
  Loop L;
  L.init(A, B, m0, n0, smem);
  // ---- prologue: blocks 0 .. PF - 1 in flight (clamped like the loop's); block 0 landed and read; block 1 landed
#pragma unroll
  for (int s = 0; s < PF; ++s) L.issue_all(min(s, nkb - 1), s, 0u);
  p3_wait_vm<(PF - 1) * Cfg::PPW>();
  __builtin_amdgcn_s_barrier();
  typename Loop::Frag f0, f1;
  L.read_all(f0, 0);
  p3_wait_vm<(PF - 2) * Cfg::PPW>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  int rslot = 1 % NS, islot = PF % NS;       // slot of block j + 1 / of block j + PF
  for (int j = 0; j < nkb; j += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h == 0) w4_block<Cfg, Loop, true, true>(L, f0, f1, acc, smem + rslot * Cfg::SLOT, min(j + PF, nkb - 1), L.slot_addr(islot), 0u);
      else w4_block<Cfg, Loop, true, true>(L, f1, f0, acc, smem + rslot * Cfg::SLOT, min(j + 1 + PF, nkb - 1), L.slot_addr(islot), 0u);
      // own pieces of block j + 2 landed: blocks j + 3 .. j + PF may fly
      p3_wait_vm<(PF - 2) * Cfg::PPW>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rslot = (rslot + 1 == NS) ? 0 : rslot + 1;
      islot = (islot + 1 == NS) ? 0 : islot + 1;
    }
  }
  p3_wait_vm<0>();   // the junk blocks of the tail have landed
  __syncthreads();   // the ring is reused by the epilogues
}

// ---- the same product for a SEQUENCE of A tiles against one B tile as ONE k-block stream (full-catalogue scoring; KC x KC): the
// ring never drains between tiles.  After the last k block of tile t every wave calls `done(t, acc)` on ITS accumulators and clears
// them.  `done` must not touch the LDS ring and must not contain workgroup barriers; global atomics / stores are fine.
template <class Cfg, class TileFn>
__device__ __forceinline__ void gemm_w4_stream(const P3Mat& A, const P3Mat& B, int K, int m_first, int n0, int n_tiles, char* smem, TileFn&& done) {
  constexpr int TM = Cfg::TM, TN = Cfg::TN, NS = Cfg::NS, PF = NS - 1;
  using Loop = W4Loop<Cfg, true, true>;
  const int nkb = K / 16;          // even
  if (n_tiles <= 0 || nkb <= 0) return;
  typename Cfg::Acc accs;
  auto& acc = accs.v;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  Loop L;
  L.init(A, B, m_first, n0, smem);
  // the issue cursor: k block i_j of item tile i_tile (clamped to the last block of the last tile: the junk blocks of the tail);
  // the A pieces of a later tile start i_tile * BM rows further down the panels
  int i_j = 0, i_tile = 0;
  unsigned i_tile_off = 0;
  auto advance = [&]() {
    if (i_j + 1 < nkb) ++i_j;
    else if (i_tile + 1 < n_tiles) { i_j = 0; ++i_tile; i_tile_off += (unsigned)(Cfg::BM * 64); }
  };
#pragma unroll
  for (int s = 0; s < PF; ++s) { L.issue_all(i_j, s, i_tile_off); advance(); }
  p3_wait_vm<(PF - 1) * Cfg::PPW>();
  __builtin_amdgcn_s_barrier();
  typename Loop::Frag f0, f1;
  L.read_all(f0, 0);
  p3_wait_vm<(PF - 2) * Cfg::PPW>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  int rslot = 1 % NS, islot = PF % NS;
  for (int tile = 0; tile < n_tiles; ++tile) {
    for (int j = 0; j < nkb; j += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 0) w4_block<Cfg, Loop, true, true>(L, f0, f1, acc, smem + rslot * Cfg::SLOT, i_j, L.slot_addr(islot), i_tile_off);
        else w4_block<Cfg, Loop, true, true>(L, f1, f0, acc, smem + rslot * Cfg::SLOT, i_j, L.slot_addr(islot), i_tile_off);
        advance();
        p3_wait_vm<(PF - 2) * Cfg::PPW>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        rslot = (rslot + 1 == NS) ? 0 : rslot + 1;
        islot = (islot + 1 == NS) ? 0 : islot + 1;
      }
    }
    done(tile, accs);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jj = 0; jj < TN; ++jj)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
  }
  p3_wait_vm<0>();
}

}  // namespace pxr
