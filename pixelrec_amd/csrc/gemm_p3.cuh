// gemm_p3.cuh -- fp32 GEMM on the bf16 matrix pipe from PRE-SPLIT operands ("planes").
//
// gemm_b3.cuh splits every fp32 operand into its three bf16 terms inside the main loop: every one of the 8-24 tiles that
// read an activation tile repeats that VALU work, the staging registers cap the occupancy, and the loads are 64-byte row
// segments (half of every cache line per request: measured 29 B/clk/CU from L2, which bounds the 64x64 tiles of the
// training step, not the matrix pipe).  Here the producers (LayerNorm / attention / GEMM epilogues / the optimizer) have
// already written each operand as three bf16 planes  hi | mid | lo  (x = hi + mid + lo exactly, p3_split2), in a PANEL
// layout that makes every global -> LDS transfer a linear 1 KiB LDS-DMA piece: the main loop is LDS-DMA + ds_read +
// v_mfma_f32_32x32x16_bf16 only -- no VALU on the operands, no staging registers.
//
// Arithmetic: identical to gemm_b3.cuh (six of the nine cross products, fp32 accumulation, the small terms in their own
// accumulators, same k order) -- the planes ARE the terms that kernel computes on the fly: results are bit-identical to it.
//
// Panel layout of a matrix X[R][C] (C % 32 == 0; `pr` >= R rows allocated per panel, pr % 32 == 0): plane q starts
// q * ps elements after the base; inside a plane, panel cb = c / 32 holds columns 32 cb .. 32 cb + 31 of ALL rows:
//     element (r, c)  at  ((cb * pr + r) * 32 + (((c >> 3) & 3) ^ ((r >> 2) & 3)) * 8 + (c & 7))      [elements]
// i.e. a 64-byte row segment per (row, panel) whose four 16-byte chunks are XOR-swizzled by the row -- the LDS image of a
// tile IS a byte copy of 1 KiB runs of a panel (16 rows): conflict-free ds_read_b128 / ds_read_b64_tr_b16 without any
// address arithmetic on the DMA source.  Rows beyond R inside a panel may hold anything finite or not: they only ever
// reach accumulator rows / columns the epilogues do not store.
//
// Operand flavours (as gemm_f32.cuh):  "KC"  k runs along the panel's 32 columns (A[M][K] / B[N][K]): fragment = one
// ds_read_b128;  "XC"  k runs along the ROWS (A stored [K][M], B stored [K][N]): the tile is 32 k-rows of BX/32 panels,
// fragments come from two ds_read_b64_tr_b16 (hardware 4x4 transpose: a 16-lane group reads a [4 k][16 x] block).
//   Y = X W^T (KC,KC) forward;  dX = dY W (KC,XC);  dW = dY^T X (XC,XC) -- all from the same panels.
//
// The DMA is issued from inline asm: hipcc orders every ds_read behind ALL outstanding LDS-DMA it knows of (vmcnt(0)),
// which would serialise the pipeline; the asm loads are invisible to it and are counted by hand (s_waitcnt vmcnt(N) +
// s_barrier before a stage is read) -- cdna_hip_programming.md §5.7.
#pragma once
#include "gemm_f32.cuh"
#include "planes.cuh"

namespace pxr {

typedef __bf16 p3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 p3_bf16x4 __attribute__((ext_vector_type(4)));

// HALF_: the operands are TWO fp16 planes (planes.cuh "h2"), three products per multiply on v_mfma_f32_32x32x16_f16, hi*hi in one
// accumulator set and the two cross terms in a second (gemm_p4.cuh has the big-tile version)
template <int BM_, int BN_, int WGM_, int WGN_, int STAGES_, bool HALF_ = false>
struct P3Cfg {
  static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = WGN_, STAGES = STAGES_;
  static constexpr bool HALF = HALF_;
  static constexpr int NPL = HALF ? 2 : 3;                            // planes per operand
  static constexpr int G = WGM * WGN, NT = 64 * G, BK = 32;
  static constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
  static constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;          // bytes per plane per stage
  static constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
  static constexpr int EPI_LD = BN + 4;                               // floats per row of the epilogue's LDS tile
  static constexpr int EPI_BYTES = BM * EPI_LD * 4;
  static constexpr int LDS_BYTES = STAGES * STAGE > EPI_BYTES ? STAGES * STAGE : EPI_BYTES;
  static constexpr int A_PIECES = NPL * BM / 16, B_PIECES = NPL * BN / 16;   // 1 KiB DMA pieces per K tile
  static constexpr int A_PPW = A_PIECES / G, B_PPW = B_PIECES / G, PPW = A_PPW + B_PPW;
  static constexpr bool PINGPONG = false;
  static_assert(A_PIECES % G == 0 && B_PIECES % G == 0, "DMA pieces must divide evenly among the waves");
  static_assert(WM % 32 == 0 && WN % 32 == 0 && TM >= 1 && TN >= 1, "wave tile = 32x32 blocks");
  static_assert(STAGES >= 2 && STAGES <= 6, "2..6 LDS stages");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(BM * EPI_LD * 4 <= LDS_BYTES, "the epilogue's fp32 tile must fit in the staging LDS");
  struct Acc {
    f32x16 v[TM][TN];
  };
};

// one LDS-DMA piece: lane l's 16 bytes from rs[voff + soff] land at LDS byte address lds_dst + 16 * l (lds_dst
// wave-uniform).  M0 carries the destination and is written in the statement that reads it (the compiler reserves M0 but
// keeps nothing live in it around an asm statement that names it as clobbered; it warns about the clobber).  s_mov, not
// s_add: an asm statement must not touch SCC, the compiler may hold a condition in it across the statement.  The s_nop
// pads the SALU-write -> VMEM-read hazard of the descriptor / soffset registers the compiler may have written just before
// the statement (5 wait states; hipcc pads nothing inside an asm string) and the M0 write -> LDS-DMA one.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void p3_dma16(bufrsrc rs, unsigned voff, unsigned soff, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
               :
               : "v"(voff), "s"(rs), "s"(lds_dst), "s"(soff)
               : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N>
__device__ __forceinline__ void p3_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// wait until at most `tiles` K tiles (PPW pieces each) of this wave's DMA are still in flight
template <int PPW>
__device__ __forceinline__ void p3_wait_tiles(int tiles) {
  static_assert(PPW <= 63, "vmcnt is a 6-bit counter");
  if constexpr (4 * PPW <= 63) {
    if (tiles >= 4) { p3_wait_vm<4 * PPW>(); return; }
  }
  if constexpr (3 * PPW <= 63) {
    if (tiles >= 3) { p3_wait_vm<3 * PPW>(); return; }
  }
  if constexpr (2 * PPW <= 63) {
    if (tiles >= 2) { p3_wait_vm<2 * PPW>(); return; }
  }
  if (tiles >= 1) p3_wait_vm<PPW>();
  else p3_wait_vm<0>();
}

// One operand's DMA plan.  BX = tile extent along the operand's x (BM or BN), PPWX = pieces this wave issues per K tile,
// KC = flavour.  A piece's source = lane * 16 (the only per-lane part: ONE VGPR) + a wave-uniform byte offset (plane,
// 1 KiB run, tile origin: SGPRs, added to the K-tile offset in the instruction's scalar offset).
template <int BX, int PPWX, bool KC, int NPL = 3>
struct P3Operand {
  bufrsrc rs;
  unsigned scal[PPWX];     // wave-uniform byte offset of each piece (loop-invariant)
  unsigned dst0;           // LDS byte offset of this wave's first piece inside a stage (pieces are 1 KiB apart)
  unsigned soff, sstep;    // byte offset of the next K tile to issue / its step
  // m: the matrix; x0: tile origin along x; area: byte offset of this operand's planes inside a stage
  __device__ __forceinline__ void init(const P3Mat& m, int x0, int wave, int lane, unsigned area) {
    rs = make_rsrc(reinterpret_cast<const float*>(m.p), m.ps * NPL * 2);
    // piece id q = wave * PPWX + j: plane-major, then 1 KiB run; its LDS offset inside the operand's area is q * 1024
    dst0 = __builtin_amdgcn_readfirstlane(area + (unsigned)(wave * PPWX * 1024));
#pragma unroll
    for (int j = 0; j < PPWX; ++j) {
      const int q = wave * PPWX + j;
      const int pl = q / (BX / 16), run = q % (BX / 16);
      unsigned o;
      if constexpr (KC) {
        // runs are 16-row groups of the tile's rows inside panel kt
        o = (unsigned)(pl * m.ps * 2) + (unsigned)((x0 + run * 16) * 64);
      } else {
        // runs are (panel run / 2, 16-k-row half run % 2) of the 32 k-rows of tile kt
        o = (unsigned)(pl * m.ps * 2) + (unsigned)(((int64_t)(x0 / 32 + run / 2) * m.pr + (run & 1) * 16) * 64);
      }
      scal[j] = __builtin_amdgcn_readfirstlane(o);
    }
    soff = 0;
    sstep = KC ? (unsigned)(m.pr * 64) : 32u * 64u;
  }
  __device__ __forceinline__ void issue(unsigned stage_base) {
    // lane * 16 is recomputed from the execution mask (two VALU ops) instead of being kept: in the 8-wave tiles every VGPR
    // is an accumulator or a fragment, and a spilled offset would come back as a scratch LOAD inside the K loop -- a VMEM
    // operation in front of which the compiler drains the whole DMA ring (vmcnt(0))
    const unsigned lo = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) << 4;
#pragma unroll
    for (int j = 0; j < PPWX; ++j)
      p3_dma16(rs, lo, __builtin_amdgcn_readfirstlane(soff + scal[j]), __builtin_amdgcn_readfirstlane(stage_base + dst0 + j * 1024));
    soff += sstep;
  }
};

// fragment addressing of one 32-row block: lane (r = lane & 31, h = lane >> 5) gets x = row r, k = 16 kb + 8 h .. + 7 of the
// stage's 32-wide K tile.
template <bool KC>
struct P3Frag {
  int off[2];     // KC: off[0] = row byte offset (the chunk is XORed in per kb); XC: the two transposed reads at kb = 0
  int swz;
  __device__ __forceinline__ void init(int blk_row0, int lane) {
    const int r = lane & 31, h = lane >> 5;
    if constexpr (KC) {
      off[0] = (blk_row0 + r) * 64;
      off[1] = h;
      swz = (r >> 2) & 3;
    } else {
      // 16-lane group g: x block (g & 1) of 16 columns, k half h = g >> 1; lane li of the group addresses k row
      // 8 h + (li >> 2) (+ 4 for the second read), 4 columns starting at 16 (g & 1) + 4 (li & 3)
      const int li = lane & 15, x16 = (lane >> 4) & 1;
      const int chunk = 2 * x16 + ((li & 3) >> 1), inner = (li & 1) * 8;
      const int panel = blk_row0 / 32;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = 8 * h + 4 * t + (li >> 2);               // k row inside the 16-wide block (kb adds 16: same swizzle phase)
        off[t] = panel * 2048 + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4) + inner;
      }
      swz = 0;
    }
  }
  __device__ __forceinline__ p3_bf16x8 read(const char* plane_base, int kb) const {
    if constexpr (KC) {
      return *reinterpret_cast<const p3_bf16x8*>(plane_base + off[0] + ((((kb << 1) + off[1]) ^ swz) << 4));
    } else {
      typedef __attribute__((address_space(3))) p3_bf16x4 lds_v4;
      // rows 16 kb + ...: (row >> 2) & 3 gains (4 kb) & 3 = 0 -> the kb = 0 offsets + 1024 kb
      const p3_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(plane_base + off[0] + kb * 1024));
      const p3_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(plane_base + off[1] + kb * 1024));
      return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }
};

// acc tile (m0, n0) = A_op x B_op over k in [0, K) (K % 32 == 0), operands as planes (see the header).
// ONES (dW only): additionally accumulate in `ones_acc[i]` the products of the A fragments with an all-ones B fragment:
// every column of that 32x32 block is the sum over k of the A rows (= the bias gradient when A = dY^T).
// dbg (timing experiments only, results are wrong): 2 = no DMA, 4 = no fragment reads / MFMAs, 8 = no barriers
template <class Cfg, bool A_KC, bool B_KC, bool EARLY, bool ONES = false>
__device__ __forceinline__ void gemm_p3_mainloop(typename Cfg::Acc& accs, const P3Mat& A, const P3Mat& B, int K, int m0, int n0,
                                                 char* smem, f32x16* ones_acc = nullptr, int dbg = 0) {
  constexpr int TM = Cfg::TM, TN = Cfg::TN, S = Cfg::STAGES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  f32x16 accm[TM][TN], accl[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { accs.v[i][j][e] = 0.f; accm[i][j][e] = 0.f; accl[i][j][e] = 0.f; }
  if constexpr (ONES) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) ones_acc[i][e] = 0.f;
  }
  const int nk = K / 32;
  if (nk <= 0) return;

  constexpr int NPL = Cfg::NPL;
  P3Operand<Cfg::BM, Cfg::A_PPW, A_KC, NPL> opA;
  P3Operand<Cfg::BN, Cfg::B_PPW, B_KC, NPL> opB;
  opA.init(A, m0, wave, lane, 0u);
  opB.init(B, n0, wave, lane, (unsigned)(NPL * Cfg::A_PLANE));
  const unsigned smem_base = (unsigned)(size_t)smem;
  unsigned i_stage = 0;                                       // stage of the next tile to issue
  auto issue = [&]() {
    const unsigned sb = smem_base + i_stage * (unsigned)Cfg::STAGE;
    if (!(dbg & 2)) {
      opA.issue(sb);
      opB.issue(sb);
    }
    i_stage = (i_stage + 1 == (unsigned)S) ? 0u : i_stage + 1;
  };

  P3Frag<A_KC> fa[TM];
  P3Frag<B_KC> fb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[i].init(wm * Cfg::WM + i * 32, lane);
#pragma unroll
  for (int j = 0; j < TN; ++j) fb[j].init(wn * Cfg::WN + j * 32, lane);
  struct Frag {
    p3_bf16x8 a[TM][NPL], b[TN][NPL];
  };
  auto read_frag = [&](Frag& f, int stage, int kb) {
    const char* sa = smem + stage * Cfg::STAGE;
    const char* sb = sa + NPL * Cfg::A_PLANE;
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f.a[i][p] = fa[i].read(sa + p * Cfg::A_PLANE, kb);
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b[j][p] = fb[j].read(sb + p * Cfg::B_PLANE, kb);
    }
  };
  p3_bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  // six products per (block, 16-wide k block); consecutive MFMAs never share an accumulator
  auto mfma = [&](const Frag& f) {
#define PXR_P3_PROD(ACC, PA, PB)                                                                                \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA], f.b[j][PB], ACC[i][j], 0, 0, 0);
#define PXR_P3_PRODH(ACC, PA, PB)                                                                               \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(p3_f16x8, f.a[i][PA]),                \
                                                         __builtin_bit_cast(p3_f16x8, f.b[j][PB]), ACC[i][j], 0, 0, 0);
    if constexpr (Cfg::HALF) {
      PXR_P3_PRODH(accm, 1, 0)       // lo * hi
      PXR_P3_PRODH(accs.v, 0, 0)     // hi * hi
      PXR_P3_PRODH(accm, 0, 1)       // hi * lo
    } else {
      PXR_P3_PROD(accl, 2, 0)        // lo  * hi
      PXR_P3_PROD(accm, 1, 0)        // mid * hi
      PXR_P3_PROD(accs.v, 0, 0)      // hi  * hi
      PXR_P3_PROD(accl, 0, 2)        // hi  * lo
      PXR_P3_PROD(accm, 0, 1)        // hi  * mid
      PXR_P3_PROD(accl, 1, 1)        // mid * mid
    }
#undef PXR_P3_PROD
#undef PXR_P3_PRODH
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
  };

  Frag f0, f1;
  int st = 0;                                               // stage of tile kt
  if constexpr (!EARLY) {
    // ---- tiles kt+1 .. kt+S-2 stay in flight across the barrier of tile kt; fragments are read after it ---------------
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
      if (s < nk) issue();
    for (int kt = 0; kt < nk; ++kt) {
      p3_wait_tiles<Cfg::PPW>(min(nk - 1 - kt, S - 2));    // this wave's pieces of tile kt have landed
      if (!(dbg & 8)) __builtin_amdgcn_s_barrier();         // ... and everybody's; everybody is done reading tile kt-1
      if (kt + S - 1 < nk) issue();                         // into the stage tile kt-1 occupied
      if (!(dbg & 4)) {
        read_frag(f0, st, 0);
        read_frag(f1, st, 1);
        mfma(f0);
        mfma(f1);
      }
      st = (st + 1 == S) ? 0 : st + 1;
    }
  } else {
    // ---- the barrier of tile kt certifies tile kt+1: its first fragments are read under the MFMAs of tile kt, so no wave
    // waits for an LDS round trip after a barrier (one LDS stage more for the same DMA lead) -----------------------------
    static_assert(!EARLY || S >= 3, "early fragment reads need 3 stages");
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
      if (s < nk) issue();
    p3_wait_tiles<Cfg::PPW>(min(nk - 1, S - 2));
    __builtin_amdgcn_s_barrier();                           // tile 0 has landed
    read_frag(f0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int nx = (st + 1 == S) ? 0 : st + 1;
      p3_wait_tiles<Cfg::PPW>(min(max(nk - 2 - kt, 0), S - 3));   // this wave's pieces of tile kt+1 have landed
      if (!(dbg & 8)) __builtin_amdgcn_s_barrier();         // ... everybody's; everybody is done reading tile kt-1
      if (kt + S - 1 < nk) issue();                         // into the stage tile kt-1 occupied
      if (!(dbg & 4)) {
        read_frag(f1, st, 1);
        mfma(f0);
        if (kt + 1 < nk) read_frag(f0, nx, 0);
        mfma(f1);
      }
      st = nx;
    }
  }
  __syncthreads();   // the staging LDS is reused by the epilogues
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) accs.v[i][j][e] += (accm[i][j][e] + accl[i][j][e]);
}

// ---- the same product for a SEQUENCE of A tiles against one B tile as ONE DMA stream (full-catalogue scoring: item tiles
// m_first, m_first + BM, ... x one block of users): the LDS ring never drains between tiles -- tile t+1's first K tiles are
// in flight while `done(t, accs)` consumes tile t's accumulators.  `done` must not touch the staging LDS [0, LDS_BYTES)
// and must not wait on vmcnt(0) more than it has to (that drains the ring); barriers inside it: p3_lds_barrier().  KC x KC.
__device__ __forceinline__ void p3_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}
template <class Cfg, class TileFn>
__device__ __forceinline__ void gemm_p3_stream(const P3Mat& A, const P3Mat& B, int K, int m_first, int n0, int n_tiles, char* smem,
                                               TileFn&& done) {
  static_assert(!Cfg::HALF, "the scoring stream runs on the three bf16 planes");
  constexpr int TM = Cfg::TM, TN = Cfg::TN, S = Cfg::STAGES;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  const int nk = K / 32;
  const int total = n_tiles * nk;
  if (total <= 0) return;
  typename Cfg::Acc accs;
  f32x16 accm[TM][TN], accl[TM][TN];
  auto zero = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { accs.v[i][j][e] = 0.f; accm[i][j][e] = 0.f; accl[i][j][e] = 0.f; }
  };
  zero();
  P3Operand<Cfg::BM, Cfg::A_PPW, true> opA;
  P3Operand<Cfg::BN, Cfg::B_PPW, true> opB;
  opA.init(A, m_first, wave, lane, 0u);
  opB.init(B, n0, wave, lane, (unsigned)(3 * Cfg::A_PLANE));
  const unsigned smem_base = (unsigned)(size_t)smem;
  const unsigned stepA = (unsigned)(A.pr * 64), stepB = (unsigned)(B.pr * 64);
  unsigned i_stage = 0, i_kt = 0, i_tile_off = 0;           // next K tile to issue: stage, k index, byte offset of its item tile
  auto issue = [&]() {
    const unsigned sb = smem_base + i_stage * (unsigned)Cfg::STAGE;
    opA.soff = i_kt * stepA + i_tile_off;
    opB.soff = i_kt * stepB;
    opA.issue(sb);
    opB.issue(sb);
    i_stage = (i_stage + 1 == (unsigned)S) ? 0u : i_stage + 1;
    if (++i_kt == (unsigned)nk) { i_kt = 0; i_tile_off += (unsigned)(Cfg::BM * 64); }
  };
  // fragment addresses: FOUR per-lane byte offsets in all (A and B, k blocks 0 and 1: row * 64 + swizzled chunk); the stage
  // is a scalar add per K tile, plane and 32-row block are immediate offsets of the ds_read (< 64 KiB)
  static_assert(2 * Cfg::A_PLANE + (TM - 1) * 2048 < 65536 && 2 * Cfg::B_PLANE + (TN - 1) * 2048 < 65536, "ds_read immediate offsets");
  int la[2], lb[2];
  {
    const int h = lane >> 5, r = lane & 31, swz = (r >> 2) & 3;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int co = (((kb << 1) + h) ^ swz) << 4;
      la[kb] = (wm * Cfg::WM + r) * 64 + co;
      lb[kb] = 3 * Cfg::A_PLANE + (wn * Cfg::WN + r) * 64 + co;
    }
  }
  struct Frag {
    p3_bf16x8 a[TM][3], b[TN][3];
  };
  auto read_frag = [&](Frag& f, int stage, int kb) {
    const char* sa = smem + stage * Cfg::STAGE + la[kb];
    const char* sb = smem + stage * Cfg::STAGE + lb[kb];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f.a[i][p] = *reinterpret_cast<const p3_bf16x8*>(sa + p * Cfg::A_PLANE + i * 2048);
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b[j][p] = *reinterpret_cast<const p3_bf16x8*>(sb + p * Cfg::B_PLANE + j * 2048);
    }
  };
  auto mfma = [&](const Frag& f) {
#define PXR_P3_PROD(ACC, PA, PB)                                                                                \
  _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
      ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA], f.b[j][PB], ACC[i][j], 0, 0, 0);
    PXR_P3_PROD(accl, 2, 0)
    PXR_P3_PROD(accm, 1, 0)
    PXR_P3_PROD(accs.v, 0, 0)
    PXR_P3_PROD(accl, 0, 2)
    PXR_P3_PROD(accm, 0, 1)
    PXR_P3_PROD(accl, 1, 1)
#undef PXR_P3_PROD
  };
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < total) issue();
  Frag f0;
  int st = 0, kt = 0, tile = 0;
  for (int it = 0; it < total; ++it) {
    p3_wait_tiles<Cfg::PPW>(min(total - 1 - it, S - 2));
    __builtin_amdgcn_s_barrier();
    if (it + S - 1 < total) issue();
    // one fragment set at a time (the 8-wave tiles this loop serves hold 192 accumulator registers per lane; the second
    // wave of the SIMD covers the LDS round trip between the two halves)
    read_frag(f0, st, 0);
    mfma(f0);
    __builtin_amdgcn_sched_barrier(0);
    read_frag(f0, st, 1);
    mfma(f0);
    st = (st + 1 == S) ? 0 : st + 1;
    if (++kt == nk) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) accs.v[i][j][e] += (accm[i][j][e] + accl[i][j][e]);
      done(tile, accs);
      zero();
      kt = 0;
      ++tile;
    }
  }
}

// ---- row epilogue: the accumulators go through LDS ([BM][BN + 4] fp32, the staging buffers are free) so that every thread
// owns 8 CONSECUTIVE columns of a row: 16-byte loads of bias / aux, 16-byte stores of C, one 16-byte store per output
// plane.  `fn(it, row, col, nv, v)` is called once per (row, 8-column chunk) that starts inside the matrix (it = the thread's
// chunk counter, see P3ChunkMap); nv = min(8, N - col) of its columns exist. --------------------------------------------------------------------------------------------------
template <class Cfg, class Fn>
__device__ __forceinline__ void p3_row_epilogue(const typename Cfg::Acc& accs, char* smem, int M, int N, int m0, int n0, Fn&& fn) {
  float* t = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN, h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * Cfg::WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        t[row * Cfg::EPI_LD + wn * Cfg::WN + j * 32 + r] = accs.v[i][j][e];
      }
  __syncthreads();
  constexpr int CPR = Cfg::BN / 8;                       // chunks per row
  constexpr int CPT = Cfg::BM * CPR / Cfg::NT;           // chunks per thread: chunk it of thread t is (row it * NT / CPR + t / CPR, t % CPR)
  static_assert(Cfg::NT % CPR == 0 && (Cfg::BM * CPR) % Cfg::NT == 0, "chunk map");
  const int c8 = (threadIdx.x % CPR) * 8;
#pragma unroll
  for (int it = 0; it < CPT; ++it) {
    const int row = it * (Cfg::NT / CPR) + threadIdx.x / CPR;
    if (m0 + row >= M || n0 + c8 >= N) continue;
    const float4 a = *reinterpret_cast<const float4*>(t + row * Cfg::EPI_LD + c8);
    const float4 b = *reinterpret_cast<const float4*>(t + row * Cfg::EPI_LD + c8 + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    fn(it, m0 + row, n0 + c8, min(8, N - n0 - c8), v);
  }
}
// the same chunk map, for operands the epilogue READS (bias, residual gradient, saved activation derivative): fetched before
// the main loop, they arrive under the MFMAs instead of stalling every thread after its last one
template <class Cfg>
struct P3ChunkMap {
  static constexpr int CPR = Cfg::BN / 8, CPT = Cfg::BM * CPR / Cfg::NT;
  static __device__ __forceinline__ int col8() { return (threadIdx.x % CPR) * 8; }
  static __device__ __forceinline__ int row(int it) { return it * (Cfg::NT / CPR) + threadIdx.x / CPR; }
};

}  // namespace pxr
