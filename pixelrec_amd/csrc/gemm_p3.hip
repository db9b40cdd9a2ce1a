// gemm_p3.hip -- fp32 GEMMs on the bf16 matrix pipe from pre-split operands ("planes" in panel layout, gemm_p3.cuh) + the
// plane producer for operands nobody has split yet (pxr_split_planes_f32: weights after an optimizer step, the item table
// before an evaluation).
//
// Reference call sites replaced: the nn.Linear forwards of the sequence block (layers.py:586-588,613,666,669), their
// autograd input / weight gradients, the full-catalogue scoring product (sasrec.py:112) -- the same products pxr_gemm_f32
// computes, from operands that were split once by their producer instead of once per reading tile.
#include "gemm_p4.cuh"

#include <cstdlib>

namespace pxr {

// one element of the epilogues of gemm_f32.cuh::epi_store (same formulas, same order of operations)
template <int EPI>
__device__ __forceinline__ float p3_epi_elem(float v, float bv, float av, float& aux_out, int act) {
  if constexpr (EPI == EPI_BIAS) {
    v += bv;
  } else if constexpr (EPI == EPI_BIAS_GELU) {
    v += bv;
    aux_out = v;
    v = gelu_erf(v);
  } else if constexpr (EPI == EPI_BIAS_GELU_GRAD) {
    v += bv;
    aux_out = dgelu_erf(v);
    v = gelu_erf(v);
  } else if constexpr (EPI == EPI_MUL) {
    v *= av;
  } else if constexpr (EPI == EPI_MUL_DGELU) {
    v *= dgelu_erf(av);
  } else if constexpr (EPI == EPI_ADD) {
    v += av;
  } else if constexpr (EPI == EPI_BIAS_ADD) {
    v = (v + bv) + av;
  } else if constexpr (EPI == EPI_BIAS_QGELU_GRAD) {
    v += bv;
    const float sg = sigmoid_1702(v);
    aux_out = sg + 1.702f * v * sg * (1.0f - sg);
    v = v * sg;
  } else if constexpr (EPI == EPI_BIAS_RELU) {
    v = fmaxf(v + bv, 0.f);
  } else if constexpr (EPI == EPI_BIAS_QGELU) {
    v += bv;
    v = v * sigmoid_1702(v);
  } else if constexpr (EPI == EPI_BIAS_ACT_GRAD) {
    v += bv;
    float dv;
    if (act == ACT_RELU) {
      dv = v > 0.f ? 1.f : 0.f; v = fmaxf(v, 0.f);
    } else if (act == ACT_SWISH) {
      const float sg = 1.0f / (1.0f + __expf(-v));
      dv = sg + v * sg * (1.0f - sg); v = v * sg;
    } else if (act == ACT_TANH) {
      const float th = tanhf(v);
      dv = 1.0f - th * th; v = th;
    } else {
      const float sg = 1.0f / (1.0f + __expf(-v));
      dv = sg * (1.0f - sg); v = sg;
    }
    aux_out = dv;
  }
  return v;
}
template <int EPI>
struct P3EpiTraits {
  static constexpr bool WRITES_AUX = (EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_GRAD || EPI == EPI_BIAS_QGELU_GRAD ||
                                      EPI == EPI_BIAS_ACT_GRAD);
};

struct P3Args {
  P3Mat A, B, Cp;          // Cp.p == nullptr: no output planes
  float* C;                // nullptr: the fp32 output is not stored (the consumer reads the planes)
  int64_t ldc;
  int M, N, K;
  const float* bias;
  float* aux;
  int64_t ldaux;
  int tiles_m, tiles_n, n_fastest, act, dbg;
  int group_m;             // > 1: tiles are walked in groups of group_m row tiles x all column tiles (see gemm_p3_kernel)
  // fp16 two-plane operands (Cfg::HALF): the operands hold A 2^a_exp and B 2^b_exp; the exponents are immediates or, when the
  // pointer is given, read from device memory (a producer chose them on the device: gradients, weights re-split every step)
  int a_exp, b_exp;
  const int* a_exp_dev;
  const int* b_exp_dev;
  const int* c_exp_dev;    // h2 output planes hold C 2^(*c_exp_dev) (null: unit scale)
  int cp_fmt;              // format of the output planes: PXR_PLANES_BF16X3 | PXR_PLANES_H2
  int32_t* status;         // status word (fp16 range check of h2 output planes) or null
};

template <class Cfg, bool PP>
struct P4Map {
  static constexpr int CPT = 1;
};
template <class Cfg>
struct P4Map<Cfg, true> : P4ChunkMap<Cfg> {};

template <class Cfg, bool B_KC, int EPI, bool EARLY>
__global__ void __launch_bounds__(Cfg::NT) gemm_p3_kernel(const P3Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  int tm = g.n_fastest ? t / g.tiles_n : t % g.tiles_m, tn = g.n_fastest ? t % g.tiles_n : t / g.tiles_m;
  if (g.group_m > 1) {
    // L2 blocking for tall GEMMs (ViT tower: M = 69 344 tokens): with the row tile slowest, the ~32 workgroups an XCD runs at
    // a time are one row tile x 24+ column tiles, so the whole B operand streams through that XCD's 4 MB L2 once per ROW TILE
    // (measured: 3.2x the algorithmic bytes over the fabric).  Walking group_m row tiles per column tile makes the concurrent
    // set group_m x (32 / group_m) tiles: B streams once per group_m row tiles.
    const int per_group = g.group_m * g.tiles_n;
    const int grp = t / per_group, r = t - grp * per_group;
    const int gsize = min(g.group_m, g.tiles_m - grp * g.group_m);
    tm = grp * g.group_m + r % gsize;
    tn = r / gsize;
  }
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
  // 16-byte accesses when every row of C / aux starts 16-byte aligned and the chunk is whole; scalar otherwise (ragged N)
  const bool vec_ok = (g.N % 8 == 0) && (g.ldc % 4 == 0) && (g.ldaux % 4 == 0);
  using Map = P3ChunkMap<Cfg>;
  // operands the epilogue reads: the lockstep tiles (few chunks per thread) request them in one go right after the main loop, the
  // ping-pong tiles (8 chunks per thread and pass) read them chunk by chunk in the epilogue
  constexpr bool PRE = !Cfg::PINGPONG;
  constexpr bool PP = Cfg::PINGPONG;
  constexpr bool PRE_AUX = EpiTraits<EPI>::READS_AUX && (PP ? (int)P4Map<Cfg, PP>::CPT <= 4 : Map::CPT <= 4);
  constexpr int APRE = !PRE_AUX ? 1 : (PP ? (int)P4Map<Cfg, PP>::CPT : (int)Map::CPT);
  float bpre[8], apre[APRE][8];
  const int pc = n0 + Map::col8();
  const bool pre_ok = PRE && vec_ok && pc + 8 <= g.N;
  typename Cfg::Acc accs;
  if constexpr (Cfg::PINGPONG) gemm_p4_mainloop<Cfg, true, B_KC>(accs, g.A, g.B, g.K, m0, n0, smem);
  else gemm_p3_mainloop<Cfg, true, B_KC, EARLY>(accs, g.A, g.B, g.K, m0, n0, smem, nullptr, g.dbg);
  if (g.dbg & 32) return;          // LAB: no epilogue at all
  // operands the epilogue reads (bias, residual / saved derivative): requested right AFTER the main loop, so that they arrive
  // under the accumulators' trip through LDS.  (Rounds 3-4 requested them BEFORE the main loop "to arrive under the MFMAs": with
  // every register an accumulator or a fragment the compiler parked them in scratch across the loop -- a store in the prologue and
  // a dependent scratch load per chunk in the epilogue, tools/isa_resources.py: 48 / 144 B per lane in the lockstep tiles.)
  if constexpr (EpiTraits<EPI>::HAS_BIAS && PRE) {
    if (pre_ok) {
      const float4 b0 = *reinterpret_cast<const float4*>(g.bias + pc), b1 = *reinterpret_cast<const float4*>(g.bias + pc + 4);
      bpre[0] = b0.x; bpre[1] = b0.y; bpre[2] = b0.z; bpre[3] = b0.w; bpre[4] = b1.x; bpre[5] = b1.y; bpre[6] = b1.z; bpre[7] = b1.w;
    }
  }
  if constexpr (PRE_AUX && PRE) {
#pragma unroll
    for (int it = 0; it < Map::CPT; ++it) {
      const int row = m0 + Map::row(it);
      if (pre_ok && row < g.M) {
        const float* ap = g.aux + (int64_t)row * g.ldaux + pc;
        const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
        apre[it][0] = a0.x; apre[it][1] = a0.y; apre[it][2] = a0.z; apre[it][3] = a0.w;
        apre[it][4] = a1.x; apre[it][5] = a1.y; apre[it][6] = a1.z; apre[it][7] = a1.w;
      }
    }
  }
  float c_scale = 1.0f;
  if constexpr (Cfg::HALF) {
    const int ea = g.a_exp_dev ? *g.a_exp_dev : g.a_exp, eb = g.b_exp_dev ? *g.b_exp_dev : g.b_exp;
    const float sc = ldexpf(1.0f, -(ea + eb));
    if (g.c_exp_dev) c_scale = ldexpf(1.0f, *g.c_exp_dev);
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accs.v[i][j][e] *= sc;
  }
  auto chunk = [&](int it, int row, int col, int nv, float (&v)[8]) __attribute__((always_inline)) {
    float bv[8], av[8], ao[8];
    const bool vec = vec_ok && nv == 8;
    if constexpr (EpiTraits<EPI>::HAS_BIAS) {
      if (vec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = bpre[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = e < nv ? g.bias[col + e] : 0.f;
      }
    }
    if constexpr (EpiTraits<EPI>::READS_AUX) {
      const float* ap = g.aux + (int64_t)row * g.ldaux + col;
      if (vec) {
        if constexpr (PRE_AUX) {
#pragma unroll
          for (int e = 0; e < 8; ++e) av[e] = apre[it][e];
        } else {
          const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
          av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = e < nv ? ap[e] : 0.f;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = p3_epi_elem<EPI>(v[e], EpiTraits<EPI>::HAS_BIAS ? bv[e] : 0.f, EpiTraits<EPI>::READS_AUX ? av[e] : 0.f, ao[e], g.act);
    if (g.dbg & 16) return;        // LAB: the epilogue's arithmetic and LDS trip, none of its stores
    if constexpr (P3EpiTraits<EPI>::WRITES_AUX) {
      float* ap = g.aux + (int64_t)row * g.ldaux + col;
      if (vec) {
        *reinterpret_cast<float4*>(ap) = make_float4(ao[0], ao[1], ao[2], ao[3]);
        *reinterpret_cast<float4*>(ap + 4) = make_float4(ao[4], ao[5], ao[6], ao[7]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nv) ap[e] = ao[e];
      }
    }
    if (g.C != nullptr) {
      float* cp = g.C + (int64_t)row * g.ldc + col;
      if (vec) {
        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nv) cp[e] = v[e];
      }
    }
    if (g.Cp.p != nullptr) {                                   // (output planes: N % 32 == 0, chunks are whole)
      if constexpr (Cfg::HALF) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= c_scale;             // (after the fp32 / aux stores: those stay unscaled)
      }
      px_store8(g.Cp, g.cp_fmt, g.status, row, col, v);
    }
  };
  if constexpr (PP) {
    p4_row_epilogue<Cfg>(accs, smem, g.M, g.N, m0, n0,
                         [&](int it, int row, int col, int nv) {
                           if (!(vec_ok && nv == 8)) return;
                           if constexpr (EpiTraits<EPI>::HAS_BIAS) {
                             if (it == 0) {          // the thread's column chunk is the same for all its rows of a pass
                               const float4 b0 = *reinterpret_cast<const float4*>(g.bias + col), b1 = *reinterpret_cast<const float4*>(g.bias + col + 4);
                               bpre[0] = b0.x; bpre[1] = b0.y; bpre[2] = b0.z; bpre[3] = b0.w; bpre[4] = b1.x; bpre[5] = b1.y; bpre[6] = b1.z; bpre[7] = b1.w;
                             }
                           }
                           if constexpr (PRE_AUX) {
                             const float* ap = g.aux + (int64_t)row * g.ldaux + col;
                             const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
                             apre[it][0] = a0.x; apre[it][1] = a0.y; apre[it][2] = a0.z; apre[it][3] = a0.w;
                             apre[it][4] = a1.x; apre[it][5] = a1.y; apre[it][6] = a1.z; apre[it][7] = a1.w;
                           }
                         },
                         chunk);
  } else {
    p3_row_epilogue<Cfg>(accs, smem, g.M, g.N, m0, n0, chunk);
  }
}

// x[rows, cols] (row stride ldx floats) -> planes (panel layout); a thread converts 8 consecutive values of a row
__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int cols8, P3Mat out,
                                                           int xcd) {
  // xcd: workgroups of one XCD (blockIdx % 8) take a CONTIGUOUS range of rows -- the range the same XCD's GEMM tiles read
  const int64_t i = (int64_t)(xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * 256 + threadIdx.x;
  if (i >= rows * cols8) return;
  const int64_t row = i / cols8;
  const int c = (int)(i % cols8) * 8;
  const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c);
  const float4 b = *reinterpret_cast<const float4*>(x + row * ldx + c + 4);
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  p3_store8(out, row, c, v);
}

// All dW[N,K] = dY[T,N]^T X[T,K] (+ db[N] = column sums of dY) of a backward pass in ONE launch, from the planes of dY and X
// (both x-contiguous operands: the token reduction runs along the panels' rows, fragments come from transposing LDS reads).
// The planes' rows T .. round_up(T, 32) - 1 must be zero.  db: the A fragments are also multiplied by an all-ones B fragment
// (3 extra MFMAs per 16 tokens in the wn == 0 waves of the tn == 0 tiles): every column of that block is the column sum.
struct P3DwProblem {
  P3Mat dy, x;
  float* dW;
  float* db;
  int T, N, K;              // tokens, out features, in features
  int tile_begin, tiles_m;
  int dy_exp, x_exp;        // fp16 two-plane operands (Cfg::HALF): the planes hold dY 2^dy_exp, X 2^x_exp (immediate or from the device)
  const int* dy_exp_dev;
  const int* x_exp_dev;
};
struct P3DwGroup {
  P3DwProblem p[DW_MAX];
  int n, total_tiles;
  unsigned* flags;         // KS == 2: one zero-initialised word per tile (left zeroed)
  unsigned* status;
};
int pxr_stream_flags(hipStream_t st, unsigned** flags, int* n_flags);      // gemm_f32.hip
// KS == 2 (split-K, ping-pong tiles only): two workgroups per tile, each reducing HALF of the tokens -- the launches whose tiles fill
// half the chip or less (the sequence block's four matrices per layer: 128 tiles of 256x128) run on all of it.  The first half STORES
// its tile and raises the tile's flag; the second half (the next logical index: dispatched later, normally on the same XCD) waits for
// it, ADDS its own sum and lowers the flag -- a fixed order (deterministic bits), no zero-filled output, no atomics on the data.
template <class Cfg, bool EARLY, int KS = 1>
__global__ void __launch_bounds__(Cfg::NT) grouped_dw_p3_kernel(const P3DwGroup g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(KS == 1 || (KS == 2 && Cfg::PINGPONG), "split-K: the ping-pong tiles");
  const int tl = xcd_remap(blockIdx.x, KS * g.total_tiles);
  const int t = tl / KS, half = tl % KS;
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < g.n; ++i)
    if (t >= g.p[i].tile_begin) pi = i;
  const P3DwProblem& P = g.p[pi];
  const int local = t - P.tile_begin;
  const int tm = local % P.tiles_m, tn = local / P.tiles_m;
  const int m0 = tm * Cfg::BM, n0 = tn * Cfg::BN;
  const int kpad = (P.T + 31) & ~31;
  typename Cfg::Acc accs;
  f32x16 ones_acc[Cfg::TM];
  const bool do_bias = (P.db != nullptr) && (tn == 0);      // block-uniform
  if constexpr (KS == 2) {
    // tokens [0, k0) | [k0, kpad), k0 a multiple of 32: a row range of both planes matrices (32 elements per panel row)
    const int k0 = (kpad / 64) * 32;
    const int kb = half ? k0 : 0, kl = half ? kpad - k0 : k0;
    const P3Mat dyh{P.dy.p + (int64_t)kb * 32, P.dy.ps, P.dy.pr}, xh{P.x.p + (int64_t)kb * 32, P.x.ps, P.x.pr};
    if (do_bias) gemm_p4_mainloop<Cfg, false, false, true>(accs, dyh, xh, kl, m0, n0, smem, ones_acc);
    else gemm_p4_mainloop<Cfg, false, false, false>(accs, dyh, xh, kl, m0, n0, smem);
    if (half) {
      if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(&g.flags[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1u << 22)) {        // ~seconds: never in a healthy launch; flag it instead of hanging the GPU
            if (g.status) atomicOr(g.status, (unsigned)PXR_STATUS_GEMM_TIMEOUT);
            break;
          }
        }
      }
      __syncthreads();                       // (the loads below bypass the caches: no acquire fence -- an agent-scope fence is a full
    }                                        //  L2 write-back + invalidate per wave, ~100 us per launch at 256 workgroups: measured)
  } else if constexpr (Cfg::PINGPONG) {
    if (do_bias) gemm_p4_mainloop<Cfg, false, false, true>(accs, P.dy, P.x, kpad, m0, n0, smem, ones_acc);
    else gemm_p4_mainloop<Cfg, false, false, false>(accs, P.dy, P.x, kpad, m0, n0, smem);
  } else {
    if (do_bias) gemm_p3_mainloop<Cfg, false, false, EARLY, true>(accs, P.dy, P.x, kpad, m0, n0, smem, ones_acc);
    else gemm_p3_mainloop<Cfg, false, false, EARLY, false>(accs, P.dy, P.x, kpad, m0, n0, smem);
  }
  if constexpr (Cfg::HALF) {
    const int ey = P.dy_exp_dev ? *P.dy_exp_dev : P.dy_exp, ex = P.x_exp_dev ? *P.x_exp_dev : P.x_exp;
    const float sc = ldexpf(1.0f, -(ey + ex)), sb = ldexpf(1.0f, -ey);
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i) {
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accs.v[i][j][e] *= sc;
      if (do_bias) {
#pragma unroll
        for (int e = 0; e < 16; ++e) ones_acc[i][e] *= sb;
      }
    }
  }
  float* dW = P.dW;
  const int64_t ldw = P.K;
  const bool vec_ok = (P.K % 8 == 0);
  // split-K hand-over: the first half's stores and the second half's loads of the partial tile go THROUGH the caches (sc1 | sc0:
  // write-through / bypass, as the stream-K partials of gemm_f32.hip), so the pair needs no fence whichever XCDs it runs on
  constexpr int COH = (1 << 4) | 1;
  const bool add = (KS == 2) && half;          // the second half of a split tile adds to what the first one stored
  const bool publish = (KS == 2) && !half;
  const bufrsrc rsw = make_rsrc(dW, (int64_t)P.N * P.K * 4);
  auto store = [&](int, int row, int col, int nv, float (&v)[8]) {
    float* cp = dW + (int64_t)row * ldw + col;
    const unsigned off = (unsigned)(((int64_t)row * ldw + col) * 4);
    if (vec_ok && nv == 8) {
      if (add) {
        const auto a = __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, COH), b = __builtin_amdgcn_raw_buffer_load_b128(rsw, off + 16, 0, COH);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(a[e]); v[4 + e] += __uint_as_float(b[e]); }
      }
      if (publish) {
        p3_u32x4 a, b;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = __float_as_uint(v[e]); b[e] = __float_as_uint(v[4 + e]); }
        __builtin_amdgcn_raw_buffer_store_b128(a, rsw, off, 0, COH);
        __builtin_amdgcn_raw_buffer_store_b128(b, rsw, off + 16, 0, COH);
      } else {
        *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < nv) {
          float o = v[e];
          if (add) o += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsw, off + 4 * e, 0, COH));
          if (publish) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rsw, off + 4 * e, 0, COH);
          else cp[e] = o;
        }
    }
  };
  if constexpr (Cfg::PINGPONG) p4_row_epilogue<Cfg>(accs, smem, P.N, P.K, m0, n0, [](int, int, int, int) {}, store);
  else p3_row_epilogue<Cfg>(accs, smem, P.N, P.K, m0, n0, store);
  if (do_bias) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN, h = lane >> 5, r = lane & 31;
    if (wn == 0 && r == 0) {
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * Cfg::WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row < P.N) {
            float o = ones_acc[i][e];
            if constexpr (KS == 2) {
              const bufrsrc rsb = make_rsrc(P.db, (int64_t)P.N * 4);
              if (add) o += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsb, (unsigned)row * 4u, 0, COH));
              if (publish) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), rsb, (unsigned)row * 4u, 0, COH);
              else P.db[row] = o;
            } else {
              P.db[row] = o;
            }
          }
        }
    }
  }
  if constexpr (KS == 2) {
    __builtin_amdgcn_s_waitcnt(0);           // every lane's stores acknowledged at the coherence point (first half); loads consumed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&g.flags[t], half ? 0u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <class Cfg, bool EARLY, int KS = 1>
static int launch_dw_p3(P3DwGroup& g, hipStream_t st) {
  int tiles = 0;
  for (int i = 0; i < g.n; ++i) {
    g.p[i].tile_begin = tiles;
    g.p[i].tiles_m = (g.p[i].N + Cfg::BM - 1) / Cfg::BM;
    tiles += g.p[i].tiles_m * ((g.p[i].K + Cfg::BN - 1) / Cfg::BN);
  }
  g.total_tiles = tiles;
  if constexpr (KS == 2) {
    if (2 * tiles > pxr_cu_count()) {
      pxr_set_error("pxr_grouped_dw_planes_f32: split-K needs both halves of its %d tiles resident at once, the device has %d CUs",
                    tiles, pxr_cu_count());
      return PXR_ERR_BAD_ARG;
    }
    int n_flags = 0;
    const int rc = pxr_stream_flags(st, &g.flags, &n_flags);
    if (rc != PXR_OK) return rc;
    if (tiles > n_flags) {
      pxr_set_error("pxr_grouped_dw_planes_f32: split-K launch with %d tiles (max %d)", tiles, n_flags);
      return PXR_ERR_BAD_ARG;
    }
    g.status = (unsigned*)pxr_status_word();
  }
  auto kern = grouped_dw_p3_kernel<Cfg, EARLY, KS>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      pxr_set_error("pxr_grouped_dw_planes_f32: cannot reserve %d bytes of LDS", Cfg::LDS_BYTES);
      return PXR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(KS * tiles)), dim3(Cfg::NT), Cfg::LDS_BYTES, st, g);
  return pxr_check_launch("pxr_grouped_dw_planes_f32");
}

// up to 16 matrices in one launch (the weight matrices of the sequence block after an optimizer step)
struct SplitMulti {
  const float* x[16]; int64_t ldx[16]; int rows[16], cols8[16]; P3Mat out[16];
  int64_t begin[17];
  int n;
  int fmt;                 // PXR_PLANES_BF16X3 | PXR_PLANES_H2 (all matrices of the launch)
  float scale[16];         // h2: the power of two each matrix is multiplied by before the split
  int32_t* status;
};
__global__ void __launch_bounds__(256) split_planes_multi_kernel(const SplitMulti m) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m.begin[m.n]) return;
  int pi = 0;
#pragma unroll 1
  for (int k = 1; k < m.n; ++k)
    if (i >= m.begin[k]) pi = k;
  const int64_t li = i - m.begin[pi];
  const int64_t row = li / m.cols8[pi];
  const int c = (int)(li % m.cols8[pi]) * 8;
  const float* src = m.x[pi] + row * m.ldx[pi] + c;
  const float4 a = *reinterpret_cast<const float4*>(src);
  const float4 b = *reinterpret_cast<const float4*>(src + 4);
  float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  if (m.fmt == PXR_PLANES_H2) {
    const float sc = m.scale[pi];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= sc;
  }
  px_store8(m.out[pi], m.fmt, m.status, row, c, v);
}

template <class Cfg, bool B_KC, int EPI, bool EARLY>
static int launch_p3(P3Args& g, hipStream_t st) {
  g.tiles_m = (g.M + Cfg::BM - 1) / Cfg::BM;
  g.tiles_n = (g.N + Cfg::BN - 1) / Cfg::BN;
  static const int xcd_env = getenv("PXR_GEMM_XCD") ? atoi(getenv("PXR_GEMM_XCD")) : -1;
  g.n_fastest = xcd_env >= 0 ? xcd_env : (g.M > g.N ? 1 : 0);
  static const int group_env = getenv("PXR_P3_GROUP_M") ? atoi(getenv("PXR_P3_GROUP_M")) : 0;
  g.group_m = (g.n_fastest && group_env > 1 && g.tiles_m >= 8 * group_env && g.tiles_n >= 4) ? group_env : 1;
  g.dbg = getenv("PXR_P3_DBG") ? atoi(getenv("PXR_P3_DBG")) : 0;   // timing experiments only (gemm_p3.cuh)
  auto kern = gemm_p3_kernel<Cfg, B_KC, EPI, EARLY>;
  static bool attr_set = false;     // > 64 KB of dynamic LDS needs the opt-in once per kernel
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES) != hipSuccess) {
      (void)hipGetLastError();
      pxr_set_error("pxr_gemm_planes_f32: cannot reserve %d bytes of LDS", Cfg::LDS_BYTES);
      return PXR_ERR_LAUNCH;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(g.tiles_m * g.tiles_n)), dim3(Cfg::NT), Cfg::LDS_BYTES, st, g);
  return pxr_check_launch("pxr_gemm_planes_f32");
}

template <class Cfg, bool EARLY>
static int epi_p3(int b_kc, int epilogue, P3Args& g, hipStream_t st) {
#define PXR_P3(BK_, E) \
  case E: return launch_p3<Cfg, BK_, E, EARLY>(g, st)
  if (b_kc) {
    switch (epilogue) {
      PXR_P3(true, EPI_NONE); PXR_P3(true, EPI_BIAS); PXR_P3(true, EPI_BIAS_GELU); PXR_P3(true, EPI_BIAS_GELU_GRAD);
      PXR_P3(true, EPI_BIAS_ACT_GRAD); PXR_P3(true, EPI_BIAS_ADD); PXR_P3(true, EPI_BIAS_QGELU); PXR_P3(true, EPI_BIAS_QGELU_GRAD);
      PXR_P3(true, EPI_BIAS_RELU);
    }
  } else {
    switch (epilogue) {
      PXR_P3(false, EPI_NONE); PXR_P3(false, EPI_ADD); PXR_P3(false, EPI_MUL);
    }
  }
#undef PXR_P3
  pxr_set_error("pxr_gemm_planes_f32: flavour b_kc=%d / epilogue %d is not instantiated", b_kc, epilogue);
  return PXR_ERR_BAD_ARG;
}

// the flavours of the fp16 two-plane GEMM: forward (weights k-contiguous) and input gradient (weights x-contiguous)
template <class Cfg>
static int epi_h2(int b_kc, int epilogue, P3Args& g, hipStream_t st) {
  if (b_kc) {
    switch (epilogue) {
      case EPI_NONE: return launch_p3<Cfg, true, EPI_NONE, false>(g, st);
      case EPI_BIAS: return launch_p3<Cfg, true, EPI_BIAS, false>(g, st);
      case EPI_BIAS_GELU: return launch_p3<Cfg, true, EPI_BIAS_GELU, false>(g, st);
      case EPI_BIAS_GELU_GRAD: return launch_p3<Cfg, true, EPI_BIAS_GELU_GRAD, false>(g, st);
      case EPI_BIAS_ACT_GRAD: return launch_p3<Cfg, true, EPI_BIAS_ACT_GRAD, false>(g, st);
      case EPI_BIAS_ADD: return launch_p3<Cfg, true, EPI_BIAS_ADD, false>(g, st);
      case EPI_BIAS_QGELU: return launch_p3<Cfg, true, EPI_BIAS_QGELU, false>(g, st);
      case EPI_BIAS_QGELU_GRAD: return launch_p3<Cfg, true, EPI_BIAS_QGELU_GRAD, false>(g, st);
      case EPI_BIAS_RELU: return launch_p3<Cfg, true, EPI_BIAS_RELU, false>(g, st);
    }
  } else {
    switch (epilogue) {
      case EPI_NONE: return launch_p3<Cfg, false, EPI_NONE, false>(g, st);
      case EPI_ADD: return launch_p3<Cfg, false, EPI_ADD, false>(g, st);
      case EPI_MUL: return launch_p3<Cfg, false, EPI_MUL, false>(g, st);
    }
  }
  pxr_set_error("pxr_gemm_h2_f32: flavour b_kc=%d / epilogue %d is not instantiated", b_kc, epilogue);
  return PXR_ERR_BAD_ARG;
}

}  // namespace pxr

using namespace pxr;

extern "C" int pxr_split_planes_f32(const float* x, int64_t rows, int64_t cols, int64_t ldx, void* planes, int64_t plane_stride,
                                    int64_t panel_rows, void* stream) {
  PXR_REQUIRE(x && planes, "pxr_split_planes_f32: null operand");
  PXR_REQUIRE(rows >= 0 && cols >= 0 && cols % 32 == 0 && ldx % 4 == 0, "pxr_split_planes_f32: cols must be a multiple of 32, ldx of 4");
  PXR_REQUIRE(panel_rows >= rows && panel_rows % 16 == 0 && plane_stride >= panel_rows * cols && plane_stride % 8 == 0,
              "pxr_split_planes_f32: panel_rows must be a multiple of 16 >= rows, plane_stride >= panel_rows * cols");
  PXR_REQUIRE((((uintptr_t)x | (uintptr_t)planes) & 15) == 0, "pxr_split_planes_f32: operands must be 16-byte aligned");
  const int64_t n = rows * (cols / 8);
  if (n == 0) return PXR_OK;
  PXR_REQUIRE((n + 255) / 256 < (1ll << 31), "pxr_split_planes_f32: too large");
  const P3Mat out{reinterpret_cast<__bf16*>(planes), plane_stride, panel_rows};
  const int xcd = getenv("PXR_SPLIT_XCD") ? atoi(getenv("PXR_SPLIT_XCD")) : 1;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, rows,
                     (int)(cols / 8), out, xcd);
  return pxr_check_launch("pxr_split_planes_f32");
}

extern "C" int pxr_split_planes_multi_f32(int n, const float* const* x, const int64_t* rows, const int64_t* cols, const int64_t* ldx,
                                          void* const* planes, const int64_t* plane_stride, const int64_t* panel_rows,
                                          void* stream) {
  PXR_REQUIRE(n >= 1 && n <= 16 && x && rows && cols && ldx && planes && plane_stride && panel_rows, "pxr_split_planes_multi_f32: bad args");
  SplitMulti m{};
  m.n = n;
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    PXR_REQUIRE(x[i] && planes[i] && rows[i] > 0 && rows[i] < (1ll << 31) && cols[i] > 0 && cols[i] % 32 == 0 && ldx[i] % 4 == 0 &&
                    (((uintptr_t)x[i]) & 15) == 0 && p3_mat_ok(planes[i], plane_stride[i], panel_rows[i], rows[i], cols[i]),
                "pxr_split_planes_multi_f32: matrix %d is bad", i);
    m.x[i] = x[i]; m.ldx[i] = ldx[i]; m.rows[i] = (int)rows[i]; m.cols8[i] = (int)(cols[i] / 8);
    m.out[i] = P3Mat{reinterpret_cast<__bf16*>(planes[i]), plane_stride[i], panel_rows[i]};
    m.begin[i] = total;
    total += rows[i] * (cols[i] / 8);
  }
  m.begin[n] = total;
  PXR_REQUIRE((total + 255) / 256 < (1ll << 31), "pxr_split_planes_multi_f32: too large");
  hipLaunchKernelGGL(split_planes_multi_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m);
  return pxr_check_launch("pxr_split_planes_multi_f32");
}

// The same into the TWO-plane fp16 format (planes.cuh "h2"): matrix i is multiplied by 2^scale_exp[i] first (exact; the caller
// picks the exponent that puts max |x| 2^e near 2^14 and hands it to pxr_gemm_h2_f32).  A scaled value beyond the fp16 range sets
// PXR_STATUS_H2_RANGE.  plane_stride: elements between the two planes (only two are written).
extern "C" int pxr_split_h2_multi_f32(int n, const float* const* x, const int64_t* rows, const int64_t* cols, const int64_t* ldx,
                                      void* const* planes, const int64_t* plane_stride, const int64_t* panel_rows,
                                      const int* scale_exp, void* stream) {
  PXR_REQUIRE(n >= 1 && n <= 16 && x && rows && cols && ldx && planes && plane_stride && panel_rows && scale_exp, "pxr_split_h2_multi_f32: bad args");
  SplitMulti m{};
  m.n = n;
  m.fmt = PXR_PLANES_H2;
  m.status = pxr_status_word();
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    PXR_REQUIRE(x[i] && planes[i] && rows[i] > 0 && rows[i] < (1ll << 31) && cols[i] > 0 && cols[i] % 32 == 0 && ldx[i] % 4 == 0 &&
                    (((uintptr_t)x[i]) & 15) == 0 && p3_mat_ok(planes[i], plane_stride[i], panel_rows[i], rows[i], cols[i]),
                "pxr_split_h2_multi_f32: matrix %d is bad", i);
    PXR_REQUIRE(scale_exp[i] >= -60 && scale_exp[i] <= 60, "pxr_split_h2_multi_f32: scale exponent %d of matrix %d", scale_exp[i], i);
    m.x[i] = x[i]; m.ldx[i] = ldx[i]; m.rows[i] = (int)rows[i]; m.cols8[i] = (int)(cols[i] / 8);
    m.out[i] = P3Mat{reinterpret_cast<__bf16*>(planes[i]), plane_stride[i], panel_rows[i]};
    m.scale[i] = ldexpf(1.0f, scale_exp[i]);
    m.begin[i] = total;
    total += rows[i] * (cols[i] / 8);
  }
  m.begin[n] = total;
  PXR_REQUIRE((total + 255) / 256 < (1ll << 31), "pxr_split_h2_multi_f32: too large");
  hipLaunchKernelGGL(split_planes_multi_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m);
  return pxr_check_launch("pxr_split_h2_multi_f32");
}

// C[M,N] = epilogue(2^-(a_exp + b_exp) * (A~ B~^T)) with both operands as TWO fp16 planes (A~ = A 2^a_exp, B~ = B 2^b_exp, [N][K]
// k-contiguous): three products per multiply on v_mfma_f32_32x32x16_f16 -- 22 significant bits per operand, the accuracy of the
// six-product bf16x3 GEMM at half its matrix-pipe work (profiles/r04/lab/h2_lab_run1.log).  Forward flavours only (epilogue:
// EPI_NONE | EPI_BIAS | EPI_BIAS_GELU | EPI_BIAS_ADD | EPI_BIAS_QGELU | EPI_BIAS_RELU); output as fp32 and / or planes in either
// format (c_fmt: 0 = three bf16 planes, 1 = two fp16 planes at unit scale).  Ping-pong tiles only (gemm_p4.cuh): meant for the tall
// GEMMs of the image tower.
extern "C" int pxr_gemm_h2_f32(int b_kc, int M, int N, int K, const void* A, int64_t a_plane_stride, int64_t a_panel_rows, int a_exp,
                               const int* a_exp_dev, const void* B, int64_t b_plane_stride, int64_t b_panel_rows, int b_exp,
                               const int* b_exp_dev, float* C, int64_t ldc, int epilogue, const float* bias, float* aux,
                               int64_t ldaux, void* c_planes, int64_t c_plane_stride, int64_t c_panel_rows, int c_fmt,
                               const int* c_exp_dev, int act, int tile_hint, void* stream) {
  PXR_REQUIRE(A && B && (C || c_planes), "pxr_gemm_h2_f32: null operand");
  PXR_REQUIRE(M >= 0 && N >= 0 && K >= 0 && K % 32 == 0, "pxr_gemm_h2_f32: K must be a multiple of 32");
  PXR_REQUIRE(b_kc || N % 32 == 0, "pxr_gemm_h2_f32: an x-contiguous B needs N %% 32 == 0");
  PXR_REQUIRE(a_panel_rows % 16 == 0 && b_panel_rows % 16 == 0 && a_plane_stride % 8 == 0 && b_plane_stride % 8 == 0,
              "pxr_gemm_h2_f32: panel rows must be multiples of 16, plane strides of 8 elements");
  PXR_REQUIRE(a_panel_rows >= M && a_plane_stride >= a_panel_rows * K &&
                  (b_kc ? (b_panel_rows >= N && b_plane_stride >= b_panel_rows * K) : (b_panel_rows >= K && b_plane_stride >= b_panel_rows * N)),
              "pxr_gemm_h2_f32: operand planes smaller than the matrices");
  PXR_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)c_planes | (uintptr_t)aux | (uintptr_t)bias) & 15) == 0,
              "pxr_gemm_h2_f32: operands must be 16-byte aligned");
  PXR_REQUIRE(a_plane_stride * 4 < 0x7FFFFFF0ll && b_plane_stride * 4 < 0x7FFFFFF0ll, "pxr_gemm_h2_f32: an operand's two planes must span less than 2 GiB");
  PXR_REQUIRE(!c_planes || (N % 32 == 0 && c_panel_rows % 16 == 0 && c_panel_rows >= M), "pxr_gemm_h2_f32: output planes need N %% 32 == 0");
  PXR_REQUIRE(c_fmt == PXR_PLANES_BF16X3 || c_fmt == PXR_PLANES_H2, "pxr_gemm_h2_f32: c_fmt");
  PXR_REQUIRE(!c_exp_dev || (c_planes && c_fmt == PXR_PLANES_H2), "pxr_gemm_h2_f32: an output exponent needs h2 output planes");
  PXR_REQUIRE(a_exp >= -60 && a_exp <= 60 && b_exp >= -60 && b_exp <= 60, "pxr_gemm_h2_f32: scale exponents");
  PXR_REQUIRE(aux || !(epilogue == EPI_BIAS_ADD || epilogue == EPI_BIAS_GELU || epilogue == EPI_BIAS_QGELU_GRAD || epilogue == EPI_ADD ||
                       epilogue == EPI_MUL || epilogue == EPI_BIAS_GELU_GRAD || epilogue == EPI_BIAS_ACT_GRAD),
              "pxr_gemm_h2_f32: epilogue %d reads / writes aux", epilogue);
  PXR_REQUIRE(bias || !b_kc || epilogue == EPI_NONE, "pxr_gemm_h2_f32: epilogue %d needs a bias", epilogue);
  if (M == 0 || N == 0) return PXR_OK;
  P3Args g;
  g.A = P3Mat{reinterpret_cast<__bf16*>(const_cast<void*>(A)), a_plane_stride, a_panel_rows};
  g.B = P3Mat{reinterpret_cast<__bf16*>(const_cast<void*>(B)), b_plane_stride, b_panel_rows};
  g.Cp = P3Mat{reinterpret_cast<__bf16*>(c_planes), c_plane_stride, c_panel_rows};
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.act = act;
  g.a_exp = a_exp; g.b_exp = b_exp; g.a_exp_dev = a_exp_dev; g.b_exp_dev = b_exp_dev; g.c_exp_dev = c_exp_dev;
  g.cp_fmt = c_fmt;
  g.status = pxr_status_word();
  hipStream_t st = (hipStream_t)stream;
  // tiles: 2 | BM | BN | ring slots | accumulator sets.  256x256 (one set) when its rounds of 256 workgroups, each 1.75x as long as
  // a round of 256x128 tiles (two sets: hi*hi apart from the cross terms; measured: profiles/r04/lab/h2_lab_run1.log -- the square
  // tile reads half as many LDS bytes per MFMA), cost no more
  // ... and, when 256-row tiles would not fill the chip twice (the reference's 3 200-token step), the lockstep tiles of the bf16x3
  // kernels on two planes: 128x64 for wide outputs, 64x64 otherwise (lab, 3 200 tokens: 25.4 -> 16.6 us fc1, 36.3 -> 22.6 qkv,
  // 14.8 -> 10.1 out-proj, 26.4 -> 17.0 fc2; profiles/r04/lab/h2_lab_run2_seq_tiles.log)
  if (tile_hint == 0) {
    const int64_t t128 = (int64_t)((M + 255) / 256) * ((N + 127) / 128), t256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    static const int env_sq = getenv("PXR_H2_SQUARE") ? atoi(getenv("PXR_H2_SQUARE")) : 1;
    static const int env_small = getenv("PXR_H2_SMALL_TILES") ? atoi(getenv("PXR_H2_SMALL_TILES")) : 1;
    if (env_small && t128 < 512) tile_hint = N >= 1024 ? 212806420 : 206406430;
    else tile_hint = (env_sq && N >= 256 && 7 * ((t256 + 255) / 256) <= 4 * ((t128 + 255) / 256)) ? 225625641 : 225612842;
  }
  if (tile_hint == 212806420) return epi_h2<P3Cfg<128, 64, 2, 2, 2, true>>(b_kc, epilogue, g, st);
  if (tile_hint == 212806430) return epi_h2<P3Cfg<128, 64, 2, 2, 3, true>>(b_kc, epilogue, g, st);
  if (tile_hint == 206406430) return epi_h2<P3Cfg<64, 64, 2, 2, 3, true>>(b_kc, epilogue, g, st);
  if (tile_hint == 225625641) return epi_h2<P4Cfg<256, 256, 4, 2, 4, 1, 0, 2, true>>(b_kc, epilogue, g, st);
  if (tile_hint == 225612842) return epi_h2<P4Cfg<256, 128, 4, 2, 4, 2, 0, 2, true>>(b_kc, epilogue, g, st);
  if (tile_hint == 225612841) return epi_h2<P4Cfg<256, 128, 4, 2, 4, 1, 0, 2, true>>(b_kc, epilogue, g, st);
  pxr_set_error("pxr_gemm_h2_f32: tile %d is not instantiated", tile_hint);
  return PXR_ERR_BAD_ARG;
}

extern "C" int pxr_gemm_planes_f32(int b_kc, int M, int N, int K, const void* A, int64_t a_plane_stride, int64_t a_panel_rows,
                                   const void* B, int64_t b_plane_stride, int64_t b_panel_rows, float* C, int64_t ldc,
                                   int epilogue, const float* bias, float* aux, int64_t ldaux, void* c_planes,
                                   int64_t c_plane_stride, int64_t c_panel_rows, int act, int tile_hint, void* stream) {
  PXR_REQUIRE(A && B && (C || c_planes), "pxr_gemm_planes_f32: null operand");
  PXR_REQUIRE(M >= 0 && N >= 0 && K >= 0 && K % 32 == 0, "pxr_gemm_planes_f32: K must be a multiple of 32");
  PXR_REQUIRE(b_kc || N % 32 == 0, "pxr_gemm_planes_f32: an x-contiguous B needs N %% 32 == 0");
  PXR_REQUIRE(a_panel_rows % 16 == 0 && b_panel_rows % 16 == 0 && a_plane_stride % 8 == 0 && b_plane_stride % 8 == 0,
              "pxr_gemm_planes_f32: panel rows must be multiples of 16, plane strides of 8 elements");
  PXR_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)c_planes | (uintptr_t)aux | (uintptr_t)bias) & 15) == 0,
              "pxr_gemm_planes_f32: operands must be 16-byte aligned");
  PXR_REQUIRE(a_plane_stride * 6 < 0x7FFFFFF0ll && b_plane_stride * 6 < 0x7FFFFFF0ll,
              "pxr_gemm_planes_f32: an operand's three planes must span less than 2 GiB");
  PXR_REQUIRE(!c_planes || (N % 32 == 0 && c_panel_rows % 16 == 0 && c_panel_rows >= M), "pxr_gemm_planes_f32: output planes need N %% 32 == 0");
  PXR_REQUIRE(aux || !(epilogue == EPI_ADD || epilogue == EPI_MUL || epilogue == EPI_BIAS_ADD || epilogue == EPI_BIAS_GELU ||
                       epilogue == EPI_BIAS_GELU_GRAD || epilogue == EPI_BIAS_QGELU_GRAD || epilogue == EPI_BIAS_ACT_GRAD),
              "pxr_gemm_planes_f32: epilogue %d reads / writes aux", epilogue);
  if (M == 0 || N == 0) return PXR_OK;
  P3Args g;
  g.A = P3Mat{reinterpret_cast<__bf16*>(const_cast<void*>(A)), a_plane_stride, a_panel_rows};
  g.B = P3Mat{reinterpret_cast<__bf16*>(const_cast<void*>(B)), b_plane_stride, b_panel_rows};
  g.Cp = P3Mat{reinterpret_cast<__bf16*>(c_planes), c_plane_stride, c_panel_rows};
  g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias = bias; g.aux = aux; g.ldaux = ldaux; g.act = act;
  g.a_exp = g.b_exp = 0; g.a_exp_dev = g.b_exp_dev = g.c_exp_dev = nullptr; g.cp_fmt = PXR_PLANES_BF16X3; g.status = nullptr;
  hipStream_t st = (hipStream_t)stream;
  // tile_hint digits: waves | BM (3) | BN (3) | stages | early fragment reads.  Heuristic (tools/p3_sweep.py on MI355X):
  // 256x128 tiles when they fill the chip more than twice; at M = B*L ~ 3200 tokens 128x64 (two workgroups per CU) for wide
  // outputs and 64x64 with early fragment reads otherwise
  if (tile_hint == 0) {
    const int64_t t256 = (int64_t)((M + 255) / 256) * ((N + 127) / 128);
    static const int env_small = getenv("PXR_P3_TILE_SMALL") ? atoi(getenv("PXR_P3_TILE_SMALL")) : 406406430;   // A/B knobs
    static const int env_wide = getenv("PXR_P3_TILE_WIDE") ? atoi(getenv("PXR_P3_TILE_WIDE")) : 412806420;
    // big problems: the ping-pong tiles (gemm_p4.cuh).  256x256 with ONE accumulator set (fp32-sequential-grade rounding, see
    // DESIGN.md) when it does not cost more rounds of 256 workgroups than 256x128 (bit-identical to the lockstep tiles) would
    static const int env_p4 = getenv("PXR_P4") ? atoi(getenv("PXR_P4")) : 1;              // 0: the round-3 lockstep 256x128 tile
    static const int env_acc1 = getenv("PXR_P4_ACC1") ? atoi(getenv("PXR_P4_ACC1")) : 1;  // 0: never the one-set 256x256 tile
    if (t256 >= 512 && env_p4) {
      const int64_t t256sq = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
      const int64_t r_sq = 2 * ((t256sq + 255) / 256), r_128 = (t256 + 255) / 256;
      tile_hint = (env_acc1 && N >= 256 && r_sq <= r_128) ? 425625631 : 425612833;
    } else {
      tile_hint = t256 >= 512 ? 825612820 : (N >= 1024 ? env_wide : env_small);
    }
  }
#define PXR_P3_TILE(CODE, EARLY, ...) \
  if (tile_hint == CODE) return epi_p3<P3Cfg<__VA_ARGS__>, EARLY>(b_kc, epilogue, g, st)
  PXR_P3_TILE(406406431, true, 64, 64, 2, 2, 3);
  PXR_P3_TILE(406406461, true, 64, 64, 2, 2, 6);
  PXR_P3_TILE(406406430, false, 64, 64, 2, 2, 3);
  PXR_P3_TILE(412806441, true, 128, 64, 2, 2, 4);
  PXR_P3_TILE(412806420, false, 128, 64, 2, 2, 2);
  PXR_P3_TILE(406412820, false, 64, 128, 2, 2, 2);
  PXR_P3_TILE(812812830, false, 128, 128, 2, 4, 3);
  PXR_P3_TILE(825612820, false, 256, 128, 4, 2, 2);
#undef PXR_P3_TILE
  // ping-pong tiles: 4 | BM | BN | ring slots | accumulator sets
  if (tile_hint == 425612833) return epi_p3<P4Cfg<256, 128, 4, 2, 3, 3>, false>(b_kc, epilogue, g, st);
  if (tile_hint == 425612832) return epi_p3<P4Cfg<256, 128, 4, 2, 3, 2>, false>(b_kc, epilogue, g, st);
  if (tile_hint == 425625631) return epi_p3<P4Cfg<256, 256, 4, 2, 3, 1>, false>(b_kc, epilogue, g, st);
  pxr_set_error("pxr_gemm_planes_f32: tile %d is not instantiated", tile_hint);
  return PXR_ERR_BAD_ARG;
}

// The same from fp16 two-plane operands (planes.cuh "h2"): dW = 2^-(dy_exp + x_exp) dY~^T X~, three products per multiply.  The
// exponents of problem i are dy_exp[i] / x_exp[i], or *dy_exp_dev[i] / *x_exp_dev[i] where that pointer is not null.
extern "C" int pxr_grouped_dw_h2_f32(int n, const void* const* dy, const int64_t* dy_plane_stride, const int64_t* dy_panel_rows,
                                     const int* dy_exp, const int* const* dy_exp_dev, const void* const* x,
                                     const int64_t* x_plane_stride, const int64_t* x_panel_rows, const int* x_exp,
                                     const int* const* x_exp_dev, float* const* dW, float* const* db, const int* T, const int* N,
                                     const int* K, int tile_hint, void* stream) {
  PXR_REQUIRE(n >= 1 && n <= DW_MAX && dy && x && dW && db && T && N && K && dy_plane_stride && dy_panel_rows && x_plane_stride &&
                  x_panel_rows && dy_exp && x_exp && dy_exp_dev && x_exp_dev, "pxr_grouped_dw_h2_f32: bad args (n=%d, max %d)", n, DW_MAX);
  P3DwGroup g{};
  g.n = n;
  int64_t t256 = 0;
  int t_min = T[0];
  for (int i = 0; i < n; ++i) {
    PXR_REQUIRE(dy[i] && x[i] && dW[i] && T[i] > 0 && N[i] > 0 && K[i] > 0 && N[i] % 32 == 0 && K[i] % 32 == 0,
                "pxr_grouped_dw_h2_f32: problem %d has a bad shape (N and K must be multiples of 32)", i);
    PXR_REQUIRE(dy_panel_rows[i] % 32 == 0 && x_panel_rows[i] % 32 == 0 && dy_panel_rows[i] >= T[i] && x_panel_rows[i] >= T[i],
                "pxr_grouped_dw_h2_f32: problem %d: panel rows must be multiples of 32 >= T", i);
    PXR_REQUIRE((((uintptr_t)dy[i] | (uintptr_t)x[i] | (uintptr_t)dW[i]) & 15) == 0, "pxr_grouped_dw_h2_f32: unaligned operand");
    PXR_REQUIRE(dy_plane_stride[i] * 4 < 0x7FFFFFF0ll && x_plane_stride[i] * 4 < 0x7FFFFFF0ll, "pxr_grouped_dw_h2_f32: planes too large");
    PXR_REQUIRE(dy_exp[i] >= -60 && dy_exp[i] <= 60 && x_exp[i] >= -60 && x_exp[i] <= 60, "pxr_grouped_dw_h2_f32: scale exponents");
    P3DwProblem& P = g.p[i];
    P.dy = P3Mat{reinterpret_cast<__bf16*>(const_cast<void*>(dy[i])), dy_plane_stride[i], dy_panel_rows[i]};
    P.x = P3Mat{reinterpret_cast<__bf16*>(const_cast<void*>(x[i])), x_plane_stride[i], x_panel_rows[i]};
    P.dW = dW[i]; P.db = db[i]; P.T = T[i]; P.N = N[i]; P.K = K[i];
    P.dy_exp = dy_exp[i]; P.x_exp = x_exp[i]; P.dy_exp_dev = dy_exp_dev[i]; P.x_exp_dev = x_exp_dev[i];
    t256 += (int64_t)((N[i] + 255) / 256) * ((K[i] + 127) / 128);
    t_min = T[i] < t_min ? T[i] : t_min;
  }
  hipStream_t st = (hipStream_t)stream;
  static const int env_splitk = getenv("PXR_DW_SPLITK") ? atoi(getenv("PXR_DW_SPLITK")) : 1;
  if (tile_hint == 0) tile_hint = (env_splitk && t256 >= 64 && t256 <= 128 && t_min >= 2048 && 2 * t256 <= pxr_cu_count()) ? 225612822 : 225612842;
  if (tile_hint == 225612842) return launch_dw_p3<P4Cfg<256, 128, 4, 2, 4, 2, 0, 2, true>, false>(g, st);
  if (tile_hint == 225612822) return launch_dw_p3<P4Cfg<256, 128, 4, 2, 4, 2, 0, 2, true>, false, 2>(g, st);
  pxr_set_error("pxr_grouped_dw_h2_f32: tile %d is not instantiated", tile_hint);
  return PXR_ERR_BAD_ARG;
}

extern "C" int pxr_grouped_dw_planes_f32(int n, const void* const* dy, const int64_t* dy_plane_stride, const int64_t* dy_panel_rows,
                                         const void* const* x, const int64_t* x_plane_stride, const int64_t* x_panel_rows,
                                         float* const* dW, float* const* db, const int* T, const int* N, const int* K,
                                         int tile_hint, void* stream) {
  PXR_REQUIRE(n >= 1 && n <= DW_MAX && dy && x && dW && db && T && N && K && dy_plane_stride && dy_panel_rows && x_plane_stride &&
                  x_panel_rows, "pxr_grouped_dw_planes_f32: bad args (n=%d, max %d)", n, DW_MAX);
  P3DwGroup g{};
  g.n = n;
  int64_t t128 = 0;
  for (int i = 0; i < n; ++i) {
    PXR_REQUIRE(dy[i] && x[i] && dW[i] && T[i] > 0 && N[i] > 0 && K[i] > 0 && N[i] % 32 == 0 && K[i] % 32 == 0,
                "pxr_grouped_dw_planes_f32: problem %d has a bad shape (N and K must be multiples of 32)", i);
    PXR_REQUIRE(dy_panel_rows[i] % 32 == 0 && x_panel_rows[i] % 32 == 0 && dy_panel_rows[i] >= T[i] && x_panel_rows[i] >= T[i],
                "pxr_grouped_dw_planes_f32: problem %d: panel rows must be multiples of 32 >= T", i);
    PXR_REQUIRE((((uintptr_t)dy[i] | (uintptr_t)x[i] | (uintptr_t)dW[i]) & 15) == 0, "pxr_grouped_dw_planes_f32: unaligned operand");
    PXR_REQUIRE(dy_plane_stride[i] * 6 < 0x7FFFFFF0ll && x_plane_stride[i] * 6 < 0x7FFFFFF0ll, "pxr_grouped_dw_planes_f32: planes too large");
    P3DwProblem& P = g.p[i];
    P.dy = P3Mat{reinterpret_cast<__bf16*>(const_cast<void*>(dy[i])), dy_plane_stride[i], dy_panel_rows[i]};
    P.x = P3Mat{reinterpret_cast<__bf16*>(const_cast<void*>(x[i])), x_plane_stride[i], x_panel_rows[i]};
    P.dW = dW[i]; P.db = db[i]; P.T = T[i]; P.N = N[i]; P.K = K[i];
    t128 += (int64_t)((N[i] + 127) / 128) * ((K[i] + 127) / 128);
  }
  hipStream_t st = (hipStream_t)stream;
  // 256x128 ping-pong tiles (two accumulator sets: the bias column sums need the third set's registers) when they fill >= 192 CUs
  // in ONE round (the ViT tower's blocks: 216 tiles); 128x128 lockstep tiles when those do; 64x64 below
  int64_t t256 = 0;
  for (int i = 0; i < n; ++i) t256 += (int64_t)((N[i] + 255) / 256) * ((K[i] + 127) / 128);
  static const int env_p4dw = getenv("PXR_P4_DW") ? atoi(getenv("PXR_P4_DW")) : 1;
  // ... and the same tiles with the token reduction split in two (grouped_dw_p3_kernel KS = 2) when the tiles fill at most half of
  // the chip and the reduction is long enough to pay for the second ramp (the sequence block: 128 tiles, >= 3200 tokens)
  static const int env_splitk = getenv("PXR_DW_SPLITK") ? atoi(getenv("PXR_DW_SPLITK")) : 1;
  int t_min = T[0];
  for (int i = 1; i < n; ++i) t_min = T[i] < t_min ? T[i] : t_min;
  if (tile_hint == 0) {
    // (split-K: the second half of a tile SPINS on the first one's flag, so both halves of every tile must be resident at once:
    // 2 x tiles workgroups of one per CU -- checked against the device's CU count, which CU masks / partition modes shrink)
    if (env_p4dw && env_splitk && t256 >= 96 && t256 <= 128 && t_min >= 2048 && 2 * t256 <= pxr_cu_count()) tile_hint = 425612822;
    else tile_hint = (env_p4dw && ((t256 >= 192 && t256 <= 256) || t256 >= 512)) ? 425612832 : (t128 >= 192 ? 412812831 : 406406431);
  }
  if (tile_hint == 425612822) return launch_dw_p3<P4Cfg<256, 128, 4, 2, 3, 2>, false, 2>(g, st);
  if (tile_hint == 425612832) return launch_dw_p3<P4Cfg<256, 128, 4, 2, 3, 2>, false>(g, st);
  if (tile_hint == 812812830) return launch_dw_p3<P3Cfg<128, 128, 2, 4, 3>, false>(g, st);
  if (tile_hint == 412812831) return launch_dw_p3<P3Cfg<128, 128, 2, 2, 3>, true>(g, st);
  if (tile_hint == 412812830) return launch_dw_p3<P3Cfg<128, 128, 2, 2, 3>, false>(g, st);
  if (tile_hint == 412806440) return launch_dw_p3<P3Cfg<128, 64, 2, 2, 4>, false>(g, st);
  if (tile_hint == 406406460) return launch_dw_p3<P3Cfg<64, 64, 2, 2, 6>, false>(g, st);
  if (tile_hint == 406406430) return launch_dw_p3<P3Cfg<64, 64, 2, 2, 3>, false>(g, st);
  if (tile_hint == 406406431) return launch_dw_p3<P3Cfg<64, 64, 2, 2, 3>, true>(g, st);
  pxr_set_error("pxr_grouped_dw_planes_f32: tile %d is not instantiated", tile_hint);
  return PXR_ERR_BAD_ARG;
}
