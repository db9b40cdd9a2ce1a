/* pxr.h -- C ABI of libpxr.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the PixelRec
 * sequential-recommender hot path (SASRec under IDNet).
 *
 * The reference (westlake-repl/PixelRec) is 100 % Python on stock PyTorch ops and has NO FFI of its own
 * (SURVEY.md §8b): the seam is the Python model-class contract.  These entry points are what a binding for this
 * path would bind -- plain pointers and sizes, no torch types -- and each one names the reference call site it
 * replaces (paths relative to /root/reference/code/REC/).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - all tensors are dense row-major fp32 unless stated; indices are int64 (torch.long) like the reference's;
 *   - every pointer is a DEVICE pointer; the library never allocates, frees or retains device memory;
 *   - `stream` is a hipStream_t; calls are asynchronous, re-entrant, and hold no global mutable state
 *     (usable from the autograd thread); all launches are hipGraph-capturable (no host syncs);
 *   - return 0 on success, <0 on error (PXR_ERR_*); pxr_last_error() gives the thread-local message;
 *   - `*_ws_bytes` functions return the scratch size the matching call needs.
 *   - dropout: Bernoulli(1-p) keep-mask = counter hash of (seed, stream_id, element index); the backward call
 *     must pass the same (p, seed, stream_id).  p = 0 disables it (eval).
 *   - `step_dev` (const int64_t*, may be NULL): a device-resident step counter.  Dropout entry points add it to
 *     `seed`; optimizer entry points take the step number from it.  With it a whole training step can be captured
 *     in a hipGraph and replayed: pxr_counter_add_i64 advances the counter on the device, no host value is baked in.
 */
#ifndef PXR_H_
#define PXR_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PXR_OK 0
#define PXR_ERR_BAD_ARG (-1)
#define PXR_ERR_LAUNCH (-2)
#define PXR_ERR_WORKSPACE (-3)

/* The ABI revision this header describes.  pxr_version() of the loaded library must EQUAL it: entries may change meaning between
 * revisions while keeping their names (0.2.0 -> 0.3.0: the `stat` buffers of pxr_ln_bwd_stat_f32 / pxr_attn_bwd_stat_f32 became
 * pxr_ln_bwd_partial_rows(rows) / 64 words instead of one caller-zeroed word, and pxr_ln_bwd_stat_f32 gained `zero`, `zero_n`),
 * so a caller built against another revision must refuse to run instead of writing out of bounds (pixelrec_amd/lib.py does). */
#define PXR_ABI_VERSION 300
int pxr_version(void);                 /* major*10000 + minor*100 + patch; == PXR_ABI_VERSION of the header it was built from */
const char* pxr_last_error(void);      /* message of the last failing call on this thread */
const char* pxr_target_arch(void);     /* "gfx950" */

/* HOST function: keep-mask bytes (1 = keep) of elements first_index .. first_index+n-1 for (seed, stream_id, p),
 * computed with the same hash the kernels use.  For tests of the dropout restatement; no device work. */
int pxr_dropout_keep_host(uint64_t seed, uint32_t stream_id, uint64_t first_index, int64_t n, float p,
                          uint8_t* keep_out);

/* Registers a caller-owned int32 in DEVICE memory as this process' status word (NULL unregisters).  Kernels that
 * gather table rows by id (pxr_embed_gather_f32, pxr_input_ln_fwd_f32) OR bit 0 into it when an id lies outside
 * [0, N) -- where the reference's nn.Embedding raises IndexError / a device-side assert (model/IDNet/sasrec.py:68) --
 * and clamp the id; pxr_merge_split_rows_f32 ORs bit 1 when a rank's row count exceeded the exchanged capacity.  The
 * host reads the word at its next synchronisation point and raises.  One process per GPU. */
int pxr_set_status_word(int32_t* dev_word);

/* ---- embedding table ---------------------------------------------------------------------------------------- */
/* out[i,:] = table[idx[i],:]                       model/IDNet/sasrec.py:68,101; model/PixelNet/mosasrec.py:102 */
int pxr_embed_gather_f32(const float* table, int64_t N, int D, const int64_t* idx, int64_t n, float* out,
                         void* stream);

/* Sparse embedding backward (replaces autograd's dense embedding_dense_backward of sasrec.py:31/68):
 * (idx[n], rows[n,D]) -> ascending uniq_idx[<=n], uniq_rows[<=n,D] = scale * sum of the rows of each id,
 * *n_uniq_dev = count.  Id 0 (padding_idx) and out-of-range ids are dropped.  Deterministic (stable sort). */
int64_t pxr_embed_grad_ws_bytes(int64_t n_occ);
int pxr_embed_grad_rows_f32(const int64_t* idx, int64_t n, const float* rows, int D, int64_t n_table, float scale,
                            int64_t* uniq_idx, float* uniq_rows, int32_t* n_uniq_dev, void* ws, int64_t ws_bytes,
                            void* stream);
/* Data-parallel merge of the W rank-local sparse gradients after the all-gather (the build's replacement of DDP's
 * dense all-reduce, run.py:40): idx_all[W,cap] / rows_all[W,cap,D], every list ascending and unique over its whole
 * cap (unused tail slots hold ids >= n_table).  No re-sort: the lowest rank holding an id owns it and adds the other
 * ranks' rows in rank order (same fixed order on every replica).  Output is not compacted: out_idx[e] = id or 0
 * (empty slot, skipped by pxr_adamw_rows_f32 / pxr_adamw_table_f32), out_rows[e,:] = scale * sum, *n_out = W*cap. */
int64_t pxr_merge_rows_ws_bytes(int W, int64_t cap);
int pxr_merge_sorted_rows_f32(const int64_t* idx_all, const float* rows_all, int W, int64_t cap, int D,
                              int64_t n_table, float scale, int64_t* out_idx, float* out_rows, int32_t* n_out_dev,
                              void* ws, int64_t ws_bytes, void* stream);
/* The same merge on the layout a ONE-collective exchange delivers: packed_all = W blocks of pxr_packed_rows_bytes(cap, D)
 * bytes, block = { int64 ids[cap] ascending over the first `count` entries; int32 count; zero padding to a 16-byte
 * boundary (rows start at pxr_packed_rows_offset(cap)); float rows[cap][D] }.  Entries at or beyond a block's count
 * are ignored whatever they hold, so a rank sends its sort/segment output as it is (no PAD fill, no second
 * all-gather for the ids).  Same output convention, same summation order as pxr_merge_sorted_rows_f32. */
int64_t pxr_packed_rows_offset(int64_t cap);
int64_t pxr_packed_rows_bytes(int64_t cap, int D);
int pxr_merge_packed_rows_f32(const void* packed_all, int W, int64_t cap, int D, int64_t n_table, float scale,
                              int64_t* out_idx, float* out_rows, int32_t* n_out_dev, void* ws, int64_t ws_bytes,
                              void* stream);
/* The exchange with a REDUCED row capacity cap_x <= cap (two collectives): heads_all = W x pxr_packed_rows_offset(cap)
 * bytes ({ids[cap], count, pad} of every rank), rows_all = [W, cap_x, D] (the first cap_x rows of every rank).  A rank
 * whose count exceeds cap_x is cut there and bit 1 of the status word (pxr_set_status_word) is set.  out_*: W*cap_x slots. */
int pxr_merge_split_rows_f32(const void* heads_all, const float* rows_all, int W, int64_t cap, int64_t cap_x, int D,
                             int64_t n_table, float scale, int64_t* out_idx, float* out_rows, int32_t* n_out_dev,
                             void* ws, int64_t ws_bytes, void* stream);
/* On-device train-batch construction (data/dataset/trainset.py:40-63): pos int64 [B,W] left-padded windows (W = L+1)
 * -> items [B,2,W] (positives | one negative per target position, uniform over [1, n_items-1] minus the window's own
 * items) and masked_index [B,W-1].  Stateless: (seed, batch_counter) select the random stream. */
int pxr_sample_negatives_i64(const int64_t* pos, int B, int W, int64_t n_items, uint64_t seed, uint64_t batch_counter,
                             int64_t* items, int64_t* masked_index, void* stream);

/* Row-sharded table, hit-row exchange as an all-to-all (model/sharded.py): split a rank's ascending unique id list (count on
 * the device) by owner (id % W) into W request lists of pp_cap slots -- req[W, pp_cap] ascending, pad_id beyond the count;
 * pos[W, pp_cap] = the id's index in the unique list (-1 for padding); counts[W].  An owner with more than pp_cap hits sets
 * status bit 16 (PXR_STATUS_SHARD_OVERFLOW) and loses its surplus.  pxr_scatter_rows_f32: dst[row_offset + pos[i], :] =
 * src[i, :] for pos[i] >= 0 (the rows that came back, put at their place in the compact block). */
int pxr_shard_bucket_ids_i64(const int64_t* ids, const int32_t* n_dev, int W, int64_t n_table, int64_t pp_cap, int64_t pad_id,
                             int64_t* req, int32_t* pos, int32_t* counts, void* stream);
int pxr_scatter_rows_f32(const float* src, const int32_t* pos, int64_t n_src, int D, float* dst, int64_t dst_rows,
                         int row_offset, void* stream);
/* Row-sharded table (north_star "embedding table optionally row-sharded", BASELINE configs[3]): owner of id =
 * id % W, its row in the owner's shard = id / W + 1 (local row 0 = all-zero dummy).  local_rows[i] = that row if this
 * rank owns ids[i] (0 < id < n_table), else 0.  pxr_ids_to_compact: out[i] = 1 + position of ids[i] in the ascending
 * unique list (0 for padding): re-indexes a batch onto the [n_uniq+1, D] block of rows fetched from the owners. */
int pxr_shard_local_rows_i64(const int64_t* ids, int64_t n, int W, int rank, int64_t n_table, int64_t* local_rows,
                             void* stream);
int pxr_ids_to_compact_i64(const int64_t* ids, int64_t n, const int64_t* uniq_idx, const int32_t* n_uniq_dev,
                           int64_t* out, void* stream);
/* ids_all[W,cap]: W ascending request lists (tail >= n_table).  Like pxr_shard_local_rows_i64, but an id requested by
 * several ranks keeps its local row only in the lowest-ranked list: the duplicate-free work list of the owner's lazy
 * AdamW catch-up. */
int pxr_shard_first_rows_i64(const int64_t* ids_all, int W, int64_t cap, int rank, int64_t n_table,
                             int64_t* local_rows, void* stream);
/* The same for the three uses of the table inside SASRec.forward (sasrec.py:68-74,88-89) without materialising the
 * [B,2,L+1,D] gather: items[B,2,L+1]; dx0 = grad of (table row + pos) [B*L,D]; out = last-layer states [B*L,D];
 * coef[B*L] from pxr_bpr_loss_bwd_f32.  n_occ = 3*B*L for the workspace size. */
int pxr_sasrec_embed_grad_f32(const int64_t* items, int B, int L, const float* dx0, const float* out,
                              const float* coef, int D, int64_t n_table, float scale, int64_t* uniq_idx,
                              float* uniq_rows, int32_t* n_uniq_dev, void* ws, int64_t ws_bytes, void* stream);

/* The two phases separately: phase 1 depends on `items` only and may run before the forward pass (the lazy table
 * optimizer brings exactly these unique rows up to date before they are read); `ws` carries the sorted occurrences to
 * phase 2 and must not be touched in between. */
int pxr_sasrec_occ_sort(const int64_t* items, int B, int L, int64_t n_table, int64_t* uniq_idx, int32_t* n_uniq_dev,
                        void* ws, int64_t ws_bytes, void* stream);
int pxr_sasrec_occ_segsum(const void* ws, int64_t ws_bytes, int B, int L, const float* dx0, const float* out,
                          const float* coef, int D, int64_t n_table, float scale, const int32_t* n_uniq_dev,
                          float* uniq_rows, void* stream);

/* Phase 2 for big batches (round 5): the same sums, with the rows of more than 1 024 occurrences (a Zipf-popular item of a
 * 2 048-sequence batch has 13 000) cut into parts of 512 occurrences that many workgroups sum, the parts of a row added in part
 * order -- three launches instead of one, worth it from ~30 000 occurrences.  Bit-reproducible; long rows are associated
 * differently from pxr_sasrec_occ_segsum (part by part).  ws2: pxr_sasrec_occ_split_ws_bytes(B, L, D) bytes (0 = shape not
 * served) whose first 256 bytes are ZERO at the first call; every call leaves them zero.  Replaces the scatter-add of the
 * embedding backward (sasrec.py:68 autograd), as pxr_sasrec_occ_segsum does. */
int64_t pxr_sasrec_occ_split_ws_bytes(int B, int L, int D);
int pxr_sasrec_occ_segsum_split(const void* ws, int64_t ws_bytes, int B, int L, const float* dx0, const float* out,
                                const float* coef, int D, int64_t n_table, float scale, const int32_t* n_uniq_dev,
                                float* uniq_rows, void* ws2, int64_t ws2_bytes, void* stream);

/* ---- LayerNorm sites ---------------------------------------------------------------------------------------- */
/* y = dropout(LN(table[idx[b*idx_bstride+t]] + pos[t]))       sasrec.py:68,77-82 (train) / :99-104 (predict).
 * xhat [B*L,D] / rstd [B*L] are saved for the backward and may be NULL for inference. */
int pxr_input_ln_fwd_f32(const float* table, int64_t n_table, const int64_t* idx, int64_t idx_bstride,
                         const float* pos, const float* gamma, const float* beta, float eps, int B, int L, int D,
                         float* y, float* xhat, float* rstd, float p_drop, uint64_t seed, uint32_t stream_id,
                         const int64_t* step_dev, void* stream);
/* y = LN(dropout(x) + res)                                      layers.py:614-615 and :670-671 */
int pxr_ln_residual_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                            int rows, int D, float* y, float* xhat, float* rstd, float p_drop, uint64_t seed,
                            uint32_t stream_id, const int64_t* step_dev, void* stream);
/* autograd of either site.  gather_mode=1: dy is w.r.t. the dropped output, dz = grad of (table row + pos).
 * gather_mode=0: dz = grad w.r.t. res, dx (optional) = grad w.r.t. x.  dgamma/dbeta are overwritten; pass both NULL to
 * defer the final reduction (partials stay in ws, reduce them with pxr_reduce_partials_multi_f32). */
int64_t pxr_ln_bwd_ws_bytes(int rows, int D);
int pxr_ln_bwd_partial_rows(int rows);   /* rows of the [P, 2*D] partial buffer left in ws when dgamma/dbeta are NULL */
int pxr_ln_bwd_f32(int gather_mode, const float* dy, const float* xhat, const float* rstd, const float* gamma,
                   int rows, int D, float* dz, float* dx, float* dgamma, float* dbeta, float p_drop, uint64_t seed,
                   uint32_t stream_id, const int64_t* step_dev, void* ws, int64_t ws_bytes, void* stream);

/* ---- fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32) --------------------------------------------------------------- */
/* General: C[M,N] = A_op x B_op; a_kc/b_kc select k-contiguous ([M][K] / [N][K]) or x-contiguous ([K][M] / [K][N])
 * storage.  epilogue: 0 none, 1 +bias[n], 2 +bias then erf-GELU (pre-activation -> aux), 3 *= gelu'(aux), 4 += aux,
 * 5 +bias then erf-GELU (gelu'(pre-activation) -> aux), 6 *= aux. */
int64_t pxr_gemm_ws_bytes(int a_kc, int b_kc, int M, int N, int K);
int pxr_gemm_f32(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb,
                 float* C, int64_t ldc, int epilogue, const float* bias, float* aux, int64_t ldaux, void* ws,
                 int64_t ws_bytes, int tile_hint, int split_hint, void* stream);
/* `batch` independent GEMMs of one shape in one launch (grid.z).  Operand z starts (z / nb2) * x1 + (z % nb2) * x2
 * floats after its base pointer (two-level strides: image n and head h of a packed [n, T, 3*heads*d] projection).  No
 * epilogue, no split-K.  The attention contractions of the ViT image encoder (HF CLIPAttention, built by the
 * reference at model/load.py:94): S = Q K^T, O = P V, dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q. */
int pxr_gemm_batched_f32(int a_kc, int b_kc, int M, int N, int K, const float* A, int64_t lda, const float* B,
                         int64_t ldb, float* C, int64_t ldc, int batch, int nb2, int64_t a1, int64_t a2, int64_t b1,
                         int64_t b2, int64_t c1, int64_t c2, int tile_hint, void* stream);

/* GEMM mode of the process (also PXR_GEMM_MODE=bf16x3|f32, default bf16x3): with bf16x3 every product pxr_gemm_f32 /
 * pxr_gemm_batched_f32 / pxr_linear_* / pxr_grouped_linear_bwd_weight_f32 would run with its heuristic tile is computed
 * on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16) after an EXACT split of each fp32 operand into three bf16 terms,
 * six cross products, fp32 accumulation (csrc/gemm_b3.cuh: error bound ~2^-25 |a||b| per product, i.e. fp32-class;
 * tests/test_gpu_gemm_b3.py).  f32 = the f32-input MFMA kernels (v_mfma_f32_32x32x2_f32).  An explicit f32 tile_hint
 * always takes the f32 kernels; tile_hint 9064 / 91281 force the bf16x3 64x64 / 128x128 tile. */
int pxr_set_gemm_mode(int bf16x3);
int pxr_get_gemm_mode(void);

/* ---- pre-split operands ("planes"): the same fp32 products on the bf16 matrix pipe, operands split ONCE ---------- */
/* An fp32 matrix X[rows, cols] (cols % 32 == 0) as three bf16 planes hi | mid | lo with X = hi + mid + lo exactly
 * (hi = bf16(X), mid = bf16(X - hi), lo = X - hi - mid).  Plane q starts q * plane_stride ELEMENTS after `planes`; inside
 * a plane the matrix is stored as cols / 32 PANELS of panel_rows (>= rows, multiple of 32; rows past `rows` must be ZERO when the rows are a GEMM's reduction dimension) rows x 32 columns:
 *     element (r, c) at ((c / 32) * panel_rows + r) * 32 + ((((c / 8) % 4) ^ ((r / 4) % 4)) * 8 + c % 8
 * (64-byte row segments, 16-byte chunks XOR-swizzled by the row: a GEMM tile is a byte copy of 1 KiB runs, csrc/gemm_p3.cuh).
 * A row range [r0, r1) with r0 % 16 == 0 is the same layout at planes + 32 r0; a column range with c0 % 32 == 0 at
 * planes + (c0 / 32) * panel_rows * 32.  pxr_split_planes_f32 writes the planes of an operand whose producer does not
 * (weights after an optimizer step, the item table before a full-sort evaluation -- model/IDNet/sasrec.py:112,115-117). */
int pxr_split_planes_f32(const float* x, int64_t rows, int64_t cols, int64_t ldx, void* planes, int64_t plane_stride,
                         int64_t panel_rows, void* stream);
/* C[M,N] = A[M,K] x B (+ epilogue as pxr_gemm_f32) with both operands given as planes: b_kc = 1: B is [N][K] (nn.Linear
 * forward model/layers.py:586-588,613,666,669; scoring model/IDNet/sasrec.py:112), b_kc = 0: B is [K][N] (the input
 * gradient dX = dY W of the same layers).  What pxr_gemm_f32 computes in mode bf16x3, bit for bit.  K % 32 == 0.
 * C may be NULL when only the output planes (c_planes: panel layout of C, N % 32 == 0) are wanted; c_planes may be NULL. */
int pxr_gemm_planes_f32(int b_kc, int M, int N, int K, const void* A, int64_t a_plane_stride, int64_t a_panel_rows,
                        const void* B, int64_t b_plane_stride, int64_t b_panel_rows, float* C, int64_t ldc, int epilogue,
                        const float* bias, float* aux, int64_t ldaux, void* c_planes, int64_t c_plane_stride,
                        int64_t c_panel_rows, int act, int tile_hint, void* stream);

/* up to 16 matrices in one launch (host arrays of n entries): the weight matrices of the block after an optimizer step */
int pxr_split_planes_multi_f32(int n, const float* const* x, const int64_t* rows, const int64_t* cols, const int64_t* ldx,
                               void* const* planes, const int64_t* plane_stride, const int64_t* panel_rows, void* stream);
/* ---- the TWO-plane fp16 operand format ("h2", csrc/planes.cuh): x 2^e = hi + lo, both fp16, 22 significant bits; a product
 * needs three MFMAs (v_mfma_f32_32x32x16_f16) instead of the six of the 3 x bf16 split -- same accuracy, half the matrix-pipe
 * work (profiles/r04/lab/h2_lab_run1.log).  fp16 has a finite range: the producer of an operand picks the power of two 2^e
 * (exact) that places its values in it and the GEMM undoes it; a value that still leaves the range sets bit 64 of the status
 * word (pxr_set_status_word) instead of silently becoming inf.  Same panel layout as the bf16 planes, planes 0 and 1.
 * Used by the forward-only blocks of the image tower (reference: the frozen CLIP blocks of code/REC/model/load.py:90-120).
 *   pxr_split_h2_multi_f32   up to 16 matrices, matrix i multiplied by 2^scale_exp[i] first
 *   pxr_gemm_h2_f32          C = epilogue(2^-(a_exp+b_exp) A~ B~), B~ [N][K] (b_kc: forward; epilogue NONE | BIAS | BIAS_GELU |
 *                            BIAS_GELU_GRAD | BIAS_ACT_GRAD | BIAS_ADD | BIAS_QGELU | BIAS_QGELU_GRAD | BIAS_RELU) or [K][N] (input gradient; NONE | ADD | MUL);
 *                            exponents immediate or read from *_exp_dev; c_fmt 0: output planes as three bf16 planes, 1: as two
 *                            fp16 planes holding C 2^(*c_exp_dev) (unit scale when null)
 *   pxr_ln_residual_fwd_h2_f32 / pxr_input_ln_fwd_h2_f32 / pxr_attn_fwd_h2_f32 / pxr_tower_attn_fwd_h2_f32: the plane-writing
 *                            producers (same arguments as their *_planes_f32 versions) with h2 planes, unit scale */
int pxr_split_h2_multi_f32(int n, const float* const* x, const int64_t* rows, const int64_t* cols, const int64_t* ldx,
                           void* const* planes, const int64_t* plane_stride, const int64_t* panel_rows, const int* scale_exp,
                           void* stream);
int pxr_gemm_h2_f32(int b_kc, int M, int N, int K, const void* A, int64_t a_plane_stride, int64_t a_panel_rows, int a_exp,
                    const int* a_exp_dev, const void* B, int64_t b_plane_stride, int64_t b_panel_rows, int b_exp,
                    const int* b_exp_dev, float* C, int64_t ldc, int epilogue, const float* bias, float* aux, int64_t ldaux,
                    void* c_planes, int64_t c_plane_stride, int64_t c_panel_rows, int c_fmt, const int* c_exp_dev, int act,
                    int tile_hint, void* stream);
/* the weight gradients of pxr_grouped_dw_planes_f32 from h2 operands (exponents per problem: immediate, or read from *_exp_dev[i]) */
int pxr_grouped_dw_h2_f32(int n, const void* const* dy, const int64_t* dy_plane_stride, const int64_t* dy_panel_rows,
                          const int* dy_exp, const int* const* dy_exp_dev, const void* const* x, const int64_t* x_plane_stride,
                          const int64_t* x_panel_rows, const int* x_exp, const int* const* x_exp_dev, float* const* dW,
                          float* const* db, const int* T, const int* N, const int* K, int tile_hint, void* stream);
/* h2 split with the scale chosen ON THE DEVICE (tensors that change every step): per matrix max |x| -> stats[2 i] (col_stats: also
 * the largest column sum of |x| -> stats[2 i + 1]; col_stats 0: rows * max instead, an upper bound; col_stats 2: stats[2 i] was
 * gathered by the producers, no statistics pass) and e = top - ceil(log2 max) -> exps[i], top = 14 unless bits 8-15 of
 * col_stats name another (8 .. 15: more headroom for tensors someone rewrites in place with the same exponent -- the weight
 * planes pxr_adamw_flat_tab_ex_f32 keeps current); no host synchronisation.
 * pxr_h2_bound_exp: *exp_out = 15 - ceil(log2(a_max[0] * b_colsum[0] * factor)) -- the exponent of an input gradient that leaves
 * a GEMM epilogue as planes before its maximum can be known (|dy W| <= max |dy| * max column sum of |W|). */
int pxr_h2_split_auto_multi_f32(int n, const float* const* x, const int64_t* rows, const int64_t* cols, const int64_t* ldx,
                                void* const* planes, const int64_t* plane_stride, const int64_t* panel_rows, int col_stats,
                                float* stats, int* exps, void* stream);
int pxr_h2_bound_exp(const float* a_max, const float* b_colsum, float factor, int* exp_out, void* stream);
/* Producers that gather the statistics themselves (col_stats = 2 of pxr_h2_split_auto_multi_f32 then skips its own pass): the
 * LayerNorm backward of a residual site / the fused attention backward, as pxr_ln_bwd_f32(gather_mode 0) / pxr_attn_bwd_f32, plus
 * PARTIAL maxima of |gradient the next GEMMs read| (dx when given, else dz) / of max(|dq|, |dk|, |dv|), reduced by
 * pxr_h2_split_parts_f32: the LayerNorm backward writes stat[w] for each of its pxr_ln_bwd_partial_rows(rows) workgroups (plain
 * stores, nothing to zero) and clears `zero_n` (<= 256) floats at `zero` on request; the attention backward raises
 * PXR_ATTN_STAT_SLOTS = 64 words the caller (or that LayerNorm launch) zeroed, one atomic per workgroup.  (Round 4 raised one
 * word once per wave: thousands of same-address atomics per launch.) */
int pxr_ln_bwd_stat_f32(const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int D, float* dz,
                        float* dx, float* dgamma, float* dbeta, float p_drop, uint64_t seed, uint32_t stream_id,
                        const int64_t* step_dev, void* ws, int64_t ws_bytes, float* stat, float* zero, int zero_n, void* stream);
/* The same site in a PRE-LN block (the image tower; HF CLIPEncoderLayer, reached from the reference's REC/model/load.py:90-120):
 * dz = res + LayerNorm-backward(dy), one launch instead of pxr_ln_bwd_f32 + pxr_add_f32 (the same bits); no dropout at these sites.
 * stat optional (NULL: no statistics). */
int pxr_ln_bwd_res_f32(const float* dy, const float* xhat, const float* rstd, const float* gamma, const float* res, int rows, int D,
                       float* dz, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, float* stat, void* stream);
int pxr_h2_split_parts_f32(const float* x, int64_t rows, int64_t cols, int64_t ldx, void* planes, int64_t plane_stride,
                           int64_t panel_rows, const float* parts, int n_parts, float* stats, int* exps,
                           const float* bound_b_colsum, float bound_factor, int* bound_exp_out, void* stream);
/* (bound_exp_out, optional: the launch also leaves the exponent pxr_h2_bound_exp would compute for the input gradient the next
 * GEMM forms from x and the weight whose column-sum statistic is *bound_b_colsum -- one one-thread launch less per use) */
int pxr_attn_bwd_stat_f32(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v, int64_t ld,
                          const float* probs, int B, int H, int L, int d, float* dq, float* dk, float* dv, int64_t ld_d,
                          float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, float* stat, void* stream);
/* ---- stale scales (round 6): the three h2 split launches per layer of a training step's backward pass removed.  The gradients a
 * backward pass hands to its GEMMs change slowly from step to step, so their producers write the planes THEMSELVES under an exponent
 * that exists before they run -- derived from the PREVIOUS step's maximum, `headroom` binades below the usual placement -- and leave
 * this step's partial maxima for the next derivation.  A value that outgrows the headroom is SATURATED to +-65504 (never inf) and
 * raises PXR_STATUS_H2_STALE (128) in the status word.
 * pxr_ln_bwd_h2s_f32: a residual LayerNorm site's backward (reference layers.py:614-615 / :670-671 under autograd); dz as fp32, the
 *   gradient the next GEMMs read (dropout applied when p_drop > 0; no fp32 copy) ONLY as two fp16 planes of gradient * 2^g_exp_dev[0];
 *   stat[pxr_ln_bwd_partial_rows(rows)] partial maxima; zero / zero_n as pxr_ln_bwd_stat_f32.  pos_score != NULL: the loss head's
 *   backward fused in as in pxr_bpr_ln_bwd_f32 (dy unused, rows == B * L); NULL: the eleven head arguments are ignored.
 * pxr_attn_bwd_h2s_f32: dq | dk | dv ONLY as such planes (column ranges as in pxr_attn_bwd_planes_f32) + the 64 spread maxima.
 * pxr_h2_sites_update: n <= 16 sites; per site m = max(maximum of its n_parts[s] partial maxima, run_max[s] * decay) (the maxima are
 *   heavy-tailed: the scale follows a decaying maximum of the recent steps; run_max persistent, zero-initialised) -> exps[s] (m 2^e in
 *   [2^(13-headroom), 2^(14-headroom))), stats[2 s ..] = (max 2^headroom, rows[s] max 2^headroom) and, where bound_b[s] (the largest
 *   column sum of |W| of the weight behind the site) is given, bexp[s] = 15 - ceil(log2(max 2^headroom * bound_b[s][0] * bound_factor)):
 *   the exponent of the planes a GEMM epilogue writes from site s (pxr_h2_bound_exp's rule).  A site without gradient keeps its entries.
 *   Run it once per step after the LAST reader of the site exponents; seed it with one exact pass (pxr_ln_bwd_stat_f32 /
 *   pxr_attn_bwd_stat_f32 + pxr_h2_split_parts_f32 leave the same partial maxima). */
int pxr_ln_bwd_h2s_f32(const float* pos_score, const float* neg_score, const float* table, int64_t n_table, const int64_t* items,
                       const int64_t* masked_index, int B, int L, float grad_scale, const float* grad_scale_dev, float* coef,
                       const float* dy, const float* xhat, const float* rstd, const float* gamma, int rows, int D, float* dz,
                       float* dgamma, float* dbeta, float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev,
                       void* ws, int64_t ws_bytes, void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows,
                       const int* g_exp_dev, float* stat, float* zero, int zero_n, void* stream);
int pxr_attn_bwd_h2s_f32(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v, int64_t ld,
                         const float* probs, int B, int H, int L, int d, float p_drop, uint64_t seed, uint32_t stream_id,
                         const int64_t* step_dev, void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows, int g_cols,
                         int col_q, int col_k, int col_v, const int* g_exp_dev, float* stat, void* stream);
int pxr_h2_sites_update(int n, const float* const* parts, const int* n_parts, const int* rows, const float* const* bound_b,
                        float bound_factor, int headroom, float decay, float* run_max, int* exps, float* stats, int* bexp, void* stream);
int pxr_ln_residual_fwd_h2_f32(const float* x, const float* res, const float* gamma, const float* beta, float eps, int rows,
                               int D, float* y, float* xhat, float* rstd, float p_drop, uint64_t seed, uint32_t stream_id,
                               const int64_t* step_dev, void* y_planes, int64_t y_plane_stride, int64_t y_panel_rows,
                               void* stream);
int pxr_input_ln_fwd_h2_f32(const float* table, int64_t n_table, const int64_t* idx, int64_t idx_bstride, const float* pos,
                            const float* gamma, const float* beta, float eps, int B, int L, int D, float* y, float* xhat,
                            float* rstd, float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev,
                            void* y_planes, int64_t y_plane_stride, int64_t y_panel_rows, void* stream);
int pxr_attn_fwd_h2_f32(const float* q, const float* k, const float* v, int64_t ld, const int64_t* keymask, int64_t km_bstride,
                        int B, int H, int L, int d, float* ctx, int64_t ld_ctx, float* probs, float p_drop, uint64_t seed,
                        uint32_t stream_id, const int64_t* step_dev, void* ctx_planes, int64_t ctx_plane_stride,
                        int64_t ctx_panel_rows, void* stream);
int pxr_tower_attn_fwd_h2_f32(const float* q, const float* k, const float* v, int64_t ld, int64_t images, int heads, int T, int d,
                              float scale, float* ctx, int64_t ld_ctx, void* ctx_planes, int64_t ctx_plane_stride,
                              int64_t ctx_panel_rows, float* lse, void* stream);

/* Producers that write their output straight as planes (same arguments as the functions they extend + the planes matrix;
 * planes == NULL: exactly the plain function).  LayerNorm sites: y as planes; LayerNorm backward (residual sites): the
 * gradient the next GEMMs read (dx when given, else dz); attention: ctx as the [B*L, H*d] matrix (ctx may then be NULL),
 * dq | dk | dv as column ranges starting at col_q / col_k / col_v of one [B*L, g_cols] matrix (dq, dk, dv may then all be
 * NULL).  pxr_attn_planes_supported(L, d): whether the fused attention kernels that can do so serve the shape. */
int pxr_input_ln_fwd_planes_f32(const float* table, int64_t n_table, const int64_t* idx, int64_t idx_bstride,
                                const float* pos, const float* gamma, const float* beta, float eps, int B, int L, int D,
                                float* y, float* xhat, float* rstd, float p_drop, uint64_t seed, uint32_t stream_id,
                                const int64_t* step_dev, void* y_planes, int64_t y_plane_stride, int64_t y_panel_rows,
                                void* stream);
int pxr_ln_residual_fwd_planes_f32(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                                   int rows, int D, float* y, float* xhat, float* rstd, float p_drop, uint64_t seed,
                                   uint32_t stream_id, const int64_t* step_dev, void* y_planes, int64_t y_plane_stride,
                                   int64_t y_panel_rows, void* stream);
int pxr_ln_bwd_planes_f32(int gather_mode, const float* dy, const float* xhat, const float* rstd, const float* gamma,
                          int rows, int D, float* dz, float* dx, float* dgamma, float* dbeta, float p_drop, uint64_t seed,
                          uint32_t stream_id, const int64_t* step_dev, void* ws, int64_t ws_bytes, void* g_planes,
                          int64_t g_plane_stride, int64_t g_panel_rows, void* stream);
int pxr_attn_planes_supported(int L, int d);
int pxr_attn_fwd_planes_f32(const float* q, const float* k, const float* v, int64_t ld, const int64_t* keymask,
                            int64_t km_bstride, int B, int H, int L, int d, float* ctx, int64_t ld_ctx, float* probs,
                            float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* ctx_planes,
                            int64_t ctx_plane_stride, int64_t ctx_panel_rows, void* stream);
int pxr_attn_bwd_planes_f32(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v, int64_t ld,
                            const float* probs, int B, int H, int L, int d, float* dq, float* dk, float* dv, int64_t ld_d,
                            float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* g_planes,
                            int64_t g_plane_stride, int64_t g_panel_rows, int g_cols, int col_q, int col_k, int col_v,
                            void* stream);
/* Measurement hook of bench.py (no reference analogue): registers two uint64 in DEVICE memory (NULL unregisters).  Every later
 * main-pass launch of the fused scoring's DEFAULT (reduced-product) threshold kernel adds the shader-clock cycles (s_memtime) and the constant 100 MHz reference ticks
 * (s_memrealtime) that its workgroup 0 lived through: clk2[0] / clk2[1] * 0.1 = the clock in GHz the part sustained INSIDE those
 * kernels (it lowers its clock under MFMA load; the nominal 2.4 GHz is what the 2.5 PFLOP/s peak assumes).  The caller zeroes the
 * buffer.  Process-wide; replaces round 5's pxr_clock_probe_f32 (a one-wave probe on a second stream, which read the idle clock). */
int pxr_score_topk_clock_out(uint64_t* clk2);
/* Host-side recovery after PXR_STATUS_GEMM_TIMEOUT (a stream-K / split-K worker gave up waiting for a partial tile): waits for
 * the device and zeroes every stream's flag words, so that later launches start from the state they expect.  No reference
 * analogue: this build's own synchronisation (gemm_f32.hip stream-K, gemm_p3.hip split-K weight gradients). */
int pxr_gemm_reset_flags(void);
/* pxr_grouped_linear_bwd_weight_f32 from planes: dW[i][N_i,K_i] = dy[i][T_i,N_i]^T x[i][T_i,K_i], db[i][N_i] = column sums
 * of dy[i] (db[i] may be NULL), all problems in one launch.  dy[i] / x[i] are planes of the [T_i, .] matrices whose panel
 * rows (multiples of 32) T_i .. panel_rows-1 are ZERO.  N_i, K_i multiples of 32.  Autograd of model/layers.py:586-588,613,
 * 666,669. */
int pxr_grouped_dw_planes_f32(int n, const void* const* dy, const int64_t* dy_plane_stride, const int64_t* dy_panel_rows,
                              const void* const* x, const int64_t* x_plane_stride, const int64_t* x_panel_rows,
                              float* const* dW, float* const* db, const int* T, const int* N, const int* K, int tile_hint,
                              void* stream);

/* ---- ViT image encoder, non-GEMM pieces (csrc/vit.hip) ------------------------------------------------------ */
/* Fused self-attention of a tower block, forward (csrc/tower_attn.hip): per (image, head)
 * ctx[b*T + t, 64 h ..] = softmax_t'(scale * q_t . k_t') v_t'; no mask, no dropout (HF CLIPAttention.forward as the item tower of
 * REC/model/modules.py runs it).  q/k/v fp32, element (b, t, h, c) at p[(b*T + t)*ld + 64 h + c]; head size 64, T <= 288
 * (pxr_tower_attn_supported).  Outputs: ctx fp32 [images*T, ld_ctx] and/or ctx as planes (at least one); lse (optional)
 * [images*heads, T] = log sum_t' exp(scale * q.k).  The score matrix never exists in memory. */
int pxr_tower_attn_supported(int T, int d);
int pxr_tower_attn_fwd_f32(const float* q, const float* k, const float* v, int64_t ld, int64_t images, int heads, int T, int d,
                           float scale, float* ctx, int64_t ld_ctx, void* ctx_planes, int64_t ctx_plane_stride,
                           int64_t ctx_panel_rows, float* lse, void* stream);
/* Backward of pxr_tower_attn_fwd_f32 (the trainable blocks of the tower): dq | dk | dv from dctx, recomputing the
 * probabilities from the forward's lse -- no [images*heads, T, T] matrix is saved or written.  ctx = the forward's fp32
 * output (for delta = rowsum(dctx o ctx)); delta_ws = [images*heads, T] floats of scratch; dq / dk / dv are addressed like
 * q / k / v with row stride ld_d (three column ranges of one [images*T, 3*heads*64] matrix in the tower).  Two launches. */
int pxr_tower_attn_bwd_f32(const float* q, const float* k, const float* v, int64_t ld, const float* dctx, const float* ctx,
                           int64_t ld_c, const float* lse, int64_t images, int heads, int T, int d, float scale, float* dq,
                           float* dk, float* dv, int64_t ld_d, float* delta_ws, void* stream);
/* in place: S[row, :T] = softmax(scale * S[row, :T]), S[row, T:ld] = 0      (HF CLIPAttention, no mask / dropout) */
int pxr_softmax_rows_f32(float* S, int64_t rows, int T, int ld, float scale, void* stream);
/* in place on dP: dS = scale * P o (dP - rowsum(dP o P))                    (autograd of the above) */
int pxr_softmax_rows_bwd_f32(const float* P, float* dP, int64_t rows, int T, int ld, float scale, void* stream);
/* out[n,t,:] = (t == 0 ? cls : patches[n,t-1,:]) + pos[t,:]                (HF CLIPVisionEmbeddings.forward) */
int pxr_vit_embed_f32(const float* patches, const float* cls, const float* pos, float* out, int64_t n, int T, int H,
                      void* stream);
/* out[n,:] = mean_t x[n,t,:]                                                (MeanItemEncoder, model/layers.py:128) */
int pxr_token_mean_f32(const float* x, float* out, int64_t n, int T, int D, void* stream);
/* dact[n,t,:] = act[n,t,:] > 0 ? dout[n,:] / T : 0                          (through the mean and rec_fc's ReLU) */
int pxr_token_mean_relu_bwd_f32(const float* dout, const float* act, float* dact, int64_t n, int T, int D, void* stream);
/* SASRec attention for MAX_ITEM_LIST_LENGTH > 128 (beyond the fused kernels behind pxr_attn_fwd_f32): the scores come
 * from a batched GEMM (S = Q K^T, unscaled, [B*H, L, ld]); this turns them in place into softmax(S / sqrt(d) + mask) with
 * the reference's additive -1e9 causal + key mask (model/layers.py:595-604, model/IDNet/sasrec.py:119-126) and writes
 * the dropped probabilities (layers.py:608; same counter hash and element numbering as pxr_attn_fwd_f32) to PD. */
int pxr_attn_rows_fwd_f32(float* S, float* PD, const int64_t* keymask, int64_t km_bstride, int B, int H, int L, int ld,
                          float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, int d, void* stream);
/* in place on dPD (gradient w.r.t. the dropped probabilities): gradient w.r.t. the unscaled scores */
int pxr_attn_rows_bwd_f32(const float* P, float* dPD, int B, int H, int L, int ld, float p_drop, uint64_t seed,
                          uint32_t stream_id, const int64_t* step_dev, int d, void* stream);
/* out = a + b (n floats, n % 4 == 0): the two branches of a residual-stream gradient */
int pxr_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream);
/* y = dropout(x): y[i] = x[i] / (1 - p) where the library's counter hash of (seed + *step_dev, stream_id, i) keeps element i, else
 * 0 (n floats, n % 4 == 0, 16-byte aligned; step_dev may be NULL).  Applied to the upstream gradient it is its own backward: the
 * mask is regenerated, not stored.  Replaces nn.Dropout on the gathered item embeddings of GRU4Rec
 * (REC/model/IDNet/gru4rec.py:26,59); masks restated for the oracle in oracle/dropout_rng.py. */
int pxr_dropout_f32(const float* x, float* y, int64_t n, float p, uint64_t seed, uint32_t stream_id, const int64_t* step_dev,
                    void* stream);

/* y = x W^T + b (act=1: erf-GELU, pre-activation saved; act=2: erf-GELU, gelu'(pre-activation) saved)
 *                                                          layers.py:586-588,613,666-667,669; sasrec.py:112 */
int pxr_linear_fwd_f32(const float* x, const float* W, const float* b, float* y, float* pre, int M, int N, int K,
                       int act, void* stream);
/* dx = dy W, optionally * gelu'(dgelu_pre) OR + add (residual gradient) OR * mul (gelu' saved by act=2);
 * dW = dy^T x  -- autograd of nn.Linear */
int pxr_linear_bwd_input_f32(const float* dy, const float* W, float* dx, const float* dgelu_pre, const float* add,
                             const float* mul, int M, int N, int K, void* stream);
int pxr_linear_bwd_weight_f32(const float* dy, const float* x, float* dW, int M, int N, int K, void* ws,
                              int64_t ws_bytes, void* stream);
/* Weight AND bias gradients of up to 16 nn.Linear layers in ONE launch (host arrays of n device pointers / sizes):
 * dW[i][N_i,K_i] = dy[i][M_i,N_i]^T x[i][M_i,K_i], db[i][N_i] = column sums of dy[i] (db[i] may be NULL).
 * No split-K, no partial buffers: all problems' 64x64 tiles share one grid.  Deterministic. */
int pxr_grouped_linear_bwd_weight_f32(int n, const float* const* dy, const float* const* x, float* const* dW,
                                      float* const* db, const int* M, const int* N, const int* K, void* stream);
/* out[n] = sum_m x[m,n]  (bias grads; position-embedding grad = colsum of dx0 viewed [B, L*D]); deterministic */
int64_t pxr_colsum_ws_bytes(int M, int N);
int pxr_colsum_partial_rows(int M);      /* rows of the [P, N] partial buffer left in ws when out is NULL */
/* out_a[i][c] (c < split[i]) / out_b[i][c-split[i]] = sum_p part[i][p][c] for up to 16 partial buffers in one launch
 * (host arrays of n entries; out_b[i] may be NULL => single output).  bump_counter (optional): a device counter
 * incremented by one by the same launch (the dropout step counter at the end of a backward pass). */
int pxr_reduce_partials_multi_f32(int n, const float* const* part, const int* P, const int* N, float* const* out_a,
                                  float* const* out_b, const int* split, int64_t* bump_counter, void* stream);
int pxr_colsum_f32(const float* x, int64_t ldx, int M, int N, float* out, void* ws, int64_t ws_bytes, void* stream);

/* ---- masked multi-head self-attention core ------------------------------------------------------------------ */
/* ctx = softmax(q k^T / sqrt(d) + mask) v with the reference's additive -1e9 causal+padding mask
 * (layers.py:590-612, sasrec.py:119-126).  q/k/v element (b,t,h,c) at p[(b*L+t)*ld + h*d + c] (fused QKV output);
 * key j of batch b is real iff keymask[b*km_bstride + j] != 0 (masked_index in training, item_seq in predict).
 * ctx is written head-merged [B*L, ld_ctx]; probs [B,H,L,L] (pre-dropout) is saved for backward, may be NULL. */
int pxr_attn_fwd_f32(const float* q, const float* k, const float* v, int64_t ld, const int64_t* keymask,
                     int64_t km_bstride, int B, int H, int L, int d, float* ctx, int64_t ld_ctx, float* probs,
                     float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* stream);
int pxr_attn_bwd_f32(const float* dctx, int64_t ld_ctx, const float* q, const float* k, const float* v, int64_t ld,
                     const float* probs, int B, int H, int L, int d, float* dq, float* dk, float* dv, int64_t ld_d,
                     float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* stream);

/* ---- GRU4Rec (code/REC/model/IDNet/gru4rec.py; torch.nn.GRU, bias=False): the gate arithmetic of one time step --------- */
/* r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z), n = tanh(gi_n + r gh_n), h = (1 - z) n + z h_prev; gi, gh [B, 3H] in
 * torch's r | z | n order, h_prev [B, H] or NULL (zeros), save (optional) [B, 4H] = r | z | n | gh_n.  The matrix products
 * around it are pxr_linear_* / pxr_gemm_f32 calls. */
int pxr_gru_gates_fwd_f32(const float* gi, const float* gh, const float* h_prev, float* h_out, float* save, int64_t B, int H,
                          void* stream);
/* dh = everything that reaches h_t  ->  dgi, dgh [B, 3H] and the direct part dh * z of d h_{t-1} (autograd of the above) */
int pxr_gru_gates_bwd_f32(const float* dh, const float* save, const float* h_prev, float* dgi, float* dgh, float* dh_prev,
                          int64_t B, int H, void* stream);

/* ---- NextItNet (code/REC/model/IDNet/nextitnet.py:160-194): the causal dilated convolution as a GEMM ------------------ */
/* xcol[b L + t, c k + j] = x[b, t - (k-1-j) dilation, c] (0 left of the sequence): conv = xcol . W.view(C_out, C_in k)^T + bias
 * with the reference's Conv2d weight [C_out, C_in, 1, k] used in place.  col2im is its transpose (the input gradient). */
int pxr_causal_im2col_f32(const float* x, float* xcol, int64_t B, int L, int C, int k, int dilation, void* stream);
int pxr_causal_col2im_f32(const float* dxcol, float* dx, int64_t B, int L, int C, int k, int dilation, void* stream);

/* ---- training head ------------------------------------------------------------------------------------------ */
/* loss = mean_b(-sum_t log(sigmoid(pos-neg)+1e-8) * mask)            sasrec.py:88-92; loss stays on the device */
int pxr_bpr_loss_fwd_f32(const float* out, const float* table, int64_t n_table, const int64_t* items,
                         const int64_t* masked_index, int B, int L, int D, float* pos_score, float* neg_score,
                         float* lossrow, float* loss, void* stream);
int pxr_bpr_loss_bwd_f32(const float* pos_score, const float* neg_score, const float* table, int64_t n_table,
                         const int64_t* items, const int64_t* masked_index, int B, int L, int D, float grad_scale,
                         const float* grad_scale_dev, float* dout, float* coef, void* stream);
/* The second stage of pxr_bpr_loss_fwd_f32 alone (loss = 1/B sum_b sum_t lossrow[b,t], fixed order). */
int pxr_bpr_loss_reduce_f32(const float* lossrow, int B, int L, float* loss, void* stream);
/* The block's LAST LayerNorm (reference layers.py:670-671, the output sasrec.py:86 names) with the loss head's forward
 * (sasrec.py:88-92) fused in: y = LN(dropout(x) + res) over B*L rows, pos / neg scores and the loss as pxr_bpr_loss_fwd_f32
 * would compute them from y -- bit-identical, one launch and one pass over y less. */
int pxr_ln_residual_bpr_fwd_f32(const float* x, const float* res, const float* gamma, const float* beta, float eps, int B, int L,
                                int D, float* y, float* xhat, float* rstd, float p_drop, uint64_t seed, uint32_t stream_id,
                                const int64_t* step_dev, const float* table, int64_t n_table, const int64_t* items,
                                const int64_t* masked_index, float* pos_score, float* neg_score, float* lossrow, float* loss,
                                void* stream);
/* Its backward: pxr_bpr_loss_bwd_f32 + pxr_ln_bwd_planes_f32 (gather_mode 0) in one launch -- the gradient w.r.t. the block's
 * output is formed per row in registers from the saved scores instead of being written and read back; coef [B*L] is written
 * for pxr_sasrec_occ_segsum.  g_planes / stat optional (the two forms of pxr_ln_bwd_planes_f32 / pxr_ln_bwd_stat_f32). */
int pxr_bpr_ln_bwd_f32(const float* pos_score, const float* neg_score, const float* table, int64_t n_table, const int64_t* items,
                       const int64_t* masked_index, int B, int L, float grad_scale, const float* grad_scale_dev, float* coef,
                       const float* xhat, const float* rstd, const float* gamma, int D, float* dz, float* dx, float* dgamma,
                       float* dbeta, float p_drop, uint64_t seed, uint32_t stream_id, const int64_t* step_dev, void* ws,
                       int64_t ws_bytes, void* g_planes, int64_t g_plane_stride, int64_t g_panel_rows, float* stat, void* stream);

/* ---- PixelNet (MOSASRec) ------------------------------------------------------------------------------------- */
/* Gradient w.r.t. the visual encoder's output viewed [B, L+1, 2, D] (pos_t | neg_t interleaved, PixelNet/
 * mosasrec.py:69-74,88-89): d_emb[b,t,0] = [t<L] dx0[b,t] + [t>=1] coef[b,t-1] out[b,t-1];  d_emb[b,t,1] = -[t>=1] ... */
int pxr_mosasrec_emb_grad_f32(const float* dx0, const float* out, const float* coef, int B, int L, int D, float* d_emb,
                              void* stream);
/* The reference's image transform on the GPU (data/dataset/trainset.py:85-96): uint8 HWC store [n_store,H,W,3] gathered
 * by item id -> fp32 [n,3,H,W] = (x/255 - 0.5)/0.5; id 0 (padding) -> zeros. */
int pxr_image_u8_to_f32(const uint8_t* store, int64_t n_store, int H, int W, const int64_t* ids, int n, float* out,
                        void* stream);

/* ---- full-sort evaluation ------------------------------------------------------------------------------------ */
/* Fused  scores = users x table^T (sasrec.py:112)  ->  scores[:,0] = -inf, scores[history] = -inf (trainer.py:333-336)
 * ->  top-K (collector.py:133): the [B,N] score matrix never reaches HBM.  users [B,D] with row stride ld_users;
 * hist_ptr int32 [B+1] + hist_items int64 = CSR of seq_eval_collate's (history_u, history_i) pairs (NULL = none);
 * K <= 32.  Outputs topk_idx int64 [B,K] / topk_val [B,K], descending. */
int64_t pxr_score_topk_ws_bytes(int B, int N, int K);
int pxr_score_topk_f32(const float* users, int64_t ld_users, int B, const float* table, int N, int D,
                       const int32_t* hist_ptr, const int64_t* hist_items, int K, int64_t* topk_idx, float* topk_val,
                       void* ws, int64_t ws_bytes, void* stream);

/* The same with both operands ALSO given as planes ("pre-split operands" above; both NULL: the plain function): on catalogues
 * that take the two-pass threshold schedule the pass over every item tile runs on the planes as one LDS-DMA stream; the
 * table's planes are made once per evaluation with pxr_split_planes_f32 (model/IDNet/sasrec.py:112,115-117). */
int pxr_score_topk_planes_f32(const float* users, int64_t ld_users, int B, const float* table, int N, int D,
                              const void* users_planes, int64_t users_plane_stride, int64_t users_panel_rows,
                              const void* table_planes, int64_t table_plane_stride, int64_t table_panel_rows,
                              const int32_t* hist_ptr, const int64_t* hist_items, int K, int64_t* topk_idx, float* topk_val,
                              void* ws, int64_t ws_bytes, void* stream);

/* ... and with the pass over the catalogue on `products` = 3 (hi*hi + mid*hi + hi*mid) or 1 (hi*hi) of the six bf16 products
 * (6: exactly the function above).  That pass only decides "score >= threshold": the threshold is lowered by a rigorous per-user
 * bound on what the dropped products and the different rounding can change (c ||user||_2 max_i ||table[i]||_2), and the few
 * survivors that can still reach the top K are re-scored with all six products in the full pass's own MFMA order -- ids and
 * values are those of pxr_score_topk_planes_f32, bit for bit, at about half (3) of its MFMA work.  Needs both operands as planes
 * and table_row_norm_max: DEVICE pointer to max_i ||table[i]||_2 (pxr_row_norm_max_f32, once per evaluation like the planes).
 * Same reference path: model/IDNet/sasrec.py:112 + trainer/trainer.py:327-337 + evaluator/collector.py:131-139. */
int pxr_score_topk_fast_f32(const float* users, int64_t ld_users, int B, const float* table, int N, int D,
                            const void* users_planes, int64_t users_plane_stride, int64_t users_panel_rows,
                            const void* table_planes, int64_t table_plane_stride, int64_t table_panel_rows,
                            const float* table_row_norm_max, int products, const int32_t* hist_ptr, const int64_t* hist_items,
                            int K, int64_t* topk_idx, float* topk_val, void* ws, int64_t ws_bytes, void* stream);
/* out[0] (device float) = max_i ||x[i, :]||_2 of x [rows, cols] fp32 (row stride ldx, cols % 4 == 0): the item-table statistic of
 * the function above (computed over what compute_item_all returns, model/IDNet/sasrec.py:115-117). */
int pxr_row_norm_max_f32(const float* x, int64_t rows, int64_t cols, int64_t ldx, float* out, void* stream);

/* ---- optimizer ---------------------------------------------------------------------------------------------- */
/* torch.optim.AdamW update (trainer.py:102,125), step is 1-based.  n must be a multiple of 4. */
int pxr_adamw_flat_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                       double beta2, double eps, double weight_decay, int64_t step, void* stream);
/* Dense-semantics AdamW over the whole table with the gradient given sparsely; slot is an int32[N] map that must be
 * all -1 on entry (pxr_slot_fill_i32 once) and is all -1 again on exit. */
int pxr_slot_fill_i32(int32_t* slot, int64_t n, int32_t value, void* stream);
int pxr_adamw_table_f32(float* table, float* m, float* v, int64_t n_rows, int D, int32_t* slot,
                        const int64_t* uniq_idx, const float* uniq_rows, const int32_t* n_uniq_dev, int64_t max_uniq,
                        double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                        void* stream);

/* Lazy (exact catch-up) form of the same dense-semantics table update: no O(N*D) sweep per step.  last int32[N]
 * holds, per row, the optimizer step through which the row is up to date; hyper (float4[capacity]) / cumlog
 * (double[capacity]) hold each step's scalars, appended once per step.  pxr_adamw_rows_f32 replays a row's missed
 * zero-gradient steps (bit-identical to the sweep for gaps <= 256 steps, closed-form weight decay beyond) through
 * t_prev and, if t_apply = t_prev+1, applies that step with gradient rows grows[i,:].  rows == NULL: all N rows
 * (flush before evaluation / checkpointing).  With step_dev the kernel takes t_prev = *step_dev + step_dev_bias from the
 * device (bias 1 = "through the step being applied": the rows of the NEXT batch, prefetched); max_blocks > 0 caps the
 * grid (a thin launch that shares the CUs with the step's GEMMs instead of flooding them).  hyper_append with advance != 0 is the end-of-step form: it first counts
 * the finished step (*step_dev += 1) and then appends the scalars of the next one, in one launch. */
int pxr_adamw_hyper_append(void* hyper, void* cumlog, int64_t capacity, int64_t step, int64_t* step_dev,
                           double lr, double beta1, double beta2, double eps, double weight_decay, int advance,
                           void* stream);
int pxr_adamw_rows_f32(float* table, float* m, float* v, int32_t* last, int64_t n_table, int D, const int64_t* rows,
                       const int32_t* n_rows_dev, int64_t max_rows, const float* grows, const void* hyper,
                       const void* cumlog, int64_t t_prev, int64_t t_apply, const int64_t* step_dev,
                       int64_t step_dev_bias, int64_t max_blocks, double beta1, double beta2, double eps,
                       void* stream);
/* The catch-up part of pxr_adamw_rows_f32 on a RAW id list (the batch's item tensor as it is): ids[n_ids] may hold
 * duplicates, 0 and out-of-range values (skipped).  The workgroup that raises last[row] to t_prev (atomicMax) replays the
 * row, the ones of its duplicates find it current -- so the rows a forward pass reads can be brought up to date BEFORE the
 * sort / unique of the batch's ids has run (it then runs beside the forward pass on a second stream). */
int pxr_adamw_rows_ids_f32(float* table, float* m, float* v, int32_t* last, int64_t n_table, int D, const int64_t* ids,
                           int64_t n_ids, const void* hyper, const void* cumlog, int64_t t_prev, const int64_t* step_dev,
                           double beta1, double beta2, double eps, void* stream);
/* The same over a 2-D window of an id tensor: n_lists rows of row_len ids, row r at ids[r * row_stride] -- the INPUT ids
 * items[:, 0, 0:L] of a batch [B, 2, L+1] (reference sasrec.py:68-70: the rows the forward pass gathers first), so that only
 * those rows stand between the batch and the first LayerNorm; the targets / negatives (read by the loss, sasrec.py:88-89) are
 * caught up beside the forward pass (pixelrec_amd/model/sasrec.py "split catch-up"). */
int pxr_adamw_rows_ids2d_f32(float* table, float* m, float* v, int32_t* last, int64_t n_table, int D, const int64_t* ids,
                             int64_t n_lists, int64_t row_len, int64_t row_stride, const void* hyper, const void* cumlog,
                             int64_t t_prev, const int64_t* step_dev, double beta1, double beta2, double eps,
                             void* cur_hyper_out, void* stream);
/* cur_hyper_out (optional, 16 bytes): the launch also copies the scalars of the optimizer step about to run (hyper entry
 * t_prev + 1) there.  pxr_adamw_flat_tab_ex_f32 = pxr_adamw_flat_tab_planes_f32 with two options: (a) seg_fmt = 1: the weight
 * segments leave the launch as fp16 two-plane operands (planes "h2"), segment i scaled by 2^seg_exps[i] (device ints: the
 * exponents the planes were last split with; the next forward needs no statistics + split launches); (b) cur_hyper != NULL: this
 * step's scalars are read from that slot and the launch CLOSES the step itself (counts it in *step_dev, appends the next entry:
 * what pxr_adamw_hyper_append(advance = 1) does in a launch of its own) -- legal because no workgroup of the launch reads the
 * counter its closing thread advances.  Reference: torch.optim.AdamW.step (trainer.py:125). */
int pxr_adamw_flat_tab_ex_f32(float* p, const float* g, float* m, float* v, int64_t n, void* hyper, void* cumlog, int64_t capacity,
                              int64_t step, int64_t* step_dev, const void* cur_hyper, double lr, double beta1, double beta2,
                              double eps, double weight_decay, int n_seg, const int64_t* seg_off, const int64_t* seg_rows,
                              const int64_t* seg_cols, void* const* seg_planes, const int64_t* seg_plane_stride,
                              const int64_t* seg_panel_rows, int seg_fmt, const int* seg_exps, void* stream);
/* pxr_adamw_flat_f32 with the step's scalars read from hyper[step] (or hyper[*step_dev + 1]). */
int pxr_adamw_flat_tab_f32(float* p, const float* g, float* m, float* v, int64_t n, const void* hyper, int64_t step,
                           const int64_t* step_dev, double beta1, double beta2, double eps, void* stream);
/* pxr_adamw_flat_tab_f32 that also writes the UPDATED values of n_seg (<= 16) weight matrices inside the flat buffer
 * ([seg_rows, seg_cols] row-major at element seg_off) as bf16x3 planes (see "pre-split operands"): the operands of the next
 * step's GEMMs come out of the optimizer (trainer.py:125) with no split launch. */
int pxr_adamw_flat_tab_planes_f32(float* p, const float* g, float* m, float* v, int64_t n, const void* hyper, int64_t step,
                                  const int64_t* step_dev, double beta1, double beta2, double eps, int n_seg,
                                  const int64_t* seg_off, const int64_t* seg_rows, const int64_t* seg_cols,
                                  void* const* seg_planes, const int64_t* seg_plane_stride, const int64_t* seg_panel_rows,
                                  void* stream);
/* *counter += delta on the device. */
int pxr_counter_add_i64(int64_t* counter, int64_t delta, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PXR_H_ */
