/*
 * gcnx.h -- C ABI of libgcnx.so: MI355X (gfx950) kernels for the GCN forward/backward
 * hot path of Sum02dean/GCN-STRING.
 *
 * The reference has no FFI or plugin interface for this path: its arithmetic runs inside
 * un-vendored Spektral/TensorFlow ops reached from src/scripts/gcn.py (call sites cited per
 * entry point below).  Each function here replaces one TensorFlow CPU op (SURVEY.md 2.3
 * "implicit kernel inventory" K1..K9 and their gradients) behind plain C types, so that the
 * Python host mirror of the Spektral call surface (gcn-string_amd/gcnx) binds it via ctypes.
 *
 * Conventions
 *   - every function returns int: 0 = GCNX_OK, otherwise a gcnx_status; the message is
 *     available from gcnx_last_error().  No C++ exception crosses this boundary.
 *   - all matrix operands are DEVICE pointers obtained from gcnx_malloc(), row-major fp32
 *     with an explicit leading dimension (elements) so that column slices of a wider buffer
 *     can be read/written in place (Spektral's connectivity="cat" skip, K7).
 *   - CSR: rowptr int32[N+1], colidx int32[nnz], vals fp32[nnz] or NULL (= all ones, the
 *     GeneralConv "sum" aggregation which ignores adjacency values).
 *   - a ctx is bound to one device and one HIP stream; calls are stream-ordered and return
 *     before the device finishes unless documented as synchronising.  A ctx is not
 *     thread-safe; distinct ctxs are independent.
 *   - the caller owns every buffer it allocates; the library keeps no user pointer after a
 *     call returns (captured graphs excepted: buffers used inside a capture must outlive it).
 */
#ifndef GCNX_H
#define GCNX_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define GCNX_API __attribute__((visibility("default")))
#else
#define GCNX_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gcnx_ctx gcnx_ctx;
typedef struct gcnx_comm gcnx_comm;
typedef struct gcnx_graph gcnx_graph;
typedef struct gcnx_event gcnx_event;
typedef struct gcnx_spmm_plan gcnx_spmm_plan;

typedef enum {
  GCNX_OK = 0,
  GCNX_ERR_INVALID = 1,     /* bad argument / shape / alignment */
  GCNX_ERR_HIP = 2,         /* HIP runtime error */
  GCNX_ERR_RCCL = 3,        /* RCCL error or librccl missing */
  GCNX_ERR_NOMEM = 4,
  GCNX_ERR_UNSUPPORTED = 5, /* valid request this build has no kernel for */
  GCNX_ERR_DATA = 6         /* device-side validation of input data failed */
} gcnx_status;

typedef enum { GCNX_ACT_NONE = 0, GCNX_ACT_RELU = 1, GCNX_ACT_PRELU = 2 } gcnx_act;
typedef enum { GCNX_POOL_SUM = 0, GCNX_POOL_AVG = 1, GCNX_POOL_MAX = 2 } gcnx_pool;
/* GEMM arithmetic: F32 = exact fp32 MFMA (v_mfma_f32_*_f32); BF16 = inputs rounded to bf16,
 * fp32 accumulate; BF16X3 = hi/lo bf16 split (16 significand bits per operand, 2^-18 relative), three MFMA passes. */
typedef enum { GCNX_PREC_F32 = 0, GCNX_PREC_BF16 = 1, GCNX_PREC_BF16X3 = 2 } gcnx_prec;
typedef enum { GCNX_NORM_SPEKTRAL = 0, GCNX_NORM_PYG = 1 } gcnx_norm_mode;
typedef enum { GCNX_RED_SUM = 0, GCNX_RED_MAX = 1 } gcnx_redop;
/* Which of Keras' two categorical_crossentropy code paths a loss entry point follows (keras.backend.
 * categorical_crossentropy, from_logits=False, as constructed at gcn.py:326):
 *   PROBS  -- the output is an EagerTensor (evaluate(), gcn.py:351-354, on TF < 2.6): p <- p / sum p, clip to
 *             [1e-7, 1 - 1e-7], -sum_c y_c log p_c; the clip passes no gradient outside its range.
 *   LOGITS -- the output is a graph tensor produced by a Softmax op (train_step runs under tf.function, gcn.py:328-335;
 *             on TF >= 2.6 also eagerly, through the output's _keras_logits): Keras takes the op's INPUT and calls
 *             softmax_cross_entropy_with_logits: loss = logsumexp(z) * sum_c y_c - sum_c y_c z_c, no renormalisation,
 *             no clip, dlogits = (p * sum_c y_c - y) / denom everywhere.
 * The two differ only where a probability saturates beyond 1e-7 (|z_i - z_j| > ~16). */
typedef enum { GCNX_CCE_PROBS = 0, GCNX_CCE_LOGITS = 1 } gcnx_cce_mode;

#define GCNX_UNIQUE_ID_BYTES 128

/* ---- lifecycle ------------------------------------------------------------------------ */
GCNX_API int gcnx_version(void);
GCNX_API int gcnx_device_count(int* n);
GCNX_API int gcnx_ctx_create(int device, gcnx_ctx** out);
GCNX_API int gcnx_ctx_destroy(gcnx_ctx* ctx);
/* Message of the last failing call on ctx (ctx may be NULL: last error of this thread). */
GCNX_API const char* gcnx_last_error(gcnx_ctx* ctx);
/* Fills name[len] with the device's gcnArchName and *cus with its compute-unit count. */
GCNX_API int gcnx_device_info(gcnx_ctx* ctx, char* name, int len, int* cus, size_t* hbm_bytes);

/* Diagnostics, not part of the drop-in contract: kernel-selection knobs of this ctx (the GCNX_* environment variables
 * set their initial values at gcnx_ctx_create; results never depend on them, only which kernel computes them).
 * Keys: "spmm_kernel" (0 auto, 1 rows, 2 tile = the round-1 tier kernels, 3 pipe = force the pipelined tile kernel),
 * "spmm_slab", "spmm_sg", "gemm_stream" (0 = bf16 GEMMs stay on the tiled kernel). */
GCNX_API int gcnx_set_tuning(gcnx_ctx* ctx, const char* key, int value);

/* ---- memory / sync (h2d, d2h synchronise the ctx stream) ------------------------------- */
GCNX_API int gcnx_malloc(gcnx_ctx* ctx, size_t bytes, void** dptr);
GCNX_API int gcnx_free(gcnx_ctx* ctx, void* dptr);
GCNX_API int gcnx_memset(gcnx_ctx* ctx, void* dptr, int value, size_t bytes);
GCNX_API int gcnx_h2d(gcnx_ctx* ctx, void* dst, const void* src, size_t bytes);
/* gcnx_h2d without the wait: the bytes are staged in pinned memory before the call returns and copied in stream order
 * (up to 16 KiB and outside capture; otherwise it is gcnx_h2d).  For per-batch descriptors: a synchronous copy makes every
 * streamed step wait for the previous one. */
GCNX_API int gcnx_h2d_async(gcnx_ctx* ctx, void* dst, const void* src, size_t bytes);
GCNX_API int gcnx_d2h(gcnx_ctx* ctx, void* dst, const void* src, size_t bytes);
GCNX_API int gcnx_d2d(gcnx_ctx* ctx, void* dst, const void* src, size_t bytes);
GCNX_API int gcnx_sync(gcnx_ctx* ctx);
/* Side sections: work launched between gcnx_side_begin and gcnx_side_end runs on a second HIP stream, after
 * everything submitted so far and concurrently with what follows on the main stream; gcnx_side_join makes the
 * main stream wait for the side sections ended so far (gcnx_sync and gcnx_capture_end join too).  Meant for the
 * gradient leaves of tape.gradient (gcn.py:337) -- dW, db -- which nothing later in the backward pass reads;
 * TensorFlow's executor runs such independent ops concurrently as well.  The caller keeps the buffers a side
 * section reads unmodified until the join.  Inside a capture the graph gets two branches. */
GCNX_API int gcnx_side_begin(gcnx_ctx* ctx);
GCNX_API int gcnx_side_end(gcnx_ctx* ctx);
GCNX_API int gcnx_side_join(gcnx_ctx* ctx);

/* ---- timing: HIP events on the ctx stream ---------------------------------------------- */
GCNX_API int gcnx_event_create(gcnx_ctx* ctx, gcnx_event** out);
GCNX_API int gcnx_event_record(gcnx_ctx* ctx, gcnx_event* ev);
/* Synchronises on `stop`, then returns stop - start in milliseconds. */
GCNX_API int gcnx_event_elapsed_ms(gcnx_ctx* ctx, gcnx_event* start, gcnx_event* stop, float* ms);
GCNX_API int gcnx_event_destroy(gcnx_ctx* ctx, gcnx_event* ev);

/* ---- HIP-graph capture of a call sequence (replaces tf.function, gcn.py:328) ----------- */
GCNX_API int gcnx_capture_begin(gcnx_ctx* ctx);
GCNX_API int gcnx_capture_end(gcnx_ctx* ctx, gcnx_graph** out);
GCNX_API int gcnx_graph_launch(gcnx_ctx* ctx, gcnx_graph* g);
GCNX_API int gcnx_graph_destroy(gcnx_ctx* ctx, gcnx_graph* g);

/* ---- graph preparation ------------------------------------------------------------------ */
/* DisjointLoader's SparseTensor (gcn.py:316-317 -> sp_matrix_to_sp_tensor + tf.sparse.reorder):
 * row-major sorted COO int64 (rows[nnz], cols[nnz]) on the DEVICE -> CSR int32.  Synchronises.
 * Fails with GCNX_ERR_DATA if rows are not non-decreasing or an index is outside [0,N). */
GCNX_API int gcnx_coo_to_csr(gcnx_ctx* ctx, const int64_t* rows, const int64_t* cols, int64_t nnz,
                    int64_t n, int32_t* rowptr, int32_t* colidx);
/* Device-side collate (what DisjointLoader's collate -- vstack / block_diag / find, gcn.py:316-317,367 -- does on
 * the host).  The dataset is resident as one disjoint union: node_ptr int32[G+1], CSR (rowptr int32[Ntot+1],
 * colidx int32, optional vals), features x[Ntot, f], labels y[G, c].  desc int32[3(b+1)] = the b selected graph
 * ids (one pad), then the batch's node offsets [b+1], then its entry offsets [b+1] (host prefix sums over the
 * selected sizes).  Writes the batch: o_x[N, f], o_rowptr[N+1], o_colidx / o_vals[nnz], o_y[b, c],
 * o_graph_ptr[b+1], and (o_node_graph may be NULL) DisjointLoader's id vector i[N] = the batch position of every row's
 * graph.  Integer outputs are exact; floats are copies. */
GCNX_API int gcnx_collate(gcnx_ctx* ctx, const int32_t* desc, int32_t b, const int32_t* node_ptr, const int32_t* rowptr,
                 const int32_t* colidx, const float* vals, const float* x, int64_t ldx, int32_t f, const float* y,
                 int32_t c, int32_t* o_rowptr, int32_t* o_colidx, float* o_vals, float* o_x, int64_t ldo, float* o_y,
                 int32_t* o_graph_ptr, int32_t* o_node_graph);
/* Spektral GCNConv.preprocess = gcn_filter on a CSR whose every row stores its
 * diagonal entry (true for the reference's data: gcn_utills.py:224-227 keeps the 0-Angstrom
 * diagonal).  vals_in NULL = ones.  SPEKTRAL: diag += 1; PYG: existing loops kept.
 * vals_out[e] = a~[e] * d^-1/2[row] * d^-1/2[col].  Synchronises; GCNX_ERR_DATA if a row has
 * no stored diagonal. */
GCNX_API int gcnx_gcn_norm(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx,
                  const float* vals_in, int32_t n, int mode, float* vals_out);
/* What the scheduling shortcuts may assume about a batch adjacency, checked once per batch on the device (the
 * reference's loader accepts any scipy matrix, gcn.py:104-116; nothing in Spektral requires symmetry).  *props
 * (host int) receives a bit set:
 *   GCNX_CSR_SYMMETRIC       A == A^T (pattern; values to 4 ulp): the backward aggregation may reuse this CSR for A^T
 *                            (otherwise build gcnx_csr_transpose);
 *   GCNX_CSR_GRAPH_PTR_OK    graph_ptr[0] == 0, graph_ptr[b] == n, non-decreasing;
 *   GCNX_CSR_BLOCK_DIAGONAL  every entry of a row of graph g has its column in [graph_ptr[g], graph_ptr[g+1]): the
 *                            premise of gcnx_spmm_plan_create and gcnx_spmm_csr_pool_bwd (sp.block_diag).
 * graph_ptr may be NULL (only symmetry is checked).  Synchronises. */
#define GCNX_CSR_SYMMETRIC 1
#define GCNX_CSR_BLOCK_DIAGONAL 2
#define GCNX_CSR_GRAPH_PTR_OK 4
GCNX_API int gcnx_csr_inspect(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, int32_t n,
                     const int32_t* graph_ptr, int32_t b, int* props);
/* CSR of the transpose (for the backward SpMM when A^ is not symmetric).  Synchronises. */
GCNX_API int gcnx_csr_transpose(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx,
                       const float* vals, int32_t n, int32_t nnz, int32_t* rowptr_t,
                       int32_t* colidx_t, float* vals_t);

/* ---- forward ----------------------------------------------------------------------------- */
/* K1 MatMul+BiasAdd (+activation): out[N,Fo] = act(X[N,Fi] * W[Fi,Fo] + bias).
 * Replaces Dense / GCNConv.kernel / GeneralConv.kernel matmuls under gcn.py:334.
 * bias may be NULL.  alpha[Fo] is the PReLU slope (NULL unless act == PRELU). */
GCNX_API int gcnx_gemm(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* w, const float* bias,
              float* out, int64_t ldo, int64_t n, int32_t fi, int32_t fo, int prec, int act,
              const float* alpha);
/* Split-bf16 weight GEMMs for mid-size batches (r3; csrc/gemm_panel.hip): the Dense products of the reference's live
 * model GeneralGNN (gcn.py:320; MatMul + BiasAdd and the MatMul grad wrt the input, gcn.py:334/337) with the weight
 * operand read from a bf16 IMAGE in MFMA-fragment order.  gcnx_wimage_elems: bf16 elements of the image of W [fi, fo]
 * (transpose = 0: the operand of X W) or of W^T (transpose = 1: the operand of dH W^T); 0 for GCNX_PREC_F32.
 * gcnx_wimage_prepare: ONE launch writes the images of up to 16 matrices (per step: the weights change with every
 * update).  gcnx_gemm_wimage: out[n, fo] = x[n, fi] W + bias (transpose = 0) or out[n, fi] (+)= x[n, fo] W^T
 * (transpose = 1; accumulate: skip-connection gradients); GCNX_ERR_UNSUPPORTED without a message when the shape is
 * not the kernel's (widths in multiples of 16 / 4 floats, 16-byte aligned operands): call gcnx_gemm / gcnx_gemm_dx.
 * bn_parts (transpose = 0, fo <= 256): the batch-norm statistics of what is written, as (rows, mean, M2) per column
 * and workgroup -- [*n_parts][3][fo] floats, room for gcnx_gemm_wimage_parts(n) parts -- which gcnx_bn_finalize_parts
 * combines (Chan's formula, part order) into mean / inv and the Keras moving-statistics update: tf.nn.moments of the
 * Dense output without a pass over it. */
typedef struct gcnx_wimage_job {
  const float* w;
  void* img;
  int32_t fi, fo;
  int32_t transpose, prec;
} gcnx_wimage_job;
GCNX_API int64_t gcnx_wimage_elems(int32_t fi, int32_t fo, int transpose, int prec);
GCNX_API int gcnx_wimage_prepare(gcnx_ctx* ctx, int32_t njobs, const gcnx_wimage_job* jobs);
GCNX_API int gcnx_gemm_wimage(gcnx_ctx* ctx, const float* x, int64_t ldx, const void* img, int32_t fi, int32_t fo, int transpose,
                     const float* bias, float* out, int64_t ldo, int64_t n, int prec, int accumulate, float* bn_parts,
                     int32_t* n_parts);
GCNX_API int64_t gcnx_gemm_wimage_parts(gcnx_ctx* ctx, int64_t n);
GCNX_API int gcnx_bn_finalize_parts(gcnx_ctx* ctx, const float* parts, int32_t nparts, int32_t f, float momentum, float eps,
                           float* mean, float* inv, float* moving_mean, float* moving_var);
/* The block-diagonal structure of a DisjointLoader batch (sp.block_diag) as a
 * scheduling plan: block_ptr int32[nblocks+1] on the device (= graph_ptr) says that rows
 * [block_ptr[g], block_ptr[g+1]) reference only columns of that same range.  Built once per
 * batch (synchronises), owned by the caller, read-only afterwards; it must describe the CSR it
 * is later used with.  It only changes how gcnx_spmm_csr schedules work, never its result. */
GCNX_API int gcnx_spmm_plan_create(gcnx_ctx* ctx, const int32_t* block_ptr, int32_t nblocks,
                                   gcnx_spmm_plan** out);
GCNX_API int gcnx_spmm_plan_destroy(gcnx_ctx* ctx, gcnx_spmm_plan* plan);
/* The tile kernels deal a graph's output rows to their lanes in degree order (longest rows first), which they read from
 * a per-rowptr record list the plan builds from a host copy of rowptr (synchronises; kept per rowptr POINTER, up to four
 * per plan: the normalised, unweighted and transposed views of one batch share the plan).  gcnx_spmm_csr and
 * gcnx_spmm_csr_pool_bwd build it on first use with a rowptr; call this (a) before capturing such a call into a HIP graph
 * unless it has run once already, (b) again whenever the row pointers behind an already bound pointer change.  Like the
 * plan itself it only changes scheduling -- and the order in which a row's output is WRITTEN, never its value. */
GCNX_API int gcnx_spmm_plan_bind(gcnx_ctx* ctx, gcnx_spmm_plan* plan, const int32_t* rowptr, int32_t n);
/* K2/K3 SparseTensorDenseMatMul / gather+unsorted_segment_sum:
 * out[t,:] = act( sum_e vals[e] * h[colidx[e],:] + bias ), e over row t.
 * GCNConv.call (bias AFTER aggregation) and GeneralConv.propagate (vals NULL).
 * plan (may be NULL) enables the LDS-resident tile kernel for graphs that fit one. */
GCNX_API int gcnx_spmm_csr(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx,
                  const float* vals, const float* h, int64_t ldh, const float* bias, float* out,
                  int64_t ldo, int32_t n, int32_t f, int act, const gcnx_spmm_plan* plan);
/* K4 SegmentSum / mean / max over sorted graph ids (GlobalSumPool, gcn.py:320 pool="sum";
 * max variant: gcn_utills.py:842).  graph_ptr int32[B+1].  argmax int32[B*F] (row index of
 * the maximum; required for MAX, else NULL). */
GCNX_API int gcnx_segment_pool(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* x, int64_t ldx,
                      float* pooled, int32_t b, int32_t f, int mode, int32_t* argmax);
/* K8 Softmax + CategoricalCrossentropy + categorical_accuracy (gcn.py:326,335,339):
 * probs = softmax(logits); loss_acc[0] += sum_g loss_g / denom; loss_acc[1] += #(argmax p == argmax y).
 * cce_mode GCNX_CCE_LOGITS (the training step, gcn.py:328-335): loss_g = sum_c y_c (logsumexp(z) - z_c),
 *   dlogits = (p * sum_c y_c - y) / denom.
 * cce_mode GCNX_CCE_PROBS (eager evaluate, gcn.py:351-354): loss_g = -sum_c y_c log clip(p_c, 1e-7, 1 - 1e-7),
 *   dlogits = (p * sum(y*m) - y*m) / denom with m = [1e-7 < p < 1 - 1e-7], the gradient of the clipped loss.
 * dlogits NULL to skip.  denom = global batch size (so that shard gradients add up to the full-batch gradient).
 * loss_acc is a device float[2] that the caller zeroes. */
GCNX_API int gcnx_softmax_cce(gcnx_ctx* ctx, const float* logits, const float* y, int32_t b, int32_t c,
                     float denom, float* probs, float* loss_acc, float* dlogits, int cce_mode);

/* The classifier head in ONE launch: probs = softmax(pooled * W + b)  (the model's last Dense,
 * gcn.py:320 activation="softmax"), loss_acc[0] = CCE(y, probs) summed over the b graphs / denom and
 * loss_acc[1] = #(argmax probs == argmax y)  (gcn.py:326,335,339; loss_acc is OVERWRITTEN, not accumulated),
 * and, when dw != NULL, the head gradients tape.gradient (gcn.py:337) produces:
 * dlogits as gcnx_softmax_cce (same cce_mode), dw[h,c] = pooled^T dlogits, db[c] = sum_g dlogits, dpooled[b,h] = dlogits W^T.
 * y == NULL: probabilities only.  c <= 32.  Same results as gcnx_gemm + gcnx_softmax_cce + gcnx_gemm_dw +
 * gcnx_act_bias_grad + gcnx_gemm_dx up to fp32 summation order; deterministic (fixed reduction order). */
GCNX_API int gcnx_dense_softmax_cce(gcnx_ctx* ctx, const float* pooled, int64_t ldp, const float* w, const float* bias,
                           const float* y, int32_t b, int32_t h, int32_t c, float denom, float* probs,
                           float* loss_acc, float* dw, float* db, float* dpooled, int64_t lddp, int cce_mode);

/* GlobalSumPool / GlobalAvgPool / GlobalMaxPool (gcn.py:319) + the head above as one call: pooled[b,h] =
 * gcnx_segment_pool(x[n,h]) is still written (the caller's saved activation), but where the pool is split into
 * row slices (few graphs: SUM / AVG) the head combines the partial sums itself while staging its operand, so the
 * pair runs as 2 launches instead of 3 (with half as many row slices: the head's workgroup reads them all).
 * Otherwise exactly gcnx_segment_pool + gcnx_dense_softmax_cce.  Results equal those of the two calls up to the
 * fp32 summation order of the pool; deterministic.  argmax: as gcnx_segment_pool (MAX only).
 * db_relu (may be NULL; needs dw and SUM / AVG): float[h] = what gcnx_pool_bwd_colsum(dpooled, y = x) returns, the
 * bias gradient of the layer that produced x through a ReLU (GCNConv, gcn.py:317) -- on the combined path the pool
 * counts the positive entries per (graph, column) in the pass it makes anyway and the head finishes
 * sum_g s_g * dPooled[g] * count[g], so that gradient costs no launch of its own. */
GCNX_API int gcnx_pool_dense_softmax_cce(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* x, int64_t ldx, int pool_mode,
                                int32_t* argmax, float* pooled, int64_t ldp, const float* w, const float* bias,
                                const float* y, int32_t b, int32_t h, int32_t c, float denom, float* probs,
                                float* loss_acc, float* dw, float* db, float* dpooled, int64_t lddp,
                                float* db_relu, int cce_mode);

/* ---- backward (what tape.gradient, gcn.py:337, generates) -------------------------------- */
/* dZ = dY * act'(Y) (mask taken from the saved output Y; PReLU uses the saved pre-activation
 * passed as y and alpha); db[f] = sum_rows dZ (BiasAddGrad).  dz may alias dy.  db/dalpha may
 * be NULL.  For PRELU dalpha[f] = sum_rows dY*min(y,0). */
GCNX_API int gcnx_act_bias_grad(gcnx_ctx* ctx, const float* dy, int64_t lddy, const float* y, int64_t ldy,
                       float* dz, int64_t lddz, int64_t n, int32_t f, int act,
                       const float* alpha, float* db, float* dalpha);
/* dW[Fi,Fo] = X^T[Fi,N] * dH[N,Fo]: deterministic two-stage split-K (no atomics). */
GCNX_API int gcnx_gemm_dw(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh,
                 float* dw, int64_t n, int32_t fi, int32_t fo, int prec);
/* ---- bf16 STORAGE of activations that only bf16-operand weight GEMMs read (r3; large batches, GCNX_PREC_BF16) ----------
 * GCNX_PREC_BF16 rounds both operands of a product to bfloat16 (nearest even) as it loads them (tf.cast(x, tf.bfloat16)
 * in front of every MatMul / MatMul grad of the Dense and GCNConv kernels, gcn.py:334, :337).  A tensor that only such
 * products read -- S1 = A X, Y1, dH2 = A^T dZ2, dZ1 of the two-layer model -- can be STORED rounded: every result stays
 * bit-identical and the producer and the consumers move half its bytes.  `x16` / `dh16` / `out16` are uint16 rows
 * (bfloat16 bit patterns), leading dimensions in ELEMENTS.
 *   gcnx_spmm_csr_bf16out           = gcnx_spmm_csr with the result stored as bf16 (tile kernels + 8-row chunks only).
 *   gcnx_spmm_csr_pool_bwd_bf16out  = gcnx_spmm_csr_pool_bwd(plan, y_bits) with the result stored as bf16.
 *   gcnx_gemm_fwd_bf16              = gcnx_gemm / gcnx_gemm_relu_bits on a bf16 input; out as bf16 (out_bf16) or fp32.
 *   gcnx_gemm_dx_bf16               = gcnx_gemm_dx / gcnx_gemm_dx_bits on a bf16 dH (mask_bits and db may be NULL).
 *   gcnx_gemm_dw_bf16               = gcnx_gemm_dw with both operands stored as bf16.
 * All five return GCNX_ERR_UNSUPPORTED -- nothing launched, gcnx_last_error untouched -- for shapes outside the streaming /
 * tile kernels (GEMMs: fi = fo = 256, n >= 32768; SpMM: weighted operator, a plan whose tile graphs all take the
 * 1024-thread shape, f a multiple of 32 above 128, no hub rows): keep fp32 storage then. */
GCNX_API int gcnx_spmm_csr_bf16out(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                          int64_t ldh, const float* bias, void* out16, int64_t ldo, int32_t n, int32_t f, int act,
                          const gcnx_spmm_plan* plan);
GCNX_API int gcnx_spmm_csr_pool_bwd_bf16out(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                          const float* y, int64_t ldy, const int32_t* graph_ptr, int32_t b, const float* dpooled, int64_t lddp,
                          void* out16, int64_t ldo, int32_t n, int32_t f, int mode, const gcnx_spmm_plan* plan, const void* y_bits);
GCNX_API int gcnx_gemm_fwd_bf16(gcnx_ctx* ctx, const void* x16, int64_t ldx, const float* w, const float* bias, void* out, int64_t ldo,
                          int out_bf16, int64_t n, int32_t fi, int32_t fo, int act, void* relu_bits, const void* wimg);
GCNX_API int gcnx_gemm_dx_bf16(gcnx_ctx* ctx, const void* dh16, int64_t lddh, const float* w, void* dx, int64_t lddx, int dx_bf16,
                          int64_t n, int32_t fi, int32_t fo, const void* mask_bits, float* db, const void* wimg);
/* `wimg` (may be NULL: the call builds it, one small launch): the bf16 image of the weight operand in MFMA-fragment order,
 * GCNX_STREAM_IMAGE_BYTES bytes, written by gcnx_gemm_stream_images -- the images of up to four operands in ONE launch, once
 * per step after the optimizer update (transpose = 1: the operand of gcnx_gemm_fwd_bf16, X W; 0: of gcnx_gemm_dx_bf16,
 * dH W^T).  A 1/8 shard of config 3 is a 0.5 ms step: three 5-us preparation launches are 3 % of it. */
#define GCNX_STREAM_IMAGE_BYTES 131072
typedef struct gcnx_stream_image_job {
  const float* w; int32_t fi, fo;   /* W [fi, fo], 256 x 256 */
  int32_t transpose;
  void* img;                        /* GCNX_STREAM_IMAGE_BYTES bytes, 16-byte aligned */
} gcnx_stream_image_job;
GCNX_API int gcnx_gemm_stream_images(gcnx_ctx* ctx, int32_t njobs, const gcnx_stream_image_job* jobs);
GCNX_API int gcnx_gemm_dw_bf16(gcnx_ctx* ctx, const void* x16, int64_t ldx, const void* dh16, int64_t lddh, float* dw, int64_t n,
                          int32_t fi, int32_t fo);
/* The ReLU mask of a Dense / GCNConv kernel product as a bit image between the forward and the backward product of
 * large batches: gcnx_gemm_relu_bits = gcnx_gemm(act = GCNX_ACT_RELU) that also writes [out > 0] to `bits` (n * 64
 * bytes reserved, 8-byte aligned; layout private to the pair, which must use the same precision), gcnx_gemm_dx_bits = gcnx_gemm_dx masked by that image instead of
 * the saved activation (ReluGrad after MatMul grad, gcn.py:337) -- 32 bytes per row read where the activation is 1 KiB.
 * Served by the streaming bf16 kernels only (GCNX_PREC_BF16 / BF16X3, fi = fo = 256, n >= 32768): GCNX_ERR_UNSUPPORTED
 * otherwise, nothing is launched and -- UNSUPPORTED being an answer, not a failure -- gcnx_last_error is left as it was. */
GCNX_API int gcnx_gemm_relu_bits(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* w, const float* bias, float* out,
                        int64_t ldo, int64_t n, int32_t fi, int32_t fo, int prec, void* bits);
GCNX_API int gcnx_gemm_dx_bits(gcnx_ctx* ctx, const float* dh, int64_t lddh, const float* w, float* dx, int64_t lddx,
                      int64_t n, int32_t fi, int32_t fo, int prec, const void* mask_bits, float* db);
/* dX[N,Fi] (+)= dH[N,Fo] * W^T.  accumulate != 0 adds into dx (skip-connection gradients).
 * y_mask (may be NULL): saved ReLU output of the layer that produced this GEMM's input; the
 * epilogue then writes dZ = dX * (y_mask > 0), i.e. the activation gradient of that layer is
 * fused.  db (may be NULL): column sums of what was written (BiasAddGrad of that layer). */
GCNX_API int gcnx_gemm_dx(gcnx_ctx* ctx, const float* dh, int64_t lddh, const float* w, float* dx,
                 int64_t lddx, int64_t n, int32_t fi, int32_t fo, int prec, int accumulate,
                 const float* y_mask, int64_t ldy, float* db);
/* The backward of one Dense / GCNConv kernel product H = X W given dH, as one call:
 *   dW[fi,fo] = X^T dH   (gcnx_gemm_dw)   and   dX[n,fi] = dH W^T (* [y_mask > 0]), db_prev = colsum(dX)   (gcnx_gemm_dx)
 * -- the two matmul gradients tape.gradient (gcn.py:337) emits for every `x @ kernel` (Spektral GCNConv.call).
 * Both read dH; fp32 with 64-column-aligned fi runs them as ONE launch (the dX tiles, and the dW split-K tiles in
 * the otherwise idle workgroup slots) plus ONE reduction launch (db_prev partials and dW slabs together): dW is a
 * gradient leaf, so on small batches it disappears behind dX instead of costing a launch pair or a second stream.
 * Other precisions / shapes: exactly the two calls.  Deterministic; y_mask / db_prev may be NULL. */
GCNX_API int gcnx_dense_bwd(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, const float* w,
                   int64_t n, int32_t fi, int32_t fo, int prec, float* dx, int64_t lddx, const float* y_mask,
                   int64_t ldy, float* db_prev, float* dw);
/* Gradient of the global pool: SUM dX[r] = dP[g(r)]; AVG /n_g; MAX routed to argmax rows.
 * If y (saved ReLU output of the last conv layer) is given its mask is fused:
 * dX[r] *= (y[r] > 0).  db (may be NULL): column sums of dX (BiasAddGrad of that layer). */
GCNX_API int gcnx_segment_pool_bwd(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* dpooled,
                          float* dx, int64_t lddx, int32_t n, int32_t b, int32_t f, int mode,
                          const int32_t* argmax, const float* y, int64_t ldy, float* db);

/* The two calls above the last conv layer's backward aggregation as ONE gather (small batches: each launch there is
 * latency): for SUM / AVG pooling and a block-diagonal operator,
 *   (A^T dZ)[i] = scale_g * dPooled[g(i)] * sum_j A^T[i,j] * [y[j] > 0],   dZ[j] = scale_g * dPooled[g(j)] * [y[j] > 0]
 * -- replaces gcnx_segment_pool_bwd(y=...) + gcnx_spmm_csr for the gradient of GlobalSumPool (gcn.py:319) through
 * the ReLU of the last GCNConv (gcn.py:317) and its aggregation; tf.GradientTape materialises dZ there (gcn.py:337).
 * (rowptr, colidx, vals) is the TRANSPOSED operator, as for the unfused backward call.  Needs f, ldy, ldo, lddp in
 * multiples of 4 floats and 16-byte aligned y / out / dpooled; MAX pooling has no such form (GCNX_ERR_INVALID).
 * plan (may be NULL): the operator's gcnx_spmm_plan built over the same graph_ptr -- large batches then run on the tile
 * kernels (the landed tile of y is masked in place, the graph's dPooled row scales the sums). */
GCNX_API int gcnx_spmm_csr_pool_bwd(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                           const float* y, int64_t ldy, const int32_t* graph_ptr, int32_t b,
                           const float* dpooled, int64_t lddp, float* out, int64_t ldo, int32_t n, int32_t f,
                           int mode, const gcnx_spmm_plan* plan, const void* y_bits);
/* gcnx_spmm_csr with act = ReLU on the tile kernels, which also write [out > 0] as a bit image: one 32-bit word per (row,
 * 32-column slab), slab-major -- relu_bits[slab * n + row], (f / 32) * n words -- for the rows of every graph the plan
 * serves with a tile (graphs taller than a tile get no bits; gcnx_spmm_csr_pool_bwd folds their rows from `out`).
 * gcnx_spmm_csr_pool_bwd(..., y_bits = relu_bits) then expands the words into its 0 / 1 tile instead of reading y:
 * 4 bytes per row and slab instead of 128.  GCNX_ERR_UNSUPPORTED when gcnx_spmm_csr would not take the tile kernels
 * (no plan, too few tile units, f % 32 != 0): call gcnx_spmm_csr then and pass y_bits = NULL. */
GCNX_API int gcnx_spmm_csr_relu_bits(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                           const float* h, int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f,
                           const gcnx_spmm_plan* plan, void* relu_bits);
/* The pooled GCNConv's forward of a TRAINING step on a tile plan without its output (r3): what the step needs of
 * Y = relu(A h + b) is [Y > 0] as the bit image (the folded backward aggregation, gcnx_spmm_csr_pool_bwd(y_bits)), the
 * graphs' pooled rows -- GlobalSumPool / GlobalAvgPool([Y, i]), gcn.py:334 -- and the positive counts per (graph, column)
 * (the layer's bias gradient).  On graphs that fit an LDS tile all three leave the aggregation's epilogue: their rows of
 * `out` are NOT written and no pool launch reads them back (2 GB of the config-3 step).  Graphs taller than a tile keep
 * the two-launch form: their rows of `out` ARE written (the folded backward gathers them) and pooled from there.
 * pooled [b, ldp], cnt [b, f].  GCNX_ERR_UNSUPPORTED (nothing launched) when the tile kernels do not serve the batch:
 * gcnx_spmm_csr_relu_bits / gcnx_spmm_csr + gcnx_pool_dense_softmax_cce then. */
GCNX_API int gcnx_spmm_csr_relu_bits_pool(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                        int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, const gcnx_spmm_plan* plan,
                        void* relu_bits, const int32_t* graph_ptr, int32_t b, int pool_mode, float* pooled, int64_t ldp, float* cnt);
/* ... and the classifier head on those pooled rows: gcnx_dense_softmax_cce + db_relu[c] = sum_g pool'(dPooled)[g][c] *
 * cnt[g][c] (the pooled layer's bias gradient; NULL to skip). */
GCNX_API int gcnx_pooled_dense_softmax_cce(gcnx_ctx* ctx, const int32_t* graph_ptr, int pool_mode, const float* pooled, int64_t ldp,
                        const float* cnt, const float* w, const float* bias, const float* y, int32_t b, int32_t h, int32_t c,
                        float denom, float* probs, float* loss_acc, float* dw, float* db, float* dpooled, int64_t lddp,
                        float* db_relu, int cce_mode);
/* db[f] = column sums of the same never-materialised dZ (BiasAddGrad of that layer):
 * sum_g scale_g * dPooled[g] * #{j in g : y[j] > 0}; deterministic (fixed summation order). */
GCNX_API int gcnx_pool_bwd_colsum(gcnx_ctx* ctx, const int32_t* graph_ptr, int32_t b, const float* dpooled,
                         int64_t lddp, const float* y, int64_t ldy, int32_t f, int mode, float* db);

/* ---- Keras BatchNormalization + PReLU around Dense (MLP / GeneralConv of GeneralGNN, gcn.py:320;
 *      axis -1, momentum 0.99, eps 1e-3, biased batch variance) ------------------- */
/* sums[0:f] = column sums of (z - shift), sums[f:2f] = column sums of (z - shift)^2 (device float[2f]);
 * shift (device float[f], may be NULL = 0).  tf.nn.moments takes the variance of the CENTRED data; to match it
 * in fp32 call twice: shift = NULL gives the mean, shift = mean gives a cancellation-free variance.  A sharded
 * caller all-reduces `sums` (and the row count) between gcnx_bn_stats and gcnx_bn_finalize for sync-BN. */
GCNX_API int gcnx_bn_stats(gcnx_ctx* ctx, const float* z, int64_t ldz, int64_t n, int32_t f, const float* shift,
                           float* sums);
/* Training (sums != NULL): d = sums/count, mean = shift + d, var = sums2/count - d^2 (biased),
 * inv = 1/sqrt(var+eps); the moving statistics (may be NULL) are updated: m <- momentum*m + (1-momentum)*batch.
 * mean may alias shift.  Inference (sums == NULL): mean/inv come from the moving statistics. */
GCNX_API int gcnx_bn_finalize(gcnx_ctx* ctx, const float* sums, float count, int32_t f, float momentum,
                              float eps, const float* shift, float* mean, float* inv, float* moving_mean,
                              float* moving_var);
/* Single-device shortcut for the training-mode moments: gcnx_bn_stats + gcnx_bn_finalize for the mean, then again
 * centred on it for the variance (the two-pass tf.nn.moments form), with each reduce + finalise pair fused: four
 * launches instead of six, bit-identical results.  moving_* (both or none) get the momentum update. */
GCNX_API int gcnx_bn_moments(gcnx_ctx* ctx, const float* z, int64_t ldz, int64_t n, int32_t f, float momentum, float eps,
                    float* mean, float* inv, float* moving_mean, float* moving_var);
/* y = act(gamma * (z - mean) * inv + beta); act NONE / RELU / PRELU(alpha[f]). */
GCNX_API int gcnx_bn_act(gcnx_ctx* ctx, const float* z, int64_t ldz, int64_t n, int32_t f, const float* mean,
                         const float* inv, const float* gamma, const float* beta, int act, const float* alpha,
                         float* y, int64_t ldy);
/* Backward of gcnx_bn_act given dY and the saved z, mean, inv: dzb = dY*act'(zb);
 * training: dZ = gamma*inv*(dzb - mean_rows(dzb) - xhat*mean_rows(dzb*xhat)); inference: gamma*inv*dzb.
 * dgamma = sum dzb*xhat, dbeta = sum dzb, dalpha = sum dY*min(zb,0) (each may be NULL).
 * sums_scratch: device float[3f].  dz may alias dy. */
GCNX_API int gcnx_bn_act_bwd(gcnx_ctx* ctx, const float* dy, int64_t lddy, const float* z, int64_t ldz, int64_t n,
                             int32_t f, const float* mean, const float* inv, const float* gamma, const float* beta,
                             int act, const float* alpha, int training, float* dz, int64_t lddz, float* dgamma,
                             float* dbeta, float* dalpha, float* sums_scratch);
/* The two halves of gcnx_bn_act_bwd for sync-BN: _stats writes the three LOCAL column sums to sums_scratch[3f]
 * (and, if given, to dbeta / dgamma / dalpha -- local parts, to be summed with the other parameter gradients);
 * the caller all-reduces sums_scratch; _apply forms dz with the reduced sums and the GLOBAL row count. */
GCNX_API int gcnx_bn_act_bwd_stats(gcnx_ctx* ctx, const float* dy, int64_t lddy, const float* z, int64_t ldz, int64_t n,
                          int32_t f, const float* mean, const float* inv, const float* gamma, const float* beta, int act,
                          const float* alpha, float* sums_scratch, float* dgamma, float* dbeta, float* dalpha);
GCNX_API int gcnx_bn_act_bwd_apply(gcnx_ctx* ctx, const float* dy, int64_t lddy, const float* z, int64_t ldz, int64_t n,
                          int32_t f, const float* mean, const float* inv, const float* gamma, const float* beta, int act,
                          const float* alpha, const float* sums, float count, int training, float* dz, int64_t lddz);

/* ---- GeneralGNN options beside gcn.py:320's defaults (csrc/elementwise.hip) ------------------ */
/* Keras Dropout(rate) in training mode (the Dropout layer of Spektral's MLP and GeneralConv):
 * out = x * keep / (1 - rate), keep ~ Bernoulli(1 - rate) per element.  The keep decision of element (row, col) is a
 * stateless hash of (seed, stream_id, *step, row * f + col) -- step may be NULL (0) -- so calling it again with the same
 * arguments on the incoming gradient IS the layer's backward pass (no stored mask), and a captured step that reads `step`
 * from device memory draws a fresh mask on every replay (gcnx_counter_add advances it).  TensorFlow's random generator
 * is not reproduced.  out may alias x. */
GCNX_API int gcnx_dropout(gcnx_ctx* ctx, const float* x, int64_t ldx, int64_t n, int32_t f, float rate, uint32_t seed,
                          uint32_t stream_id, const uint32_t* step, float* out, int64_t ldo);
/* *counter += inc on the stream (the optimizer's step count: optimizer.iterations, gcn.py:338). */
GCNX_API int gcnx_counter_add(gcnx_ctx* ctx, uint32_t* counter, uint32_t inc);
/* out = a + b, row by row (GeneralGNN(connectivity="sum"): out = z + out).  out may alias a or b. */
GCNX_API int gcnx_add(gcnx_ctx* ctx, const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo,
                      int64_t n, int32_t f);

/* GeneralConv(aggregate = "max" | "min") (tf.math.unsorted_segment_max / _min over the messages gather(h, a.indices[:,1]) of
 * every target row; adjacency values ignored): out[t, c] = max (min) over the entries (t, s) of h[s, c]; a row
 * without entries gets the lowest (largest) float, as TensorFlow does.  cnt (may be NULL; needed for the gradient) receives
 * the number of entries attaining the extremum. */
GCNX_API int gcnx_spmm_csr_minmax(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* h, int64_t ldh,
                                  float* out, int64_t ldo, float* cnt, int64_t ldc, int32_t n, int32_t f, int is_min);
/* Its gradient wrt h (TensorFlow's _UnsortedSegmentMinOrMaxGrad: the messages equal to the extremum share dy equally), from
 * the source side: rowptr_t / colidx_t is the TRANSPOSED operator (rows = sources, entries = their targets);
 * dh[s, c] = sum over targets t of [h[s, c] == out[t, c]] * dy[t, c] / cnt[t, c].  Deterministic (CSR order). */
GCNX_API int gcnx_spmm_csr_minmax_bwd(gcnx_ctx* ctx, const int32_t* rowptr_t, const int32_t* colidx_t, const float* h, int64_t ldh,
                                      const float* out, int64_t ldo, const float* cnt, int64_t ldc, const float* dy, int64_t lddy,
                                      float* dh, int64_t lddh, int32_t n, int32_t f);

/* GeneralConv(aggregate = "prod") (tf.math.unsorted_segment_prod over the messages gather(h, a.indices[:,1]) of every target row,
 * SURVEY 8.A.4; adjacency values ignored): out[t, c] = product over the entries (t, s) of h[s, c], 1 for a row without entries
 * (TensorFlow's value for an empty segment).  aux (may be NULL; needed for the gradient): out where no message of the row is zero,
 * the product of the non-zero messages where exactly one is, 0 where two or more are. */
GCNX_API int gcnx_spmm_csr_prod(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* h, int64_t ldh, float* out,
                                int64_t ldo, float* aux, int64_t lda, int32_t n, int32_t f);
/* Its gradient wrt h (TensorFlow's _UnsortedSegmentProdGrad: out / h for a non-zero message, the product of the others for the only
 * zero message of a row, 0 where a row holds two or more zeros), from the source side: rowptr_t / colidx_t is the TRANSPOSED
 * operator; dh[s, c] = sum over targets t of dy[t, c] * (h[s, c] == 0 ? aux[t, c] : out[t, c] / h[s, c]).  Deterministic. */
GCNX_API int gcnx_spmm_csr_prod_bwd(gcnx_ctx* ctx, const int32_t* rowptr_t, const int32_t* colidx_t, const float* h, int64_t ldh,
                                    const float* out, int64_t ldo, const float* aux, int64_t lda, const float* dy, int64_t lddy,
                                    float* dh, int64_t lddh, int32_t n, int32_t f);

/* ---- optimiser --------------------------------------------------------------------------- */
/* K9 Keras SGD without momentum (gcn.py:325,338): params -= lr * grads over a flat buffer. */
GCNX_API int gcnx_sgd(gcnx_ctx* ctx, float* params, const float* grads, int64_t n, float lr);
/* The learning rate of every update launch (gcnx_sgd, gcnx_gemm_dw_sgd, gcnx_gemm_dw2) read from a device scalar instead
 * of the `lr` argument (NULL: the argument again).  A learning rate passed by value is part of a captured launch; with a
 * source the same captured step serves every value of a schedule -- keras.optimizers.schedules.*, gcn.py:321-325 -- and
 * the host only rewrites 4 bytes (gcnx_h2d_async) when the value changes.  Not callable inside a capture. */
GCNX_API int gcnx_set_lr_source(gcnx_ctx* ctx, const float* lr_dev);
/* A reduction that gcnx_dense_bwd_deferred left undone: column-sum partial rows -> cout[cf] and split-K slabs ->
 * out[total], both still in the caller's scratch buffer.  All zeros = nothing pending.  Plain data, no ownership. */
typedef struct gcnx_pending_reduce {
  const float* colpart; int64_t crows; int32_t cf; float* cout;
  const float* slabs; int64_t total; int32_t nsplit; float* out;
} gcnx_pending_reduce;

/* gcnx_dense_bwd with its second launch (the reductions that finish db_prev and dw) left to the caller: both are
 * gradient LEAVES that only the optimizer reads, so the step's last launch (gcnx_gemm_dw_sgd, `pending`) can fold
 * them in -- one launch fewer on the critical path of a small batch.  The partial results stay in `scratch`
 * (>= gcnx_dense_bwd_scratch_floats(ctx, n, fi, fo) floats, 16-byte aligned, untouched until then); db_prev / dw
 * are NOT valid until that call.  If the fused form does not apply (precision, shapes, scratch too small) this is
 * exactly gcnx_dense_bwd and *pending comes back all zeros. */
GCNX_API int64_t gcnx_dense_bwd_scratch_floats(gcnx_ctx* ctx, int64_t n, int32_t fi, int32_t fo);
GCNX_API int gcnx_dense_bwd_deferred(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh,
                            const float* w, int64_t n, int32_t fi, int32_t fo, int prec, float* dx, int64_t lddx,
                            const float* y_mask, int64_t ldy, float* db_prev, float* dw, float* scratch,
                            int64_t scratch_floats, gcnx_pending_reduce* pending);

/* The last gradient of a step and the optimizer apply as one call: gcnx_gemm_dw (dw = X^T dH, written into the flat
 * gradient buffer: grads <= dw, dw + fi*fo <= grads + n_params) followed by gcnx_sgd over all n_params parameters
 * (gcn.py:337-338: tape.gradient, then optimizer.apply_gradients).  Where dW is a split-K product its reduction
 * launch also applies the update (one launch instead of two); same dW and parameter bits as the two calls.
 * pending (may be NULL): a reduction left by gcnx_dense_bwd_deferred whose results lie in the same flat gradient
 * buffer; it is finished -- and its parameters updated -- by the same launch. */
GCNX_API int gcnx_gemm_dw_sgd(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, float* dw,
                     int64_t n, int32_t fi, int32_t fo, int prec, float* params, float* grads, int64_t n_params,
                     float lr, const gcnx_pending_reduce* pending);

/* ---- GCNConv as one launch, small-feature regime (F <= 128; BASELINE config 2) --------------------------------
 * GCNConv.call (Spektral, the layer gcn.py:334 runs) is A (X W) + b; these entry points evaluate the same product
 * as (A X) W so that one workgroup owns 32 rows from the neighbour gather to the activation -- one launch per layer
 * instead of gcnx_gemm + gcnx_spmm_csr and no [N, F] intermediate between them.  Same results up to fp32 rounding
 * of the re-associated sum (tests: 1e-5 relative against the float64 oracle).
 * gcnx_gcn_conv_fused_ok: 1 if the shapes are served (fi in {32, 64, 128}, fo a multiple of 16 up to 128, ldx % 4 == 0,
 * n * ldx * 4 < 2^32); the entry points return GCNX_ERR_UNSUPPORTED otherwise -- use the two-launch form then. */
GCNX_API int gcnx_gcn_conv_fused_ok(int64_t n, int32_t fi, int32_t fo, int64_t ldx);
/* out[n, fo] = act((A x) w + bias), A in CSR (vals NULL: all ones); s (may be NULL) receives S = A x [n, fi], the
 * operand of the weight gradient dW = S^T dZ (gcnx_gemm_dw2).  wt_out (may be NULL) receives w^T [fo, fi], the layout
 * gcnx_gcn_conv_bwd_pool reads its weight operand fastest in.  act: GCNX_ACT_NONE / GCNX_ACT_RELU.  prec: GCNX_PREC_F32
 * (exact fp32 products on the fp32 MFMA) or GCNX_PREC_BF16X3 (split-bf16 on the bf16 MFMA); the gather is fp32 either way. */
GCNX_API int gcnx_gcn_conv_fwd(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                      const float* x, int64_t ldx, int32_t n, int32_t fi, const float* w, int32_t fo,
                      const float* bias, int act, float* s, int64_t lds, float* out, int64_t ldo, float* wt_out,
                      int prec);
/* gcnx_gcn_conv_fwd with the global pool's partial sums out of the same launch (GlobalSumPool / GlobalAvgPool over the
 * layer's output, gcn.py:334): node_graph [n] is the DisjointLoader id vector `i` (non-decreasing, values in [0, b)), b the
 * number of graphs.  Row t + g of tile_part / tile_cnt ((ceil(n / 32) + b) rows of fo floats each, 16-byte aligned) receives the
 * column sums / the number of positive entries of the rows of graph g inside the 32-row tile t; the other rows are not
 * written.  A graph's pooled sum is the sum of the rows t + g over its tiles t = first_row / 32 .. last_row / 32 --
 * gcnx_gcn_conv_bwd_pool(head) consumes them in that form. */
GCNX_API int gcnx_gcn_conv_fwd_pool(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                      const float* x, int64_t ldx, int32_t n, int32_t fi, const float* w, int32_t fo,
                      const float* bias, int act, float* s, int64_t lds, float* out, int64_t ldo, float* wt_out,
                      int prec, const int32_t* node_graph, int32_t b, float* tile_part, float* tile_cnt);
/* The classifier head of a small-batch step whose pool is still in per-tile partial sums (plain data, no ownership).  In
 * a latency-bound step the pool and the head -- Dense(softmax) + CCE + their gradients on a [B, H] operand: 7 + 10 us of
 * little parallel work -- sit between the forward and the backward aggregation only because that needs dPooled.  With
 * this struct gcnx_gcn_conv_bwd_pool adds up the tile partials of its tile's graphs and evaluates dPooled itself (a few
 * hundred flops per graph), writing each graph's totals to pool_sum / pool_cnt on the way (the workgroup that holds the
 * graph's first row does; rows of graphs without nodes are NOT written: zero them beforehand if there can be any), and
 * gcnx_gemm_dw2 computes everything else the head produces (probabilities, loss, accuracy, dW, db, db_relu: leaves
 * nobody in the step waits for) from those totals in the FIRST workgroup of the weight-gradient launch.
 *   tile_part / tile_cnt: gcnx_gcn_conv_fwd_pool output, tile_rows = ceil(n / 32) + b rows of h floats
 *   pool_sum / pool_cnt [b, h]: written by gcnx_gcn_conv_bwd_pool, read by gcnx_gemm_dw2 (16-byte aligned)
 *   w [h, c], bias [c], y [b, c] one-hot; c <= 2 for the in-kernel form (gcnx_gcn_conv_bwd_pool returns
 *   GCNX_ERR_UNSUPPORTED beyond that; gcnx_gemm_dw2 takes any c the head kernel does)
 *   outputs (written by gcnx_gemm_dw2): probs [b, c], loss_acc [2] (mean loss, hit count), dw [h, c], db [c],
 *   db_relu [h] (may be NULL), pooled [b, h], dpooled [b, h]. */
typedef struct gcnx_head_args {
  const float* tile_part; const float* tile_cnt; int64_t tile_rows;
  float* pool_sum; float* pool_cnt;
  const int32_t* graph_ptr; int32_t b; int32_t h; int pool_mode;
  const float* w; const float* bias; const float* y; int32_t c; float denom; int cce_mode;
  float* probs; float* loss_acc; float* dw; float* db; float* db_relu; float* pooled; float* dpooled;
} gcnx_head_args;

/* Backward from the global pool down to the pre-activation gradient of the layer below, one launch:
 *   dZ2[j] = pool'(dpooled)[graph(j)] * [y2[j] > 0]                      (GlobalSumPool / GlobalAvgPool', ReLU')
 *   dz1    = ((A^T dZ2) w2^T) * [y1 > 0]                                 (aggregation', MatMul', ReLU' of layer 1)
 *   db1    = column sums of dz1                                          (BiasAddGrad)
 * rowptr_t / colidx_t / vals_t: A^T in CSR; node_graph: the DisjointLoader id vector i[n] (graph of every row);
 * w2 [f1, f2], or with w2_transposed != 0 its transpose [f2, f1] (the forward's wt_out: coalesced operand loads);
 * y2 [n, f2], y1 [n, f1]: the saved ReLU outputs.  dz2 (may be NULL) receives dZ2 [n, f2] (the operand
 * of dW2 = S2^T dZ2).  db1 (may be NULL): with `pending` and a scratch of >= gcnx_gcn_conv_bwd_scratch_floats(n, f1)
 * floats the per-tile partial sums stay in scratch and *pending describes the reduction (gcnx_gemm_dw2 finishes it);
 * otherwise db1 is complete on return.  mode: GCNX_POOL_SUM / GCNX_POOL_AVG.  head (may be NULL): see gcnx_head_args --
 * dpooled is then not read (may be NULL). */
GCNX_API int64_t gcnx_gcn_conv_bwd_scratch_floats(int64_t n, int32_t f1);
GCNX_API int gcnx_gcn_conv_bwd_pool(gcnx_ctx* ctx, const int32_t* rowptr_t, const int32_t* colidx_t, const float* vals_t,
                      const float* y2, int64_t ldy2, const int32_t* node_graph, const int32_t* graph_ptr, int32_t b,
                      const float* dpooled, int64_t lddp, int mode, int32_t n, int32_t f2, const float* w2, int32_t f1,
                      int w2_transposed, const float* y1, int64_t ldy1, float* dz2, int64_t lddz2, float* dz1, int64_t lddz1, float* db1,
                      float* scratch, int64_t scratch_floats, gcnx_pending_reduce* pending, int prec,
                      const gcnx_head_args* head);
/* Two weight gradients in one launch, dwa = xa^T dha [fia, foa] and dwb = xb^T dhb [fib, fob] over the same n rows
 * (MatMul grads wrt the kernels, gcn.py:337), both inside the flat gradient buffer `grads`; with params != NULL the
 * reduction launch also applies p -= lr * g to all n_params parameters and finishes `pending` (column sums left by
 * gcnx_gcn_conv_bwd_pool), as gcnx_gemm_dw_sgd does.  params == NULL: gradients only.  leaf (may be NULL): the classifier
 * head's outputs (gcnx_head_args) are computed by the first workgroups of the same launch; its gradients must lie in
 * `grads` too when params != NULL (they are updated with the rest). */
GCNX_API int gcnx_gemm_dw2(gcnx_ctx* ctx, const float* xa, int64_t ldxa, const float* dha, int64_t lddha, float* dwa,
                      int32_t fia, int32_t foa, const float* xb, int64_t ldxb, const float* dhb, int64_t lddhb,
                      float* dwb, int32_t fib, int32_t fob, int64_t n, int prec, float* params, float* grads,
                      int64_t n_params, float lr, const gcnx_pending_reduce* pending, const gcnx_head_args* leaf);

/* ---- aggregation with bf16 features ("fp32 and bf16 both reported") ----
 * out[t, :] = bf16(act(sum_e vals[e] * h[colidx[e], :] + bias)): GCNConv.call's / GeneralConv's aggregation (gcn.py:334) on
 * activations stored as bf16 (uint16_t bit patterns), fp32 accumulation, round-to-nearest-even on the way out; bias fp32
 * (may be NULL), vals may be NULL (ones).  f in {64, 128, 256}, n * ldh * 2 < 2^32 (GCNX_ERR_UNSUPPORTED otherwise);
 * act: GCNX_ACT_NONE / GCNX_ACT_RELU.  A measured variant: the models keep fp32 activations.
 * gcnx_f32_to_bf16 (round to nearest even) / gcnx_bf16_to_f32: contiguous arrays of `count` elements. */
GCNX_API int gcnx_spmm_csr_bf16(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                      const uint16_t* h, int64_t ldh, const float* bias, uint16_t* out, int64_t ldo, int32_t n, int32_t f,
                      int act);
GCNX_API int gcnx_f32_to_bf16(gcnx_ctx* ctx, const float* x, uint16_t* y, int64_t count);
GCNX_API int gcnx_bf16_to_f32(gcnx_ctx* ctx, const uint16_t* x, float* y, int64_t count);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI (new capability) ---- */
GCNX_API int gcnx_comm_unique_id(char id[GCNX_UNIQUE_ID_BYTES]);
GCNX_API int gcnx_comm_init_rank(gcnx_ctx* ctx, const char id[GCNX_UNIQUE_ID_BYTES], int nranks,
                        int rank, gcnx_comm** out);
GCNX_API int gcnx_comm_destroy(gcnx_comm* comm);
/* In-place all-reduce of a device fp32 buffer on the ctx stream. */
GCNX_API int gcnx_allreduce_f32(gcnx_ctx* ctx, gcnx_comm* comm, float* buf, int64_t n, int op);

#ifdef __cplusplus
}
#endif
#endif /* GCNX_H */
