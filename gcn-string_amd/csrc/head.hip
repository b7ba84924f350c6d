// Classifier head in one launch: Dense(C, softmax) on the pooled graph vectors + CategoricalCrossentropy +
// categorical_accuracy, and the whole head backward (dW, db, dPooled).
//
// Replaces, for the last layer of the model (gcn.py:320 `activation="softmax"`; loss gcn.py:326,335; metric
// gcn.py:339; gradients gcn.py:337), the sequence MatMul+BiasAdd -> Softmax -> CCE -> MatMul grads -> BiasAddGrad
// that gcnx_gemm / gcnx_softmax_cce / gcnx_gemm_dw / gcnx_act_bias_grad / gcnx_gemm_dx run as 6-7 launches on
// matrices of a few kilobytes ([B,H] x [H,C], B = graphs per batch, C = classes): at BASELINE config 2 each of
// those launches is pure latency (4-15 us), together a fifth of the training step.
//
// One workgroup owns 64 graphs: logits, softmax, loss / accuracy terms, dlogits (gradient of the clipped
// loss, as gcnx_softmax_cce) and the dPooled rows are row-local.  dW, db and the two loss_acc sums are
// reductions over graphs: every workgroup writes its partial to a slab of the ctx workspace and the last one
// to arrive (agent-scope ticket) adds the slabs in workgroup order -- a fixed summation order, so the result
// is reproducible and independent of scheduling.
#include "common.h"
#include "head_body.h"   // the kernel itself (shared with gemm.hip, which runs it as the first workgroups of another launch)
using namespace gcnx_head;

extern "C" {

static int head_impl(gcnx_ctx* ctx, const float* pooled, int64_t ldp, const float* w, const float* bias, const float* y,
                     int32_t b, int32_t h, int32_t c, float denom, float* probs, float* loss_acc, float* dw, float* db,
                     float* dpooled, int64_t lddp, const int32_t* graph_ptr, const float* x, int64_t ldx, int pool_mode,
                     float* pooled_out, float* db_relu, int cce_mode);

// prod[g][c] = pool'(dPooled)[g][c] * cnt[g][c]: the per-graph terms of the bias gradient of the ReLU layer under the pool
__global__ __launch_bounds__(256) void dp_cnt_kernel(const float* __restrict__ dp, int64_t lddp, const float* __restrict__ cnt,
                                                     const int32_t* __restrict__ gp, int32_t b, int32_t h, int avg,
                                                     float* __restrict__ prod) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)b * h) return;
  const int g = (int)(i / h), c = (int)(i % h);
  const float sc = avg ? 1.0f / (float)max(gp[g + 1] - gp[g], 1) : 1.0f;
  prod[i] = dp[(int64_t)g * lddp + c] * sc * cnt[i];
}


int gcnx_dense_softmax_cce(gcnx_ctx* ctx, const float* pooled, int64_t ldp, const float* w, const float* bias,
                           const float* y, int32_t b, int32_t h, int32_t c, float denom, float* probs,
                           float* loss_acc, float* dw, float* db, float* dpooled, int64_t lddp, int cce_mode) {
  return head_impl(ctx, pooled, ldp, w, bias, y, b, h, c, denom, probs, loss_acc, dw, db, dpooled, lddp, nullptr, nullptr,
                   0, 0, nullptr, nullptr, cce_mode);
}

// The head on pooled rows that are already there (gcnx_spmm_csr_relu_bits_pool left them, with the positive counts):
// gcnx_dense_softmax_cce + the bias gradient of the ReLU layer under the pool from the counts,
// db_relu[c] = sum_g pool'(dPooled)[g][c] * cnt[g][c].
int gcnx_pooled_dense_softmax_cce(gcnx_ctx* ctx, const int32_t* graph_ptr, int pool_mode, const float* pooled, int64_t ldp,
                                  const float* cnt, const float* w, const float* bias, const float* y, int32_t b, int32_t h, int32_t c,
                                  float denom, float* probs, float* loss_acc, float* dw, float* db, float* dpooled, int64_t lddp,
                                  float* db_relu, int cce_mode) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "classifier head");
  GCNX_REQUIRE(ctx, b >= 0 && h >= 0 && c > 0, "gcnx_pooled_dense_softmax_cce: bad shape");
  GCNX_REQUIRE(ctx, pool_mode == GCNX_POOL_SUM || pool_mode == GCNX_POOL_AVG, "gcnx_pooled_dense_softmax_cce: SUM / AVG pooling only");
  GCNX_REQUIRE(ctx, !db_relu || (dw && dpooled && cnt && graph_ptr), "gcnx_pooled_dense_softmax_cce: db_relu needs the gradient outputs, the counts and graph_ptr");
  if (db_relu) {      // workspace for the products b x h and their reduction, reserved before the head takes its slabs from it
    const int nblk = gcnx_cdiv(b, kHeadRows);
    const size_t slab_floats = nblk > 1 ? (((size_t)nblk * ((size_t)h * c + c + 2) + 3) & ~(size_t)3) : 0;
    int rc = gcnx_ws_reserve(ctx, std::max(std::max(slab_floats * sizeof(float), gcnx_colsum_partials_ws(b, h)), (size_t)b * h * sizeof(float)));
    if (rc) return rc;
  }
  int rc = gcnx_dense_softmax_cce(ctx, pooled, ldp, w, bias, y, b, h, c, denom, probs, loss_acc, dw, db, dpooled, lddp, cce_mode);
  if (rc || !db_relu || b == 0 || h == 0) return rc;
  GCNX_REQUIRE(ctx, h % 4 == 0 && (reinterpret_cast<uintptr_t>(db_relu) & 15) == 0, "gcnx_pooled_dense_softmax_cce: db_relu needs h %% 4 == 0 and a 16-byte aligned buffer");
  hipLaunchKernelGGL(dp_cnt_kernel, dim3(gcnx_cdiv((int64_t)b * h, 256)), dim3(256), 0, ctx->stream, dpooled, lddp, cnt, graph_ptr, b, h,
                     pool_mode == GCNX_POOL_AVG ? 1 : 0, (float*)ctx->ws);
  GCNX_LAUNCH_OK(ctx);
  return gcnx_colsum_partials(ctx, b, h, db_relu);
}

int gcnx_pool_dense_softmax_cce(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* x, int64_t ldx, int pool_mode,
                                int32_t* argmax, float* pooled, int64_t ldp, const float* w, const float* bias,
                                const float* y, int32_t b, int32_t h, int32_t c, float denom, float* probs,
                                float* loss_acc, float* dw, float* db, float* dpooled, int64_t lddp, float* db_relu,
                                int cce_mode) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "pool + classifier head");
  GCNX_REQUIRE(ctx, b >= 0 && h >= 0 && c > 0, "gcnx_pool_dense_softmax_cce: bad shape");
  GCNX_REQUIRE(ctx, pool_mode >= GCNX_POOL_SUM && pool_mode <= GCNX_POOL_MAX, "gcnx_pool_dense_softmax_cce: unknown pool mode %d",
               pool_mode);
  GCNX_REQUIRE(ctx, b == 0 || h == 0 || (graph_ptr && x && pooled), "gcnx_pool_dense_softmax_cce: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= h && ldp >= h, "gcnx_pool_dense_softmax_cce: leading dimension too small");
  GCNX_REQUIRE(ctx, !db_relu || (dw && pool_mode != GCNX_POOL_MAX),
               "gcnx_pool_dense_softmax_cce: db_relu needs the gradient outputs and SUM / AVG pooling");
  const bool fused = b > 0 && h > 0 && h % 4 == 0 && c <= kHeadMaxC &&
                     head_lds_floats(h, c, db_relu != nullptr) <= (size_t)kHeadLdsFloats &&
                     (reinterpret_cast<uintptr_t>(x) & 15) == 0 && gcnx_pool_split(ctx, b, h, pool_mode, 1) > 1;
  if (!fused && db_relu && b > 0 && b <= 4096 && h > 0 && h % 4 == 0 && ldp == h && (reinterpret_cast<uintptr_t>(db_relu) & 15) == 0 &&
      gcnx_pool_split(ctx, b, h, pool_mode, 4) == 1) {   // (the row order gcnx_segment_pool sums in: same pooled bits)
    // Many graphs (a large batch): the pool's pass over x also counts the positive entries per (graph, column) -- all
    // that db_relu = sum_g pool'(dPooled)[g] * #[x_g > 0] needs -- instead of a second pass over x
    // (gcnx_pool_bwd_colsum: 1 GB at config 3).  Workspace: [head slabs, later the products b x h + their reduction's
    // scratch | counts b x h]; reserved once, up front.
    const int nblk = gcnx_cdiv(b, kHeadRows);
    const size_t slab_floats = nblk > 1 ? (((size_t)nblk * ((size_t)h * c + c + 2) + 3) & ~(size_t)3) : 0;   // as head_impl
    const size_t cnt_off = (std::max(slab_floats * sizeof(float), gcnx_colsum_partials_ws(b, h)) / sizeof(float) + 63) & ~(size_t)63;
    int rc = gcnx_ws_reserve(ctx, (cnt_off + (size_t)b * h) * sizeof(float));
    if (rc) return rc;
    float* cnt = (float*)ctx->ws + cnt_off;
    rc = gcnx_pool_partials(ctx, graph_ptr, x, ldx, b, h, pool_mode, 1, pooled, cnt, 0);
    if (rc) return rc;
    rc = gcnx_dense_softmax_cce(ctx, pooled, ldp, w, bias, y, b, h, c, denom, probs, loss_acc, dw, db, dpooled, lddp, cce_mode);
    if (rc) return rc;
    hipLaunchKernelGGL(dp_cnt_kernel, dim3(gcnx_cdiv((int64_t)b * h, 256)), dim3(256), 0, ctx->stream, dpooled, lddp,
                       (const float*)cnt, graph_ptr, b, h, pool_mode == GCNX_POOL_AVG ? 1 : 0, (float*)ctx->ws);
    GCNX_LAUNCH_OK(ctx);
    return gcnx_colsum_partials(ctx, b, h, db_relu);
  }
  if (!fused) {   // MAX pooling, many graphs (no split), operands too large for LDS: the separate calls as they are
    int rc = gcnx_segment_pool(ctx, graph_ptr, x, ldx, pooled, b, h, pool_mode, argmax);
    if (rc) return rc;
    rc = gcnx_dense_softmax_cce(ctx, pooled, ldp, w, bias, y, b, h, c, denom, probs, loss_acc, dw, db, dpooled, lddp, cce_mode);
    if (rc || !db_relu) return rc;
    return gcnx_pool_bwd_colsum(ctx, graph_ptr, b, dpooled, lddp, x, ldx, h, pool_mode, db_relu);
  }
  return head_impl(ctx, pooled, ldp, w, bias, y, b, h, c, denom, probs, loss_acc, dw, db, dpooled, lddp, graph_ptr, x, ldx,
                   pool_mode, pooled, db_relu, cce_mode);
}

}  // extern "C"

static int head_impl(gcnx_ctx* ctx, const float* pooled, int64_t ldp, const float* w, const float* bias, const float* y,
                     int32_t b, int32_t h, int32_t c, float denom, float* probs, float* loss_acc, float* dw, float* db,
                     float* dpooled, int64_t lddp, const int32_t* graph_ptr, const float* x, int64_t ldx, int pool_mode,
                     float* pooled_out, float* db_relu, int cce_mode) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, cce_mode == GCNX_CCE_PROBS || cce_mode == GCNX_CCE_LOGITS, "gcnx_dense_softmax_cce: unknown cce_mode %d", cce_mode);
  const int fl = cce_mode == GCNX_CCE_LOGITS ? 1 : 0;
  GCNX_REQUIRE(ctx, b >= 0 && h >= 0 && c > 0, "gcnx_dense_softmax_cce: bad shape");
  GCNX_REQUIRE(ctx, c <= kHeadMaxC, "gcnx_dense_softmax_cce: at most %d classes (got %d); use gcnx_gemm + gcnx_softmax_cce",
               kHeadMaxC, c);
  if (b == 0) {
    if (y && loss_acc) GCNX_HIP(ctx, hipMemsetAsync(loss_acc, 0, 2 * sizeof(float), ctx->stream));
    if (dw) GCNX_HIP(ctx, hipMemsetAsync(dw, 0, (size_t)h * c * sizeof(float), ctx->stream));
    if (dw && db) GCNX_HIP(ctx, hipMemsetAsync(db, 0, (size_t)c * sizeof(float), ctx->stream));
    return GCNX_OK;
  }
  GCNX_REQUIRE(ctx, (pooled || h == 0) && (w || h == 0) && probs, "gcnx_dense_softmax_cce: NULL pointer");
  GCNX_REQUIRE(ctx, ldp >= h, "gcnx_dense_softmax_cce: leading dimension too small");
  GCNX_REQUIRE(ctx, !y || (loss_acc && denom > 0.f), "gcnx_dense_softmax_cce: labels need loss_acc and a positive denom");
  GCNX_REQUIRE(ctx, !dw || (y && dpooled && lddp >= h), "gcnx_dense_softmax_cce: gradients need labels and dpooled");
  const int nblk = gcnx_cdiv(b, kHeadRows);
  const bool want_db = pooled_out && db_relu && dw;
  const size_t slab_floats =
      (nblk > 1 && y) ? (((size_t)nblk * ((size_t)h * c + c + 2 + (want_db ? h : 0)) + 3) & ~(size_t)3) : 0;   // partials 16-B aligned
  const int nsplit = pooled_out ? gcnx_pool_split(ctx, b, h, pool_mode, 1) : 1;       // > 1 (checked by the caller)
  const size_t one_part = pooled_out ? (size_t)nsplit * b * h : 0;
  const size_t part_floats = one_part * (want_db ? 2 : 1);
  if (slab_floats + part_floats) {
    int rc = gcnx_ws_reserve(ctx, (slab_floats + part_floats) * sizeof(float));
    if (rc) return rc;
  }
  float* slabs = slab_floats ? (float*)ctx->ws : nullptr;
  const size_t need = head_lds_floats(h, c, want_db);   // floats of the staged operands
  const int staged = need <= (size_t)kHeadLdsFloats;
  PoolParts pp{nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr};
  if (pooled_out) {
    float* part = (float*)ctx->ws + slab_floats;
    float* cnt_part = want_db ? part + one_part : nullptr;
    int rc = gcnx_pool_partials(ctx, graph_ptr, x, ldx, b, h, pool_mode, nsplit, part, cnt_part, 1);
    if (rc) return rc;
    pp = PoolParts{part, graph_ptr, pooled_out, nsplit, pool_mode == GCNX_POOL_AVG ? 1 : 0, cnt_part, want_db ? db_relu : nullptr};
#define GCNX_HEAD(S, P, C)                                                                                                 \
    hipLaunchKernelGGL((head_kernel<S, P, C>), dim3(nblk), dim3(256), need * sizeof(float), ctx->stream, pooled, ldp, w, bias, \
                       y, b, h, c, denom, probs, loss_acc, dw, db, dpooled, lddp, slabs, ctx->flag + 3, pp, fl)
    if (c == 2) GCNX_HEAD(true, true, 2); else GCNX_HEAD(true, true, 0);
  } else if (staged) {
    if (c == 2) GCNX_HEAD(true, false, 2); else GCNX_HEAD(true, false, 0);
#undef GCNX_HEAD
  } else {
    hipLaunchKernelGGL((head_kernel<false, false>), dim3(nblk), dim3(256), 0, ctx->stream, pooled, ldp, w, bias, y, b, h, c,
                       denom, probs, loss_acc, dw, db, dpooled, lddp, slabs, ctx->flag + 3, pp, fl);
  }
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

// The classifier head from the pool's partial sums (gcnx_head_args), as a launch of its own.
int gcnx_head_from_parts(gcnx_ctx* ctx, const gcnx_head_args* a) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, a != nullptr, "gcnx_head_from_parts: NULL arguments");
  const bool want_db = a->db_relu != nullptr;
  const size_t need = head_lds_floats(a->h, a->c, want_db);
  GCNX_REQUIRE(ctx, need <= (size_t)kHeadLdsFloats, "gcnx_head_from_parts: operands too large for the head kernel (h=%d c=%d)", a->h, a->c);
  const int nblk = gcnx_cdiv(a->b, kHeadRows);
  const size_t slab_floats = nblk > 1 ? (((size_t)nblk * ((size_t)a->h * a->c + a->c + 2 + (want_db ? a->h : 0)) + 3) & ~(size_t)3) : 0;
  if (slab_floats) {
    int rc = gcnx_ws_reserve(ctx, slab_floats * sizeof(float));
    if (rc) return rc;
  }
  const PoolParts pp{a->pool_sum, a->graph_ptr, a->pooled, 1, a->pool_mode == GCNX_POOL_AVG ? 1 : 0, want_db ? a->pool_cnt : nullptr,
                     want_db ? a->db_relu : nullptr};
  const int fl = a->cce_mode == GCNX_CCE_LOGITS ? 1 : 0;
  static bool attr_set = false;
  if (!attr_set) {
    GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&head_kernel<true, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kHeadLdsFloats * 4));
    GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&head_kernel<true, true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kHeadLdsFloats * 4));
    attr_set = true;
  }
  if (a->c == 2)
    hipLaunchKernelGGL((head_kernel<true, true, 2>), dim3(nblk), dim3(256), need * sizeof(float), ctx->stream, (const float*)nullptr,
                       (int64_t)a->h, a->w, a->bias, a->y, a->b, a->h, a->c, a->denom, a->probs, a->loss_acc, a->dw, a->db, a->dpooled,
                       (int64_t)a->h, slab_floats ? (float*)ctx->ws : nullptr, ctx->flag + 3, pp, fl);
  else
    hipLaunchKernelGGL((head_kernel<true, true, 0>), dim3(nblk), dim3(256), need * sizeof(float), ctx->stream, (const float*)nullptr,
                       (int64_t)a->h, a->w, a->bias, a->y, a->b, a->h, a->c, a->denom, a->probs, a->loss_acc, a->dw, a->db, a->dpooled,
                       (int64_t)a->h, slab_floats ? (float*)ctx->ws : nullptr, ctx->flag + 3, pp, fl);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

