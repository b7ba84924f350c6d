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

namespace {

constexpr int kHeadRows = 32;     // graphs per workgroup
constexpr int kHeadMaxC = 32;     // classes held in LDS per graph
constexpr int kHeadLdsFloats = 14 * 1024;   // 56 KiB for the staged operands (else they are read from global)

template <bool STAGED>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ pooled, int64_t ldp,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   const float* __restrict__ y, int32_t b, int32_t h, int32_t c,
                                                   float denom, float* __restrict__ probs,
                                                   float* __restrict__ loss_acc, float* __restrict__ dw,
                                                   float* __restrict__ db, float* __restrict__ dpooled, int64_t lddp,
                                                   float* __restrict__ slabs, int* __restrict__ ticket) {
  constexpr bool staged = STAGED;
  __shared__ float s_z[kHeadRows][kHeadMaxC + 1];   // logits, then dlogits
  __shared__ float s_y[kHeadRows * kHeadMaxC];      // labels of this workgroup's graphs, [rows][c] packed
  __shared__ float s_red[2][256];
  __shared__ int s_last;
  extern __shared__ float s_dyn[];                // staged: pooled rows [rows][h+1] | w [h*c]
  const int tid = threadIdx.x;
  const int g0 = blockIdx.x * kHeadRows;
  const int rows = min(kHeadRows, b - g0);
  const bool train = dw != nullptr;
  // Everything below walks the two small operands several times with dependent, strided reads; staged once in
  // LDS (coalesced loads, one memory latency) those walks cost LDS latency instead of L2 latency per element.
  const int ps = h + 1;                           // padded row stride: column walks (dW) stay conflict-free
  float* s_p = s_dyn;
  float* s_w = s_dyn + kHeadRows * ps;
  if (y) for (int idx = tid; idx < rows * c; idx += 256) s_y[idx] = y[(int64_t)g0 * c + idx];
  if (staged) {
    for (int i = tid >> 6; i < rows; i += 4)             // one wave per row: no integer division in the loops
      for (int j = tid & 63; j < h; j += 64) s_p[i * ps + j] = pooled[(int64_t)(g0 + i) * ldp + j];
    for (int idx = tid; idx < h * c; idx += 256) s_w[idx] = w[idx];
    __syncthreads();
  }
  auto P = [&](int i, int j) { return staged ? s_p[i * ps + j] : pooled[(int64_t)(g0 + i) * ldp + j]; };
  auto W = [&](int j, int k) { return staged ? s_w[j * c + k] : w[(int64_t)j * c + k]; };

  // logits[i][k] = pooled[i,:] . w[:,k] + bias[k]: four lanes share one output (j = q, q+4, ...), fixed combine order
  for (int idx = tid >> 2; idx < rows * c; idx += 64) {
    const int i = idx / c, k = idx % c, q = tid & 3;
    float acc = 0.f;
#pragma unroll 8
    for (int j = q; j < h; j += 4) acc = fmaf(P(i, j), W(j, k), acc);
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (q == 0) s_z[i][k] = acc + (bias ? bias[k] : 0.f);
  }
  __syncthreads();

  // per graph: softmax, clipped CCE, accuracy, dlogits (same arithmetic as softmax_cce_kernel in reduce.hip)
  float loss = 0.f, hit = 0.f;
  if (tid < rows) {
    float* z = s_z[tid];
    const int64_t g = g0 + tid;
    float m = -INFINITY;
    for (int k = 0; k < c; ++k) m = fmaxf(m, z[k]);
    float sum = 0.f;
    for (int k = 0; k < c; ++k) sum += expf(z[k] - m);
    float ymsum = 0.f, pmax = -1.f, ymax = -INFINITY;
    int pa = 0, ya = 0;
    for (int k = 0; k < c; ++k) {
      const float p = expf(z[k] - m) / sum;
      probs[g * c + k] = p;
      if (y) {
        const float yk = s_y[tid * c + k];
        if (p > 1e-7f && p < 1.0f - 1e-7f) ymsum += yk;   // clip_by_value passes no gradient outside
        if (yk > ymax) { ymax = yk; ya = k; }
        loss -= yk * logf(fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f));
      }
      if (p > pmax) { pmax = p; pa = k; }
    }
    if (y) {
      hit = (pa == ya) ? 1.f : 0.f;
      for (int k = 0; k < c; ++k) {
        const float p = expf(z[k] - m) / sum;
        const float ym = (p > 1e-7f && p < 1.0f - 1e-7f) ? s_y[tid * c + k] : 0.f;
        z[k] = (p * ymsum - ym) / denom;                  // dlogits replaces the logit in LDS
      }
    }
  }
  if (!y) return;                                         // inference: probabilities only (uniform exit)
  s_red[0][tid] = loss;
  s_red[1][tid] = hit;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { s_red[0][tid] += s_red[0][tid + off]; s_red[1][tid] += s_red[1][tid + off]; }
    __syncthreads();
  }
  const int nblk = gridDim.x;
  const int npart = h * c + c + 2;                        // dW | db | loss sum, hits
  float* part = nblk > 1 ? slabs + (int64_t)blockIdx.x * npart : nullptr;
  if (tid == 0) {
    if (part) { part[h * c + c] = s_red[0][0]; part[h * c + c + 1] = s_red[1][0]; }
    else { loss_acc[0] = s_red[0][0] / denom; loss_acc[1] = s_red[1][0]; }
  }
  if (!train) {
    if (nblk == 1) return;
  } else {
    // dPooled[i, j] = sum_k dlogits[i,k] * w[j,k]
    for (int i = tid >> 6; i < rows; i += 4)
      for (int j = tid & 63; j < h; j += 64) {
        float acc = 0.f;
        for (int k = 0; k < c; ++k) acc = fmaf(s_z[i][k], W(j, k), acc);
        dpooled[(int64_t)(g0 + i) * lddp + j] = acc;
      }
    // dW[j, k] = sum_i pooled[i,j] * dlogits[i,k];  db[k] = sum_i dlogits[i,k]   (this workgroup's graphs)
    for (int idx = tid; idx < h * c; idx += 256) {
      const int j = idx / c, k = idx % c;
      float acc = 0.f;
#pragma unroll 8
      for (int i = 0; i < rows; ++i) acc = fmaf(P(i, j), s_z[i][k], acc);
      if (part) part[idx] = acc; else dw[idx] = acc;
    }
    if (tid < c) {
      float acc = 0.f;
      for (int i = 0; i < rows; ++i) acc += s_z[i][tid];
      if (part) part[h * c + tid] = acc; else if (db) db[tid] = acc;
    }
    if (nblk == 1) return;
  }
  // several workgroups: the last one to arrive reduces the slabs in workgroup order
  __threadfence();
  __syncthreads();
  if (tid == 0) s_last = (atomicAdd(ticket, 1) == nblk - 1);
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  const int lo = train ? 0 : h * c + c;
  for (int idx = lo + tid; idx < npart; idx += 256) {
    float acc = 0.f;
    for (int q = 0; q < nblk; ++q) acc += slabs[(int64_t)q * npart + idx];
    if (idx < h * c) dw[idx] = acc;
    else if (idx < h * c + c) { if (db) db[idx - h * c] = acc; }
    else if (idx == h * c + c) loss_acc[0] = acc / denom;
    else loss_acc[1] = acc;
  }
  if (tid == 0) *ticket = 0;                              // ready for the next launch (same stream: ordered)
}

}  // namespace

extern "C" {

int gcnx_dense_softmax_cce(gcnx_ctx* ctx, const float* pooled, int64_t ldp, const float* w, const float* bias,
                           const float* y, int32_t b, int32_t h, int32_t c, float denom, float* probs,
                           float* loss_acc, float* dw, float* db, float* dpooled, int64_t lddp) {
  GCNX_CHECK_CTX(ctx);
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
  float* slabs = nullptr;
  if (nblk > 1 && y) {
    int rc = gcnx_ws_reserve(ctx, (size_t)nblk * ((size_t)h * c + c + 2) * sizeof(float));
    if (rc) return rc;
    slabs = (float*)ctx->ws;
  }
  const size_t need = (size_t)kHeadRows * (h + 1) + (size_t)h * c;   // floats of the staged operands
  const int staged = need <= (size_t)kHeadLdsFloats;
  if (staged)
    hipLaunchKernelGGL(head_kernel<true>, dim3(nblk), dim3(256), need * sizeof(float), ctx->stream, pooled, ldp, w, bias, y,
                       b, h, c, denom, probs, loss_acc, dw, db, dpooled, lddp, slabs, ctx->flag + 3);
  else
    hipLaunchKernelGGL(head_kernel<false>, dim3(nblk), dim3(256), 0, ctx->stream, pooled, ldp, w, bias, y, b, h, c, denom,
                       probs, loss_acc, dw, db, dpooled, lddp, slabs, ctx->flag + 3);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // extern "C"
