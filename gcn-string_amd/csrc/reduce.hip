// Row-reduction and elementwise kernels of the path: BiasAddGrad / activation grad, global pools
// (SegmentSum/Mean/Max) and their gradients, softmax + categorical cross-entropy, SGD.
// All HBM-bound; all reductions are two-stage and atomics-free (bitwise reproducible, so that a
// sharded run can be compared with a single-GPU run).
#include "common.h"

namespace {

constexpr int kColsumRows = 512;  // rows per first-stage workgroup

// Column sums of x[n, f] -> part[chunk][f].  Block = 64 column lanes x 4 row groups.
// Optional fused activation gradient: dz = dy * act'(y) is written and summed instead of x.
template <bool FUSE_ACT>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int32_t f,
                                                     int64_t rows_per_chunk, float* __restrict__ part,
                                                     const float* __restrict__ y, int64_t ldy, float* __restrict__ dz,
                                                     int64_t lddz, int act, const float* __restrict__ alpha,
                                                     float* __restrict__ part_alpha) {
  __shared__ float s[4][64];
  __shared__ float s2[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = min(n, r0 + rows_per_chunk);
  float acc = 0.f, acc_a = 0.f;
  if (c < f) {
    const float al = (FUSE_ACT && act == GCNX_ACT_PRELU) ? alpha[c] : 0.f;
    for (int64_t r = r0 + rg; r < r1; r += 4) {
      float v = x[r * ldx + c];
      if (FUSE_ACT) {
        const float yy = y[r * ldy + c];
        if (act == GCNX_ACT_RELU) v = yy > 0.f ? v : 0.f;
        else if (act == GCNX_ACT_PRELU) {
          acc_a += v * fminf(yy, 0.f);
          v = yy > 0.f ? v : al * v;
        }
        dz[r * lddz + c] = v;
      }
      acc += v;
    }
  }
  s[rg][cl] = acc;
  if (FUSE_ACT) s2[rg][cl] = acc_a;
  __syncthreads();
  if (rg == 0 && c < f) {
    if (part) part[(int64_t)blockIdx.y * f + c] = (s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl]);
    if (FUSE_ACT && part_alpha) part_alpha[(int64_t)blockIdx.y * f + c] = (s2[0][cl] + s2[1][cl]) + (s2[2][cl] + s2[3][cl]);
  }
}

// Pool forward: block = (column tile of 64, graph).  Rows of the graph are split over 4 groups.
__global__ __launch_bounds__(256) void pool_fwd_kernel(const int32_t* __restrict__ gp, const float* __restrict__ x,
                                                       int64_t ldx, float* __restrict__ pooled, int32_t f, int mode,
                                                       int32_t* __restrict__ argmax) {
  __shared__ float s[4][64];
  __shared__ int si[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int g = blockIdx.y;
  const int lo = gp[g], hi = gp[g + 1];
  float acc = (mode == GCNX_POOL_MAX) ? -INFINITY : 0.f;
  int arg = lo;
  if (c < f) {
    for (int r = lo + rg; r < hi; r += 4) {
      const float v = x[(int64_t)r * ldx + c];
      if (mode == GCNX_POOL_MAX) {
        if (v > acc) { acc = v; arg = r; }
      } else {
        acc += v;
      }
    }
  }
  s[rg][cl] = acc;
  si[rg][cl] = arg;
  __syncthreads();
  if (rg == 0 && c < f) {
    float out;
    if (mode == GCNX_POOL_MAX) {
      out = s[0][cl];
      int a = si[0][cl];
      // first maximal row wins (matches argmax of the oracle): strict > over ascending row groups
      // is not enough because groups interleave rows, so break ties on the smaller row index.
      for (int q = 1; q < 4; ++q) {
        const float v = s[q][cl];
        const int ai = si[q][cl];
        if (v > out || (v == out && ai < a)) { out = v; a = ai; }
      }
      if (hi == lo) { out = 0.f; a = lo; }
      if (argmax) argmax[(int64_t)g * f + c] = a;
    } else {
      out = (s[0][cl] + s[1][cl]) + (s[2][cl] + s[3][cl]);
      if (mode == GCNX_POOL_AVG && hi > lo) out /= (float)(hi - lo);
    }
    pooled[(int64_t)g * f + c] = out;
  }
}

// Pool backward (+ optional fused ReLU mask of the layer that produced the pooled tensor).
// One wave per row; the row's graph is found by binary search in graph_ptr.
__global__ __launch_bounds__(256) void pool_bwd_kernel(const int32_t* __restrict__ gp, int32_t b,
                                                       const float* __restrict__ dp, float* __restrict__ dx,
                                                       int64_t lddx, int32_t n, int32_t f, int mode,
                                                       const int32_t* __restrict__ argmax, const float* __restrict__ y,
                                                       int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  int lo = 0, hi = b;  // find g with gp[g] <= r < gp[g+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gp[mid] <= r) lo = mid; else hi = mid;
  }
  const int g = lo;
  const float scale = (mode == GCNX_POOL_AVG) ? 1.0f / (float)(gp[g + 1] - gp[g]) : 1.0f;
  for (int c = lane; c < f; c += 64) {
    float v = dp[(int64_t)g * f + c] * scale;
    if (mode == GCNX_POOL_MAX && argmax[(int64_t)g * f + c] != r) v = 0.f;
    if (y && !(y[(int64_t)r * ldy + c] > 0.f)) v = 0.f;
    dx[(int64_t)r * lddx + c] = v;
  }
}

// Softmax + CCE + accuracy over B graphs, one workgroup, deterministic tree reduction.
__global__ __launch_bounds__(256) void softmax_cce_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ y, int32_t b, int32_t c,
                                                          float denom, float* __restrict__ probs,
                                                          float* __restrict__ loss_acc, float* __restrict__ dlogits) {
  __shared__ float s_loss[256];
  __shared__ float s_hit[256];
  float loss = 0.f, hit = 0.f;
  for (int g = threadIdx.x; g < b; g += 256) {
    const float* z = logits + (int64_t)g * c;
    const float* yy = y + (int64_t)g * c;
    float m = -INFINITY;
    for (int k = 0; k < c; ++k) m = fmaxf(m, z[k]);
    float sum = 0.f;
    for (int k = 0; k < c; ++k) sum += expf(z[k] - m);
    float ymsum = 0.f, pmax = -1.f, ymax = -INFINITY, l = 0.f;
    int pa = 0, ya = 0;
    for (int k = 0; k < c; ++k) {
      const float p = expf(z[k] - m) / sum;
      probs[(int64_t)g * c + k] = p;
      if (p > 1e-7f && p < 1.0f - 1e-7f) ymsum += yy[k];  // clip_by_value passes no gradient outside
      if (p > pmax) { pmax = p; pa = k; }
      if (yy[k] > ymax) { ymax = yy[k]; ya = k; }
      const float pc = fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f);
      l -= yy[k] * logf(pc);
    }
    if (dlogits)
      for (int k = 0; k < c; ++k) {
        const float p = expf(z[k] - m) / sum;
        const float ym = (p > 1e-7f && p < 1.0f - 1e-7f) ? yy[k] : 0.f;
        dlogits[(int64_t)g * c + k] = (p * ymsum - ym) / denom;
      }
    loss += l;
    hit += (pa == ya) ? 1.f : 0.f;
  }
  s_loss[threadIdx.x] = loss;
  s_hit[threadIdx.x] = hit;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      s_loss[threadIdx.x] += s_loss[threadIdx.x + off];
      s_hit[threadIdx.x] += s_hit[threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss_acc[0] += s_loss[0] / denom;
    loss_acc[1] += s_hit[0];
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, int64_t n,
                                                  float lr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = p[i] - lr * g[i];
}

int colsum_impl(gcnx_ctx* ctx, const float* x, int64_t ldx, int64_t n, int32_t f, float* out, const float* y,
                int64_t ldy, float* dz, int64_t lddz, int act, const float* alpha, float* out_alpha) {
  const bool fuse = (dz != nullptr);
  const int nchunks = gcnx_cdiv(n, kColsumRows);
  const bool want_alpha = fuse && act == GCNX_ACT_PRELU && out_alpha;
  const size_t need = (size_t)nchunks * f * sizeof(float) * (want_alpha ? 2 : 1);
  float* part = nullptr;
  float* part_a = nullptr;
  if (out || want_alpha) {
    if (nchunks > 1) {
      int rc = gcnx_ws_reserve(ctx, need);
      if (rc) return rc;
      part = out ? (float*)ctx->ws : nullptr;
      part_a = want_alpha ? (float*)ctx->ws + (size_t)nchunks * f : nullptr;
    } else {
      part = out;
      part_a = want_alpha ? out_alpha : nullptr;
    }
  }
  dim3 grid(gcnx_cdiv(f, 64), nchunks);
  if (fuse)
    hipLaunchKernelGGL((colsum_kernel<true>), grid, dim3(256), 0, ctx->stream, x, ldx, n, f, (int64_t)kColsumRows,
                       part, y, ldy, dz, lddz, act, alpha, part_a);
  else
    hipLaunchKernelGGL((colsum_kernel<false>), grid, dim3(256), 0, ctx->stream, x, ldx, n, f, (int64_t)kColsumRows,
                       part, nullptr, (int64_t)0, nullptr, (int64_t)0, 0, nullptr, nullptr);
  GCNX_LAUNCH_OK(ctx);
  if (nchunks > 1) {
    dim3 g2(gcnx_cdiv(f, 64), 1);
    if (out) {
      hipLaunchKernelGGL((colsum_kernel<false>), g2, dim3(256), 0, ctx->stream, (const float*)part, (int64_t)f,
                         (int64_t)nchunks, f, (int64_t)nchunks, out, nullptr, (int64_t)0, nullptr, (int64_t)0, 0,
                         nullptr, nullptr);
      GCNX_LAUNCH_OK(ctx);
    }
    if (want_alpha) {
      hipLaunchKernelGGL((colsum_kernel<false>), g2, dim3(256), 0, ctx->stream, (const float*)part_a, (int64_t)f,
                         (int64_t)nchunks, f, (int64_t)nchunks, out_alpha, nullptr, (int64_t)0, nullptr, (int64_t)0,
                         0, nullptr, nullptr);
      GCNX_LAUNCH_OK(ctx);
    }
  }
  return GCNX_OK;
}

}  // namespace

int gcnx_colsum(gcnx_ctx* ctx, const float* x, int64_t ldx, int64_t n, int32_t f, float* out) {
  return colsum_impl(ctx, x, ldx, n, f, out, nullptr, 0, nullptr, 0, 0, nullptr, nullptr);
}

extern "C" {

int gcnx_act_bias_grad(gcnx_ctx* ctx, const float* dy, int64_t lddy, const float* y, int64_t ldy, float* dz,
                       int64_t lddz, int64_t n, int32_t f, int act, const float* alpha, float* db, float* dalpha) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_act_bias_grad: negative size");
  GCNX_REQUIRE(ctx, act >= GCNX_ACT_NONE && act <= GCNX_ACT_PRELU, "gcnx_act_bias_grad: unknown activation %d", act);
  if (f == 0) return GCNX_OK;
  if (n == 0) {
    if (db) GCNX_HIP(ctx, hipMemsetAsync(db, 0, (size_t)f * 4, ctx->stream));
    if (dalpha) GCNX_HIP(ctx, hipMemsetAsync(dalpha, 0, (size_t)f * 4, ctx->stream));
    return GCNX_OK;
  }
  GCNX_REQUIRE(ctx, dy && dz, "gcnx_act_bias_grad: NULL pointer");
  GCNX_REQUIRE(ctx, act == GCNX_ACT_NONE || y, "gcnx_act_bias_grad: y needed for activation gradient");
  GCNX_REQUIRE(ctx, act != GCNX_ACT_PRELU || alpha, "gcnx_act_bias_grad: alpha needed for PReLU");
  GCNX_REQUIRE(ctx, lddy >= f && lddz >= f && (!y || ldy >= f), "gcnx_act_bias_grad: leading dimension too small");
  return colsum_impl(ctx, dy, lddy, n, f, db, y ? y : dy, y ? ldy : lddy, dz, lddz, act, alpha, dalpha);
}

int gcnx_segment_pool(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* x, int64_t ldx, float* pooled, int32_t b,
                      int32_t f, int mode, int32_t* argmax) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, b >= 0 && f >= 0, "gcnx_segment_pool: negative size");
  GCNX_REQUIRE(ctx, mode >= GCNX_POOL_SUM && mode <= GCNX_POOL_MAX, "gcnx_segment_pool: unknown mode %d", mode);
  if (b == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, graph_ptr && x && pooled, "gcnx_segment_pool: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= f, "gcnx_segment_pool: leading dimension too small");
  GCNX_REQUIRE(ctx, mode != GCNX_POOL_MAX || argmax, "gcnx_segment_pool: MAX needs an argmax buffer");
  dim3 grid(gcnx_cdiv(f, 64), b);
  hipLaunchKernelGGL(pool_fwd_kernel, grid, dim3(256), 0, ctx->stream, graph_ptr, x, ldx, pooled, f, mode, argmax);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_segment_pool_bwd(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* dpooled, float* dx, int64_t lddx,
                          int32_t n, int32_t b, int32_t f, int mode, const int32_t* argmax, const float* y,
                          int64_t ldy, float* db) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && b >= 0 && f >= 0, "gcnx_segment_pool_bwd: negative size");
  GCNX_REQUIRE(ctx, mode >= GCNX_POOL_SUM && mode <= GCNX_POOL_MAX, "gcnx_segment_pool_bwd: unknown mode %d", mode);
  if (f == 0) return GCNX_OK;
  if (n == 0 || b == 0) {
    if (db) GCNX_HIP(ctx, hipMemsetAsync(db, 0, (size_t)f * 4, ctx->stream));
    return GCNX_OK;
  }
  GCNX_REQUIRE(ctx, graph_ptr && dpooled && dx, "gcnx_segment_pool_bwd: NULL pointer");
  GCNX_REQUIRE(ctx, lddx >= f && (!y || ldy >= f), "gcnx_segment_pool_bwd: leading dimension too small");
  GCNX_REQUIRE(ctx, mode != GCNX_POOL_MAX || argmax, "gcnx_segment_pool_bwd: MAX needs the argmax buffer");
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(gcnx_cdiv(n, 4)), dim3(256), 0, ctx->stream, graph_ptr, b, dpooled, dx,
                     lddx, n, f, mode, argmax, y, ldy);
  GCNX_LAUNCH_OK(ctx);
  if (db) return gcnx_colsum(ctx, dx, lddx, n, f, db);
  return GCNX_OK;
}

int gcnx_softmax_cce(gcnx_ctx* ctx, const float* logits, const float* y, int32_t b, int32_t c, float denom,
                     float* probs, float* loss_acc, float* dlogits) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, b >= 0 && c >= 1, "gcnx_softmax_cce: bad size b=%d c=%d", b, c);
  if (b == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, logits && y && probs && loss_acc, "gcnx_softmax_cce: NULL pointer");
  GCNX_REQUIRE(ctx, denom > 0.f, "gcnx_softmax_cce: denom must be positive");
  hipLaunchKernelGGL(softmax_cce_kernel, dim3(1), dim3(256), 0, ctx->stream, logits, y, b, c, denom, probs, loss_acc,
                     dlogits);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_sgd(gcnx_ctx* ctx, float* params, const float* grads, int64_t n, float lr) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0, "gcnx_sgd: negative size");
  if (n == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, params && grads, "gcnx_sgd: NULL pointer");
  int grid = gcnx_cdiv(n, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, ctx->stream, params, grads, n, lr);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // extern "C"
