// Element-wise pieces of Spektral's GeneralGNN options beside gcn.py:320's defaults (r3):
//   * Keras Dropout(rate) -- MLP and GeneralConv are Dense -> BatchNormalization -> Dropout -> activation; with the
//     positively homogeneous activations built here (PReLU, ReLU, linear) act(s u) = s act(u) for the keep / scale factor
//     s in {0, 1 / (1 - rate)}, so the layer is the fused batch-norm + activation pass followed by one multiply, and its
//     backward the same multiply on the incoming gradient;
//   * connectivity = "sum": out = z + out;
//   * aggregate = "max" | "min" and their gradient (ties share it, as TensorFlow's unsorted_segment_max gradient does).
// The keep decision of element i of (layer stream, step) is a stateless hash, so the backward pass regenerates the mask
// the forward pass used instead of storing it, and a captured step reads the step number from device memory (one
// captured graph serves every step).  TensorFlow's generator is not reproduced (PARITY UNPINNED: tests hold the
// arithmetic against the oracle fed with THIS mask, and the keep frequency against 1 - rate).
#include <cstdint>

#include "common.h"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t h) {           // murmur3's finaliser
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

// uniform 32 bits for element `idx` of stream (seed, stream_id, step)
__device__ __forceinline__ uint32_t keep_bits(uint32_t k0, uint32_t k1, uint64_t idx) {
  const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  return mix32(mix32(lo ^ k0) + (hi * 0x9E3779B9u ^ k1));
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int32_t f, uint32_t thresh,
                                                      float scale, uint32_t seed, uint32_t stream_id, const uint32_t* __restrict__ step,
                                                      float* __restrict__ out, int64_t ldo) {
  const uint32_t st = step ? *step : 0u;
  const uint32_t k0 = mix32(seed * 0x9E3779B9u + stream_id), k1 = mix32(st * 0x85EBCA6Bu + (seed ^ 0x27D4EB2Fu));
  const int64_t total = n * (int64_t)f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / f;
    const int c = (int)(i - r * f);
    const bool keep = keep_bits(k0, k1, (uint64_t)i) >= thresh;
    out[r * ldo + c] = keep ? x[r * ldx + c] * scale : 0.f;
  }
}

__global__ void counter_add_kernel(uint32_t* c, uint32_t inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                  float* __restrict__ out, int64_t ldo, int64_t n, int32_t f) {
  const int64_t total = n * (int64_t)f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / f;
    const int c = (int)(i - r * f);
    out[r * ldo + c] = a[r * lda + c] + b[r * ldb + c];
  }
}

// GeneralConv(aggregate = "max" | "min") (tf.math.unsorted_segment_max / _min over a row's messages): one wave
// per output row, lanes over the columns, the row's entries walked in CSR order.  A row without entries gets the
// reduction's identity (lowest / largest float, as TensorFlow does).  `cnt` receives the number of entries that attain the
// extremum (ties), which the gradient divides by.
template <bool MIN>
__global__ __launch_bounds__(256) void spmm_minmax_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                          const float* __restrict__ h, int64_t ldh, float* __restrict__ out, int64_t ldo,
                                                          float* __restrict__ cnt, int64_t ldc, int32_t n, int32_t f) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int e0 = rowptr[row], e1 = rowptr[row + 1];
  for (int c = lane; c < f; c += 64) {
    float m = MIN ? 3.402823466e+38f : -3.402823466e+38f, k = 0.f;
    for (int e = e0; e < e1; ++e) {
      const float v = h[(int64_t)colidx[e] * ldh + c];
      if (MIN ? v < m : v > m) { m = v; k = 1.f; }
      else if (v == m) k += 1.f;
    }
    out[row * ldo + c] = m;
    if (cnt) cnt[row * ldc + c] = k;
  }
}

// Its gradient (TensorFlow's _UnsortedSegmentMinOrMaxGrad): every message equal to the row's extremum receives
// dy / (number of such messages).  Walked from the SOURCE side over the transposed operator: dh[s] = sum over the targets t
// of row s of A^T of [h[s] == out[t]] * dy[t] / cnt[t] -- a gather in CSR order, deterministic.
__global__ __launch_bounds__(256) void spmm_minmax_bwd_kernel(const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ colidx_t,
                                                              const float* __restrict__ h, int64_t ldh, const float* __restrict__ out,
                                                              int64_t ldo, const float* __restrict__ cnt, int64_t ldc,
                                                              const float* __restrict__ dy, int64_t lddy, float* __restrict__ dh,
                                                              int64_t lddh, int32_t n, int32_t f) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int e0 = rowptr_t[row], e1 = rowptr_t[row + 1];
  for (int c = lane; c < f; c += 64) {
    const float hv = h[row * ldh + c];
    float acc = 0.f;
    for (int e = e0; e < e1; ++e) {
      const int64_t t = colidx_t[e];
      if (out[t * ldo + c] == hv) acc += dy[t * lddy + c] / cnt[t * ldc + c];
    }
    dh[row * lddh + c] = acc;
  }
}

// GeneralConv(aggregate = "prod") (tf.math.unsorted_segment_prod over a row's messages): one wave per output row, lanes over the
// columns, entries in CSR order; a row without entries gets 1 (TensorFlow's value for an empty segment).  aux (needed for the
// gradient): the product of the row's NON-ZERO messages where exactly one message is zero (then out = 0 and that message's
// derivative is this product), out itself where none is, 0 where two or more are (every derivative vanishes).
__global__ __launch_bounds__(256) void spmm_prod_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                        const float* __restrict__ h, int64_t ldh, float* __restrict__ out, int64_t ldo,
                                                        float* __restrict__ aux, int64_t lda, int32_t n, int32_t f) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int e0 = rowptr[row], e1 = rowptr[row + 1];
  for (int c = lane; c < f; c += 64) {
    float p = 1.f, pnz = 1.f;
    int nz = 0;
    for (int e = e0; e < e1; ++e) {
      const float v = h[(int64_t)colidx[e] * ldh + c];
      p *= v;
      if (v == 0.f) ++nz; else pnz *= v;
    }
    out[row * ldo + c] = p;
    if (aux) aux[row * lda + c] = nz >= 2 ? 0.f : pnz;
  }
}

// Its gradient (TensorFlow's _UnsortedSegmentProdGrad): d out[t] / d h[s] = out[t] / h[s] for a non-zero message (0 when another
// message of the row is zero: out[t] is), the product of the other messages for THE zero message of a row (aux), 0 where a row holds
// two or more zeros.  Walked from the source side over the transposed operator, CSR order: deterministic.
__global__ __launch_bounds__(256) void spmm_prod_bwd_kernel(const int32_t* __restrict__ rowptr_t, const int32_t* __restrict__ colidx_t,
                                                            const float* __restrict__ h, int64_t ldh, const float* __restrict__ out,
                                                            int64_t ldo, const float* __restrict__ aux, int64_t lda,
                                                            const float* __restrict__ dy, int64_t lddy, float* __restrict__ dh,
                                                            int64_t lddh, int32_t n, int32_t f) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int e0 = rowptr_t[row], e1 = rowptr_t[row + 1];
  for (int c = lane; c < f; c += 64) {
    const float hv = h[row * ldh + c];
    float acc = 0.f;
    for (int e = e0; e < e1; ++e) {
      const int64_t t = colidx_t[e];
      const float part = hv == 0.f ? aux[t * lda + c] : out[t * ldo + c] / hv;
      acc += dy[t * lddy + c] * part;
    }
    dh[row * lddh + c] = acc;
  }
}

int grid_for(gcnx_ctx* ctx, int64_t total) {
  int64_t g = (total + 255) / 256;
  const int64_t cap = 16LL * ctx->num_cus;
  if (g > cap) g = cap;
  return (int)(g > 0 ? g : 1);
}

}  // namespace

extern "C" {

int gcnx_dropout(gcnx_ctx* ctx, const float* x, int64_t ldx, int64_t n, int32_t f, float rate, uint32_t seed, uint32_t stream_id,
                 const uint32_t* step, float* out, int64_t ldo) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "dropout");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_dropout: negative size");
  GCNX_REQUIRE(ctx, rate >= 0.f && rate < 1.f, "gcnx_dropout: rate %g outside [0, 1)", (double)rate);
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, x && out, "gcnx_dropout: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= f && ldo >= f, "gcnx_dropout: leading dimension too small");
  // keep iff bits >= thresh, thresh = rate * 2^32 (rate 0 keeps everything and scales by 1)
  const double t = (double)rate * 4294967296.0;
  const uint32_t thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(ctx, n * (int64_t)f)), dim3(256), 0, ctx->stream, x, ldx, n, f, thresh,
                     1.0f / (1.0f - rate), seed, stream_id, step, out, ldo);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_counter_add(gcnx_ctx* ctx, uint32_t* counter, uint32_t inc) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, counter != nullptr, "gcnx_counter_add: NULL pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, ctx->stream, counter, inc);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_add(gcnx_ctx* ctx, const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int64_t n, int32_t f) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "skip connection (sum)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_add: negative size");
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, a && b && out, "gcnx_add: NULL pointer");
  GCNX_REQUIRE(ctx, lda >= f && ldb >= f && ldo >= f, "gcnx_add: leading dimension too small");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(ctx, n * (int64_t)f)), dim3(256), 0, ctx->stream, a, lda, b, ldb, out, ldo, n, f);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_spmm_csr_minmax(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* h, int64_t ldh, float* out,
                         int64_t ldo, float* cnt, int64_t ldc, int32_t n, int32_t f, int is_min) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "aggregation (max / min)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_spmm_csr_minmax: negative size");
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, rowptr && colidx && h && out, "gcnx_spmm_csr_minmax: NULL pointer");
  GCNX_REQUIRE(ctx, ldh >= f && ldo >= f && (!cnt || ldc >= f), "gcnx_spmm_csr_minmax: leading dimension too small");
  GCNX_REQUIRE(ctx, h != out, "gcnx_spmm_csr_minmax: in-place aggregation is not supported");
  const dim3 grid(gcnx_cdiv(n, 4));
  if (is_min) hipLaunchKernelGGL((spmm_minmax_kernel<true>), grid, dim3(256), 0, ctx->stream, rowptr, colidx, h, ldh, out, ldo, cnt, ldc, n, f);
  else hipLaunchKernelGGL((spmm_minmax_kernel<false>), grid, dim3(256), 0, ctx->stream, rowptr, colidx, h, ldh, out, ldo, cnt, ldc, n, f);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_spmm_csr_minmax_bwd(gcnx_ctx* ctx, const int32_t* rowptr_t, const int32_t* colidx_t, const float* h, int64_t ldh,
                             const float* out, int64_t ldo, const float* cnt, int64_t ldc, const float* dy, int64_t lddy, float* dh,
                             int64_t lddh, int32_t n, int32_t f) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "aggregation backward (max / min)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_spmm_csr_minmax_bwd: negative size");
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, rowptr_t && colidx_t && h && out && cnt && dy && dh, "gcnx_spmm_csr_minmax_bwd: NULL pointer");
  GCNX_REQUIRE(ctx, ldh >= f && ldo >= f && ldc >= f && lddy >= f && lddh >= f, "gcnx_spmm_csr_minmax_bwd: leading dimension too small");
  GCNX_REQUIRE(ctx, dh != dy && dh != h, "gcnx_spmm_csr_minmax_bwd: in-place is not supported");
  hipLaunchKernelGGL(spmm_minmax_bwd_kernel, dim3(gcnx_cdiv(n, 4)), dim3(256), 0, ctx->stream, rowptr_t, colidx_t, h, ldh, out, ldo, cnt,
                     ldc, dy, lddy, dh, lddh, n, f);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_spmm_csr_prod(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* h, int64_t ldh, float* out,
                       int64_t ldo, float* aux, int64_t lda, int32_t n, int32_t f) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "aggregation (prod)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_spmm_csr_prod: negative size");
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, rowptr && colidx && h && out, "gcnx_spmm_csr_prod: NULL pointer");
  GCNX_REQUIRE(ctx, ldh >= f && ldo >= f && (!aux || lda >= f), "gcnx_spmm_csr_prod: leading dimension too small");
  GCNX_REQUIRE(ctx, h != out, "gcnx_spmm_csr_prod: in-place aggregation is not supported");
  hipLaunchKernelGGL(spmm_prod_kernel, dim3(gcnx_cdiv(n, 4)), dim3(256), 0, ctx->stream, rowptr, colidx, h, ldh, out, ldo, aux, lda, n, f);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_spmm_csr_prod_bwd(gcnx_ctx* ctx, const int32_t* rowptr_t, const int32_t* colidx_t, const float* h, int64_t ldh,
                           const float* out, int64_t ldo, const float* aux, int64_t lda, const float* dy, int64_t lddy, float* dh,
                           int64_t lddh, int32_t n, int32_t f) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "aggregation backward (prod)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_spmm_csr_prod_bwd: negative size");
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, rowptr_t && colidx_t && h && out && aux && dy && dh, "gcnx_spmm_csr_prod_bwd: NULL pointer");
  GCNX_REQUIRE(ctx, ldh >= f && ldo >= f && lda >= f && lddy >= f && lddh >= f, "gcnx_spmm_csr_prod_bwd: leading dimension too small");
  GCNX_REQUIRE(ctx, dh != dy && dh != h, "gcnx_spmm_csr_prod_bwd: in-place is not supported");
  hipLaunchKernelGGL(spmm_prod_bwd_kernel, dim3(gcnx_cdiv(n, 4)), dim3(256), 0, ctx->stream, rowptr_t, colidx_t, h, ldh, out, ldo, aux,
                     lda, dy, lddy, dh, lddh, n, f);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // extern "C"
