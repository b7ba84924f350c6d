// Graph preparation: DisjointLoader COO -> CSR, gcn_filter normalisation, CSR transpose.
// These run once per batch (or once per dataset), not once per layer.
#include <new>
#include <vector>

#include "common.h"

namespace {

// rows[] is non-decreasing (tf.sparse.reorder order).  Thread e owns the rowptr entries of the
// rows that START at e: every r in (rows[e-1], rows[e]].  No atomics, no scan.
__global__ __launch_bounds__(256) void coo_to_csr_kernel(const int64_t* __restrict__ rows,
                                                         const int64_t* __restrict__ cols, int64_t nnz,
                                                         int64_t n, int32_t* __restrict__ rowptr,
                                                         int32_t* __restrict__ colidx, int* __restrict__ flag) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e > nnz) return;
  if (e == nnz) {  // tail: rows after the last stored row, and rowptr[n]
    const int64_t last = nnz > 0 ? rows[nnz - 1] : -1;
    if (last >= -1 && last < n)
      for (int64_t r = last + 1; r <= n; ++r) rowptr[r] = (int32_t)nnz;
    return;
  }
  const int64_t r = rows[e], c = cols[e];
  const int64_t rp = e > 0 ? rows[e - 1] : -1;
  if (r < 0 || r >= n || c < 0 || c >= n || rp > r) {
    atomicOr(flag, 1);
    return;
  }
  colidx[e] = (int32_t)c;
  for (int64_t q = (rp < -1 ? -1 : rp) + 1; q <= r; ++q) rowptr[q] = (int32_t)e;
}

// One thread per row: degree of A~ and presence of the stored diagonal.
__global__ __launch_bounds__(256) void gcn_degree_kernel(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ colidx,
                                                         const float* __restrict__ vals, int32_t n, int mode,
                                                         float* __restrict__ dinv, int* __restrict__ flag) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int a = rowptr[r], b = rowptr[r + 1];
  float deg = 0.f;
  bool has_diag = false;
  for (int e = a; e < b; ++e) {
    const float v = vals ? vals[e] : 1.0f;
    deg += v;
    has_diag |= (colidx[e] == r);
  }
  if (!has_diag) atomicOr(flag, 1);
  if (mode == GCNX_NORM_SPEKTRAL) deg += 1.0f;
  dinv[r] = deg > 0.f ? 1.0f / sqrtf(deg) : 0.f;
}

__global__ __launch_bounds__(256) void gcn_scale_kernel(const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colidx,
                                                        const float* __restrict__ vals, int32_t n, int mode,
                                                        const float* __restrict__ dinv,
                                                        float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int a = rowptr[r], b = rowptr[r + 1];
  const float dr = dinv[r];
  for (int e = a; e < b; ++e) {
    const int c = colidx[e];
    float v = vals ? vals[e] : 1.0f;
    if (mode == GCNX_NORM_SPEKTRAL && c == r) v += 1.0f;
    out[e] = v * dr * dinv[c];
  }
}


// Structure check of a batch adjacency, one wave per row: (1) is A == A^T (pattern; values to 4 ulp -- gcn_filter's
// v * d_r * d_c is rounded in a different order on either side of the diagonal), which lets the backward pass
// reuse this CSR for A^T; (2) does every entry stay inside its row's graph_ptr block, which the tile plan and the
// folded pool backward assume.  Failures clear bits of *props (device int, preset to all checks passed).
__global__ __launch_bounds__(256) void csr_inspect_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                          const float* __restrict__ vals, int32_t n,
                                                          const int32_t* __restrict__ gp, int32_t b, int* __restrict__ props) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= n) return;
  const int a = rowptr[r], e1 = rowptr[r + 1];
  int lo = 0, hi = n;
  if (gp) {                                     // the block of row r: largest g with gp[g] <= r (uniform per wave)
    int g0 = 0, g1 = b;                         // invariant: gp[g0] <= r < gp[g1]  (gp[0] == 0, gp[b] == n checked by the host)
    while (g1 - g0 > 1) { const int m = (g0 + g1) >> 1; if (gp[m] <= r) g0 = m; else g1 = m; }
    lo = gp[g0]; hi = gp[g1];
  }
  bool sym = true, blk = true;
  for (int e = a + lane; e < e1; e += 64) {
    const int c = colidx[e];
    if (c < 0 || c >= n) { sym = false; blk = false; continue; }
    if (c < lo || c >= hi) blk = false;
    if (c == r) continue;
    const float v = vals ? vals[e] : 1.0f;
    bool found = false;
    for (int q = rowptr[c]; q < rowptr[c + 1] && !found; ++q)
      if (colidx[q] == r) {
        const float w = vals ? vals[q] : 1.0f;
        found = fabsf(w - v) <= 4.8e-7f * fmaxf(fabsf(w), fabsf(v));
      }
    sym &= found;
  }
  if (!sym) atomicAnd(props, ~GCNX_CSR_SYMMETRIC);
  if (!blk) atomicAnd(props, ~GCNX_CSR_BLOCK_DIAGONAL);
}

__global__ void graph_ptr_check_kernel(const int32_t* __restrict__ gp, int32_t b, int32_t n, int* __restrict__ props) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g > b) return;
  bool ok = true;
  if (g == 0) ok = gp[0] == 0;
  if (g == b) ok = ok && gp[b] == n;
  if (g < b) ok = ok && gp[g] <= gp[g + 1];
  if (!ok) atomicAnd(props, ~(GCNX_CSR_BLOCK_DIAGONAL | GCNX_CSR_GRAPH_PTR_OK));
}

// ---- device-side collate -------------------------------------------------------------
// The dataset lives in HBM as ONE disjoint union of all its graphs (features, CSR, labels, node_ptr).  A
// batch = the graphs sel[0..b) in that order: their feature rows, CSR rows (row pointers and column indices
// re-based to the batch) and labels are gathered by one launch -- what DisjointLoader's collate does on the
// host with vstack / block_diag / find (gcn.py:316-317, 367).  Per batch only 3(b+1) ints cross PCIe.
__global__ __launch_bounds__(256) void collate_kernel(const int32_t* __restrict__ desc /* sel | bnode | bent, each b+1 */,
                                                      int32_t b, const int32_t* __restrict__ node_ptr,
                                                      const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                      const float* __restrict__ vals, const float* __restrict__ x,
                                                      int64_t ldx, int32_t f, const float* __restrict__ y, int32_t c,
                                                      int32_t* __restrict__ o_rowptr, int32_t* __restrict__ o_colidx,
                                                      float* __restrict__ o_vals, float* __restrict__ o_x, int64_t ldo,
                                                      float* __restrict__ o_y, int32_t* __restrict__ o_gp,
                                                      int32_t* __restrict__ o_ids) {
  const int g = blockIdx.y;                       // position in the batch
  const int src = desc[g];
  const int bn = desc[(b + 1) + g], be = desc[2 * (b + 1) + g];
  const int n0 = node_ptr[src], ng = node_ptr[src + 1] - n0;
  const int e0 = rowptr[n0], ne = rowptr[n0 + ng] - e0;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  // CSR rows: row pointers re-based to the batch's entry offset
  for (int i = tid; i < ng; i += nth) {
    o_rowptr[bn + i] = rowptr[n0 + i] - e0 + be;
    if (o_ids) o_ids[bn + i] = g;                 // DisjointLoader's id vector i
  }
  // entries: column indices re-based to the batch's node offset
  for (int e = tid; e < ne; e += nth) {
    o_colidx[be + e] = colidx[e0 + e] - n0 + bn;
    if (o_vals) o_vals[be + e] = vals[e0 + e];
  }
  // feature rows (f % 4 == 0 and 16-byte aligned rows are the caller's guarantee when vec4 != 0 ... handled scalar otherwise)
  const int64_t total = (int64_t)ng * f;
  for (int64_t k = tid; k < total; k += nth) {
    const int64_t i = k / f, j = k - i * f;
    o_x[(int64_t)(bn + i) * ldo + j] = x[(int64_t)(n0 + i) * ldx + j];
  }
  if (blockIdx.x == 0) {
    if (y) for (int k = threadIdx.x; k < c; k += blockDim.x) o_y[(int64_t)g * c + k] = y[(int64_t)src * c + k];
    if (threadIdx.x == 0) {
      o_gp[g] = bn;
      if (g == b - 1) { o_gp[b] = desc[(b + 1) + b]; o_rowptr[desc[(b + 1) + b]] = desc[2 * (b + 1) + b]; }
    }
  }
}

}  // namespace

extern "C" {

int gcnx_coo_to_csr(gcnx_ctx* ctx, const int64_t* rows, const int64_t* cols, int64_t nnz, int64_t n,
                    int32_t* rowptr, int32_t* colidx) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && nnz >= 0, "gcnx_coo_to_csr: negative size");
  GCNX_REQUIRE(ctx, n < 2147483647LL && nnz < 2147483647LL,
               "gcnx_coo_to_csr: N=%lld nnz=%lld do not fit the int32 CSR", (long long)n, (long long)nnz);
  GCNX_REQUIRE(ctx, rowptr != nullptr, "gcnx_coo_to_csr: rowptr is NULL");
  GCNX_REQUIRE(ctx, nnz == 0 || (rows && cols && colidx), "gcnx_coo_to_csr: NULL index array");
  GCNX_REQUIRE(ctx, !ctx->capturing, "gcnx_coo_to_csr synchronises and cannot be captured");
  GCNX_HIP(ctx, hipMemsetAsync(ctx->flag, 0, sizeof(int), ctx->stream));
  const int grid = gcnx_cdiv(nnz + 1, 256);
  hipLaunchKernelGGL(coo_to_csr_kernel, dim3(grid), dim3(256), 0, ctx->stream, rows, cols, nnz, n, rowptr,
                     colidx, ctx->flag);
  GCNX_LAUNCH_OK(ctx);
  int h = 0;
  GCNX_HIP(ctx, hipMemcpyAsync(&h, ctx->flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h) return gcnx_fail(ctx, GCNX_ERR_DATA, "gcnx_coo_to_csr: indices out of [0,%lld) or rows not sorted row-major", (long long)n);
  return GCNX_OK;
}

int gcnx_gcn_norm(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals_in,
                  int32_t n, int mode, float* vals_out) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0, "gcnx_gcn_norm: negative n");
  GCNX_REQUIRE(ctx, mode == GCNX_NORM_SPEKTRAL || mode == GCNX_NORM_PYG, "gcnx_gcn_norm: unknown mode %d", mode);
  GCNX_REQUIRE(ctx, !ctx->capturing, "gcnx_gcn_norm synchronises and cannot be captured");
  if (n == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, rowptr && colidx && vals_out, "gcnx_gcn_norm: NULL pointer");
  int rc = gcnx_ws_reserve(ctx, (size_t)n * sizeof(float));
  if (rc) return rc;
  float* dinv = (float*)ctx->ws;
  GCNX_HIP(ctx, hipMemsetAsync(ctx->flag, 0, sizeof(int), ctx->stream));
  const int grid = gcnx_cdiv(n, 256);
  hipLaunchKernelGGL(gcn_degree_kernel, dim3(grid), dim3(256), 0, ctx->stream, rowptr, colidx, vals_in, n, mode,
                     dinv, ctx->flag);
  GCNX_LAUNCH_OK(ctx);
  hipLaunchKernelGGL(gcn_scale_kernel, dim3(grid), dim3(256), 0, ctx->stream, rowptr, colidx, vals_in, n, mode,
                     dinv, vals_out);
  GCNX_LAUNCH_OK(ctx);
  int h = 0;
  GCNX_HIP(ctx, hipMemcpyAsync(&h, ctx->flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (h) return gcnx_fail(ctx, GCNX_ERR_DATA, "gcnx_gcn_norm: a row stores no diagonal entry (add self-loops on the host: GCNConv.preprocess)");
  return GCNX_OK;
}

int gcnx_csr_inspect(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, int32_t n,
                     const int32_t* graph_ptr, int32_t b, int* props) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, props != nullptr, "gcnx_csr_inspect: props is NULL");
  *props = 0;
  GCNX_REQUIRE(ctx, n >= 0 && b >= 0, "gcnx_csr_inspect: negative size");
  GCNX_REQUIRE(ctx, !ctx->capturing, "gcnx_csr_inspect synchronises and cannot be captured");
  GCNX_REQUIRE(ctx, n == 0 || (rowptr && colidx), "gcnx_csr_inspect: NULL pointer");
  int h = GCNX_CSR_SYMMETRIC | (graph_ptr ? (GCNX_CSR_BLOCK_DIAGONAL | GCNX_CSR_GRAPH_PTR_OK) : 0);
  GCNX_HIP(ctx, hipMemcpyAsync(ctx->flag + 1, &h, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));        // (h is a stack variable)
  if (graph_ptr) {
    hipLaunchKernelGGL(graph_ptr_check_kernel, dim3(gcnx_cdiv(b + 1, 256)), dim3(256), 0, ctx->stream, graph_ptr, b, n, ctx->flag + 1);
    GCNX_LAUNCH_OK(ctx);
    // the row -> block search below walks graph_ptr: only if it is well formed
    GCNX_HIP(ctx, hipMemcpyAsync(&h, ctx->flag + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!(h & GCNX_CSR_GRAPH_PTR_OK)) graph_ptr = nullptr;
  }
  if (n > 0) {
    hipLaunchKernelGGL(csr_inspect_kernel, dim3(gcnx_cdiv(n, 4)), dim3(256), 0, ctx->stream, rowptr, colidx, vals, n, graph_ptr, b,
                       ctx->flag + 1);
    GCNX_LAUNCH_OK(ctx);
  }
  GCNX_HIP(ctx, hipMemcpyAsync(&h, ctx->flag + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *props = h;
  return GCNX_OK;
}

// Prep step used only when A^ is not symmetric: stable counting sort on the host (bitwise
// reproducible entry order), staged through pageable host memory.
int gcnx_csr_transpose(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                       int32_t n, int32_t nnz, int32_t* rowptr_t, int32_t* colidx_t, float* vals_t) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && nnz >= 0, "gcnx_csr_transpose: negative size");
  GCNX_REQUIRE(ctx, rowptr && rowptr_t, "gcnx_csr_transpose: NULL rowptr");
  GCNX_REQUIRE(ctx, nnz == 0 || (colidx && colidx_t), "gcnx_csr_transpose: NULL colidx");
  GCNX_REQUIRE(ctx, (vals == nullptr) == (vals_t == nullptr), "gcnx_csr_transpose: vals and vals_t must both be given or both NULL");
  GCNX_REQUIRE(ctx, !ctx->capturing, "gcnx_csr_transpose synchronises and cannot be captured");
  try {
    std::vector<int32_t> rp((size_t)n + 1), ci((size_t)nnz), rpt((size_t)n + 1, 0), cit((size_t)nnz);
    std::vector<float> v, vt;
    GCNX_HIP(ctx, hipMemcpyAsync(rp.data(), rowptr, rp.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (nnz) GCNX_HIP(ctx, hipMemcpyAsync(ci.data(), colidx, ci.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (vals && nnz) {
      v.resize((size_t)nnz);
      vt.resize((size_t)nnz);
      GCNX_HIP(ctx, hipMemcpyAsync(v.data(), vals, v.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (rp[n] != nnz) return gcnx_fail(ctx, GCNX_ERR_DATA, "gcnx_csr_transpose: rowptr[n]=%d != nnz=%d", rp[n], nnz);
    for (int32_t e = 0; e < nnz; ++e) {
      if (ci[e] < 0 || ci[e] >= n) return gcnx_fail(ctx, GCNX_ERR_DATA, "gcnx_csr_transpose: colidx[%d]=%d out of range", e, ci[e]);
      rpt[(size_t)ci[e] + 1]++;
    }
    for (int32_t r = 0; r < n; ++r) rpt[r + 1] += rpt[r];
    std::vector<int32_t> cur(rpt.begin(), rpt.end() - 1);
    for (int32_t r = 0; r < n; ++r)
      for (int32_t e = rp[r]; e < rp[r + 1]; ++e) {
        const int32_t p = cur[ci[e]]++;
        cit[p] = r;
        if (!v.empty()) vt[p] = v[e];
      }
    GCNX_HIP(ctx, hipMemcpyAsync(rowptr_t, rpt.data(), rpt.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nnz) GCNX_HIP(ctx, hipMemcpyAsync(colidx_t, cit.data(), cit.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!v.empty()) GCNX_HIP(ctx, hipMemcpyAsync(vals_t, vt.data(), vt.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  } catch (const std::bad_alloc&) {
    return gcnx_fail(ctx, GCNX_ERR_NOMEM, "gcnx_csr_transpose: out of host memory");
  }
  return GCNX_OK;
}


int gcnx_collate(gcnx_ctx* ctx, const int32_t* desc, int32_t b, const int32_t* node_ptr, const int32_t* rowptr,
                 const int32_t* colidx, const float* vals, const float* x, int64_t ldx, int32_t f, const float* y, int32_t c,
                 int32_t* o_rowptr, int32_t* o_colidx, float* o_vals, float* o_x, int64_t ldo, float* o_y,
                 int32_t* o_graph_ptr, int32_t* o_node_graph) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "device-side collate");
  GCNX_REQUIRE(ctx, b >= 0 && f >= 0 && c >= 0, "gcnx_collate: negative size");
  if (b == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, desc && node_ptr && rowptr && colidx && o_rowptr && o_colidx && o_graph_ptr, "gcnx_collate: NULL pointer");
  GCNX_REQUIRE(ctx, f == 0 || (x && o_x && ldx >= f && ldo >= f), "gcnx_collate: bad feature buffers");
  GCNX_REQUIRE(ctx, (vals == nullptr) == (o_vals == nullptr), "gcnx_collate: values in and out go together");
  GCNX_REQUIRE(ctx, !y || o_y, "gcnx_collate: labels need an output");
  hipLaunchKernelGGL(collate_kernel, dim3(16, b), dim3(256), 0, ctx->stream, desc, b, node_ptr, rowptr, colidx, vals, x, ldx,
                     f, y, c, o_rowptr, o_colidx, o_vals, o_x, ldo, o_y, o_graph_ptr, o_node_graph);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // extern "C"
