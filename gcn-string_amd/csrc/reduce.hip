// Row-reduction and elementwise kernels of the path: BiasAddGrad / activation grad, global pools
// (SegmentSum/Mean/Max) and their gradients, softmax + categorical cross-entropy, SGD.
// All HBM-bound; all reductions are two-stage and atomics-free (bitwise reproducible, so that a
// sharded run can be compared with a single-GPU run).
#include "common.h"

namespace {

constexpr int kColsumRows = 256;  // rows per first-stage workgroup

__device__ __forceinline__ float4 ld4(const float* p, bool vec, int valid) {
  if (vec) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid > 0) v.x = p[0];
  if (valid > 1) v.y = p[1];
  if (valid > 2) v.z = p[2];
  if (valid > 3) v.w = p[3];
  return v;
}
__device__ __forceinline__ void st4(float* p, float4 v, bool vec, int valid) {
  if (vec) { *reinterpret_cast<float4*>(p) = v; return; }
  if (valid > 0) p[0] = v.x;
  if (valid > 1) p[1] = v.y;
  if (valid > 2) p[2] = v.z;
  if (valid > 3) p[3] = v.w;
}

// Column sums of x[n, f] -> part[chunk][f].  Block = 16 float4 column lanes (64 columns) x 16 row
// groups; the 16 partial rows are combined through LDS in a fixed order (deterministic).
// Optional fused activation gradient: dz = dy * act'(y) is written and summed instead of x.
template <bool FUSE_ACT>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int32_t f,
                                                     int64_t rows_per_chunk, float* __restrict__ part,
                                                     const float* __restrict__ y, int64_t ldy, float* __restrict__ dz,
                                                     int64_t lddz, int act, const float* __restrict__ alpha,
                                                     float* __restrict__ part_alpha, int vec) {
  __shared__ float4 s[16][16];
  __shared__ float4 s2[16][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cl * 4;
  const int valid = f - c;                       // columns this lane really owns (<= 0: none)
  const bool v4 = vec && valid >= 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  const int64_t r1 = min(n, r0 + rows_per_chunk);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), acc_a = acc;
  if (!FUSE_ACT && vec && (int)blockIdx.x * 64 + 64 <= f) {
    // plain column sums over aligned full tiles: four float4 loads in flight per step (the generic loop's loads each
    // sit behind ld4's vector-or-scalar test and are waited for one by one); same row order, same sums
    const float* px = x + c;
    int64_t r = r0 + rg;
    for (; r + 48 < r1; r += 64) {
      const float4 v0 = *reinterpret_cast<const float4*>(px + r * ldx);
      const float4 v1 = *reinterpret_cast<const float4*>(px + (r + 16) * ldx);
      const float4 v2 = *reinterpret_cast<const float4*>(px + (r + 32) * ldx);
      const float4 v3 = *reinterpret_cast<const float4*>(px + (r + 48) * ldx);
      acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
      acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
      acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
    }
    for (; r < r1; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(px + r * ldx);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  } else if (valid > 0) {
    float4 al = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FUSE_ACT && act == GCNX_ACT_PRELU) al = ld4(alpha + c, false, valid);
#pragma unroll 4
    for (int64_t r = r0 + rg; r < r1; r += 16) {
      float4 v = ld4(x + r * ldx + c, v4, valid);
      if (FUSE_ACT) {
        const float4 yy = ld4(y + r * ldy + c, v4, valid);
        if (act == GCNX_ACT_RELU) {
          v.x = yy.x > 0.f ? v.x : 0.f; v.y = yy.y > 0.f ? v.y : 0.f; v.z = yy.z > 0.f ? v.z : 0.f; v.w = yy.w > 0.f ? v.w : 0.f;
        } else if (act == GCNX_ACT_PRELU) {
          acc_a.x += v.x * fminf(yy.x, 0.f); acc_a.y += v.y * fminf(yy.y, 0.f);
          acc_a.z += v.z * fminf(yy.z, 0.f); acc_a.w += v.w * fminf(yy.w, 0.f);
          v.x = yy.x > 0.f ? v.x : al.x * v.x; v.y = yy.y > 0.f ? v.y : al.y * v.y;
          v.z = yy.z > 0.f ? v.z : al.z * v.z; v.w = yy.w > 0.f ? v.w : al.w * v.w;
        }
        st4(dz + r * lddz + c, v, v4, valid);
      }
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  s[rg][cl] = acc;
  if (FUSE_ACT) s2[rg][cl] = acc_a;
  __syncthreads();
  if (rg == 0 && valid > 0) {
    float4 t = s[0][cl], ta = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 1; q < 16; ++q) { t.x += s[q][cl].x; t.y += s[q][cl].y; t.z += s[q][cl].z; t.w += s[q][cl].w; }
    if (part) st4(part + (int64_t)blockIdx.y * f + c, t, false, valid);
    if (FUSE_ACT && part_alpha) {
      ta = s2[0][cl];
      for (int q = 1; q < 16; ++q) { ta.x += s2[q][cl].x; ta.y += s2[q][cl].y; ta.z += s2[q][cl].z; ta.w += s2[q][cl].w; }
      st4(part_alpha + (int64_t)blockIdx.y * f + c, ta, false, valid);
    }
  }
}

// Plain column sums of a tall matrix whose rows are at most 1 KiB (f <= 256, f / 4 a power of two): a workgroup covers
// WHOLE rows -- f / 4 float4 lanes per row, 256 / (f / 4) row groups -- so that a wave instruction reads whole rows
// (1 KiB contiguous at f = 256; the 64-column tiles of colsum_kernel read 256-byte quarters of four rows), eight loads
// in flight per thread.  Config 3 (10^6 x 256): 321 -> ~190 us per launch, twice per training step (db1, db2).  Row
// groups are combined through LDS in a fixed order; partial layout as colsum_kernel: part[chunk][f].
__global__ __launch_bounds__(256) void colsum_wide_kernel(const float* __restrict__ x, int64_t ldx, int64_t n, int32_t f,
                                                          int64_t rows_per_chunk, float* __restrict__ part) {
  __shared__ float4 s[256];
  const int lpr = f >> 2, nrg = 256 / lpr;          // lanes per row, row groups
  const int cl = threadIdx.x % lpr, rg = threadIdx.x / lpr;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
  const float* px = x + cl * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t r = r0 + rg;
  for (; r + 7 * nrg < r1; r += 8 * nrg) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(px + (r + u * nrg) * ldx);
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; r < r1; r += nrg) {
    const float4 v = *reinterpret_cast<const float4*>(px + r * ldx);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  s[threadIdx.x] = acc;
  __syncthreads();
  if (rg == 0) {
    float4 t = s[cl];
    for (int q = 1; q < nrg; ++q) { const float4 o = s[q * lpr + cl]; t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w; }
    *reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * f + cl * 4) = t;
  }
}

// Pool forward: block = (column tile of 64, graph); 16 float4 column lanes x 16 row groups.
// RG = row groups per workgroup (16 x RG threads).  16 for the stand-alone pool; 64 (1024 threads, four times the
// loads in flight per workgroup) where the slice count is kept low for the consumer's sake (the head reads them).
template <int RG>
__global__ __launch_bounds__(16 * RG) void pool_fwd_kernel(const int32_t* __restrict__ gp, const float* __restrict__ x,
                                                       int64_t ldx, float* __restrict__ pooled, int32_t f, int mode,
                                                       int32_t* __restrict__ argmax, int vec, int nsplit,
                                                       float* __restrict__ cnt, const int32_t* __restrict__ glist = nullptr,
                                                       int64_t ldp = 0) {
  // nsplit > 1 (sum/avg only): blockIdx.z owns a slice of the graph's rows and writes a partial
  // row sum to pooled + z*B*f (the caller's workspace); pool_combine_kernel adds them in order.
  // cnt (sum/avg, may be NULL): the number of positive entries per (graph, column), same layout as pooled --
  // what the bias gradient of a ReLU layer under the pool needs (gcnx_pool_dense_softmax_cce, db_relu).
  __shared__ float4 s[RG][16];
  __shared__ int4 si[RG][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cl * 4;
  const int valid = f - c;
  const bool v4 = vec && valid >= 4;
  // glist (r3): the launch covers the listed graphs only (those taller than a tile: gcnx_pool_graph_list), rows of pooled at
  // leading dimension ldp
  const int g = glist ? glist[blockIdx.y] : (int)blockIdx.y;
  int lo = gp[g], hi = gp[g + 1];
  const int glo = lo, ghi = hi;
  const int64_t prow = glist ? ldp : (int64_t)f;
  if (nsplit > 1) {
    const int per = (hi - lo + nsplit - 1) / nsplit;
    lo = min(hi, lo + (int)blockIdx.z * per);
    hi = min(hi, lo + per);
    pooled += (int64_t)blockIdx.z * gridDim.y * f;
    if (cnt) cnt += (int64_t)blockIdx.z * gridDim.y * f;
  }
  const float init = (mode == GCNX_POOL_MAX) ? -INFINITY : 0.f;
  float4 acc = make_float4(init, init, init, init);
  int4 arg = make_int4(lo, lo, lo, lo);                  // MAX: arg max rows; SUM / AVG: counts of positives
  if (mode != GCNX_POOL_MAX) arg = make_int4(0, 0, 0, 0);
  // SUM / AVG over 16-byte-aligned full column tiles (uniform per workgroup): four plain float4 loads in flight per
  // step.  In the generic loop below every load sits behind ld4's vector-or-scalar test and hipcc closes each with
  // its own wait -- the "unroll 4" there is four dependent round trips, not four loads in flight.
  const bool fast = vec && mode != GCNX_POOL_MAX && (int)blockIdx.x * 64 + 64 <= f;
  if (fast) {
    const float* px = x + c;
    int r = lo + rg;
    auto add = [&](const float4& v) {
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      arg.x += v.x > 0.f; arg.y += v.y > 0.f; arg.z += v.z > 0.f; arg.w += v.w > 0.f;
    };
    for (; r + 3 * RG < hi; r += 4 * RG) {
      const float4 v0 = *reinterpret_cast<const float4*>(px + (int64_t)r * ldx);
      const float4 v1 = *reinterpret_cast<const float4*>(px + (int64_t)(r + RG) * ldx);
      const float4 v2 = *reinterpret_cast<const float4*>(px + (int64_t)(r + 2 * RG) * ldx);
      const float4 v3 = *reinterpret_cast<const float4*>(px + (int64_t)(r + 3 * RG) * ldx);
      add(v0); add(v1); add(v2); add(v3);               // row order, as the generic loop
    }
    for (; r < hi; r += RG) add(*reinterpret_cast<const float4*>(px + (int64_t)r * ldx));
  } else if (valid > 0) {
#pragma unroll 4
    for (int r = lo + rg; r < hi; r += RG) {
      const float4 v = ld4(x + (int64_t)r * ldx + c, v4, valid);
      if (mode == GCNX_POOL_MAX) {
        if (v.x > acc.x) { acc.x = v.x; arg.x = r; }
        if (v.y > acc.y) { acc.y = v.y; arg.y = r; }
        if (v.z > acc.z) { acc.z = v.z; arg.z = r; }
        if (v.w > acc.w) { acc.w = v.w; arg.w = r; }
      } else {
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        arg.x += v.x > 0.f; arg.y += v.y > 0.f; arg.z += v.z > 0.f; arg.w += v.w > 0.f;
      }
    }
  }
  s[rg][cl] = acc;
  si[rg][cl] = arg;
  __syncthreads();
  if (RG > 16) {   // SUM / AVG only (the launcher): fold groups rg, rg + 16, ... first, then the 16 as below
    if (rg < 16) {
      float4 o = s[rg][cl];
      int4 a = si[rg][cl];
      for (int q = rg + 16; q < RG; q += 16) {
        o.x += s[q][cl].x; o.y += s[q][cl].y; o.z += s[q][cl].z; o.w += s[q][cl].w;
        a.x += si[q][cl].x; a.y += si[q][cl].y; a.z += si[q][cl].z; a.w += si[q][cl].w;
      }
      s[rg][cl] = o;
      si[rg][cl] = a;
    }
    __syncthreads();
  }
  if (rg == 0 && valid > 0) {
    float4 o = s[0][cl];
    int4 a = si[0][cl];
    if (mode == GCNX_POOL_MAX) {
      // first maximal row wins (argmax of the oracle): row groups interleave rows, so ties break
      // on the smaller row index.
      for (int q = 1; q < 16; ++q) {
        const float4 v = s[q][cl];
        const int4 ai = si[q][cl];
        if (v.x > o.x || (v.x == o.x && ai.x < a.x)) { o.x = v.x; a.x = ai.x; }
        if (v.y > o.y || (v.y == o.y && ai.y < a.y)) { o.y = v.y; a.y = ai.y; }
        if (v.z > o.z || (v.z == o.z && ai.z < a.z)) { o.z = v.z; a.z = ai.z; }
        if (v.w > o.w || (v.w == o.w && ai.w < a.w)) { o.w = v.w; a.w = ai.w; }
      }
      if (hi == lo) { o = make_float4(0.f, 0.f, 0.f, 0.f); a = make_int4(lo, lo, lo, lo); }
      if (argmax) {
        int32_t* ap = argmax + (int64_t)g * f + c;
        if (valid > 0) ap[0] = a.x;
        if (valid > 1) ap[1] = a.y;
        if (valid > 2) ap[2] = a.z;
        if (valid > 3) ap[3] = a.w;
      }
    } else {
      for (int q = 1; q < 16; ++q) { o.x += s[q][cl].x; o.y += s[q][cl].y; o.z += s[q][cl].z; o.w += s[q][cl].w; }
      if (mode == GCNX_POOL_AVG && ghi > glo && nsplit <= 1) {
        const float inv = (float)(ghi - glo);
        o.x /= inv; o.y /= inv; o.z /= inv; o.w /= inv;
      }
      if (cnt) {
        for (int q = 1; q < 16; ++q) { a.x += si[q][cl].x; a.y += si[q][cl].y; a.z += si[q][cl].z; a.w += si[q][cl].w; }
        st4(cnt + (int64_t)g * f + c, make_float4((float)a.x, (float)a.y, (float)a.z, (float)a.w), false, valid);
      }
    }
    st4(pooled + (int64_t)g * prow + c, o, false, valid);
  }
}

// Second stage of the split pool: pooled[g][c] = sum_z part[z][g][c] (z ascending), / n_g for AVG.
__global__ __launch_bounds__(256) void pool_combine_kernel(const float* __restrict__ part, const int32_t* __restrict__ gp,
                                                           float* __restrict__ pooled, int32_t b, int32_t f, int nsplit,
                                                           int mode) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)b * f) return;
  float acc = 0.f;
  for (int z = 0; z < nsplit; ++z) acc += part[(int64_t)z * b * f + i];
  if (mode == GCNX_POOL_AVG) {
    const int g = (int)(i / f);
    const int cnt = gp[g + 1] - gp[g];
    if (cnt > 0) acc /= (float)cnt;
  }
  pooled[i] = acc;
}

// Pool backward (+ optional fused ReLU mask of the layer that produced the pooled tensor).
// One wave per run of kPoolBwdRows consecutive rows (64 float4 lanes cover 256 columns per pass).  The graph of
// the run's first row is found by ONE binary search in graph_ptr (log2 B dependent loads -- per row they were
// most of this kernel's time); the following rows only step the graph index forward.
constexpr int kPoolBwdRows = 32;  // at most; small inputs take fewer rows per wave so that the grid still fills the chip
__global__ __launch_bounds__(256) void pool_bwd_kernel(const int32_t* __restrict__ gp, int32_t b,
                                                       const float* __restrict__ dp, float* __restrict__ dx,
                                                       int64_t lddx, int32_t n, int32_t f, int mode,
                                                       const int32_t* __restrict__ argmax, const float* __restrict__ y,
                                                       int64_t ldy, int vec, int rpw) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw;
  if (r0 >= n) return;
  const int r1 = min(n, r0 + rpw);
  int lo = 0, hi = b;  // find g with gp[g] <= r0 < gp[g+1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (gp[mid] <= r0) lo = mid; else hi = mid;
  }
  int g = lo;
  int gend = gp[g + 1];
  for (int r = r0; r < r1; ++r) {
    while (r >= gend && g + 1 < b) { ++g; gend = gp[g + 1]; }   // empty graphs are skipped too
    const float scale = (mode == GCNX_POOL_AVG) ? 1.0f / (float)(gend - gp[g]) : 1.0f;
    for (int c = lane * 4; c < f; c += 256) {
      const int valid = f - c;
      const bool v4 = vec && valid >= 4;
      float4 v = ld4(dp + (int64_t)g * f + c, vec && f % 4 == 0 && valid >= 4, valid);
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
      if (mode == GCNX_POOL_MAX) {
        const int32_t* ap = argmax + (int64_t)g * f + c;
        if (ap[0] != r) v.x = 0.f;
        if (valid > 1 && ap[1] != r) v.y = 0.f;
        if (valid > 2 && ap[2] != r) v.z = 0.f;
        if (valid > 3 && ap[3] != r) v.w = 0.f;
      }
      if (y) {
        const float4 yy = ld4(y + (int64_t)r * ldy + c, v4, valid);
        if (!(yy.x > 0.f)) v.x = 0.f;
        if (!(yy.y > 0.f)) v.y = 0.f;
        if (!(yy.z > 0.f)) v.z = 0.f;
        if (!(yy.w > 0.f)) v.w = 0.f;
      }
      st4(dx + (int64_t)r * lddx + c, v, v4, valid);
    }
  }
}

// Softmax + CCE + accuracy over B graphs, one workgroup, deterministic tree reduction.
__global__ __launch_bounds__(256) void softmax_cce_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ y, int32_t b, int32_t c,
                                                          float denom, float* __restrict__ probs,
                                                          float* __restrict__ loss_acc, float* __restrict__ dlogits,
                                                          int from_logits) {
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
      // LOGITS (what Keras runs inside tf.function): softmax_cross_entropy_with_logits on the Softmax op's input --
      // no renormalisation, no clip.  PROBS (eager tensors): clip_by_value passes no gradient outside [1e-7, 1-1e-7].
      const bool pass = from_logits || (p > 1e-7f && p < 1.0f - 1e-7f);
      if (pass) ymsum += yy[k];
      if (p > pmax) { pmax = p; pa = k; }
      if (yy[k] > ymax) { ymax = yy[k]; ya = k; }
      if (from_logits) l += yy[k] * ((m - z[k]) + logf(sum));
      else l -= yy[k] * logf(fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f));
    }
    if (dlogits)
      for (int k = 0; k < c; ++k) {
        const float p = expf(z[k] - m) / sum;
        const float ym = (from_logits || (p > 1e-7f && p < 1.0f - 1e-7f)) ? yy[k] : 0.f;
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
                                                  float lr_arg, const float* __restrict__ lr_dev) {
  const float lr = lr_dev ? *lr_dev : lr_arg;            // (gcnx_set_lr_source)
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = p[i] - lr * g[i];
}

// db = colsum(dZ) for dZ[j] = scale_g * dPooled[g(j)] * [y[j] > 0] WITHOUT a materialised dZ (the companion of the
// folded aggregation gcnx_spmm_csr_pool_bwd): sum_j dZ[j][c] = sum_g scale_g dPooled[g][c] * #{j in g : y[j][c] > 0}.
// Block = (column tile of 64, graph, row slice): count, times the graph's dPooled row -> one partial row each.
__global__ __launch_bounds__(256) void pool_mask_partial_kernel(const int32_t* __restrict__ gp, const float* __restrict__ y,
                                                                int64_t ldy, const float* __restrict__ dp, int64_t lddp,
                                                                float* __restrict__ part, int32_t f, int avg, int vec,
                                                                int nsplit) {
  __shared__ float4 s[16][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cl * 4;
  const int valid = f - c;
  const bool v4 = vec && valid >= 4;
  const int g = blockIdx.y;
  int lo = gp[g], hi = gp[g + 1];
  const int rows = hi - lo;
  if (nsplit > 1) {
    const int per = (hi - lo + nsplit - 1) / nsplit;
    lo = min(hi, lo + (int)blockIdx.z * per);
    hi = min(hi, lo + per);
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid > 0) {
#pragma unroll 4
    for (int r = lo + rg; r < hi; r += 16) {
      const float4 v = ld4(y + (int64_t)r * ldy + c, v4, valid);
      acc.x += v.x > 0.f ? 1.f : 0.f; acc.y += v.y > 0.f ? 1.f : 0.f;
      acc.z += v.z > 0.f ? 1.f : 0.f; acc.w += v.w > 0.f ? 1.f : 0.f;
    }
  }
  s[rg][cl] = acc;
  __syncthreads();
  if (rg == 0 && valid > 0) {
    float4 t = s[0][cl];
    for (int q = 1; q < 16; ++q) { t.x += s[q][cl].x; t.y += s[q][cl].y; t.z += s[q][cl].z; t.w += s[q][cl].w; }
    const float sc = (avg && rows > 0) ? 1.0f / (float)rows : 1.0f;
    const float4 d = ld4(dp + (int64_t)g * lddp + c, false, valid);
    t.x *= d.x * sc; t.y *= d.y * sc; t.z *= d.z * sc; t.w *= d.w * sc;
    st4(part + ((int64_t)blockIdx.z * gridDim.y + g) * f + c, t, false, valid);
  }
}

// Column sums of a few hundred partial rows in ONE launch.  colsum_kernel's shape (64 columns x 16 row groups per
// workgroup) would leave this to f/64 workgroups walking rows/16 dependent trips each (config 2: 2 workgroups, 40
// trips, 22 us); here a workgroup owns 8 columns (two float4 lanes) x 128 row groups -- f/8 workgroups, rows/128
// trips -- and folds the 128 partial sums in a fixed tree through LDS.
__global__ __launch_bounds__(256) void colpart_reduce_kernel(const float* __restrict__ part, int64_t rows, int32_t f,
                                                             float* __restrict__ out) {
  __shared__ float4 s[128][2];
  gcnx_colpart_reduce_body(part, rows, f, out, blockIdx.x, s);
}

int colsum_impl(gcnx_ctx* ctx, const float* x, int64_t ldx, int64_t n, int32_t f, float* out, const float* y,
                int64_t ldy, float* dz, int64_t lddz, int act, const float* alpha, float* out_alpha,
                size_t ws_off = 0) {   // ws_off: floats at the start of the workspace that belong to the caller
  const bool fuse = (dz != nullptr);
  auto al = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
  const int vec = al(x) && ldx % 4 == 0 && (!fuse || (al(y) && ldy % 4 == 0 && al(dz) && lddz % 4 == 0));
  // tall plain sums of rows of at most 1 KiB: whole-row workgroups, 1024-row chunks (fewer partial rows to fold)
  const bool wide = !fuse && vec && out && n >= 65536 && f >= 16 && f <= 256 && (f & (f - 1)) == 0;
  const int64_t chunk_rows = wide ? 1024 : kColsumRows;
  const int nchunks = gcnx_cdiv(n, chunk_rows);
  const bool want_alpha = fuse && act == GCNX_ACT_PRELU && out_alpha;
  const size_t need = (size_t)nchunks * f * sizeof(float) * (want_alpha ? 2 : 1);
  float* part = nullptr;
  float* part_a = nullptr;
  if (out || want_alpha) {
    if (nchunks > 1) {
      int rc = gcnx_ws_reserve(ctx, need + ws_off * sizeof(float));
      if (rc) return rc;
      part = out ? (float*)ctx->ws + ws_off : nullptr;
      part_a = want_alpha ? (float*)ctx->ws + ws_off + (size_t)nchunks * f : nullptr;
    } else {
      part = out;
      part_a = want_alpha ? out_alpha : nullptr;
    }
  }
  dim3 grid(gcnx_cdiv(f, 64), nchunks);
  if (wide)
    hipLaunchKernelGGL(colsum_wide_kernel, dim3(nchunks), dim3(256), 0, ctx->stream, x, ldx, n, f, chunk_rows, part);
  else if (fuse)
    hipLaunchKernelGGL((colsum_kernel<true>), grid, dim3(256), 0, ctx->stream, x, ldx, n, f, (int64_t)kColsumRows,
                       part, y, ldy, dz, lddz, act, alpha, part_a, vec);
  else
    hipLaunchKernelGGL((colsum_kernel<false>), grid, dim3(256), 0, ctx->stream, x, ldx, n, f, (int64_t)kColsumRows,
                       part, nullptr, (int64_t)0, nullptr, (int64_t)0, 0, nullptr, nullptr, vec);
  GCNX_LAUNCH_OK(ctx);
  if (nchunks > 1) {
    const int vec2 = f % 4 == 0;   // the partials live in the 256-B aligned workspace with row stride f
    dim3 g2(gcnx_cdiv(f, 64), 1);
    if (out && wide && nchunks <= 4096 && al(out)) {
      hipLaunchKernelGGL(colpart_reduce_kernel, dim3(gcnx_cdiv(f, 8)), dim3(256), 0, ctx->stream, (const float*)part,
                         (int64_t)nchunks, f, out);
      GCNX_LAUNCH_OK(ctx);
    } else if (out) {
      hipLaunchKernelGGL((colsum_kernel<false>), g2, dim3(256), 0, ctx->stream, (const float*)part, (int64_t)f,
                         (int64_t)nchunks, f, (int64_t)nchunks, out, nullptr, (int64_t)0, nullptr, (int64_t)0, 0,
                         nullptr, nullptr, vec2);
      GCNX_LAUNCH_OK(ctx);
    }
    if (want_alpha) {
      hipLaunchKernelGGL((colsum_kernel<false>), g2, dim3(256), 0, ctx->stream, (const float*)part_a, (int64_t)f,
                         (int64_t)nchunks, f, (int64_t)nchunks, out_alpha, nullptr, (int64_t)0, nullptr, (int64_t)0,
                         0, nullptr, nullptr, vec2);
      GCNX_LAUNCH_OK(ctx);
    }
  }
  return GCNX_OK;
}

}  // namespace

// Few graphs (an E. coli batch has 32) cannot fill 256 CUs with one workgroup per (graph, column tile): each
// graph's rows are sliced over blockIdx.z (first stage, below) and the partial sums combined in workgroup order --
// by pool_combine_kernel, or by the classifier head while it stages its operand (gcnx_pool_dense_softmax_cce).
int gcnx_pool_split(const gcnx_ctx* ctx, int32_t b, int32_t f, int mode, int half_wgs_per_cu) {
  const int base_wgs = gcnx_cdiv(f, 64) * b;
  int nsplit = 1;
  if (mode != GCNX_POOL_MAX && base_wgs < 2 * ctx->num_cus) {
    nsplit = (half_wgs_per_cu * ctx->num_cus / 2 + base_wgs - 1) / base_wgs;
    if (nsplit < 2) nsplit = 2;
    if (nsplit > 16) nsplit = 16;
    if (ctx->knob_pool_split >= 2 && ctx->knob_pool_split <= 16) nsplit = ctx->knob_pool_split;   // tuning knob, read once at context creation
  }
  return nsplit;
}

int gcnx_pool_partials(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* x, int64_t ldx, int32_t b, int32_t f,
                       int mode, int nsplit, float* part, float* cnt_part, int wide) {
  const int vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && ldx % 4 == 0;
  dim3 grid(gcnx_cdiv(f, 64), b, nsplit);
  if (wide)
    hipLaunchKernelGGL(pool_fwd_kernel<64>, grid, dim3(1024), 0, ctx->stream, graph_ptr, x, ldx, part, f, mode,
                       (int32_t*)nullptr, vec, nsplit, cnt_part);
  else
    hipLaunchKernelGGL(pool_fwd_kernel<16>, grid, dim3(256), 0, ctx->stream, graph_ptr, x, ldx, part, f, mode,
                       (int32_t*)nullptr, vec, nsplit, cnt_part);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

// SUM / AVG pool (and positive counts) of the listed graphs only, straight into their rows of pooled / cnt.
int gcnx_pool_graph_list(gcnx_ctx* ctx, const int32_t* graph_ptr, const int32_t* glist, int32_t nlist, const float* x, int64_t ldx,
                         int32_t f, int mode, float* pooled, int64_t ldp, float* cnt) {
  if (nlist <= 0) return GCNX_OK;
  const int vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && ldx % 4 == 0;
  // 1024-thread workgroups (SUM / AVG only; the callers' case): the listed graphs are the tall ones -- 1 300 to 3 000 rows each, a
  // few per shard of an 8-rank run -- and a thread of the 256-thread shape walks 80-190 rows four at a time: a chain of
  // dependent round trips (17 us for nine graphs; r4 shard trace), a quarter of it with 64 row groups
  if (mode != GCNX_POOL_MAX)
    hipLaunchKernelGGL(pool_fwd_kernel<64>, dim3(gcnx_cdiv(f, 64), nlist), dim3(1024), 0, ctx->stream, graph_ptr, x, ldx, pooled, f, mode,
                       (int32_t*)nullptr, vec, 1, cnt, glist, ldp);
  else
    hipLaunchKernelGGL(pool_fwd_kernel<16>, dim3(gcnx_cdiv(f, 64), nlist), dim3(256), 0, ctx->stream, graph_ptr, x, ldx, pooled, f, mode,
                       (int32_t*)nullptr, vec, 1, cnt, glist, ldp);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

constexpr int64_t kPartialsOneLaunch = 4096;   // partial rows one launch of colpart_reduce_kernel still walks quickly

size_t gcnx_colsum_partials_ws(int64_t rows, int32_t f) {
  const size_t mine = (size_t)rows * f;
  const size_t stage2 = (size_t)gcnx_cdiv(rows, kColsumRows) * f;
  return (mine + stage2) * sizeof(float);
}

int gcnx_colsum_partials(gcnx_ctx* ctx, int64_t rows, int32_t f, float* out) {
  const float* part = (const float*)ctx->ws;
  if (rows > kPartialsOneLaunch) return colsum_impl(ctx, part, f, rows, f, out, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, (size_t)rows * f);
  if (f % 4 != 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0)
    return colsum_impl(ctx, part, f, rows, f, out, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, (size_t)rows * f);
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3(gcnx_cdiv(f, 8)), dim3(256), 0, ctx->stream, part, rows, f, out);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

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
  if (act == GCNX_ACT_NONE && dz == dy) dz = nullptr;   // identity in place: a pure column sum, nothing is written back
  return colsum_impl(ctx, dy, lddy, n, f, db, y ? y : dy, y ? ldy : lddy, dz, lddz, act, alpha, dalpha);
}

int gcnx_segment_pool(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* x, int64_t ldx, float* pooled, int32_t b,
                      int32_t f, int mode, int32_t* argmax) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "global pool");
  GCNX_REQUIRE(ctx, b >= 0 && f >= 0, "gcnx_segment_pool: negative size");
  GCNX_REQUIRE(ctx, mode >= GCNX_POOL_SUM && mode <= GCNX_POOL_MAX, "gcnx_segment_pool: unknown mode %d", mode);
  if (b == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, graph_ptr && x && pooled, "gcnx_segment_pool: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= f, "gcnx_segment_pool: leading dimension too small");
  GCNX_REQUIRE(ctx, mode != GCNX_POOL_MAX || argmax, "gcnx_segment_pool: MAX needs an argmax buffer");
  const int vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && ldx % 4 == 0;
  const int nsplit = gcnx_pool_split(ctx, b, f, mode, 4);
  if (nsplit > 1) {
    int rc = gcnx_ws_reserve(ctx, (size_t)nsplit * b * f * sizeof(float));
    if (rc) return rc;
    rc = gcnx_pool_partials(ctx, graph_ptr, x, ldx, b, f, mode, nsplit, (float*)ctx->ws, nullptr, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(pool_combine_kernel, dim3(gcnx_cdiv((int64_t)b * f, 256)), dim3(256), 0, ctx->stream,
                       (const float*)ctx->ws, graph_ptr, pooled, b, f, nsplit, mode);
    GCNX_LAUNCH_OK(ctx);
    return GCNX_OK;
  }
  dim3 grid(gcnx_cdiv(f, 64), b);
  hipLaunchKernelGGL(pool_fwd_kernel<16>, grid, dim3(256), 0, ctx->stream, graph_ptr, x, ldx, pooled, f, mode, argmax, vec, 1,
                     (float*)nullptr);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_segment_pool_bwd(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* dpooled, float* dx, int64_t lddx,
                          int32_t n, int32_t b, int32_t f, int mode, const int32_t* argmax, const float* y,
                          int64_t ldy, float* db) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "global pool bwd");
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
  auto al = [](const void* p_) { return (reinterpret_cast<uintptr_t>(p_) & 15) == 0; };
  const int vec = al(dx) && lddx % 4 == 0 && al(dpooled) && (!y || (al(y) && ldy % 4 == 0));
  int rpw = (int)(n / (64LL * ctx->num_cus));   // rows per wave: ~16 waves per SIMD before runs get longer
  rpw = rpw < 1 ? 1 : (rpw > kPoolBwdRows ? kPoolBwdRows : rpw);
  hipLaunchKernelGGL(pool_bwd_kernel, dim3(gcnx_cdiv(n, 4 * rpw)), dim3(256), 0, ctx->stream, graph_ptr, b, dpooled, dx,
                     lddx, n, f, mode, argmax, y, ldy, vec, rpw);
  GCNX_LAUNCH_OK(ctx);
  if (db) return gcnx_colsum(ctx, dx, lddx, n, f, db);
  return GCNX_OK;
}

int gcnx_pool_bwd_colsum(gcnx_ctx* ctx, const int32_t* graph_ptr, int32_t b, const float* dpooled, int64_t lddp,
                         const float* y, int64_t ldy, int32_t f, int mode, float* db) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, b >= 0 && f >= 0, "gcnx_pool_bwd_colsum: negative size");
  GCNX_REQUIRE(ctx, mode == GCNX_POOL_SUM || mode == GCNX_POOL_AVG,
               "gcnx_pool_bwd_colsum: pool mode %d has no folded form (use gcnx_segment_pool_bwd + gcnx_act_bias_grad)", mode);
  if (f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, db, "gcnx_pool_bwd_colsum: NULL pointer");
  if (b == 0) {
    GCNX_HIP(ctx, hipMemsetAsync(db, 0, (size_t)f * 4, ctx->stream));
    return GCNX_OK;
  }
  GCNX_REQUIRE(ctx, graph_ptr && dpooled && y, "gcnx_pool_bwd_colsum: NULL pointer");
  GCNX_REQUIRE(ctx, lddp >= f && ldy >= f, "gcnx_pool_bwd_colsum: leading dimension too small");
  const int vec = (reinterpret_cast<uintptr_t>(y) & 15) == 0 && ldy % 4 == 0;
  const int base_wgs = gcnx_cdiv(f, 64) * b;            // as gcnx_segment_pool: few graphs -> slice their rows
  int nsplit = 1;
  if (base_wgs < 2 * ctx->num_cus) {
    nsplit = (2 * ctx->num_cus + base_wgs - 1) / base_wgs;
    if (nsplit > 16) nsplit = 16;
  }
  const int64_t nrows = (int64_t)nsplit * b;            // partial rows [nsplit][b][f] at the start of the workspace
  const size_t mine = (size_t)nrows * f;
  const size_t stage2 = (size_t)gcnx_cdiv(nrows, kColsumRows) * f;
  int rc = gcnx_ws_reserve(ctx, (mine + stage2) * sizeof(float));
  if (rc) return rc;
  float* part = (float*)ctx->ws;
  hipLaunchKernelGGL(pool_mask_partial_kernel, dim3(gcnx_cdiv(f, 64), b, nsplit), dim3(256), 0, ctx->stream, graph_ptr,
                     y, ldy, dpooled, lddp, part, f, mode == GCNX_POOL_AVG ? 1 : 0, vec, nsplit);
  GCNX_LAUNCH_OK(ctx);
  return colsum_impl(ctx, part, f, nrows, f, db, nullptr, 0, nullptr, 0, 0, nullptr, nullptr, mine);
}

int gcnx_softmax_cce(gcnx_ctx* ctx, const float* logits, const float* y, int32_t b, int32_t c, float denom,
                     float* probs, float* loss_acc, float* dlogits, int cce_mode) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "softmax + CCE");
  GCNX_REQUIRE(ctx, b >= 0 && c >= 1, "gcnx_softmax_cce: bad size b=%d c=%d", b, c);
  if (b == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, logits && y && probs && loss_acc, "gcnx_softmax_cce: NULL pointer");
  GCNX_REQUIRE(ctx, denom > 0.f, "gcnx_softmax_cce: denom must be positive");
  GCNX_REQUIRE(ctx, cce_mode == GCNX_CCE_PROBS || cce_mode == GCNX_CCE_LOGITS, "gcnx_softmax_cce: unknown cce_mode %d", cce_mode);
  hipLaunchKernelGGL(softmax_cce_kernel, dim3(1), dim3(256), 0, ctx->stream, logits, y, b, c, denom, probs, loss_acc,
                     dlogits, cce_mode == GCNX_CCE_LOGITS ? 1 : 0);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

// The learning rate of every update launch (gcnx_sgd, gcnx_gemm_dw_sgd, gcnx_gemm_dw2) from a device scalar instead of the
// `lr` argument: a learning rate is otherwise a kernel ARGUMENT, so a captured step is tied to one value and a schedule
// that changes every step (any keras LearningRateSchedule; the reference's PiecewiseConstantDecay has three values,
// gcn.py:321-325) would capture a graph per step.  lr_dev = NULL restores the argument.  The pointer must stay valid.
int gcnx_set_lr_source(gcnx_ctx* ctx, const float* lr_dev) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, !ctx->capturing, "gcnx_set_lr_source: not inside a capture (a captured launch keeps the source it was recorded with)");
  ctx->lr_dev = lr_dev;
  return GCNX_OK;
}

int gcnx_sgd(gcnx_ctx* ctx, float* params, const float* grads, int64_t n, float lr) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "SGD update");
  GCNX_REQUIRE(ctx, n >= 0, "gcnx_sgd: negative size");
  if (n == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, params && grads, "gcnx_sgd: NULL pointer");
  int grid = gcnx_cdiv(n, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid), dim3(256), 0, ctx->stream, params, grads, n, lr, ctx->lr_dev);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // extern "C"
