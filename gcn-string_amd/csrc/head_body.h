// Body of the classifier-head kernel (csrc/head.hip), shared with csrc/gemm.hip.  See head.hip for what it computes.
#ifndef GCNX_HEAD_BODY_H
#define GCNX_HEAD_BODY_H
#include "common.h"

namespace gcnx_head {

constexpr int kHeadRows = 32;     // graphs per workgroup
constexpr int kHeadMaxC = 32;     // classes held in LDS per graph
constexpr int kHeadLdsFloats = 14 * 1024;   // 56 KiB for the staged operands (else they are read from global)

// PARTS (with STAGED): the pooled operand arrives as the split pool's partial row sums and is combined here, in
// slice order, while it is staged (what pool_combine_kernel would do in a launch of its own); the combined rows
// are also written to `pooled_out`.
struct PoolParts {
  const float* part;        // [nsplit][b][h], row stride h
  const int32_t* gp;        // graph_ptr (AVG: row counts)
  float* pooled_out;        // [b, ldp]
  int32_t nsplit;
  int32_t avg;
  const float* cnt_part;    // with db_relu: positives per (slice, graph, column), layout of part
  float* db_relu;           // [h] or NULL: sum_g s_g * dPooled[g] * cnt[g]  (bias gradient of the ReLU layer under the pool)
};

// CT > 0: the number of classes as a compile-time constant (2 for the reference's binary labels).  With c a run-time
// value every per-class loop is a chain of scalar branches and dependent LDS reads executed by a single workgroup
// with nothing to hide them behind: the dPooled phase alone was 3.3 us of a 15 us kernel.
// The body of head_kernel as a device function: `s_dyn` is the LDS area for the staged operands, (bx, nblk) this
// workgroup's index and the number of head workgroups -- so that the same code can run as the first workgroups of
// another 256-thread launch (gemm.hip: the weight-gradient launch of a small-batch step hides the head's leaves).
// ZU: unroll of the partial-sum combine (ZU x 32 VGPRs of loads in flight: 4 as a kernel of its own, 2 inside the
// weight-gradient launch, whose tile workgroups need the 128-VGPR occupancy).
template <bool STAGED, bool PARTS = false, int CT = 0, int ZU = 4>
__device__ __forceinline__ void head_body(const float* __restrict__ pooled, int64_t ldp, const float* __restrict__ w,
                                          const float* __restrict__ bias, const float* __restrict__ y, int32_t b, int32_t h,
                                          int32_t c_rt, float denom, float* __restrict__ probs, float* __restrict__ loss_acc,
                                          float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dpooled,
                                          int64_t lddp, float* __restrict__ slabs, int* __restrict__ ticket, PoolParts pp,
                                          int from_logits, float* s_dyn, int bx, int nblk_) {
  static_assert(STAGED || !PARTS, "partials are combined into the LDS copy");
  const int c = CT > 0 ? CT : c_rt;
  constexpr bool staged = STAGED;
  constexpr int MC = CT > 0 ? CT : kHeadMaxC;       // class slots held in LDS (a compile-time class count keeps this small:
                                                    // the body also runs inside a launch whose occupancy the total decides)
  __shared__ float s_z[kHeadRows][MC + 1];          // logits, then dlogits
  __shared__ float s_y[kHeadRows * MC];             // labels of this workgroup's graphs, [rows][c] packed
  __shared__ float s_red[2][256];
  __shared__ int s_last;
  // s_dyn (argument): staged pooled rows [rows][h+1] | w [h*c] | counts | db partials
  const int tid = threadIdx.x;
  const int g0 = bx * kHeadRows;
  const int rows = min(kHeadRows, b - g0);
  const bool train = dw != nullptr;
  // Everything below walks the two small operands several times with dependent, strided reads; staged once in
  // LDS (coalesced loads, one memory latency) those walks cost LDS latency instead of L2 latency per element.
  const int ps = h + 1;                           // padded row stride: column walks (dW) stay conflict-free
  float* s_p = s_dyn;
  float* s_w = s_dyn + kHeadRows * ps;
  float* s_c = s_w + h * c;                       // PARTS with db_relu: positive counts [rows][h+1], then 4 x [h] partials
  float* s_db = s_c + kHeadRows * ps;
  __shared__ float s_sc[kHeadRows];               // AVG: 1 / rows of the graph
  const bool want_db = PARTS && pp.db_relu != nullptr && dw != nullptr;
  // One workgroup, nothing to hide a memory round trip behind: every independent load of the staging phase is
  // issued before the first result is consumed -- the first 256 labels and W entries and the graph sizes sit in
  // registers under the partial-sum loads instead of each costing a load -> wait -> LDS store trip of its own.
  const bool y_now = y && tid < rows * c, w_now = staged && tid < h * c;
  const float y_first = y_now ? y[(int64_t)g0 * c + tid] : 0.f;
  const float w_first = w_now ? w[tid] : 0.f;
  if (y) for (int idx = tid + 256; idx < rows * c; idx += 256) s_y[idx] = y[(int64_t)g0 * c + idx];
  if (staged) {
    if (PARTS) {
      // The split pool's partial row sums (and counts) are combined here, in slice order.  float4 lanes, four
      // elements per thread and pass, the slice loop unrolled: a thread's loads are in flight together -- one
      // memory latency for the whole combine at the E. coli shape.  (h % 4 == 0: checked by the host.)
      int n_first = 1;
      if (tid < rows) n_first = pp.gp[g0 + tid + 1] - pp.gp[g0 + tid];
      const int h4 = h >> 2, total = rows * h4;
      const int64_t zs = (int64_t)b * h;
      const bool st4 = (ldp & 3) == 0 && (reinterpret_cast<uintptr_t>(pp.pooled_out) & 15) == 0;
      for (int e0 = 0; e0 < total; e0 += 1024) {
        float4 acc[4], cacc[4];
        int ii[4], jj[4], nn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = min(e0 + u * 256 + tid, total - 1);       // clamped: loads stay in range, stores are guarded
          ii[u] = idx / h4;
          jj[u] = (idx - ii[u] * h4) * 4;
          acc[u] = cacc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          nn[u] = pp.avg ? pp.gp[g0 + ii[u] + 1] - pp.gp[g0 + ii[u]] : 1;
        }
#pragma unroll ZU
        for (int z = 0; z < pp.nsplit; ++z) {                       // slice order, as pool_combine_kernel
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t at = z * zs + (int64_t)(g0 + ii[u]) * h + jj[u];
            const float4 v = *reinterpret_cast<const float4*>(pp.part + at);
            acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
            if (want_db) {
              const float4 q = *reinterpret_cast<const float4*>(pp.cnt_part + at);
              cacc[u].x += q.x; cacc[u].y += q.y; cacc[u].z += q.z; cacc[u].w += q.w;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (e0 + u * 256 + tid >= total) continue;
          float4 v = acc[u];
          if (pp.avg && nn[u] > 0) { v.x /= (float)nn[u]; v.y /= (float)nn[u]; v.z /= (float)nn[u]; v.w /= (float)nn[u]; }
          float* sp = s_p + ii[u] * ps + jj[u];
          sp[0] = v.x; sp[1] = v.y; sp[2] = v.z; sp[3] = v.w;
          if (want_db) {
            float* sc = s_c + ii[u] * ps + jj[u];
            sc[0] = cacc[u].x; sc[1] = cacc[u].y; sc[2] = cacc[u].z; sc[3] = cacc[u].w;
          }
          float* po = pp.pooled_out + (int64_t)(g0 + ii[u]) * ldp + jj[u];
          if (st4) *reinterpret_cast<float4*>(po) = v;
          else { po[0] = v.x; po[1] = v.y; po[2] = v.z; po[3] = v.w; }
        }
      }
      if (tid < rows) s_sc[tid] = (pp.avg && n_first > 0) ? 1.0f / (float)n_first : 1.0f;
    } else {
      for (int i = tid >> 6; i < rows; i += 4)             // one wave per row: no integer division in the loops
        for (int j = tid & 63; j < h; j += 64) s_p[i * ps + j] = pooled[(int64_t)(g0 + i) * ldp + j];
    }
    if (w_now) s_w[tid] = w_first;
    for (int idx = tid + 256; idx < h * c; idx += 256) s_w[idx] = w[idx];
    __syncthreads();
  }
  if (y_now) s_y[tid] = y_first;                  // read after the barrier that closes the logits phase
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
        // LOGITS (tf.function): softmax_cross_entropy_with_logits, no clip; PROBS (eager): clip_by_value passes
        // no gradient outside [1e-7, 1 - 1e-7]
        if (from_logits || (p > 1e-7f && p < 1.0f - 1e-7f)) ymsum += yk;
        if (yk > ymax) { ymax = yk; ya = k; }
        if (from_logits) loss += yk * ((m - z[k]) + logf(sum));
        else loss -= yk * logf(fminf(fmaxf(p, 1e-7f), 1.0f - 1e-7f));
      }
      if (p > pmax) { pmax = p; pa = k; }
    }
    if (y) {
      hit = (pa == ya) ? 1.f : 0.f;
      for (int k = 0; k < c; ++k) {
        const float p = expf(z[k] - m) / sum;
        const float ym = (from_logits || (p > 1e-7f && p < 1.0f - 1e-7f)) ? s_y[tid * c + k] : 0.f;
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
  const int nblk = nblk_;
  const int npart = h * c + c + 2 + (want_db ? h : 0);    // dW | db | loss sum, hits | db_relu
  float* part = nblk > 1 ? slabs + (int64_t)bx * npart : nullptr;
  if (tid == 0) {
    if (part) { part[h * c + c] = s_red[0][0]; part[h * c + c + 1] = s_red[1][0]; }
    else { loss_acc[0] = s_red[0][0] / denom; loss_acc[1] = s_red[1][0]; }
  }
  if (!train) {
    if (nblk == 1) return;
  } else {
    // dPooled[i, j] = sum_k dlogits[i,k] * w[j,k]
    for (int j = tid & 63; j < h; j += 64) {
      float dbp = 0.f;                                    // db_relu: this wave's graphs (i = wave, wave + 4, ...)
      for (int i = tid >> 6; i < rows; i += 4) {
        float acc = 0.f;
        for (int k = 0; k < c; ++k) acc = fmaf(s_z[i][k], W(j, k), acc);
        dpooled[(int64_t)(g0 + i) * lddp + j] = acc;
        if (want_db) dbp = fmaf(acc * s_sc[i], s_c[i * ps + j], dbp);
      }
      if (want_db) s_db[(tid >> 6) * h + j] = dbp;
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
    if (want_db) {                                        // the four waves' sums in wave order
      __syncthreads();
      for (int j = tid; j < h; j += 256) {
        const float v = (s_db[j] + s_db[h + j]) + (s_db[2 * h + j] + s_db[3 * h + j]);
        if (part) part[h * c + c + 2 + j] = v; else pp.db_relu[j] = v;
      }
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
    int q = 0;
    for (; q + 8 <= nblk; q += 8) {                        // eight slabs' loads in flight, added in workgroup order (as a plain
      float v[8];                                          // loop the 53 slabs of a config-3 batch were 53 L2 round trips in a row)
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[(int64_t)(q + u) * npart + idx];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; q < nblk; ++q) acc += slabs[(int64_t)q * npart + idx];
    if (idx < h * c) dw[idx] = acc;
    else if (idx < h * c + c) { if (db) db[idx - h * c] = acc; }
    else if (idx == h * c + c) loss_acc[0] = acc / denom;
    else if (idx == h * c + c + 1) loss_acc[1] = acc;
    else pp.db_relu[idx - (h * c + c + 2)] = acc;
  }
  if (tid == 0) *ticket = 0;                              // ready for the next launch (same stream: ordered)
}


template <bool STAGED, bool PARTS = false, int CT = 0>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ pooled, int64_t ldp,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   const float* __restrict__ y, int32_t b, int32_t h, int32_t c_rt,
                                                   float denom, float* __restrict__ probs,
                                                   float* __restrict__ loss_acc, float* __restrict__ dw,
                                                   float* __restrict__ db, float* __restrict__ dpooled, int64_t lddp,
                                                   float* __restrict__ slabs, int* __restrict__ ticket, PoolParts pp,
                                                   int from_logits) {
  extern __shared__ float s_head_dyn[];
  head_body<STAGED, PARTS, CT>(pooled, ldp, w, bias, y, b, h, c_rt, denom, probs, loss_acc, dw, db, dpooled, lddp, slabs, ticket,
                               pp, from_logits, s_head_dyn, blockIdx.x, gridDim.x);
}

inline size_t head_lds_floats(int32_t h, int32_t c, bool with_db_relu) {
  return (size_t)kHeadRows * (h + 1) + (size_t)h * c + (with_db_relu ? (size_t)kHeadRows * (h + 1) + 4 * (size_t)h : 0);
}

}  // namespace gcnx_head
#endif
