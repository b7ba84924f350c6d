// K2/K3: CSR neighbour aggregation  out[t,:] = act(sum_e vals[e] * h[colidx[e],:] + bias).
//
// Replaces tf.sparse.sparse_dense_matmul (GCNConv.call) and gather + unsorted_segment_sum
// (GeneralConv.propagate) reached from model(inputs) at src/scripts/gcn.py:334/351.  The
// [nnz,F] message tensor TensorFlow materialises never exists here.
//
// HBM-bound: algorithmic bytes per launch = 4(N+1) + 4 nnz (+4 nnz weighted) + 2*4*N*F.
//
// Kernel "rows": one 256-thread workgroup owns a contiguous chunk of rows.  The chunk's CSR
// segment (column indices, values, row pointers) is contiguous in memory and is staged into LDS
// with coalesced loads, so the per-row work has a single dependent HBM/L2 latency (the feature
// gather) instead of two.  Inside a wave, LPR = F/4 lanes cover one feature row with 16-byte
// loads (fully coalesced: 64 lanes x 16 B = 1 KiB for F = 256); when F < 256 the wave's
// 64/LPR lane groups take different neighbours of the same row and are combined with
// __shfl_xor at the end.  Chunk ids are remapped so each XCD (private 4 MiB L2) walks one
// contiguous range of rows: in a disjoint (block-diagonal) batch the rows a chunk gathers lie
// in the same graph, hence in the same L2.
#include "common.h"

namespace {

constexpr int kRowsPerChunk = 32;    // rows per workgroup
constexpr int kStageCap = 2048;      // CSR entries staged in LDS per chunk (overflow -> global)

__device__ __forceinline__ float4 f4_fma(float v, float4 h, float4 a) {
  a.x = fmaf(v, h.x, a.x); a.y = fmaf(v, h.y, a.y); a.z = fmaf(v, h.z, a.z); a.w = fmaf(v, h.w, a.w);
  return a;
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

template <int LPR, bool WEIGHTED>
__global__ __launch_bounds__(256) void spmm_rows_kernel(const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colidx,
                                                        const float* __restrict__ vals,
                                                        const float* __restrict__ h, int64_t ldh,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int64_t ldo, int32_t n, int32_t f, int32_t col0, int act,
                                                        int nchunks) {
  constexpr int G = 64 / LPR;  // neighbour groups per wave
  __shared__ int32_t s_col[kStageCap];
  __shared__ float s_val[WEIGHTED ? kStageCap : 1];
  __shared__ int32_t s_rp[kRowsPerChunk + 1];

  const int chunk = gcnx_xcd_remap(blockIdx.x, nchunks);
  const int r0 = chunk * kRowsPerChunk;
  const int r1 = min(n, r0 + kRowsPerChunk);
  const int tid = threadIdx.x;
  if (tid <= r1 - r0) s_rp[tid] = rowptr[r0 + tid];
  const int e0 = rowptr[r0];
  const int e1 = rowptr[r1];
  const int staged = min(e1 - e0, kStageCap);
  for (int i = tid; i < staged; i += 256) {
    s_col[i] = colidx[e0 + i];
    if (WEIGHTED) s_val[i] = vals[e0 + i];
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane / LPR, sub = lane % LPR;
  const int c = col0 + sub * 4;          // first of this lane's 4 columns
  const bool col_ok = c < f;             // f % 4 == 0 is guaranteed by the dispatcher
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias && col_ok) bv = *reinterpret_cast<const float4*>(bias + c);

  for (int r = r0 + wave; r < r1; r += 4) {
    const int a = s_rp[r - r0] - e0, b = s_rp[r - r0 + 1] - e0;  // chunk-relative entry range
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col_ok) {
#pragma unroll 4
      for (int e = a + g; e < b; e += G) {
        int cidx;
        float v = 1.0f;
        if (e < kStageCap) {
          cidx = s_col[e];
          if (WEIGHTED) v = s_val[e];
        } else {
          cidx = colidx[e0 + e];
          if (WEIGHTED) v = vals[e0 + e];
        }
        const float4 hv = *reinterpret_cast<const float4*>(h + (int64_t)cidx * ldh + c);
        acc = WEIGHTED ? f4_fma(v, hv, acc) : f4_add(acc, hv);
      }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      acc.x += __shfl_xor(acc.x, off);
      acc.y += __shfl_xor(acc.y, off);
      acc.z += __shfl_xor(acc.z, off);
      acc.w += __shfl_xor(acc.w, off);
    }
    if (g == 0 && col_ok) {
      acc = f4_add(acc, bv);
      if (act == GCNX_ACT_RELU) {
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
      }
      *reinterpret_cast<float4*>(out + (int64_t)r * ldo + c) = acc;
    }
  }
}

// Fallback for widths / strides that are not multiples of 4 floats: one lane per column.
__global__ __launch_bounds__(256) void spmm_scalar_kernel(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ colidx,
                                                          const float* __restrict__ vals,
                                                          const float* __restrict__ h, int64_t ldh,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int64_t ldo, int32_t n, int32_t f, int act) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= n) return;
  const int a = rowptr[r], b = rowptr[r + 1];
  for (int c = lane; c < f; c += 64) {
    float acc = 0.f;
    for (int e = a; e < b; ++e) {
      const float v = vals ? vals[e] : 1.0f;
      acc = fmaf(v, h[(int64_t)colidx[e] * ldh + c], acc);
    }
    if (bias) acc += bias[c];
    if (act == GCNX_ACT_RELU) acc = fmaxf(acc, 0.f);
    out[(int64_t)r * ldo + c] = acc;
  }
}

template <int LPR>
void launch_rows(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                 int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int act) {
  const int nchunks = gcnx_cdiv(n, kRowsPerChunk);
  const int span = LPR * 4;
  for (int col0 = 0; col0 < f; col0 += span) {
    if (vals)
      hipLaunchKernelGGL((spmm_rows_kernel<LPR, true>), dim3(nchunks), dim3(256), 0, ctx->stream, rowptr, colidx,
                         vals, h, ldh, bias, out, ldo, n, f, col0, act, nchunks);
    else
      hipLaunchKernelGGL((spmm_rows_kernel<LPR, false>), dim3(nchunks), dim3(256), 0, ctx->stream, rowptr, colidx,
                         vals, h, ldh, bias, out, ldo, n, f, col0, act, nchunks);
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int gcnx_spmm_csr(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                             const float* h, int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n,
                             int32_t f, int act, const int32_t* block_ptr, int32_t nblocks) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_spmm_csr: negative size");
  GCNX_REQUIRE(ctx, act == GCNX_ACT_NONE || act == GCNX_ACT_RELU, "gcnx_spmm_csr: activation %d not supported here", act);
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, rowptr && colidx && h && out, "gcnx_spmm_csr: NULL pointer");
  GCNX_REQUIRE(ctx, ldh >= f && ldo >= f, "gcnx_spmm_csr: leading dimension smaller than f=%d", f);
  GCNX_REQUIRE(ctx, h != out, "gcnx_spmm_csr: in-place aggregation is not possible");
  (void)block_ptr;
  (void)nblocks;
  const bool vec = (f % 4 == 0) && (ldh % 4 == 0) && (ldo % 4 == 0) && aligned16(h) && aligned16(out) &&
                   (!bias || aligned16(bias));
  if (!vec) {
    hipLaunchKernelGGL(spmm_scalar_kernel, dim3(gcnx_cdiv(n, 4)), dim3(256), 0, ctx->stream, rowptr, colidx, vals,
                       h, ldh, bias, out, ldo, n, f, act);
    GCNX_LAUNCH_OK(ctx);
    return GCNX_OK;
  }
  const int lanes = f / 4;
  if (lanes > 32) launch_rows<64>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  else if (lanes > 16) launch_rows<32>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  else if (lanes > 8) launch_rows<16>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  else if (lanes > 4) launch_rows<8>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  else launch_rows<4>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}
