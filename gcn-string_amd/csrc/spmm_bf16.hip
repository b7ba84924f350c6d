// Aggregation with bf16 features ("fp32 and bf16 both reported"):
//     out[t, :] = bf16( act( sum_e vals[e] * h[colidx[e], :] + bias ) ),   h and out bf16 [n, f], fp32 accumulation.
// GCNConv.call / GeneralConv's aggregation (the ops behind gcn.py:334) on activations stored in bf16: half the feature
// bytes of the fp32 form (B_alg = 4 (n + 1) + 4 nnz (+ 4 nnz weighted) + 2 * 2 n f; 1.108 GB at config 3).  A measured
// variant next to the fp32 product path: the models keep fp32 activations (LOG.md section 7, item 7).
//
// One kernel, the row gather of csrc/fused.hip with 8 features per lane: a 512-thread workgroup owns 32 rows, stages their
// CSR entries in LDS as {row byte offset, weight}, f / 8 lanes cover a feature row with one 16-byte load (8 bf16), every
// row group walks its rows together, four entries each per trip, range-checked buffer loads (slots past a row's end fetch
// nothing).  bf16 -> fp32 is a shift / a mask per element; fp32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32).
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBRows = 32;        // rows per workgroup
constexpr int kBCap = 1024;       // CSR entries of a tile staged in LDS (the rest is read from global memory)

struct Acc8 { f32x2 v[4]; };

// acc += w * (the 8 bf16 values of q); element 2 i is the low half of word i
__device__ __forceinline__ void fma8(Acc8& a, float w, const u32x4 q) {
  const f32x2 ww = {w, w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 x = {__uint_as_float(q[i] << 16), __uint_as_float(q[i] & 0xffff0000u)};
    a.v[i] = __builtin_elementwise_fma(ww, x, a.v[i]);
  }
}

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 p = {(__bf16)lo, (__bf16)hi};                 // round to nearest even
  return __builtin_bit_cast(unsigned, p);
}

template <int F, bool WEIGHTED>
__global__ __launch_bounds__(512, 6) void spmm_bf16_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                                                           const float* __restrict__ vals, const uint16_t* __restrict__ h,
                                                           int64_t ldh, const float* __restrict__ bias, uint16_t* __restrict__ out,
                                                           int64_t ldo, int32_t n, int act) {
  constexpr int LPR = F / 8;                 // lanes per row
  constexpr int GW = 64 / LPR;               // row groups per wave
  constexpr int NG = 8 * GW;                 // row groups per workgroup
  constexpr int RPG = NG >= kBRows ? 1 : kBRows / NG;
  constexpr int U = 4;                       // entries per row per trip
  static_assert(F == 64 || F == 128 || F == 256, "feature width");
  __shared__ __attribute__((aligned(8))) int2 s_ent[kBCap + 2 * U];
  __shared__ int32_t s_rp[kBRows + 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = gcnx_xcd_remap(blockIdx.x, gridDim.x);
  const int r0 = t * kBRows, nr = min(n - r0, kBRows);
  if (tid <= nr) s_rp[tid] = rowptr[r0 + tid];
  const int e0 = rowptr[r0], e1 = rowptr[r0 + nr];
  const int staged = min(e1 - e0, kBCap);
  const unsigned ld2 = (unsigned)ldh * 2u;
  for (int i = tid; i < staged + 2 * U; i += 512) {       // the slack entries carry weight 0
    int2 en = make_int2(0, 0);
    if (i < staged) {
      en.x = (int)((unsigned)colidx[e0 + i] * ld2);
      en.y = WEIGHTED ? __float_as_int(vals[e0 + i]) : 0x3f800000;
    }
    s_ent[i] = en;
  }
  const int gid = wave * GW + lane / LPR, sub = lane % LPR;
  const __amdgpu_buffer_rsrc_t hr =
      __builtin_amdgcn_make_buffer_rsrc((void*)h, (short)0, (int)((unsigned)n * ld2), 0x00020000);
  __syncthreads();
  Acc8 acc[RPG];
  int ea[RPG], eb[RPG], ebf[RPG];
  int len = 0;
#pragma unroll
  for (int j = 0; j < RPG; ++j) {
    const int r = gid + j * NG;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j].v[i] = f32x2{0.f, 0.f};
    const bool live = r < nr && gid < kBRows;
    ea[j] = live ? s_rp[r] - e0 : 0;
    eb[j] = live ? s_rp[r + 1] - e0 : 0;
    ebf[j] = min(eb[j], kBCap);
    ea[j] = min(ea[j], kBCap);
    len = max(len, ebf[j] - ea[j]);
  }
  const unsigned sub16 = (unsigned)sub * 16u;
  for (int tt = 0; __builtin_amdgcn_ballot_w64(tt < len) != 0; tt += U) {
    u32x4 hv[RPG][U];
    float wv[RPG][U];
#pragma unroll
    for (int j = 0; j < RPG; ++j) {
      const int eb_ = min(ea[j] + tt, ebf[j]);          // a finished row stays at its end: the reads stay inside the slack
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = eb_ + u;
        const int2 en = s_ent[e];
        wv[j][u] = __int_as_float(en.y);
        hv[j][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(hr, e < ebf[j] ? (unsigned)en.x + sub16 : 0xFFFFFFF0u, 0, 0));
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // all loads of the trip go out before the first is consumed
#pragma unroll
    for (int j = 0; j < RPG; ++j)
#pragma unroll
      for (int u = 0; u < U; ++u) fma8(acc[j], wv[j][u], hv[j][u]);
  }
  if (e1 - e0 > kBCap) {       // uniform per workgroup, rare: entries beyond the staged ones, from global memory
#pragma unroll
    for (int j = 0; j < RPG; ++j)
      for (int e = max(eb[j] > 0 ? s_rp[gid + j * NG] - e0 : 0, kBCap); e < eb[j]; ++e) {
        const int c = colidx[e0 + e];
        const float v = WEIGHTED ? vals[e0 + e] : 1.0f;
        fma8(acc[j], v, __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(hr, (unsigned)c * ld2 + sub16, 0, 0)));
      }
  }
  if (gid >= kBRows) return;
  float b8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) b8[i] = bias ? bias[sub * 8 + i] : 0.f;
#pragma unroll
  for (int j = 0; j < RPG; ++j) {
    const int r = gid + j * NG;
    if (r >= nr) continue;
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float lo = acc[j].v[i][0] + b8[2 * i], hi = acc[j].v[i][1] + b8[2 * i + 1];
      if (act == GCNX_ACT_RELU) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
      o[i] = pack_bf16(lo, hi);
    }
    *reinterpret_cast<u32x4*>(out + (int64_t)(r0 + r) * ldo + sub * 8) = o;
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t count) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i + 1 < count) {
    *reinterpret_cast<unsigned*>(y + i) = pack_bf16(x[i], x[i + 1]);
  } else if (i < count) {
    y[i] = (uint16_t)(pack_bf16(x[i], 0.f) & 0xffffu);
  }
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const uint16_t* __restrict__ x, float* __restrict__ y, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) y[i] = __uint_as_float((unsigned)x[i] << 16);
}

}  // namespace

extern "C" {

int gcnx_spmm_csr_bf16(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const uint16_t* h,
                       int64_t ldh, const float* bias, uint16_t* out, int64_t ldo, int32_t n, int32_t f, int act) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "aggregation (bf16 features)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_spmm_csr_bf16: negative size");
  GCNX_REQUIRE(ctx, act == GCNX_ACT_NONE || act == GCNX_ACT_RELU, "gcnx_spmm_csr_bf16: activation %d not supported here", act);
  if (n == 0 || f == 0) return GCNX_OK;
  if (!(f == 64 || f == 128 || f == 256) || (uint64_t)n * (uint64_t)ldh * 2u >= 0xFFFFFFF0ull)
    return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_spmm_csr_bf16: needs f in {64, 128, 256} and n * ldh * 2 < 2^32 (got n=%d f=%d)", n, f);
  GCNX_REQUIRE(ctx, rowptr && colidx && h && out, "gcnx_spmm_csr_bf16: NULL pointer");
  GCNX_REQUIRE(ctx, ldh >= f && ldo >= f && ldh % 8 == 0 && ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(h) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               "gcnx_spmm_csr_bf16: operands must be 16-byte aligned with leading dimensions in multiples of 8 elements");
  GCNX_REQUIRE(ctx, (const void*)h != (const void*)out, "gcnx_spmm_csr_bf16: in-place aggregation is not possible");
  const int tiles = gcnx_cdiv(n, kBRows);
#define GCNX_B16_LAUNCH(F_)                                                                                                   \
  do {                                                                                                                        \
    if (vals) hipLaunchKernelGGL((spmm_bf16_kernel<F_, true>), dim3(tiles), dim3(512), 0, ctx->stream, rowptr, colidx, vals, h, ldh, \
                                 bias, out, ldo, n, act);                                                                     \
    else hipLaunchKernelGGL((spmm_bf16_kernel<F_, false>), dim3(tiles), dim3(512), 0, ctx->stream, rowptr, colidx, vals, h, ldh,    \
                            bias, out, ldo, n, act);                                                                          \
  } while (0)
  if (f == 64) GCNX_B16_LAUNCH(64);
  else if (f == 128) GCNX_B16_LAUNCH(128);
  else GCNX_B16_LAUNCH(256);
#undef GCNX_B16_LAUNCH
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_f32_to_bf16(gcnx_ctx* ctx, const float* x, uint16_t* y, int64_t count) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, count >= 0, "gcnx_f32_to_bf16: negative size");
  if (count == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, x && y && (reinterpret_cast<uintptr_t>(y) & 3) == 0, "gcnx_f32_to_bf16: NULL or unaligned pointer");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(gcnx_cdiv((count + 1) / 2, 256)), dim3(256), 0, ctx->stream, x, y, count);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_bf16_to_f32(gcnx_ctx* ctx, const uint16_t* x, float* y, int64_t count) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, count >= 0, "gcnx_bf16_to_f32: negative size");
  if (count == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, x && y, "gcnx_bf16_to_f32: NULL pointer");
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(gcnx_cdiv(count, 256)), dim3(256), 0, ctx->stream, x, y, count);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // extern "C"
