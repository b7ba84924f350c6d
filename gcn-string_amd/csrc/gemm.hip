// K1 and its gradients: the dense weight GEMMs of GCNConv / GeneralConv / Dense.
//   forward  out = act(X W + b)          (MatMul + BiasAdd, gcn.py:334)
//   dX       dX  = dH W^T  [* relu mask] (MatMul grad wrt input,  gcn.py:337)
//   dW       dW  = X^T dH                (MatMul grad wrt kernel, gcn.py:337; K = N rows)
//
// GCNX_PREC_F32: v_mfma_f32_32x32x2_f32 -- exact fp32, bit-for-bit a k-ordered fmaf chain, which
// is what the 1e-4 parity configuration (BASELINE cfg2) uses.
//
// One kernel template covers the three operand layouts.  A 256-thread workgroup (4 waves, one
// 32x32 MFMA accumulator tile each) owns a 64x64 output tile and walks K in steps of 32 through
// LDS.  LDS images are k-major ([k][m] and [k][n]) so that the MFMA operand read -- lane l takes
// element [k = 2*kk + (l>>5)][l & 31] -- is 32 consecutive dwords per half-wave: conflict-free.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32, LD = 68;  // LD*4 B = 272 B keeps rows 16-B aligned

struct Epilogue {
  const float* bias;    // [Nc] or null
  const float* alpha;   // PReLU slope [Nc] or null
  const float* mask;    // relu mask source (same shape as C) or null: C *= (mask > 0)
  int64_t ldmask;
  int act;
  int accumulate;       // C += result
};

// Loads one 64x32 (or 32x64) operand tile into a k-major LDS image s[k][LD].
//   KCONTIG:  element(i,k) = p[i*ld + k]   (row index is the M/N index; k contiguous)
//   !KCONTIG: element(i,k) = p[k*ld + i]   (k is the row index; M/N contiguous)
template <bool KCONTIG>
__device__ __forceinline__ void load_tile(const float* __restrict__ p, int64_t ld, int64_t i0, int64_t i_end,
                                          int64_t k0, int64_t k_end, float (*s)[LD], int tid, bool vec_ok) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + 256 * q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KCONTIG) {
      const int i = idx >> 3, k = (idx & 7) * 4;
      const int64_t gi = i0 + i, gk = k0 + k;
      if (gi < i_end) {
        const float* src = p + gi * ld + gk;
        if (vec_ok && gk + 3 < k_end) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          if (gk + 0 < k_end) v.x = src[0];
          if (gk + 1 < k_end) v.y = src[1];
          if (gk + 2 < k_end) v.z = src[2];
          if (gk + 3 < k_end) v.w = src[3];
        }
      }
      s[k + 0][i] = v.x; s[k + 1][i] = v.y; s[k + 2][i] = v.z; s[k + 3][i] = v.w;
    } else {
      const int k = idx >> 4, i = (idx & 15) * 4;
      const int64_t gi = i0 + i, gk = k0 + k;
      if (gk < k_end) {
        const float* src = p + gk * ld + gi;
        if (vec_ok && gi + 3 < i_end) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          if (gi + 0 < i_end) v.x = src[0];
          if (gi + 1 < i_end) v.y = src[1];
          if (gi + 2 < i_end) v.z = src[2];
          if (gi + 3 < i_end) v.w = src[3];
        }
      }
      *reinterpret_cast<float4*>(&s[k][i]) = v;
    }
  }
}

// C[M,Nc] = op(A) op(B) over k in [kz*kchunk, min(K,(kz+1)*kchunk)); blockIdx.z = kz (split-K).
// With split-K (gridDim.z > 1) the raw partial tile goes to c + kz*M*ldc (a [S][M][ldc] slab).
template <bool A_KCONTIG, bool B_KCONTIG>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ a, int64_t lda,
                                                       const float* __restrict__ b, int64_t ldb,
                                                       float* __restrict__ c, int64_t ldc, int64_t M, int32_t Nc,
                                                       int64_t K, int64_t kchunk, Epilogue ep, int vec_a, int vec_b) {
  __shared__ __attribute__((aligned(16))) float As[BK][LD];
  __shared__ __attribute__((aligned(16))) float Bs[BK][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM;
  const int64_t n0 = (int64_t)blockIdx.x * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = min(K, kbeg + kchunk);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const int fr = lane & 31, fk = lane >> 5;
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    load_tile<A_KCONTIG>(a, lda, m0, M, k0, kend, As, tid, vec_a);
    load_tile<B_KCONTIG>(b, ldb, n0, Nc, k0, kend, Bs, tid, vec_b);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float av = As[2 * kk + fk][wm * 32 + fr];
      const float bv = Bs[2 * kk + fk][wn * 32 + fr];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    __syncthreads();
  }

  // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
  const int64_t col = n0 + wn * 32 + fr;
  if (col >= Nc) return;
  float* cz = c + (gridDim.z > 1 ? (int64_t)blockIdx.z * M * ldc : 0);
  const float bias = (ep.bias ? ep.bias[col] : 0.f);
  const float alpha = (ep.alpha ? ep.alpha[col] : 0.f);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
    if (row >= M) continue;
    float v = acc[r] + bias;
    if (ep.act == GCNX_ACT_RELU) v = fmaxf(v, 0.f);
    else if (ep.act == GCNX_ACT_PRELU) v = v > 0.f ? v : alpha * v;
    if (ep.mask) v = ep.mask[row * ep.ldmask + col] > 0.f ? v : 0.f;
    float* dst = cz + row * ldc + col;
    if (ep.accumulate) v += *dst;
    *dst = v;
  }
}

// Second stage of the deterministic split-K: out[i] = sum_s part[s][i].  Block = 64 outputs x 4
// split groups; each group sums its splits in ascending order, the 4 group sums are combined in
// a fixed order -- the result does not depend on scheduling.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int64_t slab,
                                                            int nsplit, float* __restrict__ out, int64_t total) {
  __shared__ float s[4][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + el;
  float acc = 0.f;
  if (i < total) {
    const int per = (nsplit + 3) / 4;
    const int z0 = grp * per, z1 = min(nsplit, z0 + per);
#pragma unroll 4
    for (int z = z0; z < z1; ++z) acc += part[(int64_t)z * slab + i];
  }
  s[grp][el] = acc;
  __syncthreads();
  if (grp == 0 && i < total) out[i] = (s[0][el] + s[1][el]) + (s[2][el] + s[3][el]);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int gcnx_colsum(gcnx_ctx* ctx, const float* x, int64_t ldx, int64_t n, int32_t f, float* out);  // reduce.hip

extern "C" {

int gcnx_gemm(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* w, const float* bias, float* out,
              int64_t ldo, int64_t n, int32_t fi, int32_t fo, int prec, int act, const float* alpha) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm: negative size");
  GCNX_REQUIRE(ctx, act >= GCNX_ACT_NONE && act <= GCNX_ACT_PRELU, "gcnx_gemm: unknown activation %d", act);
  GCNX_REQUIRE(ctx, act != GCNX_ACT_PRELU || alpha, "gcnx_gemm: PReLU needs alpha");
  if (prec != GCNX_PREC_F32)
    return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_gemm: precision %d not built yet (only GCNX_PREC_F32)", prec);
  if (n == 0 || fo == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, x && w && out, "gcnx_gemm: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= fi && ldo >= fo, "gcnx_gemm: leading dimension too small");
  Epilogue ep{bias, act == GCNX_ACT_PRELU ? alpha : nullptr, nullptr, 0, act, 0};
  dim3 grid(gcnx_cdiv(fo, BN), gcnx_cdiv(n, BM), 1);
  const int va = al16(x) && ldx % 4 == 0, vb = al16(w) && fo % 4 == 0;
  hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, ctx->stream, x, ldx, w, (int64_t)fo, out,
                     ldo, n, fo, (int64_t)fi, (int64_t)fi + BK, ep, va, vb);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_gemm_dx(gcnx_ctx* ctx, const float* dh, int64_t lddh, const float* w, float* dx, int64_t lddx, int64_t n,
                 int32_t fi, int32_t fo, int prec, int accumulate, const float* y_mask, int64_t ldy, float* db) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm_dx: negative size");
  if (prec != GCNX_PREC_F32)
    return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_gemm_dx: precision %d not built yet (only GCNX_PREC_F32)", prec);
  GCNX_REQUIRE(ctx, !(accumulate && (y_mask || db)), "gcnx_gemm_dx: accumulate cannot be combined with mask/db");
  if (n == 0 || fi == 0) {
    if (db && fi > 0) GCNX_HIP(ctx, hipMemsetAsync(db, 0, (size_t)fi * 4, ctx->stream));
    return GCNX_OK;
  }
  GCNX_REQUIRE(ctx, dh && w && dx, "gcnx_gemm_dx: NULL pointer");
  GCNX_REQUIRE(ctx, lddh >= fo && lddx >= fi && (!y_mask || ldy >= fi), "gcnx_gemm_dx: leading dimension too small");
  // dX[n, i] = sum_o dH[n, o] * W[i, o]:  A = dH (k contiguous), B[k=o][j=i] = W[i*fo + o] (k contiguous).
  Epilogue ep{nullptr, nullptr, y_mask, ldy, GCNX_ACT_NONE, accumulate};
  dim3 grid(gcnx_cdiv(fi, BN), gcnx_cdiv(n, BM), 1);
  const int va = al16(dh) && lddh % 4 == 0, vb = al16(w) && fo % 4 == 0;
  hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, ctx->stream, dh, lddh, w, (int64_t)fo, dx,
                     lddx, n, fi, (int64_t)fo, (int64_t)fo + BK, ep, va, vb);
  GCNX_LAUNCH_OK(ctx);
  if (db) return gcnx_colsum(ctx, dx, lddx, n, fi, db);
  return GCNX_OK;
}

int gcnx_gemm_dw(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, float* dw, int64_t n,
                 int32_t fi, int32_t fo, int prec) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm_dw: negative size");
  if (prec != GCNX_PREC_F32)
    return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_gemm_dw: precision %d not built yet (only GCNX_PREC_F32)", prec);
  if (fi == 0 || fo == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, dw != nullptr, "gcnx_gemm_dw: dw is NULL");
  if (n == 0) {
    GCNX_HIP(ctx, hipMemsetAsync(dw, 0, (size_t)fi * fo * 4, ctx->stream));
    return GCNX_OK;
  }
  GCNX_REQUIRE(ctx, x && dh, "gcnx_gemm_dw: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= fi && lddh >= fo, "gcnx_gemm_dw: leading dimension too small");
  // dW[i, o] = sum_n X[n, i] * dH[n, o]: A[i][k=n] = X[n*ldx + i], B[k=n][o] = dH[n*lddh + o].
  const int tiles = gcnx_cdiv(fi, BM) * gcnx_cdiv(fo, BN);
  int nsplit = (int)((2LL * ctx->num_cus + tiles - 1) / tiles);   // ~2 workgroups per CU
  const int64_t ksteps = (n + BK - 1) / BK;
  if (nsplit > ksteps) nsplit = (int)ksteps;
  if (nsplit < 1) nsplit = 1;
  const int64_t kchunk = ((ksteps + nsplit - 1) / nsplit) * BK;
  nsplit = (int)((n + kchunk - 1) / kchunk);
  Epilogue ep{nullptr, nullptr, nullptr, 0, GCNX_ACT_NONE, 0};
  const int va = al16(x) && ldx % 4 == 0, vb = al16(dh) && lddh % 4 == 0;
  float* target = dw;
  if (nsplit > 1) {
    int rc = gcnx_ws_reserve(ctx, (size_t)nsplit * fi * fo * sizeof(float));
    if (rc) return rc;
    target = (float*)ctx->ws;
  }
  dim3 grid(gcnx_cdiv(fo, BN), gcnx_cdiv(fi, BM), nsplit);
  hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, ctx->stream, x, ldx, dh, lddh, target,
                     (int64_t)fo, (int64_t)fi, fo, n, kchunk, ep, va, vb);
  GCNX_LAUNCH_OK(ctx);
  if (nsplit > 1) {
    const int64_t total = (int64_t)fi * fo;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(gcnx_cdiv(total, 64)), dim3(256), 0, ctx->stream,
                       (const float*)ctx->ws, total, nsplit, dw, total);
    GCNX_LAUNCH_OK(ctx);
  }
  return GCNX_OK;
}

}  // extern "C"
