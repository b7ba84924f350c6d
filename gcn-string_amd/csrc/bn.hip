// Keras BatchNormalization (axis -1, momentum 0.99, eps 1e-3) + PReLU around the Dense layers of
// Spektral's MLP / GeneralConv -- what GeneralGNN(…) at src/scripts/gcn.py:320 stacks (SURVEY
// 8.A.3-8.A.5).  Training mode normalises with the batch mean and the BIASED batch variance over
// the N rows and moves the running statistics; inference uses the running statistics.
//
// All kernels are HBM-bound streaming passes over [N, F] (float4 lanes x 16 row groups, the
// column reductions two-stage and atomics-free, hence bitwise reproducible):
//   forward   z -> (sum z, sum z^2)           1 read
//             y = act(gamma*(z-mu)*inv+beta)  1 read + 1 write
//   backward  (sum dzb, sum dzb*xhat, sum dy*min(zb,0))   2 reads        dzb = dy * act'(zb)
//             dz = gamma*inv*(dzb - s1/n - xhat*s2/n)     2 reads + 1 write
// xhat and zb are recomputed from z and the saved (mu, inv): nothing but z itself is kept for
// the backward pass.
#include <type_traits>

#include "common.h"

namespace {

constexpr int kRows = 256;   // rows per first-stage workgroup

__device__ __forceinline__ float4 ld4g(const float* p, bool vec, int valid) {
  if (vec) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid > 0) v.x = p[0];
  if (valid > 1) v.y = p[1];
  if (valid > 2) v.z = p[2];
  if (valid > 3) v.w = p[3];
  return v;
}
__device__ __forceinline__ void st4g(float* p, float4 v, bool vec, int valid) {
  if (vec) { *reinterpret_cast<float4*>(p) = v; return; }
  if (valid > 0) p[0] = v.x;
  if (valid > 1) p[1] = v.y;
  if (valid > 2) p[2] = v.z;
  if (valid > 3) p[3] = v.w;
}
__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// NS running column sums per lane; each comes out of `terms(row values) -> float4[NS]`.
// part layout: [chunk][NS][f].
// `fast` (uniform per workgroup: 16-byte-aligned operands, a full 64-column tile) selects the instance whose `terms`
// loads are plain float4 loads, two rows at a time: behind ld4g's vector-or-scalar test every load is waited for on
// its own (hipcc), so the generic loop has one load in flight where this one has all of two rows'.
template <int NS, class F>
__device__ __forceinline__ void colsums(int64_t n, int32_t f, float* __restrict__ part, bool fast, F terms) {
  __shared__ float4 s[NS][16][16];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cl * 4;
  const int valid = f - c;
  const int64_t r0 = (int64_t)blockIdx.y * kRows;
  const int64_t r1 = min(n, r0 + (int64_t)kRows);
  float4 acc[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) acc[k] = f4(0.f);
  if (fast) {
    int64_t r = r0 + rg;
    for (; r + 16 < r1; r += 32) {
      float4 t0[NS], t1[NS];
      terms(r, c, 4, t0, std::true_type{});
      terms(r + 16, c, 4, t1, std::true_type{});
#pragma unroll
      for (int k = 0; k < NS; ++k) acc[k] = add4(add4(acc[k], t0[k]), t1[k]);      // row order, as below
    }
    for (; r < r1; r += 16) {
      float4 t[NS];
      terms(r, c, 4, t, std::true_type{});
#pragma unroll
      for (int k = 0; k < NS; ++k) acc[k] = add4(acc[k], t[k]);
    }
  } else if (valid > 0) {
#pragma unroll 2
    for (int64_t r = r0 + rg; r < r1; r += 16) {
      float4 t[NS];
      terms(r, c, valid, t, std::false_type{});
#pragma unroll
      for (int k = 0; k < NS; ++k) acc[k] = add4(acc[k], t[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) s[k][rg][cl] = acc[k];
  __syncthreads();
  if (rg < NS && valid > 0) {
    float4 t = s[rg][0][cl];
    for (int q = 1; q < 16; ++q) t = add4(t, s[rg][q][cl]);
    st4g(part + ((int64_t)blockIdx.y * NS + rg) * f + c, t, false, valid);
  }
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, int64_t ldz, int64_t n, int32_t f,
                                                       const float* __restrict__ shift, float* __restrict__ part, int vec) {
  const int c0 = blockIdx.x * 64 + (threadIdx.x & 15) * 4;
  float4 sh = f4(0.f);
  if (shift && f - c0 > 0) sh = ld4g(shift + c0, false, f - c0);
  colsums<2>(n, f, part, vec && (int)blockIdx.x * 64 + 64 <= f, [&](int64_t r, int c, int valid, float4 (&t)[2], auto fast) {
    float4 v = ld4g(z + r * ldz + c, decltype(fast)::value || (vec && valid >= 4), valid);
    v = make_float4(v.x - sh.x, v.y - sh.y, v.z - sh.z, v.w - sh.w);
    t[0] = v;
    t[1] = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
  });
}

// out[k][c] = sum over chunks of part[chunk][k][c].  Block = 64 outputs x 4 chunk groups: each group adds its
// chunks in ascending order, the four group sums are combined in a fixed order -- deterministic, and the chunk
// loop is a quarter as long with four loads in flight per thread (one thread per output walking all chunks took
// 16 us for 80 chunks: a serial chain of L2 round trips).  Up to three extra destinations receive copies of the
// rows k = 0, 1, 2 (the BatchNorm parameter gradients), which saves three device-to-device copies per layer.
__global__ __launch_bounds__(256) void part_reduce_kernel(const float* __restrict__ part, int nchunks, int ns, int32_t f,
                                                          float* __restrict__ out, float* __restrict__ o0,
                                                          float* __restrict__ o1, float* __restrict__ o2) {
  __shared__ float sm[4][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + el;
  const int total = ns * f;
  float acc = 0.f;
  if (i < total) {
    const int per = (nchunks + 3) / 4;
    const int c0 = grp * per, c1 = min(nchunks, c0 + per);
#pragma unroll 4
    for (int ch = c0; ch < c1; ++ch) acc += part[(int64_t)ch * total + i];
  }
  sm[grp][el] = acc;
  __syncthreads();
  if (grp == 0 && i < total) {
    const float v = (sm[0][el] + sm[1][el]) + (sm[2][el] + sm[3][el]);
    out[i] = v;
    const int k = i / f, c = i - k * f;
    float* extra = k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : nullptr));
    if (extra) extra[c] = v;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ sums, float count, int32_t f,
                                                          float momentum, float eps, const float* __restrict__ shift,
                                                          float* __restrict__ mean,
                                                          float* __restrict__ inv, float* __restrict__ moving_mean,
                                                          float* __restrict__ moving_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= f) return;
  float m, v;
  if (sums) {
    const float d = sums[c] / count;               // mean of (z - shift)
    m = (shift ? shift[c] : 0.f) + d;
    v = fmaxf(sums[f + c] / count - d * d, 0.f);   // biased variance (tf.nn.moments)
    if (moving_mean) {
      moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * m;
      moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * v;
    }
  } else {
    m = moving_mean[c];
    v = moving_var[c];
  }
  mean[c] = m;
  inv[c] = 1.0f / sqrtf(v + eps);
}

// part_reduce_kernel (ns = 2) and bn_finalize_kernel in one launch: block = 64 columns x 4 chunk groups; the
// same summation order and the same finalisation arithmetic, so gcnx_bn_moments == gcnx_bn_stats +
// gcnx_bn_finalize bit for bit, with two launches fewer per pass.
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const float* __restrict__ part, int nchunks, float count,
                                                                 int32_t f, float momentum, float eps,
                                                                 const float* __restrict__ shift, float* __restrict__ mean,
                                                                 float* __restrict__ inv, float* __restrict__ moving_mean,
                                                                 float* __restrict__ moving_var) {
  __shared__ float sm[2][4][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + el;
  float a0 = 0.f, a1 = 0.f;
  if (c < f) {
    const int per = (nchunks + 3) / 4;
    const int c0 = grp * per, c1 = min(nchunks, c0 + per);
#pragma unroll 4
    for (int ch = c0; ch < c1; ++ch) {
      a0 += part[(int64_t)ch * 2 * f + c];
      a1 += part[(int64_t)ch * 2 * f + f + c];
    }
  }
  sm[0][grp][el] = a0;
  sm[1][grp][el] = a1;
  __syncthreads();
  if (grp == 0 && c < f) {
    const float s0 = (sm[0][0][el] + sm[0][1][el]) + (sm[0][2][el] + sm[0][3][el]);
    const float s1 = (sm[1][0][el] + sm[1][1][el]) + (sm[1][2][el] + sm[1][3][el]);
    const float d = s0 / count;                      // mean of (z - shift)
    const float m = (shift ? shift[c] : 0.f) + d;
    const float v = fmaxf(s1 / count - d * d, 0.f);  // biased variance (tf.nn.moments)
    if (moving_mean) {
      moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * m;
      moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * v;
    }
    mean[c] = m;
    inv[c] = 1.0f / sqrtf(v + eps);
  }
}

// gcnx_bn_moments for a batch of at most kRows rows (the post-MLP of GeneralGNN runs on one row per graph: 32 rows at an
// E. coli-sized batch) in ONE launch: a workgroup owns 64 columns through both passes -- column sums, mean, column sums of
// the centred data, variance -- with the arithmetic of the four-launch sequence in the same order (one chunk: the chunk
// reduction adds zeros), so the results are the same bits.  `part` is scratch for one chunk ([2][f]).
__global__ __launch_bounds__(256) void bn_moments_small_kernel(const float* __restrict__ z, int64_t ldz, int64_t n, int32_t f,
                                                               float momentum, float eps, float* __restrict__ part,
                                                               float* __restrict__ mean, float* __restrict__ inv,
                                                               float* __restrict__ moving_mean, float* __restrict__ moving_var, int vec) {
  __shared__ float s_mean[64];
  const int c0 = blockIdx.x * 64 + (threadIdx.x & 15) * 4;
  const bool fast = vec && (int)blockIdx.x * 64 + 64 <= f;
  const float count = (float)n;
  colsums<2>(n, f, part, fast, [&](int64_t r, int c, int valid, float4 (&t)[2], auto fs) {
    const float4 v = ld4g(z + r * ldz + c, decltype(fs)::value || (vec && valid >= 4), valid);
    t[0] = v;
    t[1] = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
  });
  __syncthreads();                                   // this workgroup's columns of `part` were written by its own threads
  const int el = threadIdx.x, c = blockIdx.x * 64 + el;
  if (el < 64) s_mean[el] = c < f ? 0.f + ((part[c] + 0.f) + (0.f + 0.f)) / count : 0.f;      // pass 1 of bn_reduce_finalize_kernel
  __syncthreads();
  float4 sh = f4(0.f);
  if (f - c0 > 0) {
    const int q = (threadIdx.x & 15) * 4;
    sh = make_float4(s_mean[q], s_mean[q + 1], s_mean[q + 2], s_mean[q + 3]);
  }
  colsums<2>(n, f, part, fast, [&](int64_t r, int c_, int valid, float4 (&t)[2], auto fs) {
    float4 v = ld4g(z + r * ldz + c_, decltype(fs)::value || (vec && valid >= 4), valid);
    v = make_float4(v.x - sh.x, v.y - sh.y, v.z - sh.z, v.w - sh.w);
    t[0] = v;
    t[1] = make_float4(v.x * v.x, v.y * v.y, v.z * v.z, v.w * v.w);
  });
  __syncthreads();
  if (el < 64 && c < f) {
    const float s0 = (part[c] + 0.f) + (0.f + 0.f), s1 = (part[f + c] + 0.f) + (0.f + 0.f);
    const float d = s0 / count;
    const float m = s_mean[el] + d;
    const float v = fmaxf(s1 / count - d * d, 0.f);
    if (moving_mean) {
      moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * m;
      moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * v;
    }
    mean[c] = m;
    inv[c] = 1.0f / sqrtf(v + eps);
  }
}

__device__ __forceinline__ float prelu(float x, float a) { return x > 0.f ? x : a * x; }
// The activation's input, ONE expression for the forward and the backward kernels: zb = fma(z - mu, gamma * inv, beta) with
// the scale rounded to fp32 first.  The backward pass picks the PReLU / ReLU branch from the zb it recomputes, so it must be
// the very number the forward pass fed the activation -- with two different roundings (gamma * ((z - mu) * inv) + beta
// there) a pre-activation within an ulp of zero could take one branch forward and the other backward.  It also makes the
// branch reproducible on the host from (z, mu, inv, gamma, beta): sign(zb) = sign of the exact (z - mu) * sc + beta.
__device__ __forceinline__ float bn_zb(float z, float mu, float sc, float be) { return __builtin_fmaf(z - mu, sc, be); }

__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ z, int64_t ldz, int64_t n, int32_t f,
                                                     const float* __restrict__ mean, const float* __restrict__ inv,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int act, const float* __restrict__ alpha, float* __restrict__ y,
                                                     int64_t ldy, int vec) {
  const int cl = threadIdx.x & 63;
  const int c = blockIdx.x * 256 + cl * 4;
  const int valid = f - c;
  if (valid <= 0) return;
  const bool v4 = vec && valid >= 4;
  const float4 mu = ld4g(mean + c, false, valid), iv = ld4g(inv + c, false, valid);
  const float4 ga = ld4g(gamma + c, false, valid), be = ld4g(beta + c, false, valid);
  float4 al = f4(0.f);
  if (act == GCNX_ACT_PRELU) al = ld4g(alpha + c, false, valid);
  const float4 sc = make_float4(ga.x * iv.x, ga.y * iv.y, ga.z * iv.z, ga.w * iv.w);
  auto apply = [&](float4 v) {
    float4 o = make_float4(bn_zb(v.x, mu.x, sc.x, be.x), bn_zb(v.y, mu.y, sc.y, be.y), bn_zb(v.z, mu.z, sc.z, be.z),
                           bn_zb(v.w, mu.w, sc.w, be.w));
    if (act == GCNX_ACT_RELU) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
    else if (act == GCNX_ACT_PRELU) o = make_float4(prelu(o.x, al.x), prelu(o.y, al.y), prelu(o.z, al.z), prelu(o.w, al.w));
    return o;
  };
  const int64_t step = (int64_t)gridDim.y * 4;
  int64_t r = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (vec && (int)blockIdx.x * 256 + 256 <= f) {           // aligned full tile (uniform): two rows' loads in flight
    for (; r + step < n; r += 2 * step) {
      const float4 v0 = *reinterpret_cast<const float4*>(z + r * ldz + c);
      const float4 v1 = *reinterpret_cast<const float4*>(z + (r + step) * ldz + c);
      *reinterpret_cast<float4*>(y + r * ldy + c) = apply(v0);
      *reinterpret_cast<float4*>(y + (r + step) * ldy + c) = apply(v1);
    }
  }
  for (; r < n; r += step) st4g(y + r * ldy + c, apply(ld4g(z + r * ldz + c, v4, valid)), v4, valid);
}

struct BnCols { float4 mu, iv, ga, be, al; };

__device__ __forceinline__ BnCols bn_cols(const float* mean, const float* inv, const float* gamma, const float* beta,
                                          const float* alpha, int act, int c, int valid) {
  BnCols q;
  q.mu = ld4g(mean + c, false, valid); q.iv = ld4g(inv + c, false, valid);
  q.ga = ld4g(gamma + c, false, valid); q.be = ld4g(beta + c, false, valid);
  q.al = act == GCNX_ACT_PRELU ? ld4g(alpha + c, false, valid) : f4(0.f);
  return q;
}

// dzb = dy * act'(zb), zb = gamma*xhat + beta, xhat = (z - mu)*inv.  One component.
__device__ __forceinline__ void bn_bwd_terms(float dy, float z, float mu, float iv, float ga, float be, float al, int act,
                                             float& dzb, float& xhat, float& dalpha) {
  xhat = (z - mu) * iv;
  const float zb = bn_zb(z, mu, ga * iv, be);          // the forward pass's number (bn_act_kernel), not ga * xhat + be
  dalpha = 0.f;
  if (act == GCNX_ACT_RELU) dzb = zb > 0.f ? dy : 0.f;
  else if (act == GCNX_ACT_PRELU) { dzb = zb > 0.f ? dy : al * dy; dalpha = dy * fminf(zb, 0.f); }
  else dzb = dy;
}

__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ dy, int64_t lddy,
                                                           const float* __restrict__ z, int64_t ldz, int64_t n, int32_t f,
                                                           const float* __restrict__ mean, const float* __restrict__ inv,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int act, const float* __restrict__ alpha,
                                                           float* __restrict__ part, int vec) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 15) * 4;
  const int valid0 = f - c;
  BnCols q{};
  if (valid0 > 0) q = bn_cols(mean, inv, gamma, beta, alpha, act, c, valid0);
  colsums<3>(n, f, part, vec && (int)blockIdx.x * 64 + 64 <= f, [&](int64_t r, int cc, int valid, float4 (&t)[3], auto fast) {
    const bool v4 = decltype(fast)::value || (vec && valid >= 4);
    const float4 d = ld4g(dy + r * lddy + cc, v4, valid), zz = ld4g(z + r * ldz + cc, v4, valid);
    float4 xh;
    bn_bwd_terms(d.x, zz.x, q.mu.x, q.iv.x, q.ga.x, q.be.x, q.al.x, act, t[0].x, xh.x, t[2].x);
    bn_bwd_terms(d.y, zz.y, q.mu.y, q.iv.y, q.ga.y, q.be.y, q.al.y, act, t[0].y, xh.y, t[2].y);
    bn_bwd_terms(d.z, zz.z, q.mu.z, q.iv.z, q.ga.z, q.be.z, q.al.z, act, t[0].z, xh.z, t[2].z);
    bn_bwd_terms(d.w, zz.w, q.mu.w, q.iv.w, q.ga.w, q.be.w, q.al.w, act, t[0].w, xh.w, t[2].w);
    t[1] = make_float4(t[0].x * xh.x, t[0].y * xh.y, t[0].z * xh.z, t[0].w * xh.w);
  });
}

// sums = [sum dzb | sum dzb*xhat | sum dy*min(zb,0)] ; training: dz = gamma*inv*(dzb - s1/n - xhat*s2/n)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int64_t lddy,
                                                           const float* __restrict__ z, int64_t ldz, int64_t n, int32_t f,
                                                           const float* __restrict__ mean, const float* __restrict__ inv,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           int act, const float* __restrict__ alpha,
                                                           const float* __restrict__ sums, float count, int training,
                                                           float* __restrict__ dz, int64_t lddz, int vec) {
  const int cl = threadIdx.x & 63;
  const int c = blockIdx.x * 256 + cl * 4;
  const int valid = f - c;
  if (valid <= 0) return;
  const bool v4 = vec && valid >= 4;
  const BnCols q = bn_cols(mean, inv, gamma, beta, alpha, act, c, valid);
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if (training) {
    s1 = ld4g(sums + c, false, valid);
    s2 = ld4g(sums + f + c, false, valid);
    const float ic = 1.0f / count;
    s1 = make_float4(s1.x * ic, s1.y * ic, s1.z * ic, s1.w * ic);
    s2 = make_float4(s2.x * ic, s2.y * ic, s2.z * ic, s2.w * ic);
  }
  auto grad = [&](float4 d, float4 zz) {
    float4 g, xh, unused;
    bn_bwd_terms(d.x, zz.x, q.mu.x, q.iv.x, q.ga.x, q.be.x, q.al.x, act, g.x, xh.x, unused.x);
    bn_bwd_terms(d.y, zz.y, q.mu.y, q.iv.y, q.ga.y, q.be.y, q.al.y, act, g.y, xh.y, unused.y);
    bn_bwd_terms(d.z, zz.z, q.mu.z, q.iv.z, q.ga.z, q.be.z, q.al.z, act, g.z, xh.z, unused.z);
    bn_bwd_terms(d.w, zz.w, q.mu.w, q.iv.w, q.ga.w, q.be.w, q.al.w, act, g.w, xh.w, unused.w);
    float4 o;
    o.x = q.ga.x * q.iv.x * (g.x - s1.x - xh.x * s2.x);
    o.y = q.ga.y * q.iv.y * (g.y - s1.y - xh.y * s2.y);
    o.z = q.ga.z * q.iv.z * (g.z - s1.z - xh.z * s2.z);
    o.w = q.ga.w * q.iv.w * (g.w - s1.w - xh.w * s2.w);
    return o;
  };
  const int64_t step = (int64_t)gridDim.y * 4;
  int64_t r = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (vec && (int)blockIdx.x * 256 + 256 <= f) {           // aligned full tile (uniform): two rows' loads in flight
    for (; r + step < n; r += 2 * step) {
      const float4 d0 = *reinterpret_cast<const float4*>(dy + r * lddy + c);
      const float4 z0 = *reinterpret_cast<const float4*>(z + r * ldz + c);
      const float4 d1 = *reinterpret_cast<const float4*>(dy + (r + step) * lddy + c);
      const float4 z1 = *reinterpret_cast<const float4*>(z + (r + step) * ldz + c);
      *reinterpret_cast<float4*>(dz + r * lddz + c) = grad(d0, z0);
      *reinterpret_cast<float4*>(dz + (r + step) * lddz + c) = grad(d1, z1);
    }
  }
  for (; r < n; r += step)
    st4g(dz + r * lddz + c, grad(ld4g(dy + r * lddy + c, v4, valid), ld4g(z + r * ldz + c, v4, valid)), v4, valid);
}

// gcnx_bn_act_bwd for a batch of at most kRows rows in ONE launch (the post-MLP of GeneralGNN: one row per graph): a
// workgroup owns 64 columns -- the three column sums (bn_bwd_stats_kernel's terms and order), their single-chunk
// "reduction" and the parameter gradients (part_reduce_kernel's arithmetic), then dz for its columns
// (bn_bwd_apply_kernel's arithmetic per element) -- the bits of the three-launch sequence.  dz may alias dy.
__global__ __launch_bounds__(256) void bn_act_bwd_small_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ z,
                                                               int64_t ldz, int64_t n, int32_t f, const float* __restrict__ mean,
                                                               const float* __restrict__ inv, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int act, const float* __restrict__ alpha,
                                                               float* __restrict__ part, float* __restrict__ sums, float* __restrict__ dbeta,
                                                               float* __restrict__ dgamma, float* __restrict__ dalpha, int training,
                                                               float* __restrict__ dz, int64_t lddz, int vec) {
  __shared__ float s_sum[2][64];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + cl * 4;
  const int valid0 = f - c;
  BnCols q{};
  if (valid0 > 0) q = bn_cols(mean, inv, gamma, beta, alpha, act, c, valid0);
  colsums<3>(n, f, part, vec && (int)blockIdx.x * 64 + 64 <= f, [&](int64_t r, int cc, int valid, float4 (&t)[3], auto fast) {
    const bool v4 = decltype(fast)::value || (vec && valid >= 4);
    const float4 d = ld4g(dy + r * lddy + cc, v4, valid), zz = ld4g(z + r * ldz + cc, v4, valid);
    float4 xh;
    bn_bwd_terms(d.x, zz.x, q.mu.x, q.iv.x, q.ga.x, q.be.x, q.al.x, act, t[0].x, xh.x, t[2].x);
    bn_bwd_terms(d.y, zz.y, q.mu.y, q.iv.y, q.ga.y, q.be.y, q.al.y, act, t[0].y, xh.y, t[2].y);
    bn_bwd_terms(d.z, zz.z, q.mu.z, q.iv.z, q.ga.z, q.be.z, q.al.z, act, t[0].z, xh.z, t[2].z);
    bn_bwd_terms(d.w, zz.w, q.mu.w, q.iv.w, q.ga.w, q.be.w, q.al.w, act, t[0].w, xh.w, t[2].w);
    t[1] = make_float4(t[0].x * xh.x, t[0].y * xh.y, t[0].z * xh.z, t[0].w * xh.w);
  });
  __syncthreads();                                   // this workgroup's columns of `part`: written by its own threads
  if (threadIdx.x < 192) {                           // part_reduce_kernel with one chunk: (x + 0) + (0 + 0)
    const int k = threadIdx.x >> 6, el = threadIdx.x & 63, col = blockIdx.x * 64 + el;
    if (col < f) {
      const float v = ((0.f + part[(int64_t)k * f + col]) + 0.f) + (0.f + 0.f);
      sums[(int64_t)k * f + col] = v;
      float* extra = k == 0 ? dbeta : (k == 1 ? dgamma : dalpha);
      if (extra) extra[col] = v;
      if (k < 2) s_sum[k][el] = v;
    }
  }
  __syncthreads();
  if (valid0 <= 0) return;
  const bool v4 = vec && valid0 >= 4;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if (training) {
    const float ic = 1.0f / (float)n;
    float a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = j < valid0 ? s_sum[0][cl * 4 + j] * ic : 0.f; b[j] = j < valid0 ? s_sum[1][cl * 4 + j] * ic : 0.f; }
    s1 = make_float4(a[0], a[1], a[2], a[3]);
    s2 = make_float4(b[0], b[1], b[2], b[3]);
  }
  for (int64_t r = rg; r < n; r += 16) {
    const float4 d = ld4g(dy + r * lddy + c, v4, valid0), zz = ld4g(z + r * ldz + c, v4, valid0);
    float4 g, xh, unused, o;
    bn_bwd_terms(d.x, zz.x, q.mu.x, q.iv.x, q.ga.x, q.be.x, q.al.x, act, g.x, xh.x, unused.x);
    bn_bwd_terms(d.y, zz.y, q.mu.y, q.iv.y, q.ga.y, q.be.y, q.al.y, act, g.y, xh.y, unused.y);
    bn_bwd_terms(d.z, zz.z, q.mu.z, q.iv.z, q.ga.z, q.be.z, q.al.z, act, g.z, xh.z, unused.z);
    bn_bwd_terms(d.w, zz.w, q.mu.w, q.iv.w, q.ga.w, q.be.w, q.al.w, act, g.w, xh.w, unused.w);
    o.x = q.ga.x * q.iv.x * (g.x - s1.x - xh.x * s2.x);
    o.y = q.ga.y * q.iv.y * (g.y - s1.y - xh.y * s2.y);
    o.z = q.ga.z * q.iv.z * (g.z - s1.z - xh.z * s2.z);
    o.w = q.ga.w * q.iv.w * (g.w - s1.w - xh.w * s2.w);
    st4g(dz + r * lddz + c, o, v4, valid0);
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int reduce_parts(gcnx_ctx* ctx, int nchunks, int ns, int32_t f, float* out, float* o0 = nullptr, float* o1 = nullptr,
                 float* o2 = nullptr) {
  hipLaunchKernelGGL(part_reduce_kernel, dim3(gcnx_cdiv((long long)ns * f, 64)), dim3(256), 0, ctx->stream,
                     (const float*)ctx->ws, nchunks, ns, f, out, o0, o1, o2);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // namespace

extern "C" {

int gcnx_bn_stats(gcnx_ctx* ctx, const float* z, int64_t ldz, int64_t n, int32_t f, const float* shift, float* sums) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_bn_stats: negative size");
  if (f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, sums != nullptr, "gcnx_bn_stats: sums is NULL");
  if (n == 0) { GCNX_HIP(ctx, hipMemsetAsync(sums, 0, (size_t)2 * f * 4, ctx->stream)); return GCNX_OK; }
  GCNX_REQUIRE(ctx, z && ldz >= f, "gcnx_bn_stats: bad z / ldz");
  const int nchunks = gcnx_cdiv(n, kRows);
  int rc = gcnx_ws_reserve(ctx, (size_t)nchunks * 2 * f * sizeof(float));
  if (rc) return rc;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(gcnx_cdiv(f, 64), nchunks), dim3(256), 0, ctx->stream, z, ldz, n, f,
                     shift, (float*)ctx->ws, (int)(al16(z) && ldz % 4 == 0));
  GCNX_LAUNCH_OK(ctx);
  return reduce_parts(ctx, nchunks, 2, f, sums);
}

int gcnx_bn_finalize(gcnx_ctx* ctx, const float* sums, float count, int32_t f, float momentum, float eps,
                     const float* shift, float* mean, float* inv, float* moving_mean, float* moving_var) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, f >= 0, "gcnx_bn_finalize: negative size");
  if (f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, mean && inv, "gcnx_bn_finalize: NULL output");
  GCNX_REQUIRE(ctx, sums || (moving_mean && moving_var), "gcnx_bn_finalize: inference mode needs the moving statistics");
  GCNX_REQUIRE(ctx, !sums || count > 0.f, "gcnx_bn_finalize: count must be positive");
  GCNX_REQUIRE(ctx, (moving_mean == nullptr) == (moving_var == nullptr), "gcnx_bn_finalize: pass both moving buffers or none");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(gcnx_cdiv(f, 256)), dim3(256), 0, ctx->stream, sums, count, f, momentum, eps,
                     shift, mean, inv, moving_mean, moving_var);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_bn_moments(gcnx_ctx* ctx, const float* z, int64_t ldz, int64_t n, int32_t f, float momentum, float eps,
                    float* mean, float* inv, float* moving_mean, float* moving_var) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "batch norm (moments)");
  GCNX_REQUIRE(ctx, n > 0 && f >= 0, "gcnx_bn_moments: needs at least one row");
  if (f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, z && ldz >= f && mean && inv, "gcnx_bn_moments: NULL pointer / bad ldz");
  GCNX_REQUIRE(ctx, (moving_mean == nullptr) == (moving_var == nullptr), "gcnx_bn_moments: pass both moving buffers or none");
  const int nchunks = gcnx_cdiv(n, kRows);
  int rc = gcnx_ws_reserve(ctx, (size_t)nchunks * 2 * f * sizeof(float));
  if (rc) return rc;
  const int vec = (int)(al16(z) && ldz % 4 == 0);
  dim3 gs(gcnx_cdiv(f, 64), nchunks), gr(gcnx_cdiv(f, 64));
  if (nchunks == 1) {                                  // a batch of at most kRows rows: both passes in one launch, same bits
    hipLaunchKernelGGL(bn_moments_small_kernel, gr, dim3(256), 0, ctx->stream, z, ldz, n, f, momentum, eps, (float*)ctx->ws, mean, inv,
                       moving_mean, moving_var, vec);
    GCNX_LAUNCH_OK(ctx);
    return GCNX_OK;
  }
  // pass 1: mean.  pass 2: variance of the data centred on that mean (two-pass, like tf.nn.moments)
  hipLaunchKernelGGL(bn_stats_kernel, gs, dim3(256), 0, ctx->stream, z, ldz, n, f, (const float*)nullptr, (float*)ctx->ws, vec);
  GCNX_LAUNCH_OK(ctx);
  hipLaunchKernelGGL(bn_reduce_finalize_kernel, gr, dim3(256), 0, ctx->stream, (const float*)ctx->ws, nchunks, (float)n, f,
                     momentum, eps, (const float*)nullptr, mean, inv, (float*)nullptr, (float*)nullptr);
  GCNX_LAUNCH_OK(ctx);
  hipLaunchKernelGGL(bn_stats_kernel, gs, dim3(256), 0, ctx->stream, z, ldz, n, f, (const float*)mean, (float*)ctx->ws, vec);
  GCNX_LAUNCH_OK(ctx);
  hipLaunchKernelGGL(bn_reduce_finalize_kernel, gr, dim3(256), 0, ctx->stream, (const float*)ctx->ws, nchunks, (float)n, f,
                     momentum, eps, (const float*)mean, mean, inv, moving_mean, moving_var);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_bn_act(gcnx_ctx* ctx, const float* z, int64_t ldz, int64_t n, int32_t f, const float* mean, const float* inv,
                const float* gamma, const float* beta, int act, const float* alpha, float* y, int64_t ldy) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "batch norm (apply)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_bn_act: negative size");
  GCNX_REQUIRE(ctx, act >= GCNX_ACT_NONE && act <= GCNX_ACT_PRELU, "gcnx_bn_act: unknown activation %d", act);
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, z && y && mean && inv && gamma && beta, "gcnx_bn_act: NULL pointer");
  GCNX_REQUIRE(ctx, act != GCNX_ACT_PRELU || alpha, "gcnx_bn_act: PReLU needs alpha");
  GCNX_REQUIRE(ctx, ldz >= f && ldy >= f, "gcnx_bn_act: leading dimension too small");
  int gy = gcnx_cdiv(n, 4);
  if (gy > 8 * ctx->num_cus) gy = 8 * ctx->num_cus;
  const int vec = al16(z) && al16(y) && ldz % 4 == 0 && ldy % 4 == 0;
  hipLaunchKernelGGL(bn_act_kernel, dim3(gcnx_cdiv(f, 256), gy), dim3(256), 0, ctx->stream, z, ldz, n, f, mean, inv, gamma,
                     beta, act, alpha, y, ldy, vec);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

// Two halves of the BatchNorm backward, so that a multi-GPU caller can all-reduce the three column sums between
// them (sync-BN); gcnx_bn_act_bwd below is stats + apply with the local row count.
int gcnx_bn_act_bwd_stats(gcnx_ctx* ctx, const float* dy, int64_t lddy, const float* z, int64_t ldz, int64_t n, int32_t f,
                          const float* mean, const float* inv, const float* gamma, const float* beta, int act,
                          const float* alpha, float* sums_scratch, float* dgamma, float* dbeta, float* dalpha) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_bn_act_bwd: negative size");
  GCNX_REQUIRE(ctx, act >= GCNX_ACT_NONE && act <= GCNX_ACT_PRELU, "gcnx_bn_act_bwd: unknown activation %d", act);
  if (f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, sums_scratch != nullptr, "gcnx_bn_act_bwd: sums_scratch (device float[3f]) is NULL");
  if (n == 0) {   // no rows: sums and gradients are zero
    GCNX_HIP(ctx, hipMemsetAsync(sums_scratch, 0, (size_t)3 * f * 4, ctx->stream));
    if (dbeta) GCNX_HIP(ctx, hipMemsetAsync(dbeta, 0, (size_t)f * 4, ctx->stream));
    if (dgamma) GCNX_HIP(ctx, hipMemsetAsync(dgamma, 0, (size_t)f * 4, ctx->stream));
    if (dalpha) GCNX_HIP(ctx, hipMemsetAsync(dalpha, 0, (size_t)f * 4, ctx->stream));
    return GCNX_OK;
  }
  GCNX_REQUIRE(ctx, dy && z && mean && inv && gamma && beta, "gcnx_bn_act_bwd: NULL pointer");
  GCNX_REQUIRE(ctx, act != GCNX_ACT_PRELU || alpha, "gcnx_bn_act_bwd: PReLU needs alpha");
  GCNX_REQUIRE(ctx, lddy >= f && ldz >= f, "gcnx_bn_act_bwd: leading dimension too small");
  const int nchunks = gcnx_cdiv(n, kRows);
  int rc = gcnx_ws_reserve(ctx, (size_t)nchunks * 3 * f * sizeof(float));
  if (rc) return rc;
  const int vec = al16(dy) && al16(z) && lddy % 4 == 0 && ldz % 4 == 0;
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(gcnx_cdiv(f, 64), nchunks), dim3(256), 0, ctx->stream, dy, lddy, z, ldz, n,
                     f, mean, inv, gamma, beta, act, alpha, (float*)ctx->ws, vec);
  GCNX_LAUNCH_OK(ctx);
  // parameter gradients ride along: dbeta = sum dzb, dgamma = sum dzb*xhat, dalpha = sum dy*min(zb,0)
  return reduce_parts(ctx, nchunks, 3, f, sums_scratch, dbeta, dgamma, dalpha);
}

int gcnx_bn_act_bwd_apply(gcnx_ctx* ctx, const float* dy, int64_t lddy, const float* z, int64_t ldz, int64_t n, int32_t f,
                          const float* mean, const float* inv, const float* gamma, const float* beta, int act,
                          const float* alpha, const float* sums, float count, int training, float* dz, int64_t lddz) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_bn_act_bwd: negative size");
  GCNX_REQUIRE(ctx, act >= GCNX_ACT_NONE && act <= GCNX_ACT_PRELU, "gcnx_bn_act_bwd: unknown activation %d", act);
  if (f == 0 || n == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, dy && z && dz && mean && inv && gamma && beta && sums, "gcnx_bn_act_bwd: NULL pointer");
  GCNX_REQUIRE(ctx, act != GCNX_ACT_PRELU || alpha, "gcnx_bn_act_bwd: PReLU needs alpha");
  GCNX_REQUIRE(ctx, lddy >= f && ldz >= f && lddz >= f, "gcnx_bn_act_bwd: leading dimension too small");
  GCNX_REQUIRE(ctx, count > 0.f, "gcnx_bn_act_bwd: count must be positive");
  const int vec = al16(dy) && al16(z) && al16(dz) && lddy % 4 == 0 && ldz % 4 == 0 && lddz % 4 == 0;
  int gy = gcnx_cdiv(n, 4);
  if (gy > 8 * ctx->num_cus) gy = 8 * ctx->num_cus;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(gcnx_cdiv(f, 256), gy), dim3(256), 0, ctx->stream, dy, lddy, z, ldz, n, f,
                     mean, inv, gamma, beta, act, alpha, sums, count, training, dz, lddz, vec);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_bn_act_bwd(gcnx_ctx* ctx, const float* dy, int64_t lddy, const float* z, int64_t ldz, int64_t n, int32_t f,
                    const float* mean, const float* inv, const float* gamma, const float* beta, int act,
                    const float* alpha, int training, float* dz, int64_t lddz, float* dgamma, float* dbeta,
                    float* dalpha, float* sums_scratch) {
  if (ctx && n > 0 && n <= kRows && f > 0 && dy && z && dz && mean && inv && gamma && beta && sums_scratch && lddy >= f && ldz >= f &&
      lddz >= f && act >= GCNX_ACT_NONE && act <= GCNX_ACT_PRELU && (act != GCNX_ACT_PRELU || alpha)) {
    // a batch of one chunk (the post-MLP's rows): statistics, their reduction and dz in one launch, the same bits
    GCNX_CHECK_CTX(ctx);
    GCNX_RANGE(ctx, "batch norm + activation backward (small batch)");
    int rc1 = gcnx_ws_reserve(ctx, (size_t)3 * f * sizeof(float));
    if (rc1) return rc1;
    const int vec = al16(dy) && al16(z) && al16(dz) && lddy % 4 == 0 && ldz % 4 == 0 && lddz % 4 == 0;
    hipLaunchKernelGGL(bn_act_bwd_small_kernel, dim3(gcnx_cdiv(f, 64)), dim3(256), 0, ctx->stream, dy, lddy, z, ldz, n, f, mean, inv, gamma,
                       beta, act, alpha, (float*)ctx->ws, sums_scratch, dbeta, dgamma, dalpha, training, dz, lddz, vec);
    GCNX_LAUNCH_OK(ctx);
    return GCNX_OK;
  }
  int rc = gcnx_bn_act_bwd_stats(ctx, dy, lddy, z, ldz, n, f, mean, inv, gamma, beta, act, alpha, sums_scratch, dgamma,
                                 dbeta, dalpha);
  if (rc || n == 0 || f == 0) return rc;
  return gcnx_bn_act_bwd_apply(ctx, dy, lddy, z, ldz, n, f, mean, inv, gamma, beta, act, alpha, sums_scratch, (float)n,
                               training, dz, lddz);
}

}  // extern "C"
