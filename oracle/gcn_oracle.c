/*
 * CPU ORACLE (plain C, fp32, OpenMP) for the GCN forward/backward hot path.
 * TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED.
 *
 * The reference's arithmetic for this path lives in un-vendored, un-pinned Spektral/TensorFlow
 * (call sites: src/scripts/gcn.py:320 model, :326 loss, :328-340 train_step, :342-362 evaluate);
 * nothing under /root/reference can be compiled, and it holds no tests or golden vectors.  This
 * file restates, in fp32 like TensorFlow's CPU kernels, the published semantics written out in
 * SURVEY.md section 8.A (the spec of record).  It is checked in tests/ against the fp64 numpy
 * restatement (oracle/gcn_oracle.py), which in turn is cross-checked against scipy.sparse and
 * torch-CPU autograd.
 *
 * Users: tests/ (second checker), __graft_entry__.smoke(), bench.py's cpu_baseline leg ("port").
 * The product (gcn-string_amd/) never links or loads this.
 *
 * Row-parallel with OpenMP; inside a row, sums run in storage / k order, the order TF's
 * SparseTensorDenseMatMul (serial row-axpy over nnz) and Eigen's MatMul produce per element.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define API __attribute__((visibility("default")))

API int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
API void orc_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* GEMM operand model.  0: fp32 operands (TensorFlow's CPU MatMul; GCNX_PREC_F32 / BF16X3 are held to this).  1: both
 * operands of every dense product rounded to bfloat16 (round-to-nearest-even), products and sums in fp32 -- the
 * arithmetic of BASELINE config 3's "bf16 MFMA weight GEMM" (GCNX_PREC_BF16), so that the plain-bf16 path can be
 * checked against an oracle fed the same rounded operands instead of a loose bound. */
static int g_bf16 = 0;
API void orc_set_bf16_operands(int on) { g_bf16 = on; }
/* Layer 1 of orc_gcn2_step as (A X) W1 instead of A (X W1) -- the same product (gcn.py:334 computes the latter), in the
 * order the device path takes when F <= H: the layer's input needs no gradient, so with S1 = A X saved its weight
 * gradient is S1^T dZ1 and the backward aggregation of layer 1 disappears.  Only the bf16-operand MODEL needs this
 * switch (the operands that get rounded are then S1 and dZ1 instead of X and A^T dZ1); the fp32 parity checks run
 * against the reference's own order.  `s1` scratch: N * F floats, passed through orc_set_layer1_s_order. */
static float* g_s1 = NULL;
API void orc_set_layer1_s_order(float* s1_scratch) { g_s1 = s1_scratch; }
static inline float opnd(float v) {
  if (!g_bf16) return v;
  uint32_t u;
  memcpy(&u, &v, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return v;          /* inf / nan unchanged */
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  memcpy(&v, &u, 4);
  return v;
}

/* K2/K3: out[t] = act(sum_e vals[e] * h[colidx[e]] + bias); vals NULL = ones. */
API void orc_spmm_csr(const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h, int64_t ldh,
                      const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int relu) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int32_t r = 0; r < n; ++r) {
    float* o = out + (int64_t)r * ldo;
    for (int32_t c = 0; c < f; ++c) o[c] = 0.f;
    for (int32_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
      const float v = vals ? vals[e] : 1.0f;
      const float* hr = h + (int64_t)colidx[e] * ldh;
      for (int32_t c = 0; c < f; ++c) o[c] += v * hr[c];
    }
    if (bias) for (int32_t c = 0; c < f; ++c) o[c] += bias[c];
    if (relu) for (int32_t c = 0; c < f; ++c) o[c] = o[c] > 0.f ? o[c] : 0.f;
  }
}

/* K1: out[N,Fo] = act(X[N,Fi] W[Fi,Fo] + bias), k-ordered accumulation per element. */
API void orc_gemm(const float* x, int64_t ldx, const float* w, const float* bias, float* out, int64_t ldo, int64_t n,
                  int32_t fi, int32_t fo, int relu) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    float* o = out + r * ldo;
    for (int32_t j = 0; j < fo; ++j) o[j] = 0.f;
    const float* xr = x + r * ldx;
    for (int32_t k = 0; k < fi; ++k) {
      const float a = opnd(xr[k]);
      const float* wk = w + (int64_t)k * fo;
      if (g_bf16) for (int32_t j = 0; j < fo; ++j) o[j] += a * opnd(wk[j]);
      else for (int32_t j = 0; j < fo; ++j) o[j] += a * wk[j];
    }
    if (bias) for (int32_t j = 0; j < fo; ++j) o[j] += bias[j];
    if (relu) for (int32_t j = 0; j < fo; ++j) o[j] = o[j] > 0.f ? o[j] : 0.f;
  }
}

/* K1^T b: dX[N,Fi] = dH[N,Fo] W^T, optionally masked by (y > 0). */
API void orc_gemm_dx(const float* dh, int64_t lddh, const float* w, float* dx, int64_t lddx, int64_t n, int32_t fi,
                     int32_t fo, const float* y_mask, int64_t ldy) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    const float* d = dh + r * lddh;
    float* o = dx + r * lddx;
    for (int32_t i = 0; i < fi; ++i) {
      const float* wi = w + (int64_t)i * fo;
      float s = 0.f;
      if (g_bf16) for (int32_t k = 0; k < fo; ++k) s += opnd(d[k]) * opnd(wi[k]);
      else for (int32_t k = 0; k < fo; ++k) s += d[k] * wi[k];
      if (y_mask && !(y_mask[r * ldy + i] > 0.f)) s = 0.f;
      o[i] = s;
    }
  }
}

/* K1^T a: dW[Fi,Fo] = X^T dH (reduction over the N rows).  Blocked summation: fp32 products and fp32 sums inside a
 * block of 256 rows (what an fp32 kernel does), block sums combined in double, blocks in ascending row order per thread
 * and threads in thread order -- so that at N = 10^6 rows the checker's own rounding (a serial fp32 sum drifts by
 * ~sqrt(N) ulp) stays far below the 1e-4 bar it is used to judge.  Eigen's MatMul / reductions are blocked / pairwise
 * too; no summation order of TensorFlow's is being claimed. */
#define ORC_BLK 256
API void orc_gemm_dw(const float* x, int64_t ldx, const float* dh, int64_t lddh, float* dw, int64_t n, int32_t fi,
                     int32_t fo) {
  const int nt = orc_max_threads();
  const size_t sz = (size_t)fi * fo;
  double* part = (double*)calloc((size_t)nt * sz, sizeof(double));
  const int64_t nblk = (n + ORC_BLK - 1) / ORC_BLK;
#pragma omp parallel
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    double* p = part + (size_t)t * sz;
    float* blk = (float*)malloc(sz * sizeof(float));
#pragma omp for schedule(static)
    for (int64_t bi = 0; bi < nblk; ++bi) {
      const int64_t r0 = bi * ORC_BLK, r1 = r0 + ORC_BLK < n ? r0 + ORC_BLK : n;
      memset(blk, 0, sz * sizeof(float));
      for (int64_t r = r0; r < r1; ++r) {
        const float* xr = x + r * ldx;
        const float* d = dh + r * lddh;
        for (int32_t i = 0; i < fi; ++i) {
          const float a = opnd(xr[i]);
          float* pi = blk + (size_t)i * fo;
          if (g_bf16) for (int32_t j = 0; j < fo; ++j) pi[j] += a * opnd(d[j]);
          else for (int32_t j = 0; j < fo; ++j) pi[j] += a * d[j];
        }
      }
      for (size_t q = 0; q < sz; ++q) p[q] += (double)blk[q];
    }
    free(blk);
  }
  for (size_t q = 0; q < sz; ++q) {
    double s = 0.0;
    for (int t = 0; t < nt; ++t) s += part[(size_t)t * sz + q];
    dw[q] = (float)s;
  }
  free(part);
}

API void orc_colsum(const float* x, int64_t ldx, int64_t n, int32_t f, float* out) {
  double* acc = (double*)calloc((size_t)f, sizeof(double));
  float* blk = (float*)malloc((size_t)f * sizeof(float));
  for (int64_t r0 = 0; r0 < n; r0 += ORC_BLK) {
    const int64_t r1 = r0 + ORC_BLK < n ? r0 + ORC_BLK : n;
    for (int32_t c = 0; c < f; ++c) blk[c] = 0.f;
    for (int64_t r = r0; r < r1; ++r)
      for (int32_t c = 0; c < f; ++c) blk[c] += x[r * ldx + c];
    for (int32_t c = 0; c < f; ++c) acc[c] += (double)blk[c];
  }
  for (int32_t c = 0; c < f; ++c) out[c] = (float)acc[c];
  free(acc); free(blk);
}

/* K4: mode 0 sum, 1 avg, 2 max (first maximal row wins; argmax row index). */
API void orc_pool(const int32_t* gp, const float* x, int64_t ldx, float* pooled, int32_t b, int32_t f, int mode,
                  int32_t* argmax) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int32_t g = 0; g < b; ++g) {
    float* p = pooled + (int64_t)g * f;
    const int32_t lo = gp[g], hi = gp[g + 1];
    for (int32_t c = 0; c < f; ++c) p[c] = 0.f;
    if (hi == lo) { if (argmax) for (int32_t c = 0; c < f; ++c) argmax[(int64_t)g * f + c] = lo; continue; }
    if (mode == 2) {
      for (int32_t c = 0; c < f; ++c) { p[c] = x[(int64_t)lo * ldx + c]; argmax[(int64_t)g * f + c] = lo; }
      for (int32_t r = lo + 1; r < hi; ++r)
        for (int32_t c = 0; c < f; ++c) {
          const float v = x[(int64_t)r * ldx + c];
          if (v > p[c]) { p[c] = v; argmax[(int64_t)g * f + c] = r; }
        }
    } else {
      for (int32_t r = lo; r < hi; ++r)
        for (int32_t c = 0; c < f; ++c) p[c] += x[(int64_t)r * ldx + c];
      if (mode == 1) for (int32_t c = 0; c < f; ++c) p[c] /= (float)(hi - lo);
    }
  }
}

API void orc_pool_bwd(const int32_t* gp, const float* dp, float* dx, int64_t lddx, int32_t b, int32_t f, int mode,
                      const int32_t* argmax, const float* y_mask, int64_t ldy) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int32_t g = 0; g < b; ++g) {
    const int32_t lo = gp[g], hi = gp[g + 1];
    const float sc = (mode == 1 && hi > lo) ? 1.0f / (float)(hi - lo) : 1.0f;
    for (int32_t r = lo; r < hi; ++r)
      for (int32_t c = 0; c < f; ++c) {
        float v = dp[(int64_t)g * f + c] * sc;
        if (mode == 2 && argmax[(int64_t)g * f + c] != r) v = 0.f;
        if (y_mask && !(y_mask[(int64_t)r * ldy + c] > 0.f)) v = 0.f;
        dx[(int64_t)r * lddx + c] = v;
      }
  }
}

/* K8: softmax; Keras CCE; categorical accuracy.
 * from_logits = 0: the eager branch of keras.backend.categorical_crossentropy (evaluate, gcn.py:351-354): renormalise,
 *   clip [1e-7, 1-1e-7], -sum y log p; the clip passes no gradient outside.
 * from_logits = 1: what the same loss object computes inside tf.function (train_step, gcn.py:328-335): Keras takes the
 *   Softmax op's input and calls softmax_cross_entropy_with_logits: sum_c y_c (logsumexp(z) - z_c), dlogits =
 *   (p sum_c y_c - y) / denom.
 * loss_acc[0] += sum_g loss_g / denom, loss_acc[1] += #correct. */
API void orc_softmax_cce(const float* logits, const float* y, int32_t b, int32_t c, float denom, float* probs,
                         float* loss_acc, float* dlogits, int from_logits) {
  float loss = 0.f, hit = 0.f;
  for (int32_t g = 0; g < b; ++g) {
    const float* z = logits + (int64_t)g * c;
    const float* yy = y + (int64_t)g * c;
    float m = z[0];
    for (int32_t k = 1; k < c; ++k) m = z[k] > m ? z[k] : m;
    float sum = 0.f;
    for (int32_t k = 0; k < c; ++k) sum += expf(z[k] - m);
    float ymsum = 0.f, l = 0.f, pmax = -1.f, ymax = -INFINITY;
    int pa = 0, ya = 0;
    for (int32_t k = 0; k < c; ++k) {
      const float p = expf(z[k] - m) / sum;
      probs[(int64_t)g * c + k] = p;
      if (from_logits || (p > 1e-7f && p < 1.0f - 1e-7f)) ymsum += yy[k];  /* clip passes no gradient outside */
      if (p > pmax) { pmax = p; pa = k; }
      if (yy[k] > ymax) { ymax = yy[k]; ya = k; }
      if (from_logits) {
        l += yy[k] * ((m - z[k]) + logf(sum));
      } else {
        float pc = p < 1e-7f ? 1e-7f : p;
        pc = pc > 1.0f - 1e-7f ? 1.0f - 1e-7f : pc;
        l -= yy[k] * logf(pc);
      }
    }
    if (dlogits)
      for (int32_t k = 0; k < c; ++k) {
        const float p = probs[(int64_t)g * c + k];
        const float ym = (from_logits || (p > 1e-7f && p < 1.0f - 1e-7f)) ? yy[k] : 0.f;
        dlogits[(int64_t)g * c + k] = (p * ymsum - ym) / denom;
      }
    loss += l;
    hit += (pa == ya) ? 1.f : 0.f;
  }
  loss_acc[0] += loss / denom;
  loss_acc[1] += hit;
}

/* The M1 model step, order of train_step gcn.py:330-340:
 * GCNConv(F->H,relu) -> GCNConv(H->H,relu) -> GlobalSumPool -> Dense(H->C) softmax, CCE,
 * all gradients, optional SGD apply (lr > 0).  params/grads flat: w1,b1,w2,b2,w3,b3.
 * work: caller-provided scratch of 4*N*H + 2*B*H + 3*B*C floats.  Adjacency assumed symmetric
 * (the reference's data) so A^T = A.  Returns loss in out[0], #correct in out[1].
 * from_logits: the CCE branch (orc_softmax_cce); 1 is what train_step runs under tf.function. */
API void orc_gcn2_step(const int32_t* rowptr, const int32_t* colidx, const float* vals, const int32_t* gp,
                       const float* x, const float* y, int32_t n, int32_t b, int32_t f, int32_t hdim, int32_t c,
                       float* params, float* grads, float lr, float denom, float* work, float* out, int from_logits) {
  float* w1 = params;            float* b1 = w1 + (int64_t)f * hdim;
  float* w2 = b1 + hdim;         float* b2 = w2 + (int64_t)hdim * hdim;
  float* w3 = b2 + hdim;         float* b3 = w3 + (int64_t)hdim * c;
  float* gw1 = grads;            float* gb1 = gw1 + (int64_t)f * hdim;
  float* gw2 = gb1 + hdim;       float* gb2 = gw2 + (int64_t)hdim * hdim;
  float* gw3 = gb2 + hdim;       float* gb3 = gw3 + (int64_t)hdim * c;
  const int64_t nh = (int64_t)n * hdim;
  float* h = work;               float* y1 = h + nh;      float* y2 = y1 + nh;     float* dz = y2 + nh;
  float* pooled = dz + nh;       float* dpooled = pooled + (int64_t)b * hdim;
  float* logits = dpooled + (int64_t)b * hdim;
  float* probs = logits + (int64_t)b * c;  float* dlogits = probs + (int64_t)b * c;

  if (g_s1) {
    orc_spmm_csr(rowptr, colidx, vals, x, f, NULL, g_s1, f, n, f, 0);
    orc_gemm(g_s1, f, w1, b1, y1, hdim, n, f, hdim, 1);
  } else {
    orc_gemm(x, f, w1, NULL, h, hdim, n, f, hdim, 0);
    orc_spmm_csr(rowptr, colidx, vals, h, hdim, b1, y1, hdim, n, hdim, 1);
  }
  orc_gemm(y1, hdim, w2, NULL, h, hdim, n, hdim, hdim, 0);
  orc_spmm_csr(rowptr, colidx, vals, h, hdim, b2, y2, hdim, n, hdim, 1);
  orc_pool(gp, y2, hdim, pooled, b, hdim, 0, NULL);
  /* the classifier head ([B,H] x [H,C], a few kilobytes) stays fp32 in every precision mode: the bf16 operand model
   * covers the two GCNConv kernel products and their gradients only (what GCNX_PREC_BF16 selects) */
  const int bf16_saved = g_bf16;
  g_bf16 = 0;
  orc_gemm(pooled, hdim, w3, b3, logits, c, b, hdim, c, 0);
  out[0] = out[1] = 0.f;
  orc_softmax_cce(logits, y, b, c, denom, probs, out, dlogits, from_logits);

  orc_gemm_dw(pooled, hdim, dlogits, c, gw3, b, hdim, c);
  orc_colsum(dlogits, c, b, c, gb3);
  orc_gemm_dx(dlogits, c, w3, dpooled, hdim, b, hdim, c, NULL, 0);
  g_bf16 = bf16_saved;
  orc_pool_bwd(gp, dpooled, dz, hdim, b, hdim, 0, NULL, y2, hdim);
  orc_colsum(dz, hdim, n, hdim, gb2);
  orc_spmm_csr(rowptr, colidx, vals, dz, hdim, NULL, h, hdim, n, hdim, 0);
  orc_gemm_dw(y1, hdim, h, hdim, gw2, n, hdim, hdim);
  orc_gemm_dx(h, hdim, w2, dz, hdim, n, hdim, hdim, y1, hdim);
  orc_colsum(dz, hdim, n, hdim, gb1);
  if (g_s1) {
    orc_gemm_dw(g_s1, f, dz, hdim, gw1, n, f, hdim);
  } else {
    orc_spmm_csr(rowptr, colidx, vals, dz, hdim, NULL, h, hdim, n, hdim, 0);
    orc_gemm_dw(x, f, h, hdim, gw1, n, f, hdim);
  }
  if (lr > 0.f) {
    const int64_t np_ = (int64_t)f * hdim + hdim + (int64_t)hdim * hdim + hdim + (int64_t)hdim * c + c;
    for (int64_t i = 0; i < np_; ++i) params[i] -= lr * grads[i];
  }
  (void)gb3; (void)b3;
}
