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
#include "head_body.h"

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
  int vec_c;            // C (and mask) rows are 16-byte aligned: float4 epilogue stores
  float* colpart = nullptr;   // f32 kernel, float4 epilogue only: column sums of each wave's 32 rows of what it
                              // wrote -> colpart[(blockIdx.y * 2 + wm) * Nc + col] (BiasAddGrad partials)
};

// One 64x32 (or 32x64) operand tile, global -> registers (2 float4 per thread), then registers
// -> the k-major LDS image s[k][LD].  Splitting the two lets the loads of tile t+1 fly during
// the MFMAs of tile t.
//   KCONTIG:  element(i,k) = p[i*ld + k]   (row index is the M/N index; k contiguous)
//   !KCONTIG: element(i,k) = p[k*ld + i]   (k is the row index; M/N contiguous)
template <bool KCONTIG>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ p, int64_t ld, int64_t i0, int64_t i_end,
                                           int64_t k0, int64_t k_end, int tid, bool vec_ok, float4 (&v)[2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + 256 * q;
    v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KCONTIG) {
      const int i = idx >> 3, k = (idx & 7) * 4;
      const int64_t gi = i0 + i, gk = k0 + k;
      if (gi < i_end) {
        const float* src = p + gi * ld + gk;
        if (vec_ok && gk + 3 < k_end) {
          v[q] = *reinterpret_cast<const float4*>(src);
        } else {
          if (gk + 0 < k_end) v[q].x = src[0];
          if (gk + 1 < k_end) v[q].y = src[1];
          if (gk + 2 < k_end) v[q].z = src[2];
          if (gk + 3 < k_end) v[q].w = src[3];
        }
      }
    } else {
      const int k = idx >> 4, i = (idx & 15) * 4;
      const int64_t gi = i0 + i, gk = k0 + k;
      if (gk < k_end) {
        const float* src = p + gk * ld + gi;
        if (vec_ok && gi + 3 < i_end) {
          v[q] = *reinterpret_cast<const float4*>(src);
        } else {
          if (gi + 0 < i_end) v[q].x = src[0];
          if (gi + 1 < i_end) v[q].y = src[1];
          if (gi + 2 < i_end) v[q].z = src[2];
          if (gi + 3 < i_end) v[q].w = src[3];
        }
      }
    }
  }
}

// Interior tiles (checked once per workgroup: aligned operands, no ragged edge, whole K steps): plain float4 loads, one
// value at a time.  fetch_tile puts every load behind per-thread bounds tests with a scalar fallback, and hipcc ends
// each such conditional load with its own s_waitcnt vmcnt(0): the four loads of a K step -- and with them the
// prefetch of step t+1 under the MFMAs of step t -- ran one after the other (load and MFMA time were additive).
template <bool KCONTIG>
__device__ __forceinline__ float4 fetch_one(const float* __restrict__ p, int64_t ld, int64_t i0, int64_t k0, int idx) {
  if (KCONTIG) return *reinterpret_cast<const float4*>(p + (i0 + (idx >> 3)) * ld + k0 + (idx & 7) * 4);
  return *reinterpret_cast<const float4*>(p + (k0 + (idx >> 4)) * ld + i0 + (idx & 15) * 4);
}
template <bool KCONTIG>
__device__ __forceinline__ void store_one(float (*s)[LD], int idx, float4 v) {
  if (KCONTIG) {
    const int i = idx >> 3, k = (idx & 7) * 4;
    s[k + 0][i] = v.x; s[k + 1][i] = v.y; s[k + 2][i] = v.z; s[k + 3][i] = v.w;
  } else {
    *reinterpret_cast<float4*>(&s[idx >> 4][(idx & 15) * 4]) = v;
  }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_tile(float (*s)[LD], int tid, const float4 (&v)[2]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + 256 * q;
    if (KCONTIG) {
      const int i = idx >> 3, k = (idx & 7) * 4;
      s[k + 0][i] = v[q].x; s[k + 1][i] = v[q].y; s[k + 2][i] = v[q].z; s[k + 3][i] = v[q].w;
    } else {
      const int k = idx >> 4, i = (idx & 15) * 4;
      *reinterpret_cast<float4*>(&s[k][i]) = v[q];
    }
  }
}

// One 64x64 tile (bx, by) of C[M,Nc] = op(A) op(B) over k in [bz*kchunk, min(K,(bz+1)*kchunk)); bz = split-K slice
// (of nz).  With split-K (nz > 1) the raw partial tile goes to c + bz*M*ldc (a [S][M][ldc] slab).
template <bool A_KCONTIG, bool B_KCONTIG>
__device__ __forceinline__ void gemm_f32_tile(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                              int64_t ldb, float* __restrict__ c, int64_t ldc, int64_t M, int32_t Nc,
                                              int64_t K, int64_t kchunk, const Epilogue& ep, int vec_a, int vec_b,
                                              int bx, int by, int bz, int nz, float (*As)[BK][LD], float (*Bs)[BK][LD]) {
  // Two LDS images per operand (As[2], Bs[2]): the registers of K step t+1 are written to the other image while the
  // MFMAs of step t still read this one -- one barrier per K step instead of two, and the (transposing, up to 4-way
  // conflicting) LDS stores overlap other waves' MFMAs instead of standing between barriers.
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)by * BM;
  const int64_t n0 = (int64_t)bx * BN;
  const int64_t kbeg = (int64_t)bz * kchunk;
  const int64_t kend = min(K, kbeg + kchunk);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const int fr = lane & 31, fk = lane >> 5;
  const bool interior = vec_a && vec_b && m0 + BM <= M && n0 + BN <= Nc && kbeg < kend && (kend - kbeg) % BK == 0;
  if (interior) {            // uniform per workgroup
    float4 a0 = fetch_one<A_KCONTIG>(a, lda, m0, kbeg, tid), a1 = fetch_one<A_KCONTIG>(a, lda, m0, kbeg, tid + 256);
    float4 b0 = fetch_one<B_KCONTIG>(b, ldb, n0, kbeg, tid), b1 = fetch_one<B_KCONTIG>(b, ldb, n0, kbeg, tid + 256);
    store_one<A_KCONTIG>(As[0], tid, a0); store_one<A_KCONTIG>(As[0], tid + 256, a1);
    store_one<B_KCONTIG>(Bs[0], tid, b0); store_one<B_KCONTIG>(Bs[0], tid + 256, b1);
    if (kbeg + BK < kend) {
      a0 = fetch_one<A_KCONTIG>(a, lda, m0, kbeg + BK, tid); a1 = fetch_one<A_KCONTIG>(a, lda, m0, kbeg + BK, tid + 256);
      b0 = fetch_one<B_KCONTIG>(b, ldb, n0, kbeg + BK, tid); b1 = fetch_one<B_KCONTIG>(b, ldb, n0, kbeg + BK, tid + 256);
    }
    int cur = 0;
    for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
      __syncthreads();
      if (k0 + BK < kend) {
        store_one<A_KCONTIG>(As[cur ^ 1], tid, a0); store_one<A_KCONTIG>(As[cur ^ 1], tid + 256, a1);
        store_one<B_KCONTIG>(Bs[cur ^ 1], tid, b0); store_one<B_KCONTIG>(Bs[cur ^ 1], tid + 256, b1);
        if (k0 + 2 * BK < kend) {
          a0 = fetch_one<A_KCONTIG>(a, lda, m0, k0 + 2 * BK, tid); a1 = fetch_one<A_KCONTIG>(a, lda, m0, k0 + 2 * BK, tid + 256);
          b0 = fetch_one<B_KCONTIG>(b, ldb, n0, k0 + 2 * BK, tid); b1 = fetch_one<B_KCONTIG>(b, ldb, n0, k0 + 2 * BK, tid + 256);
        }
      }
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const float av = As[cur][2 * kk + fk][wm * 32 + fr];
        const float bv = Bs[cur][2 * kk + fk][wn * 32 + fr];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
      cur ^= 1;
    }
  } else {
  float4 ra[2], rb[2];
  if (kbeg < kend) {
    fetch_tile<A_KCONTIG>(a, lda, m0, M, kbeg, kend, tid, vec_a, ra);
    fetch_tile<B_KCONTIG>(b, ldb, n0, Nc, kbeg, kend, tid, vec_b, rb);
    store_tile<A_KCONTIG>(As[0], tid, ra);
    store_tile<B_KCONTIG>(Bs[0], tid, rb);
    if (kbeg + BK < kend) {
      fetch_tile<A_KCONTIG>(a, lda, m0, M, kbeg + BK, kend, tid, vec_a, ra);
      fetch_tile<B_KCONTIG>(b, ldb, n0, Nc, kbeg + BK, kend, tid, vec_b, rb);
    }
  }
  int cur = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();        // image `cur` is complete; everybody has finished reading image cur^1 (step t-1)
    if (k0 + BK < kend) {   // registers hold step t+1: into the other image, then fetch step t+2
      store_tile<A_KCONTIG>(As[cur ^ 1], tid, ra);
      store_tile<B_KCONTIG>(Bs[cur ^ 1], tid, rb);
      if (k0 + 2 * BK < kend) {
        fetch_tile<A_KCONTIG>(a, lda, m0, M, k0 + 2 * BK, kend, tid, vec_a, ra);
        fetch_tile<B_KCONTIG>(b, ldb, n0, Nc, k0 + 2 * BK, kend, tid, vec_b, rb);
      }
    }
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float av = As[cur][2 * kk + fk][wm * 32 + fr];
      const float bv = Bs[cur][2 * kk + fk][wn * 32 + fr];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    cur ^= 1;
  }
  }

  // Epilogue.  C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5): stored
  // from the accumulators that is 16 four-byte stores per lane.  With 16-byte-aligned rows each wave instead passes
  // its tile through LDS (the operand images are dead once every wave has left the K loop) and writes float4s:
  // 4 store instructions of 8 full 128-byte row segments each; bias / activation / mask / accumulate on the float4.
  float* cz = c + (nz > 1 ? (int64_t)bz * M * ldc : 0);
  const int64_t tr0 = m0 + wm * 32, tc0 = n0 + wn * 32;
  if (ep.vec_c && tc0 + 31 < Nc) {               // uniform per wave; ragged right edge: the scalar path below
    __syncthreads();
    float(*st)[36] = reinterpret_cast<float(*)[36]>(wave < 2 ? &As[0][0][0] : &Bs[0][0][0]) + (wave & 1) * 32;
    static_assert(2 * 32 * 36 <= 2 * BK * LD, "two waves' staging must fit in one operand's images");
#pragma unroll
    for (int r = 0; r < 16; ++r) st[(r & 3) + 8 * (r >> 2) + 4 * fk][fr] = acc[r];
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    const int ec = (lane & 7) * 4, er = lane >> 3;   // this lane's 4 columns and its row within each group of 8
    const int64_t gcol = tc0 + ec;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), alpha4 = bias4, csum = bias4;
    if (ep.bias) bias4 = make_float4(ep.bias[gcol], ep.bias[gcol + 1], ep.bias[gcol + 2], ep.bias[gcol + 3]);
    if (ep.alpha) alpha4 = make_float4(ep.alpha[gcol], ep.alpha[gcol + 1], ep.alpha[gcol + 2], ep.alpha[gcol + 3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int lr = q * 8 + er;
      const int64_t row = tr0 + lr;
      if (row >= M) continue;
      float4 v = *reinterpret_cast<const float4*>(&st[lr][ec]);
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (ep.act == GCNX_ACT_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      } else if (ep.act == GCNX_ACT_PRELU) {
        v.x = v.x > 0.f ? v.x : alpha4.x * v.x; v.y = v.y > 0.f ? v.y : alpha4.y * v.y;
        v.z = v.z > 0.f ? v.z : alpha4.z * v.z; v.w = v.w > 0.f ? v.w : alpha4.w * v.w;
      }
      if (ep.mask) {
        const float4 mk = *reinterpret_cast<const float4*>(ep.mask + row * ep.ldmask + gcol);
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      }
      float* dst = cz + row * ldc + gcol;
      if (ep.accumulate) {
        const float4 old = *reinterpret_cast<const float4*>(dst);
        v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
      }
      *reinterpret_cast<float4*>(dst) = v;
      csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;
    }
    if (ep.colpart) {     // rows q*8 + er summed in-lane above (q ascending), then over er: fixed order, no atomics
#pragma unroll
      for (int off = 8; off < 64; off <<= 1) {
        csum.x += __shfl_xor(csum.x, off); csum.y += __shfl_xor(csum.y, off);
        csum.z += __shfl_xor(csum.z, off); csum.w += __shfl_xor(csum.w, off);
      }
      if (lane < 8) *reinterpret_cast<float4*>(ep.colpart + ((int64_t)by * 2 + wm) * Nc + gcol) = csum;
    }
    return;
  }
  const int64_t col = n0 + wn * 32 + fr;
  if (col >= Nc) return;
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

template <bool A_KCONTIG, bool B_KCONTIG>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ a, int64_t lda,
                                                       const float* __restrict__ b, int64_t ldb,
                                                       float* __restrict__ c, int64_t ldc, int64_t M, int32_t Nc,
                                                       int64_t K, int64_t kchunk, Epilogue ep, int vec_a, int vec_b) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];
  gemm_f32_tile<A_KCONTIG, B_KCONTIG>(a, lda, b, ldb, c, ldc, M, Nc, K, kchunk, ep, vec_a, vec_b, blockIdx.x, blockIdx.y,
                                      blockIdx.z, gridDim.z, As, Bs);
}

// The backward of one Dense layer in ONE launch: workgroups [0, n_dx) are the tiles of dX = dH W^T (k-contiguous
// operands, mask / column-sum epilogue), the rest the split-K tiles of dW = X^T dH.  Both products read dH; run as
// two launches they either serialise (dW is a gradient leaf nobody downstream waits for) or need a second stream
// whose fork/join costs more than dW itself on small batches.
struct GemmJob {
  const float* a; int64_t lda;
  const float* b; int64_t ldb;
  float* c; int64_t ldc;
  int64_t M; int32_t Nc; int64_t K, kchunk;
  Epilogue ep;
  int vec_a, vec_b;
  int gx, gy, gz;     // tile grid of this job
};

__global__ __launch_bounds__(256) void gemm_f32_duo_kernel(GemmJob dx, GemmJob dw, int n_dx) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];
  int bid = blockIdx.x;                      // uniform per workgroup: the two branches below never mix in a wave
  if (bid < n_dx) {
    const int bx = bid % dx.gx, by = bid / dx.gx;
    gemm_f32_tile<true, true>(dx.a, dx.lda, dx.b, dx.ldb, dx.c, dx.ldc, dx.M, dx.Nc, dx.K, dx.kchunk, dx.ep, dx.vec_a,
                              dx.vec_b, bx, by, 0, 1, As, Bs);
  } else {
    bid -= n_dx;
    const int bx = bid % dw.gx, t = bid / dw.gx;
    gemm_f32_tile<false, false>(dw.a, dw.lda, dw.b, dw.ldb, dw.c, dw.ldc, dw.M, dw.Nc, dw.K, dw.kchunk, dw.ep, dw.vec_a,
                                dw.vec_b, bx, t % dw.gy, t / dw.gy, dw.gz, As, Bs);
  }
}

// Two weight gradients X^T dH over the same rows in one launch (gcnx_gemm_dw2): workgroups [0, n_a) are the split-K
// tiles of job a, the rest those of job b.
__global__ __launch_bounds__(256) void gemm_f32_dw2_kernel(GemmJob ja, GemmJob jb, int n_a) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];
  int bid = blockIdx.x;                      // uniform per workgroup
  if (bid < n_a) {
    const int bx = bid % ja.gx, t = bid / ja.gx;
    gemm_f32_tile<false, false>(ja.a, ja.lda, ja.b, ja.ldb, ja.c, ja.ldc, ja.M, ja.Nc, ja.K, ja.kchunk, ja.ep, ja.vec_a,
                                ja.vec_b, bx, t % ja.gy, t / ja.gy, ja.gz, As, Bs);
  } else {
    bid -= n_a;
    const int bx = bid % jb.gx, t = bid / jb.gx;
    gemm_f32_tile<false, false>(jb.a, jb.lda, jb.b, jb.ldb, jb.c, jb.ldc, jb.M, jb.Nc, jb.K, jb.kchunk, jb.ep, jb.vec_a,
                                jb.vec_b, bx, t % jb.gy, t / jb.gy, jb.gz, As, Bs);
  }
}

// gcnx_gemm_dw2 with the classifier head's leaves (gcnx_head_args): the FIRST n_head workgroups of the launch run the head
// kernel's body (csrc/head_body.h: Dense(softmax) + CCE + accuracy + dW3, db3, db_relu from the pool's partial sums) --
// 10 us of one workgroup that used to be a launch of its own between the pool and the backward aggregation; here it
// overlaps the 500-odd split-K tiles.  One LDS block serves both roles (the tile images / the head's staged operands).
struct HeadLeaf {
  const float* w; const float* bias; const float* y; int32_t b, h, c; float denom;
  float* probs; float* loss_acc; float* dw; float* db; float* dpooled; int64_t lddp; int64_t ldp;
  float* slabs; int* ticket; gcnx_head::PoolParts pp; int from_logits;
};
constexpr int kDw2HeadLds = 9216;          // floats: >= 2 x 2 x BK x LD (8704) and the head's staged operands at h = 128, c = 2 (9024)

// (The head body's combine runs with ZU = 1: at its stand-alone unroll the body holds 208 VGPRs and the WHOLE launch drops
// to two waves per SIMD -- the tiles then take 24.2 us instead of 22.3.  106 VGPRs as built; check after any change.)
__global__ __launch_bounds__(256) void gemm_f32_dw2_head_kernel(GemmJob ja, GemmJob jb, int n_a, HeadLeaf hl, int n_head) {
  struct Lds { float As[2][BK][LD]; float Bs[2][BK][LD]; float pad[kDw2HeadLds - 2 * 2 * BK * LD]; };
  __shared__ __attribute__((aligned(16))) Lds sm;
  static_assert(2 * 2 * BK * LD <= kDw2HeadLds, "tile images");
  float* smem = &sm.As[0][0][0];
  int bid = blockIdx.x;                      // uniform per workgroup
  if (bid < n_head) {
    // (two classes only -- the host checks: the generic instance keeps 32 class slots of static LDS per graph, which would
    // cost this launch a workgroup per CU)
    __builtin_amdgcn_s_setprio(3);     // one workgroup, a chain of dependent phases: its waves go first among the CU's tile waves
    gcnx_head::head_body<true, true, 2, 1>(nullptr, hl.ldp, hl.w, hl.bias, hl.y, hl.b, hl.h, hl.c, hl.denom, hl.probs, hl.loss_acc, hl.dw,
                                        hl.db, hl.dpooled, hl.lddp, hl.slabs, hl.ticket, hl.pp, hl.from_logits, smem, bid, n_head);
    return;
  }
  bid -= n_head;
  if (bid < n_a) {
    const int bx = bid % ja.gx, t = bid / ja.gx;
    gemm_f32_tile<false, false>(ja.a, ja.lda, ja.b, ja.ldb, ja.c, ja.ldc, ja.M, ja.Nc, ja.K, ja.kchunk, ja.ep, ja.vec_a,
                                ja.vec_b, bx, t % ja.gy, t / ja.gy, ja.gz, sm.As, sm.Bs);
  } else {
    bid -= n_a;
    const int bx = bid % jb.gx, t = bid / jb.gx;
    gemm_f32_tile<false, false>(jb.a, jb.lda, jb.b, jb.ldb, jb.c, jb.ldc, jb.M, jb.Nc, jb.K, jb.kchunk, jb.ep, jb.vec_a,
                                jb.vec_b, bx, t % jb.gy, t / jb.gy, jb.gz, sm.As, sm.Bs);
  }
}

// Second stage of the deterministic split-K: out[i] = sum_s part[s][i].  Block = 64 outputs x 4
// split groups; each group sums its splits in ascending order, the 4 group sums are combined in
// a fixed order -- the result does not depend on scheduling.
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ part, int64_t slab, int nsplit,
                                                   float* __restrict__ out, int64_t total, int bx, float (*s)[64]) {
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)bx * 64 + el;
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

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int64_t slab,
                                                            int nsplit, float* __restrict__ out, int64_t total) {
  __shared__ float s[4][64];
  splitk_reduce_body(part, slab, nsplit, out, total, blockIdx.x, s);
}

// The same reduction for MANY slabs (the streaming dW kernels leave one 256 KiB slab per CU): four outputs per thread
// (16-byte loads), eight slabs in flight per thread.  With one output per thread and four loads in flight the 64 slabs
// of a group were sixteen dependent round trips of 256-byte requests: 113 us for 67 MB at config 3.  Same summation
// order per output as splitk_reduce_kernel (four groups of consecutive slabs, ascending, then (s0 + s1) + (s2 + s3)):
// the results are bit-identical.  Needs total % 4 == 0, slab % 4 == 0 and 16-byte aligned pointers.
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const float* __restrict__ part, int64_t slab, int nsplit,
                                                                 float* __restrict__ out, int64_t total) {
  __shared__ float4 s[4][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = ((int64_t)blockIdx.x * 64 + el) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < total) {
    const int per = (nsplit + 3) / 4;
    const int z0 = grp * per, z1 = min(nsplit, z0 + per);
    int z = z0;
    for (; z + 8 <= z1; z += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(part + (int64_t)(z + u) * slab + i);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; z < z1; ++z) {
      const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)z * slab + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  s[grp][el] = acc;
  __syncthreads();
  if (grp == 0 && i < total) {
    const float4 a = s[0][el], b = s[1][el], c = s[2][el], d = s[3][el];
    *reinterpret_cast<float4*>(out + i) = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y),
                                                      (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
  }
}

// The last gradient and the optimizer step in one launch (gcnx_gemm_dw_sgd).  Workgroups, in order:
//   [0, n_s)        fold this call's split-K slabs into dW (an interval of the flat gradient buffer) and update it;
//   [.., + n_pc)    fold a PENDING column-sum reduction (gcnx_dense_bwd_deferred) and update its parameters;
//   [.., + n_ps)    fold a PENDING split-K reduction and update its parameters;
//   the rest        p -= lr * g for every parameter outside those three intervals, 256 per workgroup.
// params == NULL: the reductions only (gcnx_gemm_dw2 in a multi-GPU step: the all-reduce comes before the update).
struct SgdPending {
  const float* cpart; int64_t crows; int32_t cf; int64_t coff;      // partial rows -> grads[coff, coff + cf)
  const float* slabs; int64_t total; int32_t nsplit; int64_t soff;  // slabs -> grads[soff, soff + total)
  int n_pc, n_ps;
};

__device__ __forceinline__ float splitk_sum(const float* __restrict__ part, int64_t slab, int nsplit, int64_t i,
                                            int64_t total, float (*s)[64]) {
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  float acc = 0.f;
  if (i < total) {
    const int per = (nsplit + 3) / 4;
    const int z0 = grp * per, z1 = min(nsplit, z0 + per);
#pragma unroll 4
    for (int z = z0; z < z1; ++z) acc += part[(int64_t)z * slab + i];
  }
  s[grp][el] = acc;
  __syncthreads();
  return (s[0][el] + s[1][el]) + (s[2][el] + s[3][el]);              // the order of splitk_reduce_kernel
}

__global__ __launch_bounds__(256) void reduce_sgd_kernel(const float* __restrict__ part, int64_t slab, int nsplit,
                                                         int64_t total, int n_s, float* __restrict__ params,
                                                         float* __restrict__ grads, int64_t off, int64_t n_params,
                                                         float lr_arg, SgdPending pd, const float* __restrict__ lr_dev) {
  const float lr = lr_dev ? *lr_dev : lr_arg;            // (gcnx_set_lr_source)
  __shared__ float4 s4[128][2];
  float (*s)[64] = reinterpret_cast<float(*)[64]>(&s4[0][0]);
  int bid = blockIdx.x;
  if (bid < n_s) {
    const int64_t i = (int64_t)bid * 64 + (threadIdx.x & 63);
    const float g = splitk_sum(part, slab, nsplit, i, total, s);
    if ((threadIdx.x >> 6) == 0 && i < total) { grads[off + i] = g; if (params) params[off + i] = params[off + i] - lr * g; }
    return;
  }
  bid -= n_s;
  if (bid < pd.n_pc) {
    gcnx_colpart_reduce_body(pd.cpart, pd.crows, pd.cf, grads + pd.coff, bid, s4);       // writes grads (all threads sync inside)
    const int cl = threadIdx.x & 1, c = bid * 8 + cl * 4;
    if ((threadIdx.x >> 1) == 0 && c < pd.cf && params) {
      const float4 g = s4[0][cl];
      float* p = params + pd.coff + c;
      p[0] -= lr * g.x; p[1] -= lr * g.y; p[2] -= lr * g.z; p[3] -= lr * g.w;
    }
    return;
  }
  bid -= pd.n_pc;
  if (bid < pd.n_ps) {
    const int64_t i = (int64_t)bid * 64 + (threadIdx.x & 63);
    const float g = splitk_sum(pd.slabs, pd.total, pd.nsplit, i, pd.total, s);
    if ((threadIdx.x >> 6) == 0 && i < pd.total) { grads[pd.soff + i] = g; if (params) params[pd.soff + i] = params[pd.soff + i] - lr * g; }
    return;
  }
  bid -= pd.n_ps;
  const int64_t i = (int64_t)bid * 256 + threadIdx.x;
  if (i >= n_params || !params) return;
  if (i >= off && i < off + total) return;
  if (pd.n_pc && i >= pd.coff && i < pd.coff + pd.cf) return;
  if (pd.n_ps && i >= pd.soff && i < pd.soff + pd.total) return;
  params[i] = params[i] - lr * grads[i];
}

// Both second stages of gcnx_dense_bwd in one launch: workgroups [0, n_c) fold the dX epilogue's partial column
// sums into db, the rest the split-K slabs into dW.
__global__ __launch_bounds__(256) void reduce_duo_kernel(const float* __restrict__ cpart, int64_t crows, int32_t cf,
                                                         float* __restrict__ cout, int n_c,
                                                         const float* __restrict__ part, int64_t slab, int nsplit,
                                                         float* __restrict__ out, int64_t total) {
  __shared__ float4 s4[128][2];
  if ((int)blockIdx.x < n_c) gcnx_colpart_reduce_body(cpart, crows, cf, cout, blockIdx.x, s4);
  else splitk_reduce_body(part, slab, nsplit, out, total, blockIdx.x - n_c, reinterpret_cast<float(*)[64]>(&s4[0][0]));
}

// ----------------------------------------------------------------------------------------------
// bf16 MFMA path (GCNX_PREC_BF16, GCNX_PREC_BF16X3): v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//
// BF16X3 splits every fp32 operand into hi = bf16(x), lo = bf16(x - hi) and accumulates
// hi*hi + hi*lo + lo*hi (the dropped lo*lo term is 2^-18 relative): fp32-grade results at three
// MFMA passes, still far below the HBM time of these shapes (F <= a few hundred: AI 64-128
// flop/B against a bf16 machine balance of ~300).  128x128 block tile, 4 waves of 64x64
// (4x4 MFMA tiles, 64 accumulator registers), K steps of 32.  LDS images are [row][k] bf16 with
// an 80-byte row stride: the operand read -- lane l takes the 16 bytes k = 8*(l>>4).. of row
// l&15 -- is one ds_read_b128 per 16x32 fragment, bank-conflict-free.
// The streamed operand is converted in registers on its way to LDS; the small weight operand is
// converted (and transposed for X*W) once per call into a [col][k] bf16 image in the workspace.
// X^T*dH (K = the N rows) transposes both operands 4x4 in registers.
// ----------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int HM = 128, HN = 128, HK = 32, HLD = 40;   // HLD bf16 elements = 80 B row stride

__device__ __forceinline__ void split4(float4 v, bf16x4& hi, bf16x4& lo) {
  hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
  lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
  lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
}

// Weight operand -> [col][kpad] bf16 hi/lo images.  TRANSPOSE: out[o][i] = W[i][o] (X*W);
// otherwise out[i][o] = W[i][o] (dH*W^T).  Columns k >= K are zero.
__global__ __launch_bounds__(256) void wprep_kernel(const float* __restrict__ w, int fi, int fo, int transpose,
                                                    int kpad, __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
  const int ncol = transpose ? fo : fi, K = transpose ? fi : fo;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)ncol * kpad) return;
  const int col = (int)(idx / kpad), k = (int)(idx % kpad);
  float v = 0.f;
  if (k < K) v = transpose ? w[(int64_t)k * fo + col] : w[(int64_t)col * fo + k];
  const __bf16 h = (__bf16)v;
  hi[idx] = h;
  lo[idx] = (__bf16)(v - (float)h);
}

// MODE 0: C = A(fp32 [M,K], k contiguous) * B(bf16 image [Nc][kpad])           (X*W and dH*W^T)
// MODE 1: C = A^T * B with A = fp32 [K,M] rows, B = fp32 [K,Nc] rows, split-K   (X^T*dH)
template <int MODE, bool X3>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(const float* __restrict__ a, int64_t lda,
                                                        const float* __restrict__ bf, int64_t ldb,
                                                        const __bf16* __restrict__ bhi, const __bf16* __restrict__ blo,
                                                        int kpad, float* __restrict__ c, int64_t ldc, int64_t M,
                                                        int32_t Nc, int64_t K, int64_t kchunk, Epilogue ep, int vec_a,
                                                        int vec_b) {
  // one LDS object: A image | B image (hi then lo planes); the epilogue reuses it as fp32 staging
  constexpr int NP = X3 ? 2 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 smem[NP * (HM + HN) * HLD];
  __bf16(*As)[HM][HLD] = reinterpret_cast<__bf16(*)[HM][HLD]>(smem);
  __bf16(*Bs)[HN][HLD] = reinterpret_cast<__bf16(*)[HN][HLD]>(smem + NP * HM * HLD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * HM;
  const int64_t n0 = (int64_t)blockIdx.x * HN;
  const int64_t kbeg = (int64_t)blockIdx.z * kchunk;
  const int64_t kend = min(K, kbeg + kchunk);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  for (int64_t k0 = kbeg; k0 < kend; k0 += HK) {
    if (MODE == 0) {
      // A tile 128 rows x 32 k (fp32): thread -> rows (tid>>3) + 32*q, k = (tid&7)*4
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = (tid >> 3) + 32 * q, k = (tid & 7) * 4;
        const int64_t gr = m0 + r, gk = k0 + k;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < M) {
          const float* src = a + gr * lda + gk;
          if (vec_a && gk + 3 < kend) v = *reinterpret_cast<const float4*>(src);
          else {
            if (gk + 0 < kend) v.x = src[0];
            if (gk + 1 < kend) v.y = src[1];
            if (gk + 2 < kend) v.z = src[2];
            if (gk + 3 < kend) v.w = src[3];
          }
        }
        bf16x4 hi, lo;
        split4(v, hi, lo);
        *reinterpret_cast<bf16x4*>(&As[0][r][k]) = hi;
        if (X3) *reinterpret_cast<bf16x4*>(&As[X3 ? 1 : 0][r][k]) = lo;
      }
      // B tile 128 cols x 32 k from the bf16 images: thread -> col (tid>>2) + 64*q, k = (tid&3)*8
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int col = (tid >> 2) + 64 * q, k = (tid & 3) * 8;
        const int64_t gc = n0 + col;
        bf16x8 h = {0, 0, 0, 0, 0, 0, 0, 0}, l = h;
        if (gc < Nc) {   // kpad is a multiple of 32 and zero padded: no k guard
          h = *reinterpret_cast<const bf16x8*>(bhi + gc * kpad + k0 + k);
          if (X3) l = *reinterpret_cast<const bf16x8*>(blo + gc * kpad + k0 + k);
        }
        *reinterpret_cast<bf16x8*>(&Bs[0][col][k]) = h;
        if (X3) *reinterpret_cast<bf16x8*>(&Bs[X3 ? 1 : 0][col][k]) = l;
      }
    } else {
      // both operands are [k][m] fp32 rows: 4 k-rows x 4 columns per thread, transposed in registers
#pragma unroll
      for (int which = 0; which < 2; ++which) {
        const float* p = which == 0 ? a : bf;
        const int64_t ld = which == 0 ? lda : ldb;
        const int64_t i0 = which == 0 ? m0 : n0, iend = which == 0 ? M : (int64_t)Nc;
        const bool vec = which == 0 ? vec_a : vec_b;
        const int kq = (tid >> 5) * 4, i = (tid & 31) * 4;
        float4 v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t gk = k0 + kq + r, gi = i0 + i;
          v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (gk < kend) {
            const float* src = p + gk * ld + gi;
            if (vec && gi + 3 < iend) v[r] = *reinterpret_cast<const float4*>(src);
            else {
              if (gi + 0 < iend) v[r].x = src[0];
              if (gi + 1 < iend) v[r].y = src[1];
              if (gi + 2 < iend) v[r].z = src[2];
              if (gi + 3 < iend) v[r].w = src[3];
            }
          }
        }
        const float4 t0 = make_float4(v[0].x, v[1].x, v[2].x, v[3].x), t1 = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
        const float4 t2 = make_float4(v[0].z, v[1].z, v[2].z, v[3].z), t3 = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
        const float4 tt[4] = {t0, t1, t2, t3};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x4 hi, lo;
          split4(tt[j], hi, lo);
          __bf16(*dst)[HM][HLD] = which == 0 ? As : Bs;
          *reinterpret_cast<bf16x4*>(&dst[0][i + j][kq]) = hi;
          if (X3) *reinterpret_cast<bf16x4*>(&dst[X3 ? 1 : 0][i + j][kq]) = lo;
        }
      }
    }
    __syncthreads();
    bf16x8 ah[4], bh[4], al[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(&As[0][wm * 64 + i * 16 + fr][fq * 8]);
      bh[i] = *reinterpret_cast<const bf16x8*>(&Bs[0][wn * 64 + i * 16 + fr][fq * 8]);
      if (X3) {
        al[i] = *reinterpret_cast<const bf16x8*>(&As[X3 ? 1 : 0][wm * 64 + i * 16 + fr][fq * 8]);
        bl[i] = *reinterpret_cast<const bf16x8*>(&Bs[X3 ? 1 : 0][wn * 64 + i * 16 + fr][fq * 8]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (X3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
      }
    __syncthreads();
  }

  // Epilogue.  C/D map of a 16x16 tile: col = lane & 15, row = (lane >> 4)*4 + reg -- stored straight from the
  // accumulators that is 4 bytes per lane in 64-byte pieces, store-issue bound for a [N, F] output.  Each wave
  // instead passes its 16 x 64 strips through LDS (the operand images are dead by now) and writes whole 256-byte
  // row segments, one float4 per lane; bias / activation / mask / accumulate are applied on that float4.
  float* cz = c + (gridDim.z > 1 ? (int64_t)blockIdx.z * M * ldc : 0);
  float(*stage)[68] = reinterpret_cast<float(*)[68]>(reinterpret_cast<char*>(smem) + wave * (16 * 68 * 4));
  static_assert(sizeof(smem) >= 4 * 16 * 68 * 4, "epilogue staging must fit in the operand images");
  const int ec = (lane & 15) * 4;            // this lane's 4 columns of the 64-column strip
  const int er = lane >> 4;                  // and its row within each group of 4 rows
  const int64_t gcol = n0 + wn * 64 + ec;
  const bool col_vec = ep.vec_c && gcol + 3 < Nc;
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), alpha4 = bias4;
  if (ep.bias) {
    if (gcol + 0 < Nc) bias4.x = ep.bias[gcol + 0];
    if (gcol + 1 < Nc) bias4.y = ep.bias[gcol + 1];
    if (gcol + 2 < Nc) bias4.z = ep.bias[gcol + 2];
    if (gcol + 3 < Nc) bias4.w = ep.bias[gcol + 3];
  }
  if (ep.alpha) {
    if (gcol + 0 < Nc) alpha4.x = ep.alpha[gcol + 0];
    if (gcol + 1 < Nc) alpha4.y = ep.alpha[gcol + 1];
    if (gcol + 2 < Nc) alpha4.z = ep.alpha[gcol + 2];
    if (gcol + 3 < Nc) alpha4.w = ep.alpha[gcol + 3];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // accumulators of the strip -> LDS (own region of this wave: no workgroup barrier needed)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[fq * 4 + r][j * 16 + fr] = acc[i][j][r];
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int lr = q * 4 + er;             // row of the strip
      const int64_t row = m0 + wm * 64 + i * 16 + lr;
      float4 v = *reinterpret_cast<const float4*>(&stage[lr][ec]);
      if (row < M && gcol < Nc) {
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (ep.act == GCNX_ACT_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (ep.act == GCNX_ACT_PRELU) {
          v.x = v.x > 0.f ? v.x : alpha4.x * v.x; v.y = v.y > 0.f ? v.y : alpha4.y * v.y;
          v.z = v.z > 0.f ? v.z : alpha4.z * v.z; v.w = v.w > 0.f ? v.w : alpha4.w * v.w;
        }
        float* dst = cz + row * ldc + gcol;
        if (col_vec) {
          if (ep.mask) {
            const float4 mk = *reinterpret_cast<const float4*>(ep.mask + row * ep.ldmask + gcol);
            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
          }
          if (ep.accumulate) {
            const float4 old = *reinterpret_cast<const float4*>(dst);
            v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
          }
          *reinterpret_cast<float4*>(dst) = v;
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (gcol + t < Nc) {
              float o = vv[t];
              if (ep.mask) o = ep.mask[row * ep.ldmask + gcol + t] > 0.f ? o : 0.f;
              if (ep.accumulate) o += dst[t];
              dst[t] = o;
            }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();         // the strip is consumed before the next one overwrites it
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int gcnx_colsum(gcnx_ctx* ctx, const float* x, int64_t ldx, int64_t n, int32_t f, float* out);  // reduce.hip

namespace {

// X*W (transpose = 1) and dH*W^T (transpose = 0) on the bf16 MFMA path.
// colsum_out / colsum_done (may be NULL): column sums of c wanted / set to 1 if this call produced them (the streaming
// kernel sums what it writes; otherwise the caller runs a column-sum pass).
int launch_bf16_nn(gcnx_ctx* ctx, const float* a, int64_t lda, const float* w, int fi, int fo, int transpose,
                   float* c, int64_t ldc, int64_t m, int prec, const Epilogue& ep, float* colsum_out = nullptr,
                   int* colsum_done = nullptr) {
  const int ncol = transpose ? fo : fi, K = transpose ? fi : fo;
  if (colsum_done) *colsum_done = 0;
  if (ep.vec_c && !ep.colpart) {            // tall activations: the streaming kernel (weights resident in LDS)
    int rs = gcnx_gemm_stream_nn(ctx, a, lda, w, fi, fo, transpose, c, ldc, m, prec, ep.bias, ep.alpha, ep.act, ep.mask, ep.ldmask,
                                 ep.accumulate, colsum_done ? colsum_out : nullptr);
    if (rs == GCNX_OK && colsum_done && colsum_out) *colsum_done = 1;
    if (rs == GCNX_ERR_UNSUPPORTED && colsum_done && colsum_out)   // (e.g. an unaligned db): without the sums
      rs = gcnx_gemm_stream_nn(ctx, a, lda, w, fi, fo, transpose, c, ldc, m, prec, ep.bias, ep.alpha, ep.act, ep.mask, ep.ldmask,
                               ep.accumulate, nullptr);
    if (rs != GCNX_ERR_UNSUPPORTED) return rs;
  }
  const int kpad = ((K + HK - 1) / HK) * HK;
  const size_t img = (size_t)ncol * kpad;   // elements per image
  int rc = gcnx_ws_reserve(ctx, 2 * img * sizeof(__bf16) + 256);
  if (rc) return rc;
  __bf16* hi = (__bf16*)ctx->ws;
  __bf16* lo = hi + ((img + 127) / 128) * 128;
  hipLaunchKernelGGL(wprep_kernel, dim3(gcnx_cdiv((long long)img, 256)), dim3(256), 0, ctx->stream, w, fi, fo, transpose,
                     kpad, hi, lo);
  GCNX_LAUNCH_OK(ctx);
  dim3 grid(gcnx_cdiv(ncol, HN), gcnx_cdiv(m, HM), 1);
  const int va = al16(a) && lda % 4 == 0;
  if (prec == GCNX_PREC_BF16X3)
    hipLaunchKernelGGL((gemm_bf16_kernel<0, true>), grid, dim3(256), 0, ctx->stream, a, lda, (const float*)nullptr,
                       (int64_t)0, hi, lo, kpad, c, ldc, m, ncol, (int64_t)K, (int64_t)kpad + HK, ep, va, 0);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<0, false>), grid, dim3(256), 0, ctx->stream, a, lda, (const float*)nullptr,
                       (int64_t)0, hi, lo, kpad, c, ldc, m, ncol, (int64_t)K, (int64_t)kpad + HK, ep, va, 0);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // namespace

// ----------------------------------------------------------------------------------------------
// fp32 row-tile GEMM, weights in registers (r2): X W and dH W^T for K <= 256, <= 256 output columns.
// The 64 x 64-tile kernel above reaches 51-66 % of the fp32 MFMA peak at config 3 and 30 % at GeneralGNN's sizes
// (N = 22 576, 256 x 256: 62 us for 2.96 GFLOP): per K step every workgroup pays a barrier, two transposing LDS stores
// and 2 LDS reads per MFMA.  Here the small operand never touches LDS: wave w owns output columns [16w, 16w + 16) and
// keeps its K x 16 slice of W in REGISTERS (K / 4 per lane, the v_mfma_f32_16x16x4_f32 B layout: lane l holds
// k = 4 kk + (l >> 4), column l & 15), loaded once per persistent workgroup.  The streamed operand goes through LDS
// untransposed -- a 32-row tile, row stride K + 4 floats, so that the A read S[l & 15][4 kk + (l >> 4)] is
// conflict-free -- double-buffered: the rows of tile t + 1 are in flight while tile t's MFMAs run, one barrier per tile.
// One LDS read per MFMA, epilogue (bias, ReLU / PReLU, ReLU mask, accumulate, column sums) from the accumulator
// layout.  The same structure as the product phase of csrc/fused.hip, which measures 80 % of the fp32 MFMA peak.
// ----------------------------------------------------------------------------------------------
constexpr int kRtRows = 32;

template <int K>
__global__ __launch_bounds__(1024, 4) void gemm_f32_rowtile_kernel(const float* __restrict__ a, int64_t lda,
                                                                   const float* __restrict__ w, int64_t ldw,
                                                                   float* __restrict__ c, int64_t ldc, int64_t M, int32_t nc,
                                                                   Epilogue ep, int ntiles) {
  constexpr int LD = K + 4;
  constexpr int F4 = kRtRows * K / 4;                  // float4 pieces of a tile
  constexpr int NL = (F4 + 1023) / 1024;               // ... per thread
  extern __shared__ __attribute__((aligned(16))) float rt_lds[];
  float (*tile)[kRtRows][LD] = reinterpret_cast<float (*)[kRtRows][LD]>(rt_lds);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, kq = lane >> 4;
  const int col = 16 * wave + c16;
  const bool wave_on = 16 * wave < nc;
  typedef float rtf4 __attribute__((ext_vector_type(4)));

  float wreg[K / 4];
  if (wave_on) {
#pragma unroll
    for (int kk = 0; kk < K / 4; ++kk)
      wreg[kk] = w[(int64_t)(4 * kk + kq) * ldw + col];   // 64-byte row pieces (dH W^T hands in a transposed copy of W)
  }
  const float bcol = (ep.bias && wave_on) ? ep.bias[col] : 0.f;
  const float acol = (ep.alpha && wave_on) ? ep.alpha[col] : 0.f;

  float4 pa[NL];
  auto fetch = [&](int t) {
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int i = tid + q * 1024;
      const int row = i / (K / 4), c4 = i % (K / 4);
      const int64_t gr = (int64_t)t * kRtRows + row;
      pa[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < F4 && t < ntiles && gr < M) pa[q] = *reinterpret_cast<const float4*>(a + gr * lda + 4 * c4);
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int i = tid + q * 1024;
      if (i < F4) *reinterpret_cast<float4*>(&tile[buf][i / (K / 4)][4 * (i % (K / 4))]) = pa[q];
    }
  };
  int t = blockIdx.x;
  fetch(t);
  stash(0);
  __syncthreads();
  float cs = 0.f;                                       // column sum over this workgroup's tiles (ep.colpart)
  int cur = 0;
  for (; t < ntiles; t += gridDim.x) {
    fetch(t + gridDim.x);                               // the next tile's rows fly under this tile's MFMAs
    const int64_t r0 = (int64_t)t * kRtRows;
    float mk[8];
    if (ep.mask && wave_on) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int64_t row = r0 + 16 * (r >> 2) + 4 * kq + (r & 3);
        mk[r] = row < M ? ep.mask[row * ep.ldmask + col] : 0.f;
      }
    }
    float old[8];
    if (ep.accumulate && wave_on) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int64_t row = r0 + 16 * (r >> 2) + 4 * kq + (r & 3);
        old[r] = row < M ? c[row * ldc + col] : 0.f;
      }
    }
    rtf4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
    if (wave_on) {
#pragma unroll
      for (int kk = 0; kk < K / 4; ++kk) {
        const float a0 = tile[cur][c16][4 * kk + kq];
        const float a1 = tile[cur][16 + c16][4 * kk + kq];
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wreg[kk], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, wreg[kk], c1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int64_t row = r0 + 16 * (r >> 2) + 4 * kq + (r & 3);
        float v = (r < 4 ? c0[r & 3] : c1[r & 3]) + bcol;
        if (ep.act == GCNX_ACT_RELU) v = fmaxf(v, 0.f);
        else if (ep.act == GCNX_ACT_PRELU) v = v > 0.f ? v : acol * v;
        if (ep.mask) v = mk[r] > 0.f ? v : 0.f;
        if (ep.accumulate) v += old[r];
        if (row < M) { c[row * ldc + col] = v; cs += v; }
      }
    }
    stash(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  if (ep.colpart && wave_on) {                          // rows ascending within the lane, then the four row groups: fixed order
    cs += __shfl_xor(cs, 16);
    cs += __shfl_xor(cs, 32);
    if (lane < 16) ep.colpart[(int64_t)blockIdx.x * nc + col] = cs;
  }
}

// 1 if the row-tile kernel serves the shape (dispatches below); K is the reduction width, nc the output width
static bool rowtile_ok(const gcnx_ctx* ctx, int64_t m, int k, int nc, const float* a, int64_t lda) {
  // (>= 192 columns: at least 12 of the 16 waves have a column tile; narrower products measured no faster than the tiles)
  return ctx->knob_gemm_stream && m >= 2048 && (k == 16 || k == 32 || k == 64 || k == 128 || k == 256) && nc >= 192 && nc <= 256 &&
         nc % 16 == 0 && lda % 4 == 0 && al16(a);
}

// out[o][i] = w[i][o]: the [K, nc] operand of the row-tile kernel for dH W^T (a strided read of W in the kernel costs
// 16 lines per load instruction, once per workgroup: 20 us at N = 22 576 where a workgroup sees three tiles)
__global__ __launch_bounds__(256) void transpose_small_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows * cols) out[(int64_t)(i % cols) * rows + i / cols] = w[i];
}

static int launch_rowtile(gcnx_ctx* ctx, const float* a, int64_t lda, const float* w, int64_t ldw, float* c, int64_t ldc,
                          int64_t m, int k, int nc, const Epilogue& ep, int grid) {
  const int ntiles = gcnx_cdiv(m, kRtRows);
#define GCNX_RT(K_)                                                                                                      \
  do {                                                                                                                   \
    constexpr int lds_bytes = 2 * kRtRows * (K_ + 4) * 4;                                                                \
    static bool attr_set = false;                                                                                        \
    if (!attr_set) {                                                                                                     \
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_rowtile_kernel<K_>),                     \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));                         \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
    hipLaunchKernelGGL((gemm_f32_rowtile_kernel<K_>), dim3(grid), dim3(1024), lds_bytes, ctx->stream, a, lda, w,          \
                       ldw, c, ldc, m, nc, ep, ntiles);                                                                  \
  } while (0)
  switch (k) {
    case 16: GCNX_RT(16); break;
    case 32: GCNX_RT(32); break;
    case 64: GCNX_RT(64); break;
    case 128: GCNX_RT(128); break;
    default: GCNX_RT(256); break;
  }
#undef GCNX_RT
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

// Split-K slices of the f32 dW products are at least this many K steps deep (32 rows each): with loads and MFMAs
// overlapped, short slices only add slab traffic and reduction work (config 2, 641 steps: 256 slices of 2-3 steps ->
// 64 of 10: step 0.1496 -> 0.143 ms); long inputs still get ~4 workgroups per CU.
constexpr int64_t kMinSliceSteps = 10;

extern "C" {

int gcnx_gemm(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* w, const float* bias, float* out,
              int64_t ldo, int64_t n, int32_t fi, int32_t fo, int prec, int act, const float* alpha) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMM (X W)");
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm: negative size");
  GCNX_REQUIRE(ctx, act >= GCNX_ACT_NONE && act <= GCNX_ACT_PRELU, "gcnx_gemm: unknown activation %d", act);
  GCNX_REQUIRE(ctx, act != GCNX_ACT_PRELU || alpha, "gcnx_gemm: PReLU needs alpha");
  GCNX_REQUIRE(ctx, prec >= GCNX_PREC_F32 && prec <= GCNX_PREC_BF16X3, "gcnx_gemm: unknown precision %d", prec);
  if (n == 0 || fo == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, x && w && out, "gcnx_gemm: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= fi && ldo >= fo, "gcnx_gemm: leading dimension too small");
  Epilogue ep{bias, act == GCNX_ACT_PRELU ? alpha : nullptr, nullptr, 0, act, 0, al16(out) && ldo % 4 == 0};
  if (prec != GCNX_PREC_F32) return launch_bf16_nn(ctx, x, ldx, w, fi, fo, 1, out, ldo, n, prec, ep);
  if (rowtile_ok(ctx, n, fi, fo, x, ldx))
    return launch_rowtile(ctx, x, ldx, w, (int64_t)fo, out, ldo, n, fi, fo, ep, std::min(gcnx_cdiv(n, kRtRows), ctx->num_cus));
  dim3 grid(gcnx_cdiv(fo, BN), gcnx_cdiv(n, BM), 1);
  const int va = al16(x) && ldx % 4 == 0, vb = al16(w) && fo % 4 == 0;
  hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, ctx->stream, x, ldx, w, (int64_t)fo, out,
                     ldo, n, fo, (int64_t)fi, (int64_t)fi + BK, ep, va, vb);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

// The ReLU mask of a Dense layer as a BIT IMAGE between its forward and its backward product (large batches on the
// streaming bf16 kernel): gcnx_gemm_relu_bits is gcnx_gemm(act = RELU) that also writes [out > 0] -- 32 bytes per row, in
// the kernel's lane order (private to the pair) -- and gcnx_gemm_dx_bits masks dX with it instead of reading the saved
// activation again: 1 GB -> 32 MB per step at config 3.  GCNX_ERR_UNSUPPORTED (no launch) when the shape / precision is
// not the streaming kernel's one-plane form: use gcnx_gemm / gcnx_gemm_dx then.
int gcnx_gemm_relu_bits(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* w, const float* bias, float* out, int64_t ldo,
                        int64_t n, int32_t fi, int32_t fo, int prec, void* bits) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMM (X W)");
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm_relu_bits: negative size");
  GCNX_REQUIRE(ctx, x && w && out && bits, "gcnx_gemm_relu_bits: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= fi && ldo >= fo, "gcnx_gemm_relu_bits: leading dimension too small");
  int rc = GCNX_ERR_UNSUPPORTED;
  if (prec != GCNX_PREC_F32 && fo == 256 && al16(out) && ldo % 4 == 0)
    rc = gcnx_gemm_stream_nn(ctx, x, ldx, w, fi, fo, 1, out, ldo, n, prec, bias, nullptr, GCNX_ACT_RELU, nullptr, 0, 0, nullptr, nullptr, bits);
  // (UNSUPPORTED is an answer, not a failure: returned without a message, ctx's last error stays what it was -- a caller
  // that probes every step must not pay for formatting one, nor find a stale "unsupported" text after its fallback worked)
  return rc;
}

int gcnx_gemm_dx_bits(gcnx_ctx* ctx, const float* dh, int64_t lddh, const float* w, float* dx, int64_t lddx, int64_t n,
                      int32_t fi, int32_t fo, int prec, const void* mask_bits, float* db) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMM (dX)");
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm_dx_bits: negative size");
  GCNX_REQUIRE(ctx, dh && w && dx && mask_bits, "gcnx_gemm_dx_bits: NULL pointer");
  GCNX_REQUIRE(ctx, lddh >= fo && lddx >= fi, "gcnx_gemm_dx_bits: leading dimension too small");
  int rc = GCNX_ERR_UNSUPPORTED;
  if (prec != GCNX_PREC_F32 && fi == 256 && al16(dx) && lddx % 4 == 0 && (!db || al16(db)))
    rc = gcnx_gemm_stream_nn(ctx, dh, lddh, w, fi, fo, 0, dx, lddx, n, prec, nullptr, nullptr, GCNX_ACT_NONE, nullptr, 0, 0, db, mask_bits,
                             nullptr);
  return rc;                                   // (UNSUPPORTED without a message, as above)
}

// bf16 STORAGE between bf16-operand weight GEMMs (r3).  GCNX_PREC_BF16 rounds both operands of every product to bf16 when
// it loads them; an activation that only such products read can therefore be STORED as bf16 (round to nearest even at the
// producer) without changing a single bit of any result, and every kernel on the chain moves half the bytes.  These are
// the three products with the streamed operand(s) read as bf16 (uint16 rows, leading dimensions in elements) and, for the
// first two, the result written as bf16 (out_bf16) or fp32.  Streaming kernels only: GCNX_ERR_UNSUPPORTED, without a
// message and with nothing launched, for any other shape (fi = fo = 256, n >= 32768, 16-byte aligned rows).
int gcnx_gemm_stream_images(gcnx_ctx* ctx, int32_t njobs, const gcnx_stream_image_job* jobs) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, njobs >= 0 && njobs <= 4 && (njobs == 0 || jobs), "gcnx_gemm_stream_images: 0 .. 4 jobs");
  if (njobs == 0) return GCNX_OK;
  const float* w[4]; int tr[4]; void* img[4];
  for (int j = 0; j < njobs; ++j) {
    GCNX_REQUIRE(ctx, jobs[j].w && jobs[j].img && (reinterpret_cast<uintptr_t>(jobs[j].img) & 15) == 0,
                 "gcnx_gemm_stream_images: job %d: NULL weight / image, or an image that is not 16-byte aligned", j);
    GCNX_REQUIRE(ctx, jobs[j].fi == 256 && jobs[j].fo == 256, "gcnx_gemm_stream_images: job %d: the streaming bf16 kernels take 256 x 256 "
                 "operands (got %d x %d)", j, jobs[j].fi, jobs[j].fo);
    w[j] = jobs[j].w; tr[j] = jobs[j].transpose ? 1 : 0; img[j] = jobs[j].img;
  }
  return gcnx_gemm_stream_images_impl(ctx, njobs, w, tr, img);
}

int gcnx_gemm_fwd_bf16(gcnx_ctx* ctx, const void* x16, int64_t ldx, const float* w, const float* bias, void* out, int64_t ldo,
                       int out_bf16, int64_t n, int32_t fi, int32_t fo, int act, void* relu_bits, const void* wimg) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMM (X W)");
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm_fwd_bf16: negative size");
  GCNX_REQUIRE(ctx, x16 && w && out, "gcnx_gemm_fwd_bf16: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= fi && ldo >= fo, "gcnx_gemm_fwd_bf16: leading dimension too small");
  GCNX_REQUIRE(ctx, act == GCNX_ACT_NONE || act == GCNX_ACT_RELU, "gcnx_gemm_fwd_bf16: activation %d not supported here", act);
  GCNX_REQUIRE(ctx, !relu_bits || act == GCNX_ACT_RELU, "gcnx_gemm_fwd_bf16: the bit image is that of a ReLU output");
  if (fi != 256 || fo != 256 || (bias && !al16(bias))) return GCNX_ERR_UNSUPPORTED;
  return gcnx_gemm_stream_bf16(ctx, x16, ldx, w, fi, fo, 1, out, ldo, out_bf16, n, bias, act, nullptr, nullptr, relu_bits, wimg);
}

int gcnx_gemm_dx_bf16(gcnx_ctx* ctx, const void* dh16, int64_t lddh, const float* w, void* dx, int64_t lddx, int dx_bf16, int64_t n,
                      int32_t fi, int32_t fo, const void* mask_bits, float* db, const void* wimg) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMM (dX)");
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm_dx_bf16: negative size");
  GCNX_REQUIRE(ctx, dh16 && w && dx, "gcnx_gemm_dx_bf16: NULL pointer");
  GCNX_REQUIRE(ctx, lddh >= fo && lddx >= fi, "gcnx_gemm_dx_bf16: leading dimension too small");
  if (fi != 256 || fo != 256) return GCNX_ERR_UNSUPPORTED;
  return gcnx_gemm_stream_bf16(ctx, dh16, lddh, w, fi, fo, 0, dx, lddx, dx_bf16, n, nullptr, GCNX_ACT_NONE, db, mask_bits, nullptr, wimg);
}

int gcnx_gemm_dw_bf16(gcnx_ctx* ctx, const void* x16, int64_t ldx, const void* dh16, int64_t lddh, float* dw, int64_t n,
                      int32_t fi, int32_t fo) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMM (dW)");
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm_dw_bf16: negative size");
  GCNX_REQUIRE(ctx, x16 && dh16 && dw, "gcnx_gemm_dw_bf16: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= fi && lddh >= fo, "gcnx_gemm_dw_bf16: leading dimension too small");
  if (fi != 256 || fo != 256 || n < 32 * 1024 || !ctx->knob_gemm_stream || !al16(dw)) return GCNX_ERR_UNSUPPORTED;
  const int max_slices = ctx->num_cus;
  int rc = gcnx_ws_reserve(ctx, (size_t)max_slices * 65536 * sizeof(float));
  if (rc) return rc;
  const int ns = gcnx_gemm_dw_stream16(ctx, x16, ldx, dh16, lddh, (float*)ctx->ws, n, fi, fo, max_slices);
  if (ns < 0) return gcnx_fail(ctx, GCNX_ERR_HIP, "gcnx_gemm_dw_bf16: streaming kernel launch failed");
  if (ns == 0) return GCNX_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(gcnx_cdiv(65536, 256)), dim3(256), 0, ctx->stream, (const float*)ctx->ws, (int64_t)65536,
                     ns, dw, (int64_t)65536);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_gemm_dx(gcnx_ctx* ctx, const float* dh, int64_t lddh, const float* w, float* dx, int64_t lddx, int64_t n,
                 int32_t fi, int32_t fo, int prec, int accumulate, const float* y_mask, int64_t ldy, float* db) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMM (dX)");
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm_dx: negative size");
  GCNX_REQUIRE(ctx, prec >= GCNX_PREC_F32 && prec <= GCNX_PREC_BF16X3, "gcnx_gemm_dx: unknown precision %d", prec);
  GCNX_REQUIRE(ctx, !(accumulate && (y_mask || db)), "gcnx_gemm_dx: accumulate cannot be combined with mask/db");
  if (n == 0 || fi == 0) {
    if (db && fi > 0) GCNX_HIP(ctx, hipMemsetAsync(db, 0, (size_t)fi * 4, ctx->stream));
    return GCNX_OK;
  }
  GCNX_REQUIRE(ctx, dh && w && dx, "gcnx_gemm_dx: NULL pointer");
  GCNX_REQUIRE(ctx, lddh >= fo && lddx >= fi && (!y_mask || ldy >= fi), "gcnx_gemm_dx: leading dimension too small");
  // dX[n, i] = sum_o dH[n, o] * W[i, o]:  A = dH (k contiguous), B[k=o][j=i] = W[i*fo + o] (k contiguous).
  Epilogue ep{nullptr, nullptr, y_mask, ldy, GCNX_ACT_NONE, accumulate,
              al16(dx) && lddx % 4 == 0 && (!y_mask || (al16(y_mask) && ldy % 4 == 0))};
  if (prec != GCNX_PREC_F32) {
    int db_done = 0;
    int rc = launch_bf16_nn(ctx, dh, lddh, w, fi, fo, 0, dx, lddx, n, prec, ep, db, &db_done);
    if (rc) return rc;
    if (db && !db_done) return gcnx_colsum(ctx, dx, lddx, n, fi, db);
    return GCNX_OK;
  }
  if (rowtile_ok(ctx, n, fo, fi, dh, lddh) && (!db || al16(db))) {   // dX[n, i] = sum_o dH[n, o] W[i, o]: K = fo, columns = fi
    const int wgs = std::min(gcnx_cdiv(n, kRtRows), ctx->num_cus);
    // workspace: [db partial rows + their reduction's scratch | W^T]
    const size_t part_bytes = db ? (((size_t)gcnx_colsum_partials_ws(wgs, fi) + 255) & ~(size_t)255) : 0;
    int rc = gcnx_ws_reserve(ctx, part_bytes + (size_t)fi * fo * sizeof(float));
    if (rc) return rc;
    if (db) ep.colpart = (float*)ctx->ws;
    float* wt = (float*)((char*)ctx->ws + part_bytes);
    hipLaunchKernelGGL(transpose_small_kernel, dim3(gcnx_cdiv((int64_t)fi * fo, 256)), dim3(256), 0, ctx->stream, w, fi, fo, wt);
    GCNX_LAUNCH_OK(ctx);
    rc = launch_rowtile(ctx, dh, lddh, wt, (int64_t)fi, dx, lddx, n, fo, fi, ep, wgs);
    if (rc || !db) return rc;
    return gcnx_colsum_partials(ctx, wgs, fi, db);
  }
  dim3 grid(gcnx_cdiv(fi, BN), gcnx_cdiv(n, BM), 1);
  const int va = al16(dh) && lddh % 4 == 0, vb = al16(w) && fo % 4 == 0;
  // db: every wave adds up the columns of the 32 rows it writes (float4 epilogue, all column tiles full) and a
  // second launch sums those n/32 partial rows -- instead of a column-sum pass that reads dX back (2 launches over
  // the whole matrix; at config 2 they were 17 us of the critical path, the partial reduce is 5).
  const bool fused_db = db && ep.vec_c && fi % BN == 0;
  const int64_t prow = 2LL * grid.y;
  if (fused_db) {
    int rc = gcnx_ws_reserve(ctx, (size_t)gcnx_colsum_partials_ws(prow, fi));
    if (rc) return rc;
    ep.colpart = (float*)ctx->ws;
  }
  hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, ctx->stream, dh, lddh, w, (int64_t)fo, dx,
                     lddx, n, fi, (int64_t)fo, (int64_t)fo + BK, ep, va, vb);
  GCNX_LAUNCH_OK(ctx);
  if (fused_db) return gcnx_colsum_partials(ctx, prow, fi, db);
  if (db) return gcnx_colsum(ctx, dx, lddx, n, fi, db);
  return GCNX_OK;
}

}  // extern "C"

// nsplit / kchunk of the dW part of the fused dense backward (shared by the sizing helper and the launcher)
static int dense_bwd_split(const gcnx_ctx* ctx, int64_t n, int32_t fi, int32_t fo, int64_t* kchunk_out) {
  const int64_t gy = gcnx_cdiv(n, BM);
  const int n_dx = (int)gy * (fi / BN);
  const int tiles = gcnx_cdiv(fi, BM) * gcnx_cdiv(fo, BN);
  const int slots = 4 * ctx->num_cus;
  int spare = slots - n_dx % slots;
  if (spare < slots / 4) spare += slots;
  int nsplit = spare / tiles;
  const int64_t ksteps = (n + BK - 1) / BK;
  if (nsplit > ksteps / kMinSliceSteps) nsplit = (int)(ksteps / kMinSliceSteps);
  if (nsplit < 1) nsplit = 1;
  const int64_t kchunk = ((ksteps + nsplit - 1) / nsplit) * BK;
  if (kchunk_out) *kchunk_out = kchunk;
  return (int)((n + kchunk - 1) / kchunk);
}

static int dense_bwd_impl(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, const float* w, int64_t n,
                          int32_t fi, int32_t fo, int prec, float* dx, int64_t lddx, const float* y_mask, int64_t ldy,
                          float* db_prev, float* dw, float* scratch, int64_t scratch_floats, gcnx_pending_reduce* pending);

extern "C" {

int gcnx_dense_bwd(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, const float* w, int64_t n,
                   int32_t fi, int32_t fo, int prec, float* dx, int64_t lddx, const float* y_mask, int64_t ldy,
                   float* db_prev, float* dw) {
  return dense_bwd_impl(ctx, x, ldx, dh, lddh, w, n, fi, fo, prec, dx, lddx, y_mask, ldy, db_prev, dw, nullptr, 0, nullptr);
}

int64_t gcnx_dense_bwd_scratch_floats(gcnx_ctx* ctx, int64_t n, int32_t fi, int32_t fo) {
  if (!ctx || n <= 0 || fi <= 0 || fo <= 0 || fi % BN != 0) return 0;
  const int nsplit = dense_bwd_split(ctx, n, fi, fo, nullptr);
  return ((2 * (int64_t)gcnx_cdiv(n, BM) * fi + 63) & ~(int64_t)63) + (int64_t)nsplit * fi * fo;
}

int gcnx_dense_bwd_deferred(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, const float* w,
                            int64_t n, int32_t fi, int32_t fo, int prec, float* dx, int64_t lddx, const float* y_mask,
                            int64_t ldy, float* db_prev, float* dw, float* scratch, int64_t scratch_floats,
                            gcnx_pending_reduce* pending) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "dense backward (dX + dW)");
  GCNX_REQUIRE(ctx, pending != nullptr, "gcnx_dense_bwd_deferred: pending is NULL");
  return dense_bwd_impl(ctx, x, ldx, dh, lddh, w, n, fi, fo, prec, dx, lddx, y_mask, ldy, db_prev, dw, scratch, scratch_floats,
                        pending);
}

}  // extern "C"

static int dense_bwd_impl(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, const float* w, int64_t n,
                          int32_t fi, int32_t fo, int prec, float* dx, int64_t lddx, const float* y_mask, int64_t ldy,
                          float* db_prev, float* dw, float* scratch, int64_t scratch_floats, gcnx_pending_reduce* pending) {
  GCNX_CHECK_CTX(ctx);
  if (pending) *pending = gcnx_pending_reduce{nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr};
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_dense_bwd: negative size");
  GCNX_REQUIRE(ctx, prec >= GCNX_PREC_F32 && prec <= GCNX_PREC_BF16X3, "gcnx_dense_bwd: unknown precision %d", prec);
  GCNX_REQUIRE(ctx, dx && dw, "gcnx_dense_bwd: dx and dw are both required (use gcnx_gemm_dx / gcnx_gemm_dw for one of them)");
  const int64_t gy = gcnx_cdiv(n, BM);
  const bool fused = prec == GCNX_PREC_F32 && n > 0 && fi > 0 && fo > 0 && x && dh && w && fi % BN == 0 && al16(dx) &&
                     lddx % 4 == 0 && (!y_mask || (al16(y_mask) && ldy % 4 == 0)) && fo % 4 == 0 &&
                     (!db_prev || (al16(db_prev) && 2 * gy <= 4096)) && gy * (fi / BN) < (1 << 30);
  if (!fused) {   // bf16 paths, ragged widths, empty inputs: the two products as separate calls
    int rc = gcnx_gemm_dx(ctx, dh, lddh, w, dx, lddx, n, fi, fo, prec, 0, y_mask, ldy, db_prev);
    if (rc) return rc;
    return gcnx_gemm_dw(ctx, x, ldx, dh, lddh, dw, n, fi, fo, prec);
  }
  GCNX_REQUIRE(ctx, ldx >= fi && lddh >= fo && lddx >= fi && (!y_mask || ldy >= fi), "gcnx_dense_bwd: leading dimension too small");
  // dX tiles, then as many dW split-K slices as fill the rest of the resident-workgroup slots (4 per CU)
  const int n_dx = (int)gy * (fi / BN);
  int64_t kchunk = 0;
  const int nsplit = dense_bwd_split(ctx, n, fi, fo, &kchunk);
  // partial results: [dX column-sum partials (2 per row tile) | dW slabs] -- in the ctx workspace, or (deferred
  // reduction) in the caller's scratch, where they stay until gcnx_gemm_dw_sgd folds them
  const int64_t prow = db_prev ? 2 * gy : 0;
  const size_t part_floats = ((size_t)prow * fi + 63) & ~(size_t)63;
  const size_t slab_floats = nsplit > 1 ? (size_t)nsplit * fi * fo : 0;
  const bool defer = pending && scratch && al16(scratch) && (size_t)scratch_floats >= part_floats + slab_floats &&
                     (prow > 0 || nsplit > 1);
  float* base = scratch;
  if (!defer) {
    if (part_floats + slab_floats) {
      int rc = gcnx_ws_reserve(ctx, (part_floats + slab_floats) * sizeof(float));
      if (rc) return rc;
    }
    base = (float*)ctx->ws;
  }
  float* colpart = db_prev ? base : nullptr;
  float* slabs = base + part_floats;
  GemmJob jx{dh, lddh, w, (int64_t)fo, dx, lddx, n, fi, (int64_t)fo, (int64_t)fo + BK,
             Epilogue{nullptr, nullptr, y_mask, ldy, GCNX_ACT_NONE, 0, 1, colpart},
             al16(dh) && lddh % 4 == 0, al16(w) && fo % 4 == 0, fi / BN, (int)gy, 1};
  GemmJob jw{x, ldx, dh, lddh, nsplit > 1 ? slabs : dw, (int64_t)fo, (int64_t)fi, fo, n, kchunk,
             Epilogue{nullptr, nullptr, nullptr, 0, GCNX_ACT_NONE, 0, nsplit > 1 || al16(dw), nullptr},
             al16(x) && ldx % 4 == 0, al16(dh) && lddh % 4 == 0, gcnx_cdiv(fo, BN), gcnx_cdiv(fi, BM), nsplit};
  hipLaunchKernelGGL(gemm_f32_duo_kernel, dim3(n_dx + jw.gx * jw.gy * jw.gz), dim3(256), 0, ctx->stream, jx, jw, n_dx);
  GCNX_LAUNCH_OK(ctx);
  const int n_c = db_prev ? gcnx_cdiv(fi, 8) : 0;
  const int64_t total = (int64_t)fi * fo;
  const int n_s = nsplit > 1 ? gcnx_cdiv(total, 64) : 0;
  if (defer) {
    *pending = gcnx_pending_reduce{colpart, prow, db_prev ? fi : 0, db_prev, nsplit > 1 ? slabs : nullptr,
                                   nsplit > 1 ? total : 0, nsplit > 1 ? nsplit : 0, nsplit > 1 ? dw : nullptr};
    return GCNX_OK;
  }
  if (n_c + n_s > 0) {
    hipLaunchKernelGGL(reduce_duo_kernel, dim3(n_c + n_s), dim3(256), 0, ctx->stream, (const float*)colpart, prow, fi,
                       db_prev, n_c, (const float*)slabs, total, nsplit, dw, total);
    GCNX_LAUNCH_OK(ctx);
  }
  return GCNX_OK;
}

extern "C" {

int gcnx_gemm_dw(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, float* dw, int64_t n,
                 int32_t fi, int32_t fo, int prec) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMM (dW)");
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gemm_dw: negative size");
  GCNX_REQUIRE(ctx, prec >= GCNX_PREC_F32 && prec <= GCNX_PREC_BF16X3, "gcnx_gemm_dw: unknown precision %d", prec);
  if (fi == 0 || fo == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, dw != nullptr, "gcnx_gemm_dw: dw is NULL");
  if (n == 0) {
    GCNX_HIP(ctx, hipMemsetAsync(dw, 0, (size_t)fi * fo * 4, ctx->stream));
    return GCNX_OK;
  }
  GCNX_REQUIRE(ctx, x && dh, "gcnx_gemm_dw: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= fi && lddh >= fo, "gcnx_gemm_dw: leading dimension too small");
  // dW[i, o] = sum_n X[n, i] * dH[n, o]: A[i][k=n] = X[n*ldx + i], B[k=n][o] = dH[n*lddh + o].
  if (prec != GCNX_PREC_F32) {
    if (fi == 256 && fo == 256 && n >= 32 * 1024 && ctx->knob_gemm_stream) {   // tall: one slice per CU, whole product in registers
      const int max_slices = ctx->num_cus;
      int rc = gcnx_ws_reserve(ctx, (size_t)max_slices * 65536 * sizeof(float));
      if (rc) return rc;
      const int ns = gcnx_gemm_dw_stream(ctx, x, ldx, dh, lddh, (float*)ctx->ws, n, fi, fo, prec, max_slices);
      if (ns < 0) return gcnx_fail(ctx, GCNX_ERR_HIP, "gcnx_gemm_dw: streaming kernel launch failed");
      if (ns > 0) {
        if (al16(dw))
          hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(gcnx_cdiv(65536, 256)), dim3(256), 0, ctx->stream, (const float*)ctx->ws,
                             (int64_t)65536, ns, dw, (int64_t)65536);
        else
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(gcnx_cdiv(65536, 64)), dim3(256), 0, ctx->stream, (const float*)ctx->ws,
                           (int64_t)65536, ns, dw, (int64_t)65536);
        GCNX_LAUNCH_OK(ctx);
        return GCNX_OK;
      }
    }
    if (n < 32 * 1024 || fi != 256) {      // mid-size batches / wide inputs (GeneralGNN): panels of the streaming kernel
      const int pr = gcnx_gemm_dw_panels(ctx, x, ldx, dh, lddh, dw, n, fi, fo, prec);
      if (pr < 0) return gcnx_fail(ctx, GCNX_ERR_HIP, "gcnx_gemm_dw: streaming kernel (panels) launch failed");
      if (pr > 0) return GCNX_OK;
    }
    const int tiles_h = gcnx_cdiv(fi, HM) * gcnx_cdiv(fo, HN);
    int ns = (int)((4LL * ctx->num_cus + tiles_h - 1) / tiles_h);
    const int64_t ksteps_h = (n + HK - 1) / HK;
    if (ns > ksteps_h) ns = (int)ksteps_h;
    if (ns < 1) ns = 1;
    const int64_t kchunk_h = ((ksteps_h + ns - 1) / ns) * HK;
    ns = (int)((n + kchunk_h - 1) / kchunk_h);
    Epilogue eph{nullptr, nullptr, nullptr, 0, GCNX_ACT_NONE, 0, fo % 4 == 0 && (ns > 1 || al16(dw))};
    float* tgt = dw;
    if (ns > 1) {
      int rc = gcnx_ws_reserve(ctx, (size_t)ns * fi * fo * sizeof(float));
      if (rc) return rc;
      tgt = (float*)ctx->ws;
    }
    dim3 gridh(gcnx_cdiv(fo, HN), gcnx_cdiv(fi, HM), ns);
    const int vah = al16(x) && ldx % 4 == 0, vbh = al16(dh) && lddh % 4 == 0;
    if (prec == GCNX_PREC_BF16X3)
      hipLaunchKernelGGL((gemm_bf16_kernel<1, true>), gridh, dim3(256), 0, ctx->stream, x, ldx, dh, lddh,
                         (const __bf16*)nullptr, (const __bf16*)nullptr, 0, tgt, (int64_t)fo, (int64_t)fi, fo, n, kchunk_h,
                         eph, vah, vbh);
    else
      hipLaunchKernelGGL((gemm_bf16_kernel<1, false>), gridh, dim3(256), 0, ctx->stream, x, ldx, dh, lddh,
                         (const __bf16*)nullptr, (const __bf16*)nullptr, 0, tgt, (int64_t)fo, (int64_t)fi, fo, n, kchunk_h,
                         eph, vah, vbh);
    GCNX_LAUNCH_OK(ctx);
    if (ns > 1) {
      const int64_t total = (int64_t)fi * fo;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(gcnx_cdiv(total, 64)), dim3(256), 0, ctx->stream, (const float*)ctx->ws,
                         total, ns, dw, total);
      GCNX_LAUNCH_OK(ctx);
    }
    return GCNX_OK;
  }
  const int tiles = gcnx_cdiv(fi, BM) * gcnx_cdiv(fo, BN);
  int nsplit = (int)((4LL * ctx->num_cus + tiles - 1) / tiles);   // ~4 workgroups per CU
  const int64_t ksteps = (n + BK - 1) / BK;
  if (nsplit > ksteps / kMinSliceSteps) nsplit = (int)(ksteps / kMinSliceSteps);
  if (nsplit < 1) nsplit = 1;
  const int64_t kchunk = ((ksteps + nsplit - 1) / nsplit) * BK;
  nsplit = (int)((n + kchunk - 1) / kchunk);
  Epilogue ep{nullptr, nullptr, nullptr, 0, GCNX_ACT_NONE, 0, 0};
  const int va = al16(x) && ldx % 4 == 0, vb = al16(dh) && lddh % 4 == 0;
  float* target = dw;
  ep.vec_c = fo % 4 == 0 && (nsplit > 1 || al16(dw));   // partial slabs live in the 256-byte aligned workspace
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

// Runs a pending reduction on its own (the fallback paths of gcnx_gemm_dw_sgd).
static int flush_pending(gcnx_ctx* ctx, const gcnx_pending_reduce* pd) {
  if (!pd || (!pd->colpart && !pd->slabs)) return GCNX_OK;
  const int n_c = pd->colpart ? gcnx_cdiv(pd->cf, 8) : 0;
  const int n_s = pd->slabs ? gcnx_cdiv(pd->total, 64) : 0;
  hipLaunchKernelGGL(reduce_duo_kernel, dim3(n_c + n_s), dim3(256), 0, ctx->stream, pd->colpart, pd->crows, pd->cf, pd->cout,
                     n_c, pd->slabs, pd->total, pd->nsplit, pd->out, pd->total);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_gemm_dw_sgd(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, float* dw, int64_t n,
                     int32_t fi, int32_t fo, int prec, float* params, float* grads, int64_t n_params, float lr,
                     const gcnx_pending_reduce* pending) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMM (dW) + update");
  if (pending && !pending->colpart && !pending->slabs) pending = nullptr;
  if (pending) {
    GCNX_REQUIRE(ctx, !pending->colpart || (pending->cout >= grads && pending->cout + pending->cf <= grads + n_params &&
                                            pending->cf % 4 == 0 && al16(pending->cout)),
                 "gcnx_gemm_dw_sgd: the pending column sums must land (16-byte aligned) inside the flat gradient buffer");
    GCNX_REQUIRE(ctx, !pending->slabs || (pending->out >= grads && pending->out + pending->total <= grads + n_params),
                 "gcnx_gemm_dw_sgd: the pending split-K result must land inside the flat gradient buffer");
  }
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0 && n_params >= 0, "gcnx_gemm_dw_sgd: negative size");
  GCNX_REQUIRE(ctx, prec >= GCNX_PREC_F32 && prec <= GCNX_PREC_BF16X3, "gcnx_gemm_dw_sgd: unknown precision %d", prec);
  GCNX_REQUIRE(ctx, n_params == 0 || (params && grads), "gcnx_gemm_dw_sgd: NULL pointer");
  const int64_t total = (int64_t)fi * fo;
  GCNX_REQUIRE(ctx, total == 0 || (dw >= grads && dw + total <= grads + n_params),
               "gcnx_gemm_dw_sgd: dw must lie inside the flat gradient buffer");
  // split-K as gcnx_gemm_dw (the same slices, so the same dW bits)
  int nsplit = 1;
  int64_t kchunk = 0;
  if (prec == GCNX_PREC_F32 && n > 0 && total > 0) {
    const int tiles = gcnx_cdiv(fi, BM) * gcnx_cdiv(fo, BN);
    nsplit = (int)((4LL * ctx->num_cus + tiles - 1) / tiles);
    const int64_t ksteps = (n + BK - 1) / BK;
    if (nsplit > ksteps / kMinSliceSteps) nsplit = (int)(ksteps / kMinSliceSteps);
    if (nsplit < 1) nsplit = 1;
    kchunk = ((ksteps + nsplit - 1) / nsplit) * BK;
    nsplit = (int)((n + kchunk - 1) / kchunk);
  }
  if (nsplit <= 1 || fo % 4 != 0) {   // nothing to reduce (or bf16 / ragged): the separate calls
    int rc = flush_pending(ctx, pending);
    if (rc) return rc;
    rc = gcnx_gemm_dw(ctx, x, ldx, dh, lddh, dw, n, fi, fo, prec);
    if (rc) return rc;
    return gcnx_sgd(ctx, params, grads, n_params, lr);
  }
  GCNX_REQUIRE(ctx, x && dh, "gcnx_gemm_dw_sgd: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= fi && lddh >= fo, "gcnx_gemm_dw_sgd: leading dimension too small");
  int rc = gcnx_ws_reserve(ctx, (size_t)nsplit * total * sizeof(float));
  if (rc) return rc;
  Epilogue ep{nullptr, nullptr, nullptr, 0, GCNX_ACT_NONE, 0, 1};
  const int va = al16(x) && ldx % 4 == 0, vb = al16(dh) && lddh % 4 == 0;
  dim3 grid(gcnx_cdiv(fo, BN), gcnx_cdiv(fi, BM), nsplit);
  hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, ctx->stream, x, ldx, dh, lddh, (float*)ctx->ws,
                     (int64_t)fo, (int64_t)fi, fo, n, kchunk, ep, va, vb);
  GCNX_LAUNCH_OK(ctx);
  SgdPending pd{nullptr, 0, 0, 0, nullptr, 0, 0, 0, 0, 0};
  if (pending) {
    if (pending->colpart) { pd.cpart = pending->colpart; pd.crows = pending->crows; pd.cf = pending->cf;
                            pd.coff = pending->cout - grads; pd.n_pc = gcnx_cdiv(pending->cf, 8); }
    if (pending->slabs) { pd.slabs = pending->slabs; pd.total = pending->total; pd.nsplit = pending->nsplit;
                          pd.soff = pending->out - grads; pd.n_ps = gcnx_cdiv(pending->total, 64); }
  }
  const int n_s = gcnx_cdiv(total, 64), n_o = gcnx_cdiv(n_params, 256);
  hipLaunchKernelGGL(reduce_sgd_kernel, dim3(n_s + pd.n_pc + pd.n_ps + n_o), dim3(256), 0, ctx->stream, (const float*)ctx->ws,
                     total, nsplit, total, n_s, params, grads, (int64_t)(dw - grads), n_params, lr, pd, ctx->lr_dev);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int gcnx_gemm_dw2(gcnx_ctx* ctx, const float* xa, int64_t ldxa, const float* dha, int64_t lddha, float* dwa, int32_t fia,
                  int32_t foa, const float* xb, int64_t ldxb, const float* dhb, int64_t lddhb, float* dwb, int32_t fib,
                  int32_t fob, int64_t n, int prec, float* params, float* grads, int64_t n_params, float lr,
                  const gcnx_pending_reduce* pending, const gcnx_head_args* leaf) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight GEMMs (dW1 + dW2) + head leaves + update");
  if (leaf) {
    GCNX_REQUIRE(ctx, leaf->pool_sum && leaf->graph_ptr && leaf->w && leaf->y && leaf->probs && leaf->loss_acc && leaf->dw && leaf->dpooled &&
                          leaf->pooled && leaf->b > 0 && leaf->h > 0 && leaf->h % 4 == 0 && leaf->c >= 1 && leaf->c <= gcnx_head::kHeadMaxC &&
                          leaf->denom > 0.f && (leaf->pool_mode == GCNX_POOL_SUM || leaf->pool_mode == GCNX_POOL_AVG) &&
                          (leaf->cce_mode == GCNX_CCE_PROBS || leaf->cce_mode == GCNX_CCE_LOGITS) && (!leaf->db_relu || leaf->pool_cnt),
                 "gcnx_gemm_dw2: inconsistent head arguments");
    GCNX_REQUIRE(ctx, !params || (leaf->dw >= grads && leaf->dw + (int64_t)leaf->h * leaf->c <= grads + n_params),
                 "gcnx_gemm_dw2: the head's gradients must lie inside the flat gradient buffer");
  }
  if (pending && !pending->colpart && !pending->slabs) pending = nullptr;
  GCNX_REQUIRE(ctx, n >= 0 && fia >= 0 && foa >= 0 && fib >= 0 && fob >= 0 && n_params >= 0, "gcnx_gemm_dw2: negative size");
  GCNX_REQUIRE(ctx, prec >= GCNX_PREC_F32 && prec <= GCNX_PREC_BF16X3, "gcnx_gemm_dw2: unknown precision %d", prec);
  const int64_t ta = (int64_t)fia * foa, tb = (int64_t)fib * fob;
  GCNX_REQUIRE(ctx, (ta == 0 || dwa) && (tb == 0 || dwb), "gcnx_gemm_dw2: NULL gradient pointer");
  if (params) {
    GCNX_REQUIRE(ctx, grads != nullptr, "gcnx_gemm_dw2: params without grads");
    GCNX_REQUIRE(ctx, (ta == 0 || (dwa >= grads && dwa + ta <= grads + n_params)) &&
                          (tb == 0 || (dwb >= grads && dwb + tb <= grads + n_params)),
                 "gcnx_gemm_dw2: both gradients must lie inside the flat gradient buffer");
    GCNX_REQUIRE(ctx, !pending || !pending->colpart || (pending->cout >= grads && pending->cout + pending->cf <= grads + n_params &&
                                                        pending->cf % 4 == 0 && al16(pending->cout)),
                 "gcnx_gemm_dw2: the pending column sums must land (16-byte aligned) inside the flat gradient buffer");
  }
  int nsplit = 1;
  int64_t kchunk = 0;
  const bool shapes_ok = prec == GCNX_PREC_F32 && n > 0 && ta > 0 && tb > 0 && foa % 4 == 0 && fob % 4 == 0 &&
                         (!pending || !pending->slabs);
  if (shapes_ok) {
    const int tiles = gcnx_cdiv(fia, BM) * gcnx_cdiv(foa, BN) + gcnx_cdiv(fib, BM) * gcnx_cdiv(fob, BN);
    nsplit = (int)((4LL * ctx->num_cus + tiles - 1) / tiles);   // ~4 workgroups per CU over both products
    const int64_t ksteps = (n + BK - 1) / BK;
    if (nsplit > ksteps / kMinSliceSteps) nsplit = (int)(ksteps / kMinSliceSteps);
    if (nsplit < 1) nsplit = 1;
    kchunk = ((ksteps + nsplit - 1) / nsplit) * BK;
    nsplit = (int)((n + kchunk - 1) / kchunk);
  }
  if (!shapes_ok || nsplit <= 1) {   // bf16 precisions, empty or short inputs: the separate calls
    int rc = flush_pending(ctx, pending);
    if (rc) return rc;
    if (leaf) { rc = gcnx_head_from_parts(ctx, leaf); if (rc) return rc; }
    rc = gcnx_gemm_dw(ctx, xa, ldxa, dha, lddha, dwa, n, fia, foa, prec);
    if (rc) return rc;
    rc = gcnx_gemm_dw(ctx, xb, ldxb, dhb, lddhb, dwb, n, fib, fob, prec);
    if (rc) return rc;
    return params ? gcnx_sgd(ctx, params, grads, n_params, lr) : GCNX_OK;
  }
  GCNX_REQUIRE(ctx, xa && dha && xb && dhb, "gcnx_gemm_dw2: NULL pointer");
  GCNX_REQUIRE(ctx, ldxa >= fia && lddha >= foa && ldxb >= fib && lddhb >= fob, "gcnx_gemm_dw2: leading dimension too small");
  const size_t slab_a = ((size_t)nsplit * ta + 63) & ~(size_t)63;
  int rc = gcnx_ws_reserve(ctx, (slab_a + (size_t)nsplit * tb) * sizeof(float));
  if (rc) return rc;
  float* sa = (float*)ctx->ws;
  float* sb = sa + slab_a;
  const Epilogue ep{nullptr, nullptr, nullptr, 0, GCNX_ACT_NONE, 0, 1, nullptr};
  GemmJob ja{xa, ldxa, dha, lddha, sa, (int64_t)foa, (int64_t)fia, foa, n, kchunk, ep,
             al16(xa) && ldxa % 4 == 0, al16(dha) && lddha % 4 == 0, gcnx_cdiv(foa, BN), gcnx_cdiv(fia, BM), nsplit};
  GemmJob jb{xb, ldxb, dhb, lddhb, sb, (int64_t)fob, (int64_t)fib, fob, n, kchunk, ep,
             al16(xb) && ldxb % 4 == 0, al16(dhb) && lddhb % 4 == 0, gcnx_cdiv(fob, BN), gcnx_cdiv(fib, BM), nsplit};
  const int n_a = ja.gx * ja.gy * ja.gz, n_b = jb.gx * jb.gy * jb.gz;
  const bool want_db = leaf && leaf->db_relu;
  const bool merged = leaf && leaf->c == 2 && gcnx_head::head_lds_floats(leaf->h, leaf->c, want_db) <= (size_t)kDw2HeadLds &&
                      leaf->b <= gcnx_head::kHeadRows;
  if (leaf && !merged) { rc = gcnx_head_from_parts(ctx, leaf); if (rc) return rc; }   // (many graphs / wide operands: its own launch)
  if (merged) {
    HeadLeaf hl{leaf->w, leaf->bias, leaf->y, leaf->b, leaf->h, leaf->c, leaf->denom, leaf->probs, leaf->loss_acc, leaf->dw, leaf->db,
                leaf->dpooled, (int64_t)leaf->h, (int64_t)leaf->h, nullptr, ctx->flag + 3,
                gcnx_head::PoolParts{leaf->pool_sum, leaf->graph_ptr, leaf->pooled, 1, leaf->pool_mode == GCNX_POOL_AVG ? 1 : 0,
                                     want_db ? leaf->pool_cnt : nullptr, want_db ? leaf->db_relu : nullptr},
                leaf->cce_mode == GCNX_CCE_LOGITS ? 1 : 0};
    hipLaunchKernelGGL(gemm_f32_dw2_head_kernel, dim3(1 + n_a + n_b), dim3(256), 0, ctx->stream, ja, jb, n_a, hl, 1);
  } else {
    hipLaunchKernelGGL(gemm_f32_dw2_kernel, dim3(n_a + n_b), dim3(256), 0, ctx->stream, ja, jb, n_a);
  }
  GCNX_LAUNCH_OK(ctx);
  // one reduction launch either way: with params == NULL reduce_sgd_kernel only folds (offsets relative to `base`)
  float* base = params ? grads : std::min(dwa, dwb);
  if (!params && pending && pending->colpart) base = std::min(base, pending->cout);
  SgdPending pd{nullptr, 0, 0, 0, nullptr, 0, 0, 0, 0, 0};
  if (pending && pending->colpart) { pd.cpart = pending->colpart; pd.crows = pending->crows; pd.cf = pending->cf;
                                     pd.coff = pending->cout - base; pd.n_pc = gcnx_cdiv(pending->cf, 8); }
  pd.slabs = sb; pd.total = tb; pd.nsplit = nsplit; pd.soff = dwb - base; pd.n_ps = gcnx_cdiv(tb, 64);
  const int n_s = gcnx_cdiv(ta, 64), n_o = params ? gcnx_cdiv(n_params, 256) : 0;
  hipLaunchKernelGGL(reduce_sgd_kernel, dim3(n_s + pd.n_pc + pd.n_ps + n_o), dim3(256), 0, ctx->stream, (const float*)sa,
                     ta, nsplit, ta, n_s, params, base, (int64_t)(dwa - base), n_params, lr, pd, ctx->lr_dev);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // extern "C"
