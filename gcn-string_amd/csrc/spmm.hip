// K2/K3: CSR neighbour aggregation  out[t,:] = act(sum_e vals[e] * h[colidx[e],:] + bias).
//
// Replaces tf.sparse.sparse_dense_matmul (GCNConv.call) and gather + unsorted_segment_sum
// (GeneralConv.propagate) reached from model(inputs) at src/scripts/gcn.py:334/351.  The
// [nnz,F] message tensor TensorFlow materialises never exists here.
//
// HBM-bound: algorithmic bytes per launch = 4(N+1) + 4 nnz (+4 nnz weighted) + 2*4*N*F.
//
// Two kernels.
//
// Kernel "block" (used when the caller passes the diagonal blocks of the disjoint batch, i.e.
// graph_ptr): the gather never leaves the CU.  A workgroup owns (graph g, 32-column slab): it
// copies the slab of g's feature rows H[rows of g, slab] into LDS once (coalesced 128-B row
// segments), then every output row of g is a sum of LDS rows.  HBM/L2 see each feature element
// once (compulsory traffic); the ~10x re-read of the gather is served by LDS.  Each group of
// LPR = 8 lanes owns one output row (8 rows per wave in flight); a row's CSR entries are fetched
// 8 at a time into the group's registers and broadcast inside the group with __shfl, so there
// is no cross-lane reduction at all.  80 KiB of LDS per workgroup -> 2 workgroups per CU: one
// streams its tile in while the other computes.  Graphs too large for a 32-column tile are
// done in 2 (4) passes of 16 (8) columns; beyond that the same loop gathers from global memory.
//
// Kernel "rows" (no block structure known / odd widths): one 256-thread workgroup owns a contiguous chunk of rows.  The chunk's CSR
// segment (column indices, values, row pointers) is contiguous in memory and is staged into LDS
// with coalesced loads, so the per-row work has a single dependent HBM/L2 latency (the feature
// gather) instead of two.  Inside a wave, LPR = F/4 lanes cover one feature row with 16-byte
// loads (fully coalesced: 64 lanes x 16 B = 1 KiB for F = 256); when F < 256 the wave's
// 64/LPR lane groups take different neighbours of the same row and are combined with
// __shfl_xor at the end.  Chunk ids are remapped so each XCD (private 4 MiB L2) walks one
// contiguous range of rows: in a disjoint (block-diagonal) batch the rows a chunk gathers lie
// in the same graph, hence in the same L2.
#include <cstdlib>

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


// ----------------------------------------------------------------------------------------------
// Block kernel
// ----------------------------------------------------------------------------------------------
constexpr int kBlkLdsBytes = 80 * 1024;   // 2 workgroups per CU (160 KiB LDS)
constexpr int kSlab = 32;                 // columns per work item

// Broadcast of lane J's value to the lanes of its row group as a DPP quad_perm move (VALU; no
// LDS traffic).  LPR >= 4: every lane of the quad reads lane J.  LPR == 2: two rows share a
// quad, lanes {0,1} read lane J and lanes {2,3} read lane 2+J.
template <int LPR, int J>
__device__ __forceinline__ int quad_bcast(int v) {
  constexpr int ctrl = LPR >= 4 ? (J | (J << 2) | (J << 4) | (J << 6)) : (J | (J << 2) | ((2 + J) << 4) | ((2 + J) << 6));
  return __builtin_amdgcn_mov_dpp(v, ctrl, 0xF, 0xF, true);
}

struct __attribute__((aligned(4))) I4u { int x, y, z, w; };     // 16-byte load from a 4-byte-aligned address
struct __attribute__((aligned(4))) F4u { float x, y, z, w; };
struct __attribute__((aligned(4))) I2u { int x, y; };

// A row's next (up to) 16 CSR entries in ONE vector-memory instruction per array: lane `slot`
// of the row's quad fetches entries [e0 + 4*slot, +4) as a dwordx4.  Slots past the row end are
// redirected to the zero row `pad` (cols) / 0 (vals).  `last4` = nnz - 4 guards the array end.
template <bool WEIGHTED>
__device__ __forceinline__ void fetch_entries(const int32_t* __restrict__ colidx, const float* __restrict__ vals,
                                              int e0, int slot, int b, int row0, int pad, int last4, int (&mc)[4],
                                              float (&mv)[4]) {
  const int e = e0 + 4 * slot;
  int c0 = pad, c1 = pad, c2 = pad, c3 = pad;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (e < b) {
    if (e <= last4) {
      const I4u c = *reinterpret_cast<const I4u*>(colidx + e);
      c0 = c.x; c1 = c.y; c2 = c.z; c3 = c.w;
      if (WEIGHTED) { const F4u v = *reinterpret_cast<const F4u*>(vals + e); v0 = v.x; v1 = v.y; v2 = v.z; v3 = v.w; }
    } else {  // the last three entries of the whole matrix
      c0 = colidx[e]; if (e + 1 < b) c1 = colidx[e + 1]; if (e + 2 < b) c2 = colidx[e + 2];
      if (WEIGHTED) { v0 = vals[e]; if (e + 1 < b) v1 = vals[e + 1]; if (e + 2 < b) v2 = vals[e + 2]; }
    }
  }
  mc[0] = e + 0 < b ? c0 - row0 : pad;  mv[0] = e + 0 < b ? v0 : 0.f;
  mc[1] = e + 1 < b ? c1 - row0 : pad;  mv[1] = e + 1 < b ? v1 : 0.f;
  mc[2] = e + 2 < b ? c2 - row0 : pad;  mv[2] = e + 2 < b ? v2 : 0.f;
  mc[3] = e + 3 < b ? c3 - row0 : pad;  mv[3] = e + 3 < b ? v3 : 0.f;
}

// One pass over FT = 4*LPR columns [c0, c0+FT) of block rows [row0, row0+ng).
//
// The tile holds ng feature rows plus one all-zero row at index ng: CSR slots past the end of a
// row point there (value 0), so the inner loop is branch-free per group of 4 entries and adds
// exact zeros for the padding (no 0*inf hazards).  Each LPR-lane group owns one output row.  The
// per-(row, slab) index traffic is what limits this kernel (8 slabs re-read the CSR), so it is
// kept to 4 vector-memory instructions per 8 rows: one dwordx2 for the row pointers, one
// dwordx4 each for 16 column indices and 16 values (fetch_entries), one store.  Entries reach
// the lanes of the group through DPP quad broadcasts.  Index traffic is software-pipelined two
// row groups ahead: while group i is reduced, the entries of group i+1 and the row pointers of
// group i+2 are in flight.
template <int THREADS, int LPR, bool WEIGHTED, bool FROM_LDS>
__device__ __forceinline__ void block_pass(float* __restrict__ tile, const int32_t* __restrict__ rowptr,
                                           const int32_t* __restrict__ colidx, const float* __restrict__ vals,
                                           const float* __restrict__ h, int64_t ldh, const float* __restrict__ bias,
                                           float* __restrict__ out, int64_t ldo, int row0, int ng, int c0, int act,
                                           int last4, int ablate) {
  constexpr int FT = LPR * 4;
  constexpr int RPW = 64 / LPR;           // rows per wave iteration (one row per LPR-lane group)
  constexpr int QL = LPR >= 4 ? 4 : 2;    // lanes of a row that fetch entries (4 entries each)
  constexpr int EPB = 4 * QL;             // CSR entries fetched per row per batch
  constexpr int STRIDE = (THREADS / 64) * RPW;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int grp = lane / LPR, sub = lane % LPR;
  const int slot = sub % QL;
  const int pad = FROM_LDS ? ng : 0;      // local row used by padding slots

  // ---- stage the feature tile: every load is in flight before the first LDS write
  if (FROM_LDS && !(ablate & 1)) {
    constexpr int U = 5 * 1024 / THREADS;  // ceil((rows that fit) * LPR / THREADS)
    const int total = ng * LPR;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(tid + u * THREADS, total - 1);   // clamped: branch-free
      v[u] = *reinterpret_cast<const float4*>(h + (int64_t)(row0 + i / LPR) * ldh + c0 + (i % LPR) * 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * THREADS;
      if (i < total) *reinterpret_cast<float4*>(tile + i * 4) = v[u];   // tile[r][q*4..]: r*FT + q*4 == i*4
    }
    if (tid < LPR) *reinterpret_cast<float4*>(tile + ng * FT + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- pipeline prologue: row pointers of this wave's first two row groups, entries of the first
  int r = wave * RPW + grp;
  int a = 0, b = 0, a1 = 0, b1 = 0;
  if (r < ng) { const I2u p = *reinterpret_cast<const I2u*>(rowptr + row0 + r); a = p.x; b = p.y; }
  if (r + STRIDE < ng) { const I2u p = *reinterpret_cast<const I2u*>(rowptr + row0 + r + STRIDE); a1 = p.x; b1 = p.y; }
  int mc[4];
  float mv[4];
  fetch_entries<WEIGHTED>(colidx, vals, a, slot, b, row0, pad, last4, mc, mv);
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) bv = *reinterpret_cast<const float4*>(bias + c0 + sub * 4);
  const float* hcol = h + c0 + sub * 4;
  const float* trow = tile + sub * 4;
  if (FROM_LDS) __syncthreads();

  for (int rb = wave * RPW; rb < ng; rb += STRIDE) {
    r = rb + grp;
    // prefetch: entries of the next row group, row pointers of the one after
    int mc_n[4];
    float mv_n[4];
    fetch_entries<WEIGHTED>(colidx, vals, a1, slot, b1, row0, pad, last4, mc_n, mv_n);
    int a2 = 0, b2 = 0;
    if (r + 2 * STRIDE < ng) { const I2u p = *reinterpret_cast<const I2u*>(rowptr + row0 + r + 2 * STRIDE); a2 = p.x; b2 = p.y; }

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int base = a;
    while (!(ablate & 2)) {   // one batch of EPB entries per trip; trip count uniform inside a lane group
#define GCNX_STEP4(J)                                                                                          \
      if ((J) == 0 || __builtin_amdgcn_ballot_w64(base + 4 * (J) < b) != 0) {   /* wave-uniform skip */            \
        int c[4];                                                                                              \
        float w[4];                                                                                            \
        float4 hv[4];                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
          c[i] = quad_bcast<LPR, (J)>(mc[i]);                                                                    \
          if (WEIGHTED) w[i] = __int_as_float(quad_bcast<LPR, (J)>(__float_as_int(mv[i])));                      \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                        \
          if (FROM_LDS) {                                                                                      \
            hv[i] = *reinterpret_cast<const float4*>(trow + c[i] * FT);                                        \
          } else {                                                                                             \
            hv[i] = *reinterpret_cast<const float4*>(hcol + (int64_t)(row0 + c[i]) * ldh);                     \
            if (base + 4 * (J) + i >= b) hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                \
          }                                                                                                    \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) acc = WEIGHTED ? f4_fma(w[i], hv[i], acc) : f4_add(acc, hv[i]); \
      }
      GCNX_STEP4(0)
      GCNX_STEP4(1)
      if (QL == 4) {
        GCNX_STEP4(2 % QL)
        GCNX_STEP4(3 % QL)
      }
#undef GCNX_STEP4
      base += EPB;
      if (base >= b) break;
      fetch_entries<WEIGHTED>(colidx, vals, base, slot, b, row0, pad, last4, mc, mv);   // rows longer than EPB
    }
    if (r < ng) {
      acc = f4_add(acc, bv);
      if (act == GCNX_ACT_RELU) {
        acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
      }
      *reinterpret_cast<float4*>(out + (int64_t)(row0 + r) * ldo + c0 + sub * 4) = acc;
    }
    a = a1; b = b1; a1 = a2; b1 = b2;
#pragma unroll
    for (int q = 0; q < 4; ++q) { mc[q] = mc_n[q]; mv[q] = mv_n[q]; }
  }
  if (FROM_LDS) __syncthreads();  // the next pass overwrites the tile
}

template <int THREADS, bool WEIGHTED>
__global__ __launch_bounds__(THREADS, THREADS / 128) void spmm_block_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const float* __restrict__ vals,
    const float* __restrict__ h, int64_t ldh, const float* __restrict__ bias, float* __restrict__ out, int64_t ldo,
    const int32_t* __restrict__ block_ptr, int nslabs, int act, int nwork, int cap32, int n, int ablate) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int w = gcnx_xcd_remap(blockIdx.x, nwork);  // slabs of one graph stay on one XCD, adjacent in time
  const int g = w / nslabs, s = w % nslabs;
  const int row0 = block_ptr[g];
  const int ng = block_ptr[g + 1] - row0;
  if (ng <= 0) return;
  const int c0 = s * kSlab;
  const int last4 = rowptr[n] - 4;   // last entry index from which a 4-entry vector load stays in bounds
  // cap32 = rows that fit (plus the zero row) with a 32-column tile
  if (ng <= cap32) {
    block_pass<THREADS, 8, WEIGHTED, true>(tile, rowptr, colidx, vals, h, ldh, bias, out, ldo, row0, ng, c0, act, last4, ablate);
  } else if (ng <= 2 * cap32) {
#pragma unroll 1
    for (int p = 0; p < 2; ++p)
      block_pass<THREADS, 4, WEIGHTED, true>(tile, rowptr, colidx, vals, h, ldh, bias, out, ldo, row0, ng, c0 + 16 * p, act, last4, ablate);
  } else if (ng <= 4 * cap32) {
#pragma unroll 1
    for (int p = 0; p < 4; ++p)
      block_pass<THREADS, 2, WEIGHTED, true>(tile, rowptr, colidx, vals, h, ldh, bias, out, ldo, row0, ng, c0 + 8 * p, act, last4, ablate);
  } else {
    block_pass<THREADS, 8, WEIGHTED, false>(tile, rowptr, colidx, vals, h, ldh, bias, out, ldo, row0, ng, c0, act, last4, ablate);
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
  const bool vec = (f % 4 == 0) && (ldh % 4 == 0) && (ldo % 4 == 0) && aligned16(h) && aligned16(out) &&
                   (!bias || aligned16(bias));
  if (!vec) {
    hipLaunchKernelGGL(spmm_scalar_kernel, dim3(gcnx_cdiv(n, 4)), dim3(256), 0, ctx->stream, rowptr, colidx, vals,
                       h, ldh, bias, out, ldo, n, f, act);
    GCNX_LAUNCH_OK(ctx);
    return GCNX_OK;
  }
  const char* force = getenv("GCNX_SPMM_KERNEL");  // tuning knob: "rows" disables the block kernel
  if (block_ptr && nblocks > 0 && f % kSlab == 0 && !(force && force[0] == 'r')) {
    int lds_bytes = kBlkLdsBytes, threads = 512;
    if (const char* e = getenv("GCNX_SPMM_LDS_KB")) {   // tuning knobs
      const int kb = atoi(e);
      if (kb >= 8 && kb * 1024 <= kBlkLdsBytes) lds_bytes = kb * 1024;
    }
    int ablate = 0;  // timing-only ablation (wrong results): 1 = skip tile load, 2 = skip reduction
    if (const char* e = getenv("GCNX_SPMM_ABLATE")) ablate = atoi(e);
    if (const char* e = getenv("GCNX_SPMM_THREADS")) threads = atoi(e) == 1024 ? 1024 : 512;
    const int cap32 = lds_bytes / (kSlab * 4) - 1;
    const int nslabs = f / kSlab;
    const long long nwork = (long long)nblocks * nslabs;
    GCNX_REQUIRE(ctx, nwork < 2147483647LL, "gcnx_spmm_csr: too many (block, slab) work items");
#define GCNX_LAUNCH_BLOCK(T, W)                                                                                   \
  do {                                                                                                            \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) {                                                                                              \
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_block_kernel<T, W>),                  \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kBlkLdsBytes));               \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    hipLaunchKernelGGL((spmm_block_kernel<T, W>), dim3((unsigned)nwork), dim3(T), lds_bytes, ctx->stream, rowptr, \
                       colidx, vals, h, ldh, bias, out, ldo, block_ptr, nslabs, act, (int)nwork, cap32, n, ablate); \
  } while (0)
    if (threads == 1024) {
      if (vals) GCNX_LAUNCH_BLOCK(1024, true); else GCNX_LAUNCH_BLOCK(1024, false);
    } else {
      if (vals) GCNX_LAUNCH_BLOCK(512, true); else GCNX_LAUNCH_BLOCK(512, false);
    }
#undef GCNX_LAUNCH_BLOCK
    GCNX_LAUNCH_OK(ctx);
    return GCNX_OK;
  }
  int lanes = f / 4;
  // Debug/tuning knob (not part of the ABI contract): GCNX_SPMM_SLAB = column-slab width in
  // floats (64/128/256) forces the lanes-per-row split; results are identical.
  if (const char* e = getenv("GCNX_SPMM_SLAB")) {
    const int slab = atoi(e);
    if (slab >= 16 && slab / 4 < lanes) lanes = slab / 4;
  }
  if (lanes > 32) launch_rows<64>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  else if (lanes > 16) launch_rows<32>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  else if (lanes > 8) launch_rows<16>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  else if (lanes > 4) launch_rows<8>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  else launch_rows<4>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}
