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
// Tile kernel (used when the caller passes the diagonal blocks of the disjoint batch, i.e. a plan over graph_ptr,
// and the batch has enough (graph, slab) units to fill the chip): the gather never leaves the CU.  A workgroup
// owns (graph g, 32-column slab): it copies the slab of g's feature rows H[rows of g, slab] into LDS once by
// LDS-DMA (128-byte row pieces), then every output row of g is a sum of LDS rows.  HBM sees each feature element
// once (compulsory traffic); the ~10x re-read of the gather is served by LDS.  A quad of lanes owns one output
// row; a row's CSR entries reach the quad's lanes by DPP broadcasts, so there is no cross-lane reduction.
// Graphs of up to 604 rows: 79.5 KiB of LDS per 512-thread workgroup, two workgroups per CU; up to 1236 rows: one
// 1024-thread workgroup with all 160 KiB; taller graphs: plan-listed row chunks on the rows kernel.
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
#include <algorithm>
#include <cstdlib>
#include <new>
#include <queue>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"

namespace {

constexpr int kRowsPerChunk = 32;    // rows per workgroup (large inputs; plan chunks)
constexpr int kRowsPerChunkSmall = 8;   // small inputs: more, shorter workgroups (launch-latency regime)
constexpr int kSlab = 32;             // widest column slab of the tile kernel
constexpr int kStageCap = 2048;      // CSR entries staged in LDS per chunk (overflow -> global)
constexpr int kLongRowSmall = 256;    // rows with more entries are split over the workgroup's four waves: 8-row chunks
constexpr int kLongRowLarge = 32;     // (latency regime: a chunk below the limit skips the hub phase outright) / 32-row chunks

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
// Packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two lanes' worth of fp32 per instruction at full rate).  The row gather
// is not pure memory time: at config 2 each launch issues ~150 k wave-level float4 loads, and with one VALU
// instruction per component the accumulation alone was >1 us of the kernel (the folded backward's mask: 3 us).
__device__ __forceinline__ float4 f4_fma(float v, float4 h, float4 a) {
  const f32x2 w = {v, v};
  const f32x2 lo = __builtin_elementwise_fma(w, f32x2{h.x, h.y}, f32x2{a.x, a.y});
  const f32x2 hi = __builtin_elementwise_fma(w, f32x2{h.z, h.w}, f32x2{a.z, a.w});
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
// [a > 0] for a >= 0 (a saved ReLU output): clamp01(a * 2^127 * 2^127) -- 0 stays 0, every positive value down to the
// smallest denormal (2^-149 -> 2^105) saturates to 1.  Two packed multiplies per two components (the second with the
// VOP3P clamp bit) instead of a compare + select per component.
__device__ __forceinline__ float4 f4_step(float4 a) {
  const f32x2 big = {0x1p127f, 0x1p127f};
  f32x2 lo = {a.x, a.y}, hi = {a.z, a.w}, t0, t1;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(lo), "v"(big));
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(hi), "v"(big));
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(lo) : "v"(t0), "v"(big));
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(hi) : "v"(t1), "v"(big));
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
__device__ __forceinline__ float4 buf4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const f32x4v r = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  return make_float4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  const f32x2 lo = f32x2{a.x, a.y} + f32x2{b.x, b.y}, hi = f32x2{a.z, a.w} + f32x2{b.z, b.w};
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}

// Pool' + ReLU' folded into the backward aggregation (FOLD): with dZ[j] = dPooled[graph(j)] * [y[j] > 0] and a
// block-diagonal operator, (A^T dZ)[i] = dPooled[graph(i)] * sum_j A^T[i,j] [y[j] > 0] -- the gather reads the saved
// layer output y instead of a materialised dZ, and the row's dPooled vector multiplies the finished sum.
struct FoldArgs {
  const int32_t* gp;   // graph_ptr [b + 1]
  const float* dp;     // dPooled [b, f]
  int64_t lddp;
  int32_t b;
  int32_t avg;         // 1: GlobalAvgPool (row scale 1 / n_g)
};

// Tuning builds only (make TUNING=1): per-wave lifetimes of the row-gather kernel, for the load-imbalance figure SURVEY
// 8(d) asks for at config 5 (max / mean wave time).  g_wave_stamps: [blocks * 4] durations in s_memrealtime ticks (100 MHz),
// set through gcnx_tuning_wave_stamps; NULL = off.
#ifdef GCNX_TUNING
__device__ unsigned long long* g_wave_stamps = nullptr;
#define GCNX_WSTAMP_BEGIN const unsigned long long wst0_ = __builtin_amdgcn_s_memrealtime();
#define GCNX_WSTAMP_END                                                                              \
  if (g_wave_stamps && (threadIdx.x & 63) == 0)                                                      \
    g_wave_stamps[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime() - wst0_;
#else
#define GCNX_WSTAMP_BEGIN
#define GCNX_WSTAMP_END
#endif

// OUT16 forms (r3): the result row is stored as bf16, round to nearest even -- for outputs that only bf16-operand weight
// GEMMs read, which would round an fp32 copy in exactly this way (gcnx_spmm_csr_bf16out).  `out` then points at uint16 rows
// and ldo counts elements.
__device__ __forceinline__ void store4_bf16(float* out, int64_t elem, const float4& v) {
  typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
  const b16x4 o = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  *reinterpret_cast<b16x4*>(reinterpret_cast<unsigned short*>(out) + elem) = o;
}

// Result rows of the row gather.  NT (the HUBS instances: power-law batches, whose graphs do not fit an XCD's L2): stored with
// the non-temporal policy -- the rows are not read again by this launch and should not push the gathered rows out of L2
// (r4, config 5, same box: 730 -> 709 us; no effect at config 3, and measured SLOWER on the tile kernels' stores: 520 -> 560 us,
// as was the nt policy on their LDS-DMA reads: no change -- scripts/r04/nt_ab.sh, profiles/r04/cache_policy_ab.txt).
template <bool NT>
__device__ __forceinline__ void rows_store4(float* p, const float4& v) {
  if constexpr (NT) __builtin_nontemporal_store(f32x4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4v*>(p));
  else *reinterpret_cast<float4*>(p) = v;
}

template <int LPR, bool WEIGHTED, int RPC, bool FOLD = false, bool HUBS = false, bool OUT16 = false>
__global__ __launch_bounds__(256, (LPR == 64 ? 7 : 8)) void spmm_rows_kernel(   // 8 waves per SIMD = at most 64 VGPRs (latency regime)
   const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ colidx,
                                                        const float* __restrict__ vals,
                                                        const float* __restrict__ h, int64_t ldh,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int64_t ldo, int32_t n, int32_t f, int32_t col0, int act,
                                                        int nchunks, const int2* __restrict__ chunk_list, FoldArgs fo, int hub_deg_rt) {
  // (a template flag: with the test in the row loop of every instance the 64-VGPR latency-regime instance -- config 2 -- ran
  // 16.3 us instead of 11.9)
  const int hub_deg = HUBS ? hub_deg_rt : 0;
  // hub_deg > 0 (r3): rows with more entries are NOT this kernel's -- spmm_hub_seg_kernel / spmm_hub_combine_kernel walk them
  // as 256-entry segments on workgroups of their own (a plan lists them); such a row is skipped here, store included.
  constexpr int G = 64 / LPR;  // neighbour groups per wave
  __shared__ int32_t s_g[FOLD ? RPC : 1];   // FOLD: graph of each row of the chunk, and its pool scale
  __shared__ float s_sc[FOLD ? RPC : 1];
  __shared__ float4 s_d[FOLD ? 2 : 1][FOLD ? LPR : 1];   // FOLD: dPooled rows of the first row's graph and the next
  __shared__ int32_t s_col[kStageCap];
  __shared__ float s_val[WEIGHTED ? kStageCap : 1];
  __shared__ int32_t s_rp[RPC + 1];
  __shared__ float4 s_long[4][LPR];     // per-wave partial sums of a long row

  GCNX_WSTAMP_BEGIN
  const int chunk = gcnx_xcd_remap(blockIdx.x, nchunks);
  int r0 = chunk * RPC;
  int r1 = min(n, r0 + RPC);
  if (chunk_list) { r0 = chunk_list[chunk].x; r1 = chunk_list[chunk].y; }   // row ranges picked by a plan
  const int tid = threadIdx.x;
  // FOLD, wave 3 (the wave with the least staging to do): the graph of the chunk's first row by a 64-ary search --
  // one load level per factor 64 of b, where a binary search's log2 b dependent loads would outlast the staging --
  // with its first probe issued here, next to the row-pointer loads, and its second level (the dPooled rows, the
  // next graph boundary) next to the entry loads: the fold adds no load level in front of the barrier.
  const bool fw = FOLD && tid >= 192;
  const int fl = tid - 192;
  int f_stride = FOLD ? (fo.b + 63) >> 6 : 0, f_probe = 0;
  if (fw) f_probe = fl * f_stride < fo.b ? fo.gp[fl * f_stride] : INT_MAX;
  if (tid <= r1 - r0) s_rp[tid] = rowptr[r0 + tid];
  const int e0 = rowptr[r0];
  const int e1 = rowptr[r1];
  int f_base = 0, f_nxt = 0;
  float4 f_d0 = make_float4(0.f, 0.f, 0.f, 0.f), f_d1 = f_d0;
  if (fw) {
    int span = fo.b;                                  // gp[f_base] <= r0, answer in [f_base, f_base + span)
    while (true) {
      const int k = max(__popcll(__builtin_amdgcn_ballot_w64(f_probe <= r0)) - 1, 0);   // gp is non-decreasing: a lane prefix
                                                                                        // (gp[0] > r0, a malformed graph_ptr, stays in range)
      span = min(f_stride, span - k * f_stride);
      f_base += k * f_stride;
      if (span <= 1) break;
      f_stride = (span + 63) >> 6;
      f_probe = fl * f_stride < span ? fo.gp[f_base + fl * f_stride] : INT_MAX;
    }
    // A chunk rarely spans more than two graphs: their dPooled rows (this launch's columns) wait in LDS, so the
    // epilogue's multiply costs an LDS read instead of a dependent trip to L2 at the end of every row.
    if (fl < LPR && col0 + fl * 4 < f) {
      f_d0 = *reinterpret_cast<const float4*>(fo.dp + (int64_t)f_base * fo.lddp + col0 + fl * 4);
      if (f_base + 1 < fo.b) f_d1 = *reinterpret_cast<const float4*>(fo.dp + (int64_t)(f_base + 1) * fo.lddp + col0 + fl * 4);
    }
    f_nxt = fo.gp[f_base + 1];
  }
  const int staged = min(e1 - e0, kStageCap);
  for (int i = tid; i < staged; i += 256) {
    s_col[i] = colidx[e0 + i];
    if (WEIGHTED) s_val[i] = vals[e0 + i];
  }
  if (fw) {
    if (fl < LPR) { s_d[0][fl] = f_d0; s_d[1][fl] = f_d1; }
    if (fl < r1 - r0) {
      const int r = r0 + fl;
      int gq = f_base;
      while (f_nxt <= r && gq + 1 < fo.b) { ++gq; f_nxt = fo.gp[gq + 1]; }   // gp[b] = n > r ends it; the guard keeps a
                                                                             // malformed graph_ptr from walking off the array
      s_g[fl] = gq;
      s_sc[fl] = fo.avg ? 1.0f / (float)(f_nxt - fo.gp[gq]) : 1.0f;
    }
  }
  __syncthreads();

  const int lane = tid & 63, wave = tid >> 6;
  const int g = lane / LPR, sub = lane % LPR;
  const int c = col0 + sub * 4;          // first of this lane's 4 columns
  const bool col_ok = c < f;             // f % 4 == 0 is guaranteed by the dispatcher
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias && col_ok) bv = *reinterpret_cast<const float4*>(bias + c);

  // Two rows per trip: their gathers are independent, so both sets of loads are in flight
  // before either is accumulated (the per-row dependent latency is what bounds small inputs).
  auto entry = [&](int e, int& cidx, float& v) {
    v = 1.0f;
    if (e < kStageCap) {
      cidx = s_col[e];
      if (WEIGHTED) v = s_val[e];
    } else {
      cidx = colidx[e0 + e];
      if (WEIGHTED) v = vals[e0 + e];
    }
  };
  auto fold_row = [&](int r) {                 // FOLD: the row's (scaled) dPooled vector
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (FOLD && col_ok && r < r1) {
      const int gi = s_g[r - r0] - s_g[0];
      d = s_d[min(gi, 1)][sub];            // (two statements: a select between an LDS and a global address trips hipcc)
      if (gi >= 2) d = *reinterpret_cast<const float4*>(fo.dp + (int64_t)s_g[r - r0] * fo.lddp + c);
      const float sc = s_sc[r - r0];
      d.x *= sc; d.y *= sc; d.z *= sc; d.w *= sc;
    }
    return d;
  };
  auto finish = [&](float4 acc, int r) {
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      acc.x += __shfl_xor(acc.x, off);
      acc.y += __shfl_xor(acc.y, off);
      acc.z += __shfl_xor(acc.z, off);
      acc.w += __shfl_xor(acc.w, off);
    }
    if (g == 0 && col_ok && r < r1) {
      if (FOLD) {
        const float4 d = fold_row(r);
        acc.x *= d.x; acc.y *= d.y; acc.z *= d.z; acc.w *= d.w;
      } else {
        acc = f4_add(acc, bv);
        if (act == GCNX_ACT_RELU) {
          acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        }
      }
      if constexpr (OUT16) store4_bf16(out, (int64_t)r * ldo + c, acc);
      else rows_store4<HUBS>(out + (int64_t)r * ldo + c, acc);
    }
  };
  constexpr int kLongRow = RPC <= 8 ? kLongRowSmall : kLongRowLarge;
  // h as a raw buffer resource (range-checked loads for the row tails); usable while byte offsets fit 32 bits
  const bool use_buf = (uint64_t)n * (uint64_t)ldh * 4u < 0xFFFFFF00ull;
  const __amdgpu_buffer_rsrc_t hbuf =
      __builtin_amdgcn_make_buffer_rsrc((void*)h, (short)0, use_buf ? (int)((uint64_t)n * (uint64_t)ldh * 4u) : 0, 0x00020000);
  const unsigned ld32 = (unsigned)ldh;
  bool saw_long = false;
  // Two instances of the row loop: chunks whose entries all fit the LDS stage (every chunk but those with hub rows)
  // take the one WITHOUT the global-index fallback.  With the fallback in the loop, hipcc's wait-count pass must
  // assume a pending index load on every trip and puts s_waitcnt vmcnt(0) after each gather load -- the trip's four
  // loads ran one after the other instead of together.
  auto rows_pass = [&](auto all_staged) {
  constexpr bool kAllStaged = decltype(all_staged)::value;
  auto entry_of = [&](int e, int& cidx, float& v) {
    if (kAllStaged) { cidx = s_col[e]; v = WEIGHTED ? s_val[e] : 1.0f; }
    else entry(e, cidx, v);
  };
  for (int r = r0 + wave; r < r1; r += 8) {
    const int rB = r + 4;                                          // second row of this trip (may be past the chunk)
    int aA = s_rp[r - r0] - e0, bA = s_rp[r - r0 + 1] - e0;        // chunk-relative entry ranges
    int aB = rB < r1 ? s_rp[rB - r0] - e0 : 0, bB = rB < r1 ? s_rp[rB - r0 + 1] - e0 : 0;
    const bool hubA = hub_deg > 0 && bA - aA > hub_deg, hubB = hub_deg > 0 && bB - aB > hub_deg;   // another kernel's rows
    const bool longA = bA - aA > kLongRow, longB = bB - aB > kLongRow;   // long rows: done by all four waves below
    saw_long |= (longA & !hubA) | (longB & !hubB);
    if (longA) bA = aA;
    if (longB) bB = aB;
    float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA;
    if (col_ok) {
      int eA = aA + g, eB = aB + g;
      if (kAllStaged) {
        // Full trips first: 2G entries of both rows, no conditions -- four gather loads issued back to back and one
        // wait.  (The conditional trip below is what remains for the row tails: each of its loads sits in a branch
        // of its own, and hipcc ends every such branch with a wait or a register copy.)
        const int nfull = min(bA - aA, bB - aB) / (2 * G);        // uniform per wave
        for (int t = 0; t < nfull; ++t) {
          const int cA0 = s_col[eA], cA1 = s_col[eA + G], cB0 = s_col[eB], cB1 = s_col[eB + G];
          // (32-bit offsets from a scalar base: a third of the address arithmetic of 64-bit pointers, no pointer VGPRs)
          const float4 hA0 = buf4(hbuf, ((unsigned)cA0 * ld32 + (unsigned)c) * 4u);
          const float4 hA1 = buf4(hbuf, ((unsigned)cA1 * ld32 + (unsigned)c) * 4u);
          const float4 hB0 = buf4(hbuf, ((unsigned)cB0 * ld32 + (unsigned)c) * 4u);
          const float4 hB1 = buf4(hbuf, ((unsigned)cB1 * ld32 + (unsigned)c) * 4u);
          if (WEIGHTED) {
            const float wA0 = s_val[eA], wA1 = s_val[eA + G], wB0 = s_val[eB], wB1 = s_val[eB + G];
            accA = f4_fma(wA0, FOLD ? f4_step(hA0) : hA0, accA);
            accB = f4_fma(wB0, FOLD ? f4_step(hB0) : hB0, accB);
            accA = f4_fma(wA1, FOLD ? f4_step(hA1) : hA1, accA);
            accB = f4_fma(wB1, FOLD ? f4_step(hB1) : hB1, accB);
          } else {
            accA = f4_add(accA, FOLD ? f4_step(hA0) : hA0);
            accB = f4_add(accB, FOLD ? f4_step(hB0) : hB0);
            accA = f4_add(accA, FOLD ? f4_step(hA1) : hA1);
            accB = f4_add(accB, FOLD ? f4_step(hB1) : hB1);
          }
          eA += 2 * G;
          eB += 2 * G;
        }
        // The row tails, branch-free: h is read through a raw buffer resource, a missing entry gets an
        // out-of-range offset and comes back as zeros -- no fetch, no branch, nothing to discard -- so the tail's
        // loads (2 rows x 2 neighbour groups) are issued together like a full trip's.
        const int rem = max(bA - aA, bB - aB) - nfull * 2 * G;     // uniform per wave
        const int last = max(staged - 1, 0);
        if (use_buf)
        for (int t = 0; t < rem; t += 2 * G) {
          const bool okA0 = eA < bA, okA1 = eA + G < bA, okB0 = eB < bB, okB1 = eB + G < bB;
          const unsigned oA0 = okA0 ? ((unsigned)s_col[min(eA, last)] * ld32 + (unsigned)c) * 4u : 0xFFFFFFF0u;
          const unsigned oA1 = okA1 ? ((unsigned)s_col[min(eA + G, last)] * ld32 + (unsigned)c) * 4u : 0xFFFFFFF0u;
          const unsigned oB0 = okB0 ? ((unsigned)s_col[min(eB, last)] * ld32 + (unsigned)c) * 4u : 0xFFFFFFF0u;
          const unsigned oB1 = okB1 ? ((unsigned)s_col[min(eB + G, last)] * ld32 + (unsigned)c) * 4u : 0xFFFFFFF0u;
          const f32x4v rA0 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(hbuf, oA0, 0, 0));
          const f32x4v rA1 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(hbuf, oA1, 0, 0));
          const f32x4v rB0 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(hbuf, oB0, 0, 0));
          const f32x4v rB1 = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(hbuf, oB1, 0, 0));
          float wA0 = 1.f, wA1 = 1.f, wB0 = 1.f, wB1 = 1.f;
          if (WEIGHTED) {
            wA0 = s_val[min(eA, last)]; wA1 = s_val[min(eA + G, last)];
            wB0 = s_val[min(eB, last)]; wB1 = s_val[min(eB + G, last)];
          }
          float4 hA0 = make_float4(rA0.x, rA0.y, rA0.z, rA0.w), hA1 = make_float4(rA1.x, rA1.y, rA1.z, rA1.w);
          float4 hB0 = make_float4(rB0.x, rB0.y, rB0.z, rB0.w), hB1 = make_float4(rB1.x, rB1.y, rB1.z, rB1.w);
          if (FOLD) { hA0 = f4_step(hA0); hA1 = f4_step(hA1); hB0 = f4_step(hB0); hB1 = f4_step(hB1); }
          if (WEIGHTED) {
            accA = f4_fma(wA0, hA0, accA); accB = f4_fma(wB0, hB0, accB);
            accA = f4_fma(wA1, hA1, accA); accB = f4_fma(wB1, hB1, accB);
          } else {
            accA = f4_add(accA, hA0); accB = f4_add(accB, hB0);
            accA = f4_add(accA, hA1); accB = f4_add(accB, hB1);
          }
          eA += 2 * G;
          eB += 2 * G;
        }
      }
      if (!kAllStaged || !use_buf)
      while (eA < bA || eB < bB) {                                 // (the instance with the global-index fallback)
        float4 hA[2], hB[2];
        float vA[2], vB[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          int cA = 0, cB = 0;
          vA[u] = vB[u] = 0.f;
          hA[u] = hB[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (eA + u * G < bA) { entry_of(eA + u * G, cA, vA[u]); hA[u] = *reinterpret_cast<const float4*>(h + (int64_t)cA * ldh + c); }
          if (eB + u * G < bB) { entry_of(eB + u * G, cB, vB[u]); hB[u] = *reinterpret_cast<const float4*>(h + (int64_t)cB * ldh + c); }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          // (FOLD: the mask is taken here, next to the accumulation -- applied where the value is loaded it made
          // hipcc wait after every load instead of after the trip's four)
          if (eA + u * G < bA) { const float4 t = FOLD ? f4_step(hA[u]) : hA[u]; accA = WEIGHTED ? f4_fma(vA[u], t, accA) : f4_add(accA, t); }
          if (eB + u * G < bB) { const float4 t = FOLD ? f4_step(hB[u]) : hB[u]; accB = WEIGHTED ? f4_fma(vB[u], t, accB) : f4_add(accB, t); }
        }
        eA += 2 * G;
        eB += 2 * G;
      }
    }
    finish(accA, longA ? r1 : r);               // r1: outside the chunk = no store (the merge shuffles still run)
    finish(accB, longB ? r1 : rB);
  }
  };
  if (e1 - e0 <= kStageCap && use_buf) rows_pass(std::true_type{}); else rows_pass(std::false_type{});   // uniform per workgroup
  // Hub rows (power-law batches: one row of a chunk can hold thousands of entries).  Walked by a single wave such a
  // row alone set the kernel's duration (config 5: a 4096-entry row = 2 ms); here every wave of the workgroup takes
  // a quarter of its entries and the four partial sums are combined in wave order (deterministic).
  if (e1 - e0 <= kLongRow) { GCNX_WSTAMP_END return; }           // no row of this chunk can be that long (uniform)
  if (!__syncthreads_or(saw_long)) { GCNX_WSTAMP_END return; }   // ... and none was (one barrier; the scan below costs more)
  const int nnz_all = rowptr[n];
  for (int r = r0; r < r1; ++r) {                // uniform over the workgroup
    const int a = s_rp[r - r0] - e0, b = s_rp[r - r0 + 1] - e0;
    if (b - a <= kLongRow || (hub_deg > 0 && b - a > hub_deg)) continue;
    const int per = (b - a + 3) / 4;
    const int wa = a + wave * per, wb = min(b, wa + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RPC > 8 && col_ok && use_buf) {          // (not in the 8-row instance: its registers are spoken for)
      // Branch-free through range-checked buffer loads (indices straight from the CSR arrays, then the feature
      // rows): a slot past this wave's share gets out-of-range offsets and contributes zeros.  HU entries per
      // lane group and trip, so HU index loads and then HU gathers are in flight -- as conditional loads they
      // ran one by one, and a 4096-entry row is 128+ trips.
      const __amdgpu_buffer_rsrc_t cbuf = __builtin_amdgcn_make_buffer_rsrc((void*)colidx, (short)0, nnz_all * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t vbuf =
          __builtin_amdgcn_make_buffer_rsrc((void*)(WEIGHTED ? (const void*)vals : (const void*)colidx), (short)0, nnz_all * 4, 0x00020000);
      constexpr int HU = LPR == 64 ? 8 : 4;          // entries per lane group and trip (register budget)
      for (int e = wa + g; e < wb; e += HU * G) {
        int ci[HU];
        float wv[HU];
#pragma unroll
        for (int u = 0; u < HU; ++u) {
          const unsigned off = e + u * G < wb ? (unsigned)(e0 + e + u * G) * 4u : 0xFFFFFFF0u;
          ci[u] = __builtin_amdgcn_raw_buffer_load_b32(cbuf, off, 0, 0);
          wv[u] = WEIGHTED ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(vbuf, off, 0, 0)) : 1.0f;
        }
        f32x4v hv[HU];
#pragma unroll
        for (int u = 0; u < HU; ++u) {
          const unsigned off = e + u * G < wb ? ((unsigned)ci[u] * ld32 + (unsigned)c) * 4u : 0xFFFFFFF0u;
          hv[u] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(hbuf, off, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < HU; ++u) {
          float4 t = make_float4(hv[u].x, hv[u].y, hv[u].z, hv[u].w);
          if (FOLD) t = f4_step(t);
          acc = WEIGHTED ? f4_fma(wv[u], t, acc) : f4_add(acc, t);
        }
      }
    } else if (col_ok) {
      for (int e = wa + g; e < wb; e += 4 * G) {      // four neighbours in flight per lane
        float4 hv[4];
        float vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int cc = 0;
          vv[u] = 0.f;
          hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e + u * G < wb) { entry(e + u * G, cc, vv[u]); hv[u] = *reinterpret_cast<const float4*>(h + (int64_t)cc * ldh + c); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (e + u * G < wb) acc = WEIGHTED ? f4_fma(vv[u], FOLD ? f4_step(hv[u]) : hv[u], acc) : f4_add(acc, FOLD ? f4_step(hv[u]) : hv[u]);
      }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
      acc.x += __shfl_xor(acc.x, off);
      acc.y += __shfl_xor(acc.y, off);
      acc.z += __shfl_xor(acc.z, off);
      acc.w += __shfl_xor(acc.w, off);
    }
    __syncthreads();                             // s_long free (previous long row consumed)
    if (g == 0) s_long[wave][sub] = acc;
    __syncthreads();
    if (wave == 0 && g == 0 && col_ok) {
      float4 t = f4_add(f4_add(s_long[0][sub], s_long[1][sub]), f4_add(s_long[2][sub], s_long[3][sub]));
      if (FOLD) {
        const float4 d = fold_row(r);
        t.x *= d.x; t.y *= d.y; t.z *= d.z; t.w *= d.w;
      } else {
        t = f4_add(t, bv);
        if (act == GCNX_ACT_RELU) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
      }
      if constexpr (OUT16) store4_bf16(out, (int64_t)r * ldo + c, t);
      else *reinterpret_cast<float4*>(out + (int64_t)r * ldo + c) = t;
    }
  }
  GCNX_WSTAMP_END
}

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
// SCALE: the stored value is (col - row0) * SCALE -- the byte offset of the tile row, computed once per
// fetched entry instead of once per lane that consumes the broadcast.
__device__ __forceinline__ int tile_off(int row, int scale) { return row * scale; }

// The two index arrays as raw buffer resources (base, size in bytes): buffer loads are range-checked per dword by
// the memory pipeline -- a lane whose slot is past the row end is given an out-of-range offset and reads zeros
// WITHOUT a fetch and without a branch, and a dwordx4 that straddles the end of the array returns its valid dwords.
struct EntryBufs {
  __amdgpu_buffer_rsrc_t col, val;
};
__device__ __forceinline__ EntryBufs entry_bufs(const int32_t* colidx, const float* vals, int nnz) {
  EntryBufs r;
  r.col = __builtin_amdgcn_make_buffer_rsrc((void*)colidx, (short)0, nnz * 4, 0x00020000);
  r.val = __builtin_amdgcn_make_buffer_rsrc((void*)(vals ? (const void*)vals : (const void*)colidx), (short)0, nnz * 4, 0x00020000);
  return r;
}

template <bool WEIGHTED, int SCALE = 1>
__device__ __forceinline__ void fetch_entries(const EntryBufs& eb, int e0, int slot, int b, int row0, int pad,
                                              int (&mc)[4], float (&mv)[4]) {
  // Branch-free: with each load inside `if (slot has entries)` hipcc closed every one of a unit's 2 x NI index
  // loads with its own s_waitcnt vmcnt(0) -- ten dependent round trips per unit where one is needed.  (Clamping the
  // address instead measured 10 % slower: the slots past the row end then fetch too; out-of-range buffer lanes don't.)
  const int e = e0 + 4 * slot;
  const unsigned off = e < b ? (unsigned)e * 4u : 0xFFFFFFF0u;
  const i32x4 c = __builtin_amdgcn_raw_buffer_load_b128(eb.col, off, 0, 0);
  const int c0 = c.x, c1 = c.y, c2 = c.z, c3 = c.w;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  if (WEIGHTED) {
    const i32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(eb.val, off, 0, 0);
    v0 = __int_as_float(vb.x); v1 = __int_as_float(vb.y); v2 = __int_as_float(vb.z); v3 = __int_as_float(vb.w);
  }
  mc[0] = tile_off(e + 0 < b ? c0 - row0 : pad, SCALE);  mv[0] = e + 0 < b ? v0 : 0.f;
  mc[1] = tile_off(e + 1 < b ? c1 - row0 : pad, SCALE);  mv[1] = e + 1 < b ? v1 : 0.f;
  mc[2] = tile_off(e + 2 < b ? c2 - row0 : pad, SCALE);  mv[2] = e + 2 < b ? v2 : 0.f;
  mc[3] = tile_off(e + 3 < b ? c3 - row0 : pad, SCALE);  mv[3] = e + 3 < b ? v3 : 0.f;
}

// ----------------------------------------------------------------------------------------------
// Building blocks of the tile kernel: the index burst (row pointers + the first 16 entries of every row, in
// registers) and the LDS-DMA of a tile.
// ----------------------------------------------------------------------------------------------
// Index burst of an item, part 1: row pointers of every row this quad owns (NI wave iterations),
// branch-free so that all loads are in flight at once.
template <int NI, int SPAN, int LPR>
__device__ __forceinline__ void tile_load_rowptr(const int32_t* __restrict__ rowptr, int row0, int ng, int (&a)[NI],
                                                 int (&b)[NI], int rlo = 0) {
  // rows [rlo, ng) of the block; ng doubles as the (exclusive) end of the row range
  const int rbase = rlo + (threadIdx.x >> 6) * (64 / LPR) + ((threadIdx.x & 63) / LPR);
#pragma unroll
  for (int t = 0; t < NI; ++t) {
    const int r = rbase + t * SPAN;
    const I2u p = *reinterpret_cast<const I2u*>(rowptr + row0 + min(r, ng - 1));   // clamped: always in bounds
    a[t] = p.x;
    b[t] = r < ng ? p.y : p.x;   // rows past the range end become empty
  }
}

// Part 2: the first 16 entries of each of those rows.
template <int NI, bool WEIGHTED, int SCALE>
__device__ __forceinline__ void tile_load_entries(const EntryBufs& eb, int row0, int pad, const int (&a)[NI],
                                                  const int (&b)[NI], int (&mc)[NI][4], float (&mv)[NI][4]) {
#pragma unroll
  for (int t = 0; t < NI; ++t)
    fetch_entries<WEIGHTED, SCALE>(eb, a[t], threadIdx.x & 3, b[t], row0, pad, mc[t], mv[t]);
}


// Tile DMA with a uniform 64-bit base (SGPR pair) and a 32-bit per-lane element offset: one register per piece
// instead of a 64-bit pointer (hoisted 64-bit row addresses were spilled, and each reload put a vmcnt(0) in
// front of its piece, serialising the tile stream).  Needs ng * ldh < 2^32.
template <int PPR, int THREADS, int PIECES>
__device__ __forceinline__ void tile_dma32(float* buf, const float* __restrict__ gbase, unsigned ld32, int ng) {
  const int tid = threadIdx.x;
  const int total = ng * PPR;
#pragma unroll
  for (int u = 0; u < (PIECES + THREADS - 1) / THREADS; ++u) {
    const int i = tid + u * THREADS;
    if (i < total) {
      const unsigned off = (unsigned)(i / PPR) * ld32 + (unsigned)(i % PPR) * 4u;
      float* dst = buf + (u * THREADS + (tid & ~63)) * 4;   // wave-uniform base; the DMA adds lane*16 bytes
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gbase + off),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  }
}

// The same with the LDS-DMA as inline asm.  For the builtin hipcc (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every
// later LDS read -- it cannot tell two LDS buffers apart -- which serialises a double-buffered loop; the asm form is
// invisible to its wait-count pass, so the CALLER waits (s_waitcnt vmcnt(0) before the barrier that precedes the reads).
// lds_byte: LDS byte address of the tile's first row.
template <int PPR, int THREADS, int PIECES>
__device__ __forceinline__ void tile_dma32_asm(unsigned lds_byte, const float* __restrict__ gbase, unsigned ld32, int ng) {
  const int tid = threadIdx.x;
  const int total = ng * PPR;
#pragma unroll
  for (int u = 0; u < (PIECES + THREADS - 1) / THREADS; ++u) {
    const int i = tid + u * THREADS;
    if (i < total) {
      const unsigned off = (unsigned)(i / PPR) * ld32 + (unsigned)(i % PPR) * 4u;
      const float* src = gbase + off;
      const unsigned base = __builtin_amdgcn_readfirstlane(lds_byte + (unsigned)(u * THREADS + (tid & ~63)) * 16u);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(base) : "memory", "m0");
    }
  }
}

// ----------------------------------------------------------------------------------------------
// Tile kernel (default with a plan): independent workgroups, no pipeline inside a workgroup.  Per work unit
// (graph, sg column slabs) one index burst -- row pointers and the first 16 entries of every row, kept in registers
// and shared by the unit's slabs -- then per slab
//     barrier | LDS-DMA of the tile -> LDS | barrier | reduce + store.
// Nothing in a workgroup is consumed while its own DMA is in flight, so the in-order vmcnt problem of a
// double-buffered design (the first version of this kernel) does not arise: a row's entries past 16 are simply
// fetched on demand.
// Units are dealt statically in snake order over the size-sorted graph list; workgroup ids are folded so that the
// slabs of one graph (consecutive units) run on ONE XCD and share its L2 for the index arrays.
// Measured on config 3 (same box): tier 1 0.53-0.57 ns/row against 0.71-0.73 for that first version; copying the
// tile in and zeros out (no index burst, no reduction) alone runs at 5.1 TB/s.  An L2 prefetch of the next tile
// during the reduction (one dword per row piece) and a chunk-order swizzle against LDS bank conflicts were both
// measured: -6 % and 0 %.
// ----------------------------------------------------------------------------------------------
// Two shapes of the same kernel: THREADS = 512 with 79.5 KiB (two workgroups per CU; graphs up to kDuoCap32 rows)
// and THREADS = 1024 with all 160 KiB (one workgroup per CU; graphs up to kSoloCap32 rows).
constexpr int kDuoLdsFloats = 20352;             // 81 408 B: two workgroups fit one CU's 160 KiB
constexpr int kSoloLdsFloats = 40960;            // 160 KiB
constexpr int kDuoCap32 = 632;                   // rows of a 32-column tile (+ zero row + bias)
constexpr int kSoloCap32 = 1276;
template <int THREADS, int FT> struct DuoShape {
  static constexpr int LDSF = THREADS == 512 ? kDuoLdsFloats : kSoloLdsFloats;
  static constexpr int CAP = (FT == 32 ? (THREADS == 512 ? kDuoCap32 : kSoloCap32) : ((LDSF / FT) - 2) & ~3);
};

// Row order of the tile kernels (r3).  A graph's output rows are dealt to the quads in DEGREE order (longest first, ties
// in row order; gcnx_spmm_plan binds it to a rowptr): the 16 rows of a wave then have about the same number of entries,
// so the 4-entry steps a wave skips are the steps NO row of it needs (at mean degree 10 a wave of rows in natural order
// nearly always holds one row of 13+ entries and walks all 16 slots: 37 % of the LDS reads fetched the zero row; and 29 %
// of the waves held a row of more than 16 entries and stalled on its on-demand fetch in the middle of the reduction).
// Record of position p of graph g (p = 0 .. ng - 1): the row's first CSR entry and (local row | degree << 16).
struct __attribute__((aligned(8))) RowRec { int a; unsigned w; };

// LDS byte address of a tile row: base + (16-bit half of `packed`) * row_bytes in one instruction.
template <int HI>
__device__ __forceinline__ unsigned tile_addr(unsigned packed, unsigned row_bytes, unsigned base) {
  unsigned a;
  if (HI) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(a) : "v"(packed), "v"(row_bytes), "v"(base));
  else asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(a) : "v"(packed), "v"(row_bytes), "v"(base));
  return a;
}

// One 4-entry step of a row group: the entries [4 J, 4 J + 4) (+ FIRST) of every row live in lane J of the row's quad as two
// packed 16-bit tile-row pairs (PC) and four weights (MV); a DPP quad broadcast hands them to the quad's lanes.  The
// step runs if any row of the wave has an entry in it (DEG: the lane's row degree).
#define GCNX_DSTEP4(J, PC, MV, DEG, FIRST)                                                                     \
  if (((J) == 0 && (FIRST) == 0) || __builtin_amdgcn_ballot_w64((DEG) > (FIRST) + 4 * (J)) != 0) {              \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                            \
      const unsigned pk = (unsigned)quad_bcast<4, (J)>((int)PC[k]);                                            \
      _Pragma("unroll") for (int hh = 0; hh < 2; ++hh) {                                                       \
        const unsigned a0 = hh ? tile_addr<1>(pk, RB, tb0) : tile_addr<0>(pk, RB, tb0);                        \
        f32x2 w2 = f32x2{1.f, 1.f};                                                                            \
        if (WEIGHTED) { const float w = __int_as_float(quad_bcast<4, (J)>(__float_as_int(MV[2 * k + hh]))); w2 = f32x2{w, w}; } \
        _Pragma("unroll") for (int j = 0; j < CPL; ++j) {                                                      \
          const f32x4v hq = *reinterpret_cast<const __attribute__((address_space(3))) f32x4v*>(a0 + (j ? tbd : 0u)); \
          if (WEIGHTED) {                                                                                      \
            acc[j][0] = __builtin_elementwise_fma(w2, f32x2{hq.x, hq.y}, acc[j][0]);                           \
            acc[j][1] = __builtin_elementwise_fma(w2, f32x2{hq.z, hq.w}, acc[j][1]);                           \
          } else {                                                                                             \
            acc[j][0] += f32x2{hq.x, hq.y};                                                                    \
            acc[j][1] += f32x2{hq.z, hq.w};                                                                    \
          }                                                                                                    \
        }                                                                                                      \
      }                                                                                                        \
    }                                                                                                          \
  }

// FOLD (gcnx_spmm_csr_pool_bwd with a plan): h is the saved ReLU output Y of the pooled layer and the kernel computes
// A^T (pool'(dPooled) * [Y > 0]) -- the landed tile is turned into its 0 / 1 mask in place (one pass over the tile:
// every element is gathered ~degree times, so masking it once is that much cheaper than masking every use), and the
// graph's dPooled row (x 1 / n_g for the average pool) multiplies the finished sums where the bias is added otherwise.
// MODE 2 / 3, the bit image of a ReLU output (r2): the forward launch of the pooled layer also writes [out > 0] as one
// 32-bit word per (row, 32-column slab), slab-major (`bits[slab * n + row]`: a unit's words are contiguous), and the folded
// backward expands those words into the 0 / 1 tile instead of DMA-ing the fp32 slab and masking it -- 4 bytes per row
// and slab where the fp32 source is 128 (config 3: 1 GB of reads per step become 32 MB).
// Sum over the 16 lanes l, l + 4, ..., l + 60 of a wave (the lanes of the 16 quads that hold the same columns), VALU only.
__device__ __forceinline__ float wave_quads_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));   // row_ror:8
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__device__ __forceinline__ unsigned wave_quads_sum_u(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);
  const auto a = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = a[0] + a[1];
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return b[0] + b[1];
}

struct DuoFold {
  const int32_t* gids;     // graph index of every entry of `graphs`
  const float* dp;         // dPooled [b, f]
  int64_t lddp;
  int avg;
  uint32_t* bits;          // MODE 2, 4: written; MODE 3: read
  float* ppart = nullptr;  // MODE 4: [tile graphs][waves][f] column sums of the rows each wave computed, and ...
  float* cpart = nullptr;  // ... the number of positive entries among them (same layout)
};
// kDuoBitsPool (r3): the pooled layer's forward WITHOUT its output -- what the step needs of Y = relu(A H + b) on a tile
// graph is the bit image [Y > 0] (folded backward aggregation) and the graph's column sums and positive counts (global
// pool, db of the layer): a unit owns the whole graph, so both leave the epilogue and Y is neither written (1 GB at
// config 3) nor read back by a pool launch (1 GB).  Per lane: sums of its 8 columns over the row groups, counts as two
// registers of four byte counters; one 16-quad shuffle tree per slab; lanes 0..3 of every wave write the wave's partial
// row; a small launch adds the 16 waves' rows in order (pool_parts_reduce_kernel).
enum { kDuoPlain = 0, kDuoFold = 1, kDuoBitsOut = 2, kDuoFoldBits = 3, kDuoBitsPool = 4 };

template <int THREADS, int FT, int LPR, bool WEIGHTED, int MODE = kDuoPlain, bool OUT16 = false>
__global__ __launch_bounds__(THREADS, 4) void spmm_duo_kernel(
    const int32_t* __restrict__ rowptr, const RowRec* __restrict__ rowrec, const int32_t* __restrict__ colidx,
    const float* __restrict__ vals, const float* __restrict__ h, int64_t ldh, const float* __restrict__ bias,
    float* __restrict__ out, int64_t ldo, const int2* __restrict__ graphs /* (row0, ng), largest first */,
    int upg /* units per graph */, int sg /* column slabs per unit */, int act, int nunits, int n, int dbg_rt, DuoFold fo) {
  // Phase-ablation bits (results are wrong by design when set).  The HOST passes 0 unless this is a tuning build
  // (launch_duo), so a release library cannot be talked into wrong results; the tests stay in the kernel as uniform
  // branches on purpose: with them folded away at compile time hipcc's allocation of this 128-VGPR kernel spilled 6.
  const int dbg = dbg_rt;
  constexpr bool FOLD = MODE == kDuoFold || MODE == kDuoFoldBits;
  static_assert(FT == 32 && LPR == 4, "one quad per row, two 64-byte halves per tile row (the xor-64 chunk addressing)");
  constexpr int CPL = FT / (4 * LPR);           // float4 chunks per lane
  constexpr int RPW = 64 / LPR;                 // rows per wave
  constexpr int SPAN = (THREADS / 64) * RPW;
  constexpr int CAP = DuoShape<THREADS, FT>::CAP;
  constexpr int NI = (CAP + SPAN - 1) / SPAN;   // row groups per unit: their first 16 entries live in registers
  constexpr unsigned RB = FT * 4;               // bytes per tile row
  static_assert((CAP + 2) * FT <= DuoShape<THREADS, FT>::LDSF, "tile + zero row + bias must fit");
  static_assert(NI <= 5, "index registers");
  static_assert(CAP < 65536, "16-bit tile rows");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lbias = lds + (CAP + 1) * FT;          // this step's bias slice
  const int tid = threadIdx.x, lane = tid & 63;
  const int sub = lane % LPR, slot = sub & 3;
  const int rbase = (tid >> 6) * RPW + lane / LPR;
  // ds_read_b128 is served in four 16-lane groups {quads 0,3,5,6}, {1,2,4,7} (+8 for the upper half-wave) over
  // 64 banks: a 128-byte tile row covers half a bank row, a 64-byte chunk a quarter.  With one quad per row and
  // two chunks per lane (CPL = 2), reading chunk j in the same order everywhere leaves the 4 rows of a group only
  // the 2 quarters (row parity, j): expected 2.75 LDS cycles per read.  Quads 2..5 therefore read their chunks in
  // the opposite order (chunk j ^ 1 first): each group then has two quads on either chunk, 1.75 expected.
  const int csw = (CPL == 2) ? ((((lane >> 2) + 2) >> 2) & 1) : 0;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned tbl = lds0 + (sub + LPR * csw) * 16;   // LDS address of this lane's FIRST chunk of tile row 0 (buffer 0)
  const unsigned tbd = csw ? (unsigned)-64 : 64u;       // ... and the distance to its second chunk
  // workgroup id folded per XCD (ids are dealt round-robin to the 8 XCDs): consecutive virtual ids share an L2
  const int G = gridDim.x;
  const int w = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
  auto unit_of = [&](int round) { return round * G + ((round & 1) ? (G - 1 - w) : w); };
  int u = unit_of(0);
  if (u >= nunits) return;                      // uniform per workgroup
  if (tid < FT) lds[CAP * FT + tid] = 0.f;      // the all-zero row padding entries point at
  const EntryBufs ebufs = entry_bufs(colidx, vals, rowptr[n]);

  // Double buffering (r3; the 1024-thread shape, DMA-fed modes): a graph of up to kDblCap rows leaves room for TWO tiles, so
  // the tile of the next slab (or of the next unit's first slab) streams in while this one is reduced -- the DMA wait, a
  // third of a slab step, disappears for such graphs (half of config 3's rows).  The DMA is inline asm throughout (see
  // tile_dma32_asm); the wait for it is the explicit vmcnt(0) in front of the barrier that precedes the reduction.
  constexpr int kDblCap = 624;
  constexpr unsigned kDblBuf = (kDblCap + 1) * RB;          // tile + its zero row
  constexpr bool kCanDbl = THREADS == 1024 && (MODE == kDuoPlain || MODE == kDuoBitsOut || MODE == kDuoBitsPool);
  static_assert(!kCanDbl || 2 * kDblBuf <= (unsigned)(CAP + 1) * RB, "two small tiles must stay below the bias slice");
  int cur = 0;                                   // buffer the current tile is in (double-buffered units)
  bool prefetched = false, zero_rows_ok = false; // this step's tile already requested; the small tiles' zero rows written

  int2 g = graphs[u / upg];
  for (int round = 0;; ++round) {
    const int cbase = (u % upg) * sg * FT;
    const int un = unit_of(round + 1);
    const bool has_next = un < nunits;
    const int2 gn = has_next ? graphs[un / upg] : g;
    const bool dbl = kCanDbl && g.y <= kDblCap && !(dbg_rt & 16);   // (bit 16 of the tuning word: single tiles only, for A/B runs)
    const int pad = dbl ? kDblCap : CAP;           // tile row padding entries point at (all zeros)
    if (!dbl) { cur = 0; zero_rows_ok = false; }
    // index burst of the unit: (local row, degree) and the first 16 entries of every row this quad owns -- 32 for the
    // first row group, where the degree order puts the long rows -- once for all sg slabs (they share rows and
    // entries).  In flight together with the first tile.
    unsigned rw[NI];                            // local row | degree << 16 (0 entries for positions past the graph)
    unsigned pc[NI][2], pc2[2];                 // tile rows of the entries, 16-bit pairs: entry 2k | entry 2k + 1 << 16
    float mv[NI][4], mv2[4];
    {
      int a[NI];
#pragma unroll
      for (int t = 0; t < NI; ++t) {
        const int pos = rbase + t * SPAN;
        const RowRec rr = rowrec[g.x + min(pos, g.y - 1)];    // clamped: always in bounds
        a[t] = rr.a;
        rw[t] = pos < g.y ? rr.w : (rr.w & 0xFFFFu);          // positions past the graph: no entries, no store
      }
#pragma unroll
      for (int t = 0; t < NI; ++t) {
        int c4[4];
        if (!(dbg & 8)) fetch_entries<WEIGHTED, 1>(ebufs, a[t], slot, a[t] + (int)(rw[t] >> 16), g.x, pad, c4, mv[t]);
        else { c4[0] = c4[1] = c4[2] = c4[3] = pad; mv[t][0] = mv[t][1] = mv[t][2] = mv[t][3] = 0.f; }
        pc[t][0] = (unsigned)c4[0] | ((unsigned)c4[1] << 16);
        pc[t][1] = (unsigned)c4[2] | ((unsigned)c4[3] << 16);
      }
      {
        int c4[4];
        if (!(dbg & 8)) fetch_entries<WEIGHTED, 1>(ebufs, a[0] + 16, slot, a[0] + (int)(rw[0] >> 16), g.x, pad, c4, mv2);
        else { c4[0] = c4[1] = c4[2] = c4[3] = pad; mv2[0] = mv2[1] = mv2[2] = mv2[3] = 0.f; }
        pc2[0] = (unsigned)c4[0] | ((unsigned)c4[1] << 16);
        pc2[1] = (unsigned)c4[2] | ((unsigned)c4[3] << 16);
      }
    }
    for (int s = 0; s < sg; ++s) {
      const int c0 = cbase + s * FT;
      // The previous reduction no longer reads the tile / bias.  Its LDS reads were consumed by
      // the arithmetic, so a bare s_barrier is enough: __syncthreads() would also drain the output stores
      // (vmcnt(0) of its release fence) before the next tile may even be requested.
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      unsigned ld32 = (unsigned)ldh;
      asm volatile("" : "+s"(ld32));            // opaque per step: the piece offsets are recomputed, not hoisted and spilled
      if constexpr (MODE == kDuoFoldBits) {     // the 0 / 1 tile straight from the bit image: 8 threads per row, one float4 each
        const uint32_t* bw = fo.bits + (size_t)(c0 / FT) * (size_t)n + g.x;
        float4* t4 = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < g.y * (FT / 4); i += THREADS) {
          const unsigned nib = bw[i >> 3] >> ((i & 7) * 4);
          t4[i] = make_float4((nib & 1u) ? 1.f : 0.f, (nib & 2u) ? 1.f : 0.f, (nib & 4u) ? 1.f : 0.f, (nib & 8u) ? 1.f : 0.f);
        }
      } else if (!(dbg & 1) && !prefetched)
        tile_dma32_asm<FT / 4, THREADS, CAP * (FT / 4)>(lds0 + cur * kDblBuf, h + (int64_t)g.x * ldh + c0, ld32, g.y);
      if (dbl && !zero_rows_ok) {               // (the two small tiles' zero rows: a single-tile unit may have overwritten them)
        if (tid < 2 * FT) lds[(tid / FT) * (kDblBuf / 4) + kDblCap * FT + tid % FT] = 0.f;
        zero_rows_ok = true;
      }
      if (FOLD) {
        if (tid < FT) lbias[tid] = fo.dp[(int64_t)fo.gids[u / upg] * fo.lddp + c0 + tid] * (fo.avg ? 1.0f / (float)g.y : 1.0f);
      } else {
        if (tid < FT) lbias[tid] = bias ? bias[c0 + tid] : 0.f;
      }
      // the tile has landed (asm DMA: hipcc does not count it), the bias slice is written.  (A counted wait that leaves the
      // previous reduction's stores in flight -- range-checked buffer stores, a fixed number per wave -- measured no faster.)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      prefetched = false;
      if (dbl && !(dbg & 1)) {                  // the next tile streams in under this reduction
        if (s + 1 < sg) {
          tile_dma32_asm<FT / 4, THREADS, kDblCap * (FT / 4)>(lds0 + (cur ^ 1) * kDblBuf, h + (int64_t)g.x * ldh + c0 + FT, ld32, g.y);
          prefetched = true;
        } else if (has_next && gn.y <= kDblCap) {
          tile_dma32_asm<FT / 4, THREADS, kDblCap * (FT / 4)>(lds0 + (cur ^ 1) * kDblBuf, h + (int64_t)gn.x * ldh + (un % upg) * sg * FT, ld32, gn.y);
          prefetched = true;
        }
      }
      [[maybe_unused]] unsigned touch = 0;
      if constexpr (MODE == kDuoFoldBits) {
        // The next step's words of the bit image, touched now (one lane per 64-byte piece: at most one load per thread): the
        // expansion at the top of the next step then starts from L2 instead of HBM -- every word is read once per launch, so
        // it is never there by itself.  The value is only kept until the end of this step (a register-held prefetch of the
        // whole slab, 10 words per thread, spilled 26 registers of this 128-VGPR kernel).
        const bool same = s + 1 < sg;
        const int nrow0 = same ? g.x : gn.x, nrows = (same || has_next) ? (same ? g.y : gn.y) : 0;
        const int nc0 = same ? c0 + FT : (un % upg) * sg * FT;
        const uint32_t* bwn = fo.bits + (size_t)(nc0 / FT) * (size_t)n + nrow0;
        if (tid * 16 < nrows && !(dbg_rt & 32)) touch = bwn[tid * 16];
      }
      const unsigned tb0 = tbl + cur * kDblBuf;  // this lane's first chunk of tile row 0 in the current buffer
      if constexpr (MODE == kDuoFold) {         // Y -> [Y > 0], in place
        float4* t4 = reinterpret_cast<float4*>(lds);
        for (int i = tid; i < g.y * (FT / 4); i += THREADS) t4[i] = f4_step(t4[i]);
        __syncthreads();
      }
      [[maybe_unused]] float ps[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // kDuoBitsPool: this lane's column sums of the slab
      [[maybe_unused]] unsigned pk[2] = {0u, 0u};                                 // ... and positive counts, a byte per column
#pragma unroll
      for (int t = 0; t < NI; ++t) if (t * SPAN < g.y) {
        __builtin_amdgcn_sched_barrier(0);      // one row group at a time: bounded live ranges
        const int pos = rbase + t * SPAN;
        const int deg = (int)(rw[t] >> 16);
        f32x2 acc[CPL][2];
#pragma unroll
        for (int j = 0; j < CPL; ++j) acc[j][0] = acc[j][1] = f32x2{0.f, 0.f};
        if (!(dbg & 2)) {
          GCNX_DSTEP4(0, pc[t], mv[t], deg, 0)
          GCNX_DSTEP4(1, pc[t], mv[t], deg, 0)
          GCNX_DSTEP4(2, pc[t], mv[t], deg, 0)
          GCNX_DSTEP4(3, pc[t], mv[t], deg, 0)
          int base = 16;
          if (t == 0) {                         // (the degree order puts a graph's long rows here: 16 more entries in registers)
            if (__builtin_amdgcn_ballot_w64(deg > 16) != 0) {
              GCNX_DSTEP4(0, pc2, mv2, deg, 16)
              GCNX_DSTEP4(1, pc2, mv2, deg, 16)
              GCNX_DSTEP4(2, pc2, mv2, deg, 16)
              GCNX_DSTEP4(3, pc2, mv2, deg, 16)
            }
            base = 32;
          }
          if (__builtin_amdgcn_ballot_w64(deg > base) != 0) {   // longer rows: the rest on demand (power-law batches)
            const int ea = rowrec[g.x + min(pos, g.y - 1)].a;
            while (__builtin_amdgcn_ballot_w64(deg > base) != 0) {
              int xc[4];
              float xv[4];
              fetch_entries<WEIGHTED, 1>(ebufs, ea + base, slot, ea + deg, g.x, pad, xc, xv);
              unsigned xp[2] = {(unsigned)xc[0] | ((unsigned)xc[1] << 16), (unsigned)xc[2] | ((unsigned)xc[3] << 16)};
              GCNX_DSTEP4(0, xp, xv, deg, base)
              GCNX_DSTEP4(1, xp, xv, deg, base)
              GCNX_DSTEP4(2, xp, xv, deg, base)
              GCNX_DSTEP4(3, xp, xv, deg, base)
              base += 16;
            }
          }
        }
        [[maybe_unused]] unsigned bword = 0;
        const int r = (int)(rw[t] & 0xFFFFu);   // the row of the graph this quad owns in this group
        if (pos < g.y && !(dbg & 4)) {
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            const float4 bvj = *reinterpret_cast<const float4*>(lbias + (sub + LPR * (j ^ csw)) * 4);
            float4 o = FOLD ? make_float4(acc[j][0][0] * bvj.x, acc[j][0][1] * bvj.y, acc[j][1][0] * bvj.z, acc[j][1][1] * bvj.w)
                            : make_float4(acc[j][0][0] + bvj.x, acc[j][0][1] + bvj.y, acc[j][1][0] + bvj.z, acc[j][1][1] + bvj.w);
            if (!FOLD && act == GCNX_ACT_RELU) {
              o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            }
            if constexpr (MODE == kDuoBitsPool) {
              const unsigned b4 = (o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u);
              bword |= b4 << (4 * (sub + LPR * (j ^ csw)));
              ps[4 * j] += o.x; ps[4 * j + 1] += o.y; ps[4 * j + 2] += o.z; ps[4 * j + 3] += o.w;
              pk[j] += (b4 * 0x00204081u) & 0x01010101u;           // bit k -> byte k
            } else {
              if constexpr (OUT16) store4_bf16(out, (int64_t)(g.x + r) * ldo + c0 + (sub + LPR * (j ^ csw)) * 4, o);
              else *reinterpret_cast<float4*>(out + (int64_t)(g.x + r) * ldo + c0 + (sub + LPR * (j ^ csw)) * 4) = o;
            }
            if constexpr (MODE == kDuoBitsOut)
              bword |= ((o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u))
                       << (4 * (sub + LPR * (j ^ csw)));
          }
        }
        if constexpr (MODE == kDuoBitsOut || MODE == kDuoBitsPool) {      // the quad's four lanes hold the row's 32 columns: one word per row
          // (DPP quad permutes: pure VALU -- __shfl_xor compiles to ds_bpermute_b32, an LDS-pipe instruction, and this kernel is
          // bound by its LDS reads)
          bword |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)bword, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]: lane ^ 1
          bword |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)bword, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]: lane ^ 2
          if (sub == 0 && pos < g.y) fo.bits[(size_t)(c0 / FT) * (size_t)n + g.x + r] = bword;
        }
      }
      if constexpr (MODE == kDuoBitsPool) {     // the wave's 16 quads (fixed tree); lanes 0..3 write the wave's partial row
        if (csw) {                              // quads 2..5 hold their two chunks in the opposite order (bank swizzle): undo it
#pragma unroll
          for (int i = 0; i < 4; ++i) { const float t_ = ps[i]; ps[i] = ps[4 + i]; ps[4 + i] = t_; }
          const unsigned t_ = pk[0]; pk[0] = pk[1]; pk[1] = t_;
        }
        // the 16 quads' sums without the LDS pipe (r4; 40 ds_bpermute_b32 per wave and slab before): inside a 16-lane row two DPP
        // rotations (lane l ends with l, l + 4, l + 8, l + 12), across the rows v_permlane16_swap / v_permlane32_swap (gfx950);
        // lanes 0..3 hold the wave's sums in a fixed association
#pragma unroll
        for (int i = 0; i < 8; ++i) ps[i] = wave_quads_sum(ps[i]);
        pk[0] = wave_quads_sum_u(pk[0]);
        pk[1] = wave_quads_sum_u(pk[1]);
        if (lane < 4) {
          const size_t prow = ((size_t)(u / upg) * (THREADS / 64) + (tid >> 6)) * (size_t)(upg * sg * FT) + c0;
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            const int col = (sub + LPR * j) * 4;
            *reinterpret_cast<float4*>(fo.ppart + prow + col) = make_float4(ps[4 * j], ps[4 * j + 1], ps[4 * j + 2], ps[4 * j + 3]);
            *reinterpret_cast<float4*>(fo.cpart + prow + col) = make_float4((float)(pk[j] & 255u), (float)((pk[j] >> 8) & 255u),
                                                                            (float)((pk[j] >> 16) & 255u), (float)(pk[j] >> 24));
          }
        }
      }
      if constexpr (MODE == kDuoFoldBits) asm volatile("" ::"v"(touch));   // (the touch load has landed; nothing reads it)
      if (dbl) cur ^= 1;                        // (the next slab's tile -- or the next unit's first -- is in the other buffer)
    }
    if (!has_next) break;
    g = gn;
    u = un;
  }
}
#undef GCNX_DSTEP4

// ----------------------------------------------------------------------------------------------
// Pipelined tile kernel: the tile kernel above with its phases OVERLAPPED.
//
// What the round-1 ablation of the tile kernel showed: tile in + result out run at the HBM rate, the LDS reduction
// costs less than that -- but inside a workgroup they ADD (DMA, barrier, reduce, store), and two workgroups per CU
// cover each other only a little.  Here one 1024-thread workgroup per CU holds TWO source buffers of 78 KiB:
//   * a phase = (graph, column slab).  Graphs of up to 624 rows use 32-column slabs, graphs of up to 1024 rows
//     16-column slabs (25 % more index work per column, but no graph is gathered in pieces); taller graphs stay with
//     the row gather (plan-listed chunks).
//   * while a phase is reduced, the rows of the NEXT phase stream into the other buffer by LDS-DMA, issued a piece at a
//     time between the row groups (a burst blocks the issuing waves for as long as the memory pipeline takes to accept
//     it: 90-150 us of a launch were spent "issuing").  The DMA is inline asm ON PURPOSE: for the builtin, hipcc
//     (ROCm 7.2) puts s_waitcnt vmcnt(0) in front of every later LDS read -- it cannot tell the two buffers apart --
//     which is what serialised every double-buffered version before this one (scripts/micro/lds_dma_overlap.hip shows the
//     hardware overlaps the two perfectly).  The wait for the pieces is placed by hand at the end of the phase,
//     counted past the last row group's stores (range-checked buffer stores: every wave issues the same number).
//   * a row's first 32 entries live in registers for the whole graph: column indices as 16-bit pairs (entry k and
//     entry 16 + k of the row share a register; 0xFFFF... no: padding entries hold the index of an all-zero row), so a
//     quad broadcast (DPP) and ONE v_mad_u32_u16 (op_sel picks the half) give the LDS address; weights as fp32.  Rows of
//     up to 16 entries (98 % at degree 10) fetch nothing for the second half (range-checked buffer loads); rows with
//     more than 32 add the rest after the reduction straight from global memory.
//   * one barrier per phase.
// ----------------------------------------------------------------------------------------------
constexpr int kPipeBufBytes = 624 * 128;              // = 1248 * 64: source rows of a phase
constexpr int kPipeBuf = kPipeBufBytes + 128;         // + the all-zero row padding entries point at
constexpr int kPipeCap32 = 624, kPipeCap16 = 1024;    // graph rows per phase with 32- / 16-column slabs (16: 4 row groups of 256:
                                                      // the registers that hold 32 entries per row allow no more)
constexpr int kPipeMaxF = 512;                        // bias slice kept in LDS
constexpr int kPipeLds = 2 * kPipeBuf + kPipeMaxF * 4;

struct PipeItem { int row0, ng; };                    // graph rows [row0, row0 + ng)

// LDS-DMA of one 16-byte piece per lane (see above why not the builtin).  lds_base: wave-uniform byte address (M0); the
// DMA adds lane * 16.
__device__ __forceinline__ void pipe_dma16(const float* gptr, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_base) : "memory", "m0");
}
// LDS address of an entry's row: base + (16-bit half of packed) * row_bytes, one instruction.
template <int HI>
__device__ __forceinline__ unsigned pipe_addr(unsigned packed, unsigned row_bytes, unsigned base) {
  unsigned a;
  if (HI) asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(a) : "v"(packed), "v"(row_bytes), "v"(base));
  else asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(a) : "v"(packed), "v"(row_bytes), "v"(base));
  return a;
}

// One phase's reduction of row group T: 16 (+16) entries, FT columns per row (CPL float4 per lane of the quad).
#define GCNX_PSTEP4(J, HI, MV, T)                                                                              \
  if (__builtin_amdgcn_ballot_w64(slot == (J) && (HI ? cnt > 16 + 4 * (J) : cnt > 4 * (J))) != 0) {             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                            \
      const unsigned cp = (unsigned)quad_bcast<4, (J)>((int)colp[T][i]);                                        \
      const unsigned a = pipe_addr<HI>(cp, FT * 4u, lbase);                                                     \
      f32x2 w2 = f32x2{1.f, 1.f};                                                                              \
      if (WEIGHTED) { const float w = __int_as_float(quad_bcast<4, (J)>(__float_as_int(MV[T][i]))); w2 = f32x2{w, w}; } \
      _Pragma("unroll") for (int j = 0; j < CPL; ++j) {                                                        \
        const f32x4v hq = *reinterpret_cast<const __attribute__((address_space(3))) f32x4v*>(a + j * 64u);      \
        const float4 hv = make_float4(hq.x, hq.y, hq.z, hq.w);                                                 \
        if (WEIGHTED) {                                                                                        \
          acc[T][j][0] = __builtin_elementwise_fma(w2, f32x2{hv.x, hv.y}, acc[T][j][0]);                       \
          acc[T][j][1] = __builtin_elementwise_fma(w2, f32x2{hv.z, hv.w}, acc[T][j][1]);                       \
        } else {                                                                                               \
          acc[T][j][0] += f32x2{hv.x, hv.y};                                                                   \
          acc[T][j][1] += f32x2{hv.z, hv.w};                                                                   \
        }                                                                                                      \
      }                                                                                                        \
    }                                                                                                          \
  }

struct PipeCursor { int round, ph, nph, row0, ng, cbase, ft; bool live; };

template <bool WEIGHTED, int KFT>
__global__ __launch_bounds__(1024, 4) void spmm_pipe_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx, const float* __restrict__ vals,
    const float* __restrict__ h, int64_t ldh, const float* __restrict__ bias, float* __restrict__ out, int64_t ldo,
    const PipeItem* __restrict__ items, int upg /* slab groups per graph */, int sg /* 32-column slabs per group */, int act,
    int nunits, int n, int f, int dbg_rt, unsigned long long* __restrict__ stamps) {
#ifdef GCNX_TUNING
#define PIPE_STAMP(K) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); seg[K] += t_ - tlast; tlast = t_; }
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memrealtime();
  const int dbg = dbg_rt;                             // phase-ablation bits of a tuning build (results wrong by design):
#else                                                 // 1 no DMA, 2 no reduction, 4 no stores, 8 no index burst
  constexpr int dbg = 0;
  (void)dbg_rt; (void)stamps;
#define PIPE_STAMP(K)
#endif
  extern __shared__ __attribute__((aligned(16))) char plds[];
  float* lbias = reinterpret_cast<float*>(plds + 2 * kPipeBuf);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x;
  const int w = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;   // consecutive virtual ids share an XCD
  auto unit_of = [&](int round) { return round * G + ((round & 1) ? (G - 1 - w) : w); };
  if (unit_of(0) >= nunits) return;                   // uniform per workgroup: before any barrier
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)plds;

  // Phase sequence of this workgroup: unit -> (graph, slab group); the group's sg 32-column slabs as sg phases of 32
  // columns (graphs of up to 624 rows) or 2 sg phases of 16.  A cursor walks it one phase ahead for the DMA.
  auto load_unit = [&](PipeCursor& k) {
    const int u = unit_of(k.round);
    k.live = u < nunits;
    if (!k.live) return;
    const PipeItem it = items[u / upg];
    k.row0 = __builtin_amdgcn_readfirstlane(it.row0); k.ng = __builtin_amdgcn_readfirstlane(it.ng);
    k.cbase = (u % upg) * sg * 32;
    k.ft = KFT;
    k.nph = sg * (32 / KFT);
    k.ph = 0;
  };
  auto advance = [&](PipeCursor& k) {
    if (++k.ph < k.nph) return;
    ++k.round;
    load_unit(k);
  };
  // Piece step kk (0..4) of the phase under cursor k into buffer buf: one 1-KiB instruction per wave = 8 rows of 128 B
  // (32-column slabs) or 16 rows of 64 B (16-column slabs); wave v takes rows [8 | 16] (v + 16 kk) ...
  auto dma_step = [&](const PipeCursor& k, int buf, int kk) {
    if (!k.live || (dbg & 1)) return;
    const int c0 = k.cbase + k.ph * k.ft;
    if (KFT == 32) {
      const int r = wave * 8 + kk * 128 + (lane >> 3);
      if (wave * 8 + kk * 128 < k.ng) {               // uniform per wave
        const float* src = h + (int64_t)(k.row0 + min(r, k.ng - 1)) * ldh + c0 + (lane & 7) * 4;
        if (r < k.ng) pipe_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + buf * kPipeBuf + (wave * 8 + kk * 128) * 128));
      }
    } else {
      const int r = wave * 16 + kk * 256 + (lane >> 2);
      if (wave * 16 + kk * 256 < k.ng) {
        const float* src = h + (int64_t)(k.row0 + min(r, k.ng - 1)) * ldh + c0 + (lane & 3) * 4;
        if (r < k.ng) pipe_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + buf * kPipeBuf + (wave * 16 + kk * 256) * 64));
      }
    }
  };

  const int sub = lane & 3, slot = sub;               // lane of the quad: entries [4 slot, 4 slot + 4) (+16) of its row
  const int rbase = wave * 16 + (lane >> 2);          // this quad's row within a pass of 256
  for (int i = tid; i < 32; i += 1024) {              // the two zero rows
    reinterpret_cast<float*>(plds + kPipeBufBytes)[i] = 0.f;
    reinterpret_cast<float*>(plds + kPipeBuf + kPipeBufBytes)[i] = 0.f;
  }
  for (int i = tid; i < f; i += 1024) lbias[i] = bias ? bias[i] : 0.f;
  const int nnz = rowptr[n];
  const EntryBufs ebufs = entry_bufs(colidx, vals, nnz);
  const __amdgpu_buffer_rsrc_t obuf =
      __builtin_amdgcn_make_buffer_rsrc((void*)out, (short)0, (int)min((uint64_t)n * (uint64_t)ldo * 4u, (uint64_t)0xFFFFFFF0u), 0x00020000);
  const unsigned ldo4 = (unsigned)ldo * 4u;

  PipeCursor nxt{0, 0, 1, 0, 0, 0, 32, false};
  load_unit(nxt);
#pragma unroll
  for (int kk = 0; kk < 5; ++kk) dma_step(nxt, 0, kk);   // phase 0
  advance(nxt);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  PIPE_STAMP(0)                                       // prologue
  int buf = 0;

  // The body of one graph's phases, for one slab width: FT columns per phase, NI row groups of 256 rows.
#define GCNX_PIPE_GRAPH(FT_, NI_)                                                                               \
  {                                                                                                             \
    constexpr int FT = FT_, NI = NI_, CPL = FT / 16;                                                            \
    constexpr unsigned ZROW = kPipeBufBytes / (FT * 4);      /* index of the all-zero row in either buffer */   \
    unsigned colp[NI][4], rcnt[NI];                                                                             \
    float mv[NI][4], mv2[NI][4];                                                                                \
    _Pragma("unroll") for (int t = 0; t < NI; ++t) {                                                            \
      const int r = rbase + t * 256;                                                                            \
      const I2u p = *reinterpret_cast<const I2u*>(rowptr + it.row0 + min(r, it.ng - 1));                        \
      const int ea_t = p.x;                                                                                     \
      rcnt[t] = r < it.ng ? (unsigned)(p.y - p.x) : 0u;                                                         \
      int c1[4], c2[4];                                                                                         \
      const int eb = ea_t + (int)rcnt[t];                                                                       \
      if (!(dbg & 8)) {                                                                                         \
        fetch_entries<WEIGHTED, 1>(ebufs, ea_t, slot, eb, it.row0, (int)ZROW, c1, mv[t]);                       \
        fetch_entries<WEIGHTED, 1>(ebufs, ea_t + 16, slot, eb, it.row0, (int)ZROW, c2, mv2[t]);                 \
      } else {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { c1[i] = c2[i] = (int)ZROW; mv[t][i] = mv2[t][i] = 0.f; } \
      }                                                                                                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) colp[t][i] = (unsigned)c1[i] | ((unsigned)c2[i] << 16);     \
    }                                                                                                           \
    PIPE_STAMP(1)                                     /* index burst of the graph */                           \
    for (int ph = 0; ph < nph; ++ph) {                                                                          \
      const int c0 = cbase + ph * FT;                                                                           \
      const unsigned lbase = lds0 + buf * kPipeBuf + sub * 16;                                                  \
      f32x2 acc[NI][CPL][2];                                                                                    \
      _Pragma("unroll") for (int t = 0; t < NI; ++t)                                                            \
        _Pragma("unroll") for (int j = 0; j < CPL; ++j) acc[t][j][0] = acc[t][j][1] = f32x2{0.f, 0.f};          \
      _Pragma("unroll") for (int t = 0; t < NI; ++t) {                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        /* the next phase's rows, a piece at a time: all five pieces are out before the last row group's stores */ \
        if (NI == 3) { if (t == 0) { dma_step(nxt, buf ^ 1, 0); dma_step(nxt, buf ^ 1, 1); }                     \
                       if (t == 1) { dma_step(nxt, buf ^ 1, 2); dma_step(nxt, buf ^ 1, 3); }                     \
                       if (t == 2) dma_step(nxt, buf ^ 1, 4); }                                                 \
        else { dma_step(nxt, buf ^ 1, t); if (t == NI - 1) dma_step(nxt, buf ^ 1, NI); }                         \
        const unsigned cnt = rcnt[t];                                                                           \
        if (t * 256 < it.ng && !(dbg & 2)) {                                                                    \
          GCNX_PSTEP4(0, 0, mv, t) GCNX_PSTEP4(1, 0, mv, t) GCNX_PSTEP4(2, 0, mv, t) GCNX_PSTEP4(3, 0, mv, t)     \
          if (__builtin_amdgcn_ballot_w64(cnt > 16) != 0) {                                                     \
            GCNX_PSTEP4(0, 1, mv2, t) GCNX_PSTEP4(1, 1, mv2, t) GCNX_PSTEP4(2, 1, mv2, t) GCNX_PSTEP4(3, 1, mv2, t) \
          }                                                                                                     \
          if (__builtin_amdgcn_ballot_w64(cnt > 32) != 0) {   /* the rest straight from global memory (rare) */  \
            const int ea_t = rowptr[it.row0 + min(rbase + t * 256, it.ng - 1)];                                 \
            for (int e = ea_t + 32; e < ea_t + (int)cnt; ++e) {                                                 \
              const int col = colidx[e];                                                                        \
              const float v = WEIGHTED ? vals[e] : 1.0f;                                                        \
              const float* hr = h + (int64_t)col * ldh + c0 + sub * 4;                                          \
              _Pragma("unroll") for (int j = 0; j < CPL; ++j) {                                                 \
                const float4 hv = *reinterpret_cast<const float4*>(hr + j * 16);                                \
                acc[t][j][0] = __builtin_elementwise_fma(f32x2{v, v}, f32x2{hv.x, hv.y}, acc[t][j][0]);         \
                acc[t][j][1] = __builtin_elementwise_fma(f32x2{v, v}, f32x2{hv.z, hv.w}, acc[t][j][1]);         \
              }                                                                                                 \
            }                                                                                                   \
          }                                                                                                     \
        }                                                                                                       \
        /* epilogue of the row group right behind its reduction: bias, activation, FT * 4 bytes per row.  Range-  \
           checked buffer stores (a quad without a row gets an out-of-range offset): every wave issues exactly CPL \
           per row group, the count the hand-placed vmcnt below relies on. */                                    \
        const int r = rbase + t * 256;                                                                          \
        _Pragma("unroll") for (int j = 0; j < CPL; ++j) {                                                       \
          const int cc = c0 + sub * 4 + j * 16;                                                                 \
          const float4 bvj = *reinterpret_cast<const float4*>(lbias + cc);                                      \
          float4 o = make_float4(acc[t][j][0][0] + bvj.x, acc[t][j][0][1] + bvj.y, acc[t][j][1][0] + bvj.z, acc[t][j][1][1] + bvj.w); \
          if (act == GCNX_ACT_RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); } \
          const unsigned off = (r < it.ng && !(dbg & 4)) ? (unsigned)(it.row0 + r) * ldo4 + (unsigned)cc * 4u : 0xFFFFFFF0u; \
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, f32x4v{o.x, o.y, o.z, o.w}), obuf, off, 0, 0); \
        }                                                                                                       \
      }                                                                                                         \
      advance(nxt);                                                                                             \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
      PIPE_STAMP(3)                                   /* reduction + epilogue issue (+ DMA issue) */           \
      /* every DMA piece of the next phase is older than the last row group's CPL stores */                     \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CPL) : "memory");                                     \
      PIPE_STAMP(4)                                   /* wait for the next phase's rows */                     \
      __builtin_amdgcn_s_barrier();                                                                             \
      PIPE_STAMP(5)                                   /* barrier */                                            \
      buf ^= 1;                                                                                                 \
    }                                                                                                           \
  }

  for (int round = 0;; ++round) {
    const int u = unit_of(round);
    if (u >= nunits) break;
    const PipeItem it = items[u / upg];
    const int cbase = (u % upg) * sg * 32;
    const int nph = sg * (32 / KFT);
    if (KFT == 32) GCNX_PIPE_GRAPH(32, 3)
    else GCNX_PIPE_GRAPH(16, 4)
  }
#undef GCNX_PIPE_GRAPH
#ifdef GCNX_TUNING
  if (stamps && (tid & 63) == 0)
    for (int k = 0; k < 8; ++k) stamps[((size_t)blockIdx.x * 16 + wave) * 8 + k] = seg[k];
#endif
#undef PIPE_STAMP
}
#undef GCNX_PSTEP4

// ----------------------------------------------------------------------------------------------
// Hub rows (r3; power-law batches, BASELINE config 5: degrees up to 4096).  Inside spmm_rows_kernel a long row is walked
// by the four waves of ONE workgroup: the 4096-entry row took 518 us of a 736 us launch, max / mean wave lifetime 15.6.
// A plan now lists every row of more than kHubDeg entries as segments of kHubSeg entries; a workgroup of
// spmm_hub_seg_kernel sums one segment (four waves, 64 entries each, 8 gathers in flight per lane, partial sums merged in
// wave order) into a partial row, spmm_hub_combine_kernel adds a row's partials in segment order and applies the
// epilogue.  Deterministic; the row gather skips these rows (hub_deg).
// ----------------------------------------------------------------------------------------------
constexpr int kHubDeg = 256;
constexpr int kHubSeg = 256;
struct HubSeg { int row, a, b, slot; };          // entries [a, b) of `row` -> partial row `slot`
struct HubRow { int row, slot0, nseg, pad; };

template <bool WEIGHTED, bool FOLD>
__global__ __launch_bounds__(256) void spmm_hub_seg_kernel(const HubSeg* __restrict__ segs, const int32_t* __restrict__ colidx,
                                                           const float* __restrict__ vals, const float* __restrict__ h, int64_t ldh,
                                                           float* __restrict__ part, int32_t n, int32_t f, int nnz) {
  __shared__ float4 s_part[4][64];
  const HubSeg sg = segs[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (sg.b - sg.a + 3) / 4;
  const int wa = sg.a + wave * per, wb = min(sg.b, wa + per);
  const __amdgpu_buffer_rsrc_t cbuf = __builtin_amdgcn_make_buffer_rsrc((void*)colidx, (short)0, nnz * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t vbuf =
      __builtin_amdgcn_make_buffer_rsrc((void*)(WEIGHTED ? (const void*)vals : (const void*)colidx), (short)0, nnz * 4, 0x00020000);
  const bool use_buf = (uint64_t)n * (uint64_t)ldh * 4u < 0xFFFFFF00ull;
  const __amdgpu_buffer_rsrc_t hbuf =
      __builtin_amdgcn_make_buffer_rsrc((void*)h, (short)0, use_buf ? (int)((uint64_t)n * (uint64_t)ldh * 4u) : 0, 0x00020000);
  const unsigned ld32 = (unsigned)ldh;
  for (int c0 = 0; c0 < f; c0 += 256) {                   // 64 lanes x float4 per pass
    const int c = c0 + lane * 4;
    const bool col_ok = c < f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int HU = 8;
    for (int e = wa; e < wb; e += HU) {                   // (uniform per wave)
      int ci[HU];
      float wv[HU];
#pragma unroll
      for (int u = 0; u < HU; ++u) {
        const unsigned off = e + u < wb ? (unsigned)(e + u) * 4u : 0xFFFFFFF0u;
        ci[u] = __builtin_amdgcn_raw_buffer_load_b32(cbuf, off, 0, 0);
        wv[u] = WEIGHTED ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(vbuf, off, 0, 0)) : 1.0f;
      }
      f32x4v hv[HU];
#pragma unroll
      for (int u = 0; u < HU; ++u) {
        if (use_buf) {
          const unsigned off = (e + u < wb && col_ok) ? ((unsigned)ci[u] * ld32 + (unsigned)c) * 4u : 0xFFFFFFF0u;
          hv[u] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(hbuf, off, 0, 0));
        } else {
          hv[u] = f32x4v{0.f, 0.f, 0.f, 0.f};
          if (e + u < wb && col_ok) { const float4 t = *reinterpret_cast<const float4*>(h + (int64_t)ci[u] * ldh + c); hv[u] = f32x4v{t.x, t.y, t.z, t.w}; }
        }
      }
#pragma unroll
      for (int u = 0; u < HU; ++u) {
        float4 t = make_float4(hv[u].x, hv[u].y, hv[u].z, hv[u].w);
        if (FOLD) t = f4_step(t);
        if (e + u < wb) acc = WEIGHTED ? f4_fma(wv[u], t, acc) : f4_add(acc, t);
      }
    }
    __syncthreads();                                      // (s_part free: the previous column pass was consumed)
    s_part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && col_ok) {
      const float4 t = f4_add(f4_add(s_part[0][lane], s_part[1][lane]), f4_add(s_part[2][lane], s_part[3][lane]));
      *reinterpret_cast<float4*>(part + (size_t)sg.slot * f + c) = t;
    }
  }
}

template <bool FOLD>
__global__ __launch_bounds__(64) void spmm_hub_combine_kernel(const HubRow* __restrict__ rows, const float* __restrict__ part,
                                                              const float* __restrict__ bias, float* __restrict__ out, int64_t ldo,
                                                              int32_t f, int act, FoldArgs fo) {
  const HubRow hr = rows[blockIdx.x];
  const int lane = threadIdx.x;
  int g = 0;
  float sc = 1.0f;
  if (FOLD) {                                             // graph of the row: binary search in graph_ptr (a few hundred rows in all)
    int lo = 0, hi = fo.b;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (fo.gp[mid] <= hr.row) lo = mid; else hi = mid; }
    g = lo;
    if (fo.avg) sc = 1.0f / (float)(fo.gp[g + 1] - fo.gp[g]);
  }
  for (int c = lane * 4; c < f; c += 256) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < hr.nseg; ++k) t = f4_add(t, *reinterpret_cast<const float4*>(part + (size_t)(hr.slot0 + k) * f + c));
    if (FOLD) {
      const float4 d = *reinterpret_cast<const float4*>(fo.dp + (int64_t)g * fo.lddp + c);
      t.x *= d.x * sc; t.y *= d.y * sc; t.z *= d.z * sc; t.w *= d.w * sc;
    } else {
      if (bias) { const float4 bv = *reinterpret_cast<const float4*>(bias + c); t = f4_add(t, bv); }
      if (act == GCNX_ACT_RELU) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
    }
    *reinterpret_cast<float4*>(out + (int64_t)hr.row * ldo + c) = t;
  }
}

// ----------------------------------------------------------------------------------------------
// Column-block row gather (r4; BASELINE config 5).  A graph whose feature rows do not fit an XCD's L2 -- 8 192 nodes x
// 1 KiB = 8 MiB against 4 MiB -- is gathered ~degree times out of a working set that keeps falling out of the cache (r3 PMC:
// L2 hit rate 0.52, 1.88 x the compulsory HBM traffic).  Here such a graph is walked one 64-COLUMN block at a time: all its
// rows for columns [0, 64), then [64, 128), ... -- consecutive work items, which the workgroup-id remap hands to ONE XCD -- so
// the block's source rows (8 192 x 256 B = 2 MiB) stay in that XCD's L2 while they are gathered, and the graph's index
// arrays (re-read once per block) stay there too.
// A 64-column block is 16 lanes x float4, so a wave holds four lane groups, and in a power-law batch 83 % of the rows have at
// most three entries while a third of the ENTRIES sit in rows of hundreds.  The first version gave the four groups four
// rows in row order: the traffic fell as planned and the launch got slower (895 us against 705), because a wave walked
// max(length of its rows) trips with most lanes idle and the long rows ran one after the other behind barriers.  So the
// rows of such a graph are taken in DEGREE order (the plan's RowRec list, as for the tile kernels) and an item is one of:
//   kind 3  32 rows of at most kCbShort (32) entries, 8 per wave, two per lane group (kCbRpg) -- neighbours in the order have
//           (nearly) the same length, so a wave's entry slots are all useful;
//   kind 1  4 rows of kCbShort + 1 .. kCbHub entries: a wave per row, 64 entries per step (4 groups x 16 slots);
//   kind 0  1 row of more than kCbHub entries: the four waves take a quarter each (fixed combination order).
// The long rows' items are dealt among the short rows' (they gather out of L2, the short rows stream from HBM; GCNX_SPMM_CB=2:
// heaviest first -- no difference measured).  Range-checked buffer loads throughout: a missing entry costs no fetch and adds an
// exact zero.  A row of kind 3 adds its entries in CSR order -- the row gather's order at f = 256, bit for bit.
// What the first version cost and why (profiles/r04/config5_column_blocks_ablation.txt): every entry was its own 4-byte broadcast
// load per lane group -- sixteen vector-memory instructions per wave for eight rows -- and a CU's vector-memory pipeline takes ONE
// instruction at a time whatever its width: with gathers and stores compiled out, those index loads were 259 of the short rows'
// 607 us; and that pipeline's time ADDS to the HBM time instead of hiding under it.  Hence few, wide index instructions: one
// record load per wave, one entry load per array and four slots (kind 3) or 64 entries (kinds 0 / 1), slots handed to their lane
// group by DPP row broadcasts; result rows stored with the non-temporal policy.  Config 5 (same boxes): 605-640 us against 682-714
// for the row gather + hub segments (frac 0.41-0.43 against 0.37), HBM traffic 1.04 x the algorithmic bytes against 1.88 x.  The default for such graphs since (GCNX_SPMM_CB=0: the r3 path).  Measured on the way and dropped: items of
// 128 rows software-pipelined over four octets per wave (776-1000 us), short rows left in row order (937 us), other hub
// thresholds / entries per trip (no effect), fewer workgroups per CU for a higher L2 hit rate (slower at every setting).
// FOLD (gcnx_spmm_csr_pool_bwd): h is the saved ReLU output of the pooled layer -- summed as its 0 / 1 mask -- and the graph's
// dPooled row (x 1 / n_g for the average pool) multiplies the finished sum where the bias is added otherwise.
// ----------------------------------------------------------------------------------------------
constexpr int kCbCols = 64;          // columns per block
constexpr int kCbMinRows = 4096;     // graphs of at least this many rows are walked this way (and fewer than 65 536: RowRec)
constexpr int kCbShort = 32;         // kind 3 (short rows) up to here
#ifndef GCNX_CB_RPG
#define GCNX_CB_RPG 2
#endif
// rows per lane group of a kind-3 item: 2 = 8 rows per wave, 32 per item; 4 = 16 / 64.  With 4 a wave issues fewer index instructions
// per row, but 16 k rows are then in flight per XCD -- two (graph, block)s, 4 MiB of source rows against the 4-MiB L2: measured
// (same box, profiles/r04/config5_column_blocks_rows_per_group.txt) 4: 620 us, 1.22 x the compulsory HBM bytes; 2: 605 us, 1.04 x
constexpr int kCbRpg = GCNX_CB_RPG;
constexpr int kCbSps = 16 / kCbRpg;  // entry slots one load fetches per row
#ifndef GCNX_CB_HUB
#define GCNX_CB_HUB 512
#endif
constexpr int kCbHub = GCNX_CB_HUB;  // kind 1 up to here
constexpr int kCbMinF = 128;         // narrower features: the whole graph fits L2 anyway

// Lane N of every 16-lane row, to all lanes of that row (DPP row_newbcast: one VALU move, no LDS).
template <int N>
__device__ __forceinline__ int row_bcast16(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xF, 0xF, false); }

template <bool WEIGHTED, bool FOLD, int S, int K>
__device__ __forceinline__ void cb_slot_row(const __amdgpu_buffer_rsrc_t hbuf, unsigned ld32, unsigned c, bool no_gather, int ci, float wv,
                                            int deg, float4& acc) {
  const int col = row_bcast16<K + kCbRpg * (S % kCbSps)>(ci);
  const unsigned off = (S < deg && !no_gather) ? ((unsigned)col * ld32 + c) * 4u : 0xFFFFFFF0u;
  float4 hv = buf4(hbuf, off);
  if (FOLD) hv = f4_step(hv);                 // (the gathered operand is the saved ReLU output: its 0 / 1 mask is what is summed)
  if (WEIGHTED) acc = f4_fma(__int_as_float(row_bcast16<K + kCbRpg * (S % kCbSps)>(__float_as_int(wv))), hv, acc);
  else acc = f4_add(acc, hv);
}

// Entry slots [J0, J) of four rows (spmm_cb_kernel, kind 3): slot s of row k is held by lane k + 4 (s & 3) of the lane group, in
// register set s >> 2.
template <bool WEIGHTED, bool FOLD, int J, int J0, int T>
__device__ __forceinline__ void cb_slots(const __amdgpu_buffer_rsrc_t hbuf, unsigned ld32, unsigned c, bool no_gather, const int (&ci)[T],
                                         const float (&wv)[T], const int (&deg)[kCbRpg], float4 (&acc)[kCbRpg]) {
  if constexpr (J0 < J) {
    cb_slot_row<WEIGHTED, FOLD, J0, 0>(hbuf, ld32, c, no_gather, ci[J0 / kCbSps], wv[J0 / kCbSps], deg[0], acc[0]);
    cb_slot_row<WEIGHTED, FOLD, J0, 1>(hbuf, ld32, c, no_gather, ci[J0 / kCbSps], wv[J0 / kCbSps], deg[1], acc[1]);
    if constexpr (kCbRpg == 4) {
      cb_slot_row<WEIGHTED, FOLD, J0, 2>(hbuf, ld32, c, no_gather, ci[J0 / kCbSps], wv[J0 / kCbSps], deg[2], acc[2]);
      cb_slot_row<WEIGHTED, FOLD, J0, 3>(hbuf, ld32, c, no_gather, ci[J0 / kCbSps], wv[J0 / kCbSps], deg[3], acc[3]);
    }
    // (eight gathers in flight: more hoisted together spill; FOLD -- the mask arithmetic needs registers of its own -- four)
    if constexpr ((((J0 + 1) * kCbRpg) % ((FOLD || kCbRpg == 2) ? 4 : 8)) == 0 && J0 + 1 < J) __builtin_amdgcn_sched_barrier(0);
    cb_slots<WEIGHTED, FOLD, J, J0 + 1, T>(hbuf, ld32, c, no_gather, ci, wv, deg, acc);
  }
}

// The same for ONE row walked by a whole wave (kinds 0 and 1): slot j of lane group g is entry 4 j + g of the current 64-entry step.
template <bool WEIGHTED, bool FOLD, int J, int J0>
__device__ __forceinline__ void cb_slots1(const __amdgpu_buffer_rsrc_t hbuf, unsigned ld32, unsigned c, bool no_gather, int ci, float wv,
                                          int left, float4& acc) {
  if constexpr (J0 < J) {
    const int col = row_bcast16<J0>(ci);
    const unsigned off = (J0 < left && !no_gather) ? ((unsigned)col * ld32 + c) * 4u : 0xFFFFFFF0u;
    float4 hv = buf4(hbuf, off);
    if (FOLD) hv = f4_step(hv);
    if (WEIGHTED) acc = f4_fma(__int_as_float(row_bcast16<J0>(__float_as_int(wv))), hv, acc);
    else acc = f4_add(acc, hv);
    cb_slots1<WEIGHTED, FOLD, J, J0 + 1>(hbuf, ld32, c, no_gather, ci, wv, left, acc);
  }
}

#ifndef GCNX_CB_OCC
#define GCNX_CB_OCC 8
#endif
template <bool WEIGHTED, bool FOLD = false>
__global__ __launch_bounds__(256, GCNX_CB_OCC) void spmm_cb_kernel(const RowRec* __restrict__ rowrec, const int32_t* __restrict__ colidx,
                                                         const float* __restrict__ vals, const float* __restrict__ h, int64_t ldh,
                                                         const float* __restrict__ bias, float* __restrict__ out, int64_t ldo,
                                                         int32_t n, int32_t nnz, int act, int nitems, const int4* __restrict__ items, int dbg,
                                                         FoldArgs fo, const int32_t* __restrict__ cb_gids) {
  __shared__ float4 s_long[4][16];
  const int4 it = items[gcnx_xcd_remap(blockIdx.x, nitems)];
  // (item: first position, rows | kind << 16, first column | index among the column-block graphs << 16, the graph's first row)
  const int p0 = it.x, cnt = it.y & 0xFFFF, kind = it.y >> 16, col0 = it.z & 0xFFFF, row0 = it.w;
  if (dbg & 16) { if (dbg == 0x7fffffff) out[0] = (float)p0; return; }      // (tuning: launch + item load only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, sub = lane & 15;
  const int c = col0 + sub * 4;
  // bv: the bias slice of this lane's four columns; FOLD (gcnx_spmm_csr_pool_bwd): the graph's dPooled row (x 1 / n_g for the average
  // pool) instead, which MULTIPLIES the finished sum of the gathered ReLU masks
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (FOLD) {
    const int gid = cb_gids[(unsigned)it.z >> 16];
    bv = *reinterpret_cast<const float4*>(fo.dp + (int64_t)gid * fo.lddp + c);
    if (fo.avg) { const float sc = 1.0f / (float)(fo.gp[gid + 1] - fo.gp[gid]); bv.x *= sc; bv.y *= sc; bv.z *= sc; bv.w *= sc; }
  } else if (bias) bv = *reinterpret_cast<const float4*>(bias + c);
  // (the host checks that byte offsets into h fit 32 bits)
  const __amdgpu_buffer_rsrc_t hbuf = __builtin_amdgcn_make_buffer_rsrc((void*)h, (short)0, (int)((uint64_t)n * (uint64_t)ldh * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t cbuf = __builtin_amdgcn_make_buffer_rsrc((void*)colidx, (short)0, nnz * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t vbuf =
      __builtin_amdgcn_make_buffer_rsrc((void*)(WEIGHTED ? (const void*)vals : (const void*)colidx), (short)0, nnz * 4, 0x00020000);
  const unsigned ld32 = (unsigned)ldh;
  constexpr unsigned kOob = 0xFFFFFFF0u;
  auto epilogue = [&](float4 acc, int r) {
    if (FOLD) { acc.x *= bv.x; acc.y *= bv.y; acc.z *= bv.z; acc.w *= bv.w; }
    else {
      acc = f4_add(acc, bv);
      if (act == GCNX_ACT_RELU) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    }
    if (dbg & 1) return;
    // non-temporal: a result row is not read again by this launch and should not push the block's source rows out of L2
    if (!(dbg & 8)) __builtin_nontemporal_store(f32x4v{acc.x, acc.y, acc.z, acc.w}, reinterpret_cast<f32x4v*>(out + (int64_t)r * ldo + c));
    else *reinterpret_cast<float4*>(out + (int64_t)r * ldo + c) = acc;
  };
  auto ld_col = [&](unsigned off) { return __builtin_amdgcn_raw_buffer_load_b32(cbuf, off, 0, 0); };
  auto ld_val = [&](unsigned off) { return WEIGHTED ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(vbuf, off, 0, 0)) : 1.0f; };
  auto h_off = [&](bool ok, int col) { return (ok && !(dbg & 4)) ? ((unsigned)col * ld32 + (unsigned)c) * 4u : kOob; };
  if (kind == 3) {
    // 16 kCbRpg rows of at most kCbShort entries, 4 kCbRpg per wave: lane group g owns positions wave * 4 kCbRpg + g + 4 k
    // (k < kCbRpg), and lane j of a group stands for (row k = j % kCbRpg, entry slot j / kCbRpg) -- ONE record load per wave, and
    // one load per array fetches 16 / kCbRpg entry slots of every row of every group (rows of at most 4 entries are 79 % of a
    // power-law graph's rows).  Slot s of row k then lives in lane k + kCbRpg (s % kCbSps) of register set s / kCbSps and reaches
    // the group's lanes by a DPP row broadcast; the gathers of the J entry slots are straight-line code (a slot past a row's end is
    // an out-of-range buffer offset), so a wave has one dependent round trip per stage -- records, entries, gathers, stores.
    const int kq = sub % kCbRpg, sq = sub / kCbRpg;
    const int posq = wave * (4 * kCbRpg) + g + 4 * kq;
    const RowRec rq = rowrec[p0 + min(posq, cnt - 1)];
    const int aq = rq.a, dq = posq < cnt ? (int)(rq.w >> 16) : 0, rlq = posq < cnt ? (int)(rq.w & 0xFFFFu) : -1;
    if (dbg & 32) { if (aq == 0x7fffffff) out[0] = 1.f; return; }
    int deg[kCbRpg];
    deg[0] = row_bcast16<0>(dq); deg[1] = row_bcast16<1>(dq);
    if constexpr (kCbRpg == 4) { deg[2] = row_bcast16<2>(dq); deg[3] = row_bcast16<3>(dq); }
    const int maxd = __builtin_amdgcn_readfirstlane(dq);          // degree order: the wave's first position holds its longest row
    float4 acc[kCbRpg];
#pragma unroll
    for (int k = 0; k < kCbRpg; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    // entry slots [base, base + J) of the group's rows: J <= 8
    auto slots = [&](auto jt, int base) {
      constexpr int J = decltype(jt)::value;
      constexpr int T = (J + kCbSps - 1) / kCbSps;
      int ci[T];
      float wv[T];
      int dg[kCbRpg];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const unsigned off = (base + kCbSps * t + sq < dq && !(dbg & 64)) ? (unsigned)(aq + base + kCbSps * t + sq) * 4u : kOob;
        ci[t] = ld_col(off); wv[t] = ld_val(off);
      }
#pragma unroll
      for (int k = 0; k < kCbRpg; ++k) dg[k] = deg[k] - base;
      cb_slots<WEIGHTED, FOLD, J, 0, T>(hbuf, ld32, (unsigned)c, (dbg & 4) != 0, ci, wv, dg, acc);
    };
    if (maxd <= 1) slots(std::integral_constant<int, 1>{}, 0);
    else if (maxd <= 2) slots(std::integral_constant<int, 2>{}, 0);
    else if (maxd <= 3) slots(std::integral_constant<int, 3>{}, 0);
    else if (maxd <= 4) slots(std::integral_constant<int, 4>{}, 0);
    else if (maxd <= 6) slots(std::integral_constant<int, 6>{}, 0);
    else if (maxd <= 8) slots(std::integral_constant<int, 8>{}, 0);
    else for (int base = 0; base < maxd; base += 8) {           // (6.6 % of the rows; uniform trip count)
      slots(std::integral_constant<int, 8>{}, base);
      __builtin_amdgcn_sched_barrier(0);
    }
    { const int r = row_bcast16<0>(rlq); if (r >= 0) epilogue(acc[0], row0 + r); }
    { const int r = row_bcast16<1>(rlq); if (r >= 0) epilogue(acc[1], row0 + r); }
    if constexpr (kCbRpg == 4) {
      { const int r = row_bcast16<2>(rlq); if (r >= 0) epilogue(acc[2], row0 + r); }
      { const int r = row_bcast16<3>(rlq); if (r >= 0) epilogue(acc[3], row0 + r); }
    }
    return;
  }
  // kinds 1 and 0: a wave walks entries [wa, wb) of one row, its four lane groups taking every fourth entry, 4 per group and trip
  const RowRec rr = rowrec[p0 + (kind == 1 ? min(wave, cnt - 1) : 0)];
  const int deg = (int)(rr.w >> 16);
  int wa = rr.a, wb = rr.a + (kind == 1 && wave >= cnt ? 0 : deg);
  if (kind == 0) { const int per = (deg + 3) / 4; wa = rr.a + wave * per; wb = min(rr.a + deg, wa + per); }
  // 64 entries per step: lane 16 g + j takes entry 4 j + g of the step -- ONE 256-byte load per array and step, where the first
  // version issued a 4-byte broadcast load per entry and lane group (its index loads were 112 of the long rows' 202 us) -- and
  // slot j of a lane group is a DPP row broadcast of lane j.  Group g still adds the entries = g (mod 4) in CSR order: the
  // same association as before, bit for bit.  The next step's entries load under this step's gathers.
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto ent_off = [&](int e0) { const int my = e0 + 4 * sub + g; return my < wb ? (unsigned)my * 4u : kOob; };
  int cn = ld_col(ent_off(wa));
  float wn = ld_val(ent_off(wa));
  for (int e0 = wa; e0 < wb; e0 += 64) {                 // (uniform per wave)
    const int ci1 = cn;
    const float wv1 = wn;
    cn = ld_col(ent_off(e0 + 64)); wn = ld_val(ent_off(e0 + 64));
    const int left = (wb - e0 - g + 3) >> 2;            // this group's entries from here on: slot j exists while j < left
    // (four slots at a time, more only where the step has them: all 16 gathers hoisted together spilled 8 registers)
    cb_slots1<WEIGHTED, FOLD, 4, 0>(hbuf, ld32, (unsigned)c, (dbg & 4) != 0, ci1, wv1, left, acc);
    if (wb - e0 > 16) {
      __builtin_amdgcn_sched_barrier(0);
      cb_slots1<WEIGHTED, FOLD, 8, 4>(hbuf, ld32, (unsigned)c, (dbg & 4) != 0, ci1, wv1, left, acc);
      if (wb - e0 > 32) {
        __builtin_amdgcn_sched_barrier(0);
        cb_slots1<WEIGHTED, FOLD, 12, 8>(hbuf, ld32, (unsigned)c, (dbg & 4) != 0, ci1, wv1, left, acc);
        __builtin_amdgcn_sched_barrier(0);
        cb_slots1<WEIGHTED, FOLD, 16, 12>(hbuf, ld32, (unsigned)c, (dbg & 4) != 0, ci1, wv1, left, acc);
      }
    }
  }
#pragma unroll
  for (int off = 16; off < 64; off <<= 1) {          // the wave's four groups, fixed order
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
    acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
  }
  const int row = row0 + (int)(rr.w & 0xFFFFu);
  if (kind == 1) {
    if (g == 0 && wave < cnt) epilogue(acc, row);
    return;
  }
  if (g == 0) s_long[wave][sub] = acc;                // kind 0: the four waves' quarters, in wave order
  __syncthreads();
  if (wave == 0 && g == 0) epilogue(f4_add(f4_add(s_long[0][sub], s_long[1][sub]), f4_add(s_long[2][sub], s_long[3][sub])), row);
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
                 int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int act,
                 const int2* chunk_list = nullptr, int list_len = 0, const FoldArgs* fold = nullptr, int list_rpc = kRowsPerChunk,
                 int hub_deg = 0, int out16 = 0) {
  const FoldArgs fo = fold ? *fold : FoldArgs{nullptr, nullptr, 0, 0, 0};
  if (out16) {   // (bf16 result rows: the one shape the bf16-storage path of large batches needs -- the caller has checked it)
    if constexpr (LPR == 64) {
      for (int col0 = 0; col0 < f; col0 += LPR * 4) {
        if (fold)
          hipLaunchKernelGGL((spmm_rows_kernel<64, true, kRowsPerChunkSmall, true, false, true>), dim3(list_len), dim3(256), 0, ctx->stream, rowptr,
                             colidx, vals, h, ldh, bias, out, ldo, n, f, col0, act, list_len, chunk_list, fo, 0);
        else
          hipLaunchKernelGGL((spmm_rows_kernel<64, true, kRowsPerChunkSmall, false, false, true>), dim3(list_len), dim3(256), 0, ctx->stream, rowptr,
                             colidx, vals, h, ldh, bias, out, ldo, n, f, col0, act, list_len, chunk_list, fo, 0);
      }
    }
    return;
  }
  const bool small = chunk_list ? list_rpc <= kRowsPerChunkSmall : n < 16 * 1024 * kRowsPerChunk / 4;   // < 128k rows
  const int nchunks = chunk_list ? list_len : gcnx_cdiv(n, small ? kRowsPerChunkSmall : kRowsPerChunk);
  const int span = LPR * 4;
  for (int col0 = 0; col0 < f; col0 += span) {
#define GCNX_ROWS_F(W, R, F, H)                                                                                      \
    hipLaunchKernelGGL((spmm_rows_kernel<LPR, W, R, F, H>), dim3(nchunks), dim3(256), 0, ctx->stream, rowptr, colidx, vals, \
                       h, ldh, bias, out, ldo, n, f, col0, act, nchunks, chunk_list, fo, hub_deg)
#define GCNX_ROWS(W, R) do { if (hub_deg > 0) { if (fold) GCNX_ROWS_F(W, R, true, true); else GCNX_ROWS_F(W, R, false, true); }     \
                             else { if (fold) GCNX_ROWS_F(W, R, true, false); else GCNX_ROWS_F(W, R, false, false); } } while (0)
    if (small) { if (vals) GCNX_ROWS(true, kRowsPerChunkSmall); else GCNX_ROWS(false, kRowsPerChunkSmall); }
    else { if (vals) GCNX_ROWS(true, kRowsPerChunk); else GCNX_ROWS(false, kRowsPerChunk); }
#undef GCNX_ROWS_F
#undef GCNX_ROWS
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

void dispatch_rows(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                   int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int act,
                   const int2* chunk_list, int list_len, const FoldArgs* fold = nullptr, int list_rpc = kRowsPerChunk, int hub_deg = 0,
                   int out16 = 0) {
  if (out16) { launch_rows<64>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, chunk_list, list_len, fold, list_rpc, 0, 1); return; }
  int lanes = f / 4;
  // Tuning knob (not part of the ABI contract): GCNX_SPMM_SLAB = column-slab width in floats
  // forces the lanes-per-row split of the rows kernel; results are identical.
  if (ctx->knob_spmm_slab >= 16 && ctx->knob_spmm_slab / 4 < lanes) lanes = ctx->knob_spmm_slab / 4;
  if (lanes > 32) launch_rows<64>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, chunk_list, list_len, fold, list_rpc, hub_deg);
  else if (lanes > 16) launch_rows<32>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, chunk_list, list_len, fold, list_rpc, hub_deg);
  else if (lanes > 8) launch_rows<16>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, chunk_list, list_len, fold, list_rpc, hub_deg);
  else if (lanes > 4) launch_rows<8>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, chunk_list, list_len, fold, list_rpc, hub_deg);
  else launch_rows<4>(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, chunk_list, list_len, fold, list_rpc, hub_deg);
}

template <int THREADS, int FT, int LPR>
int launch_duo(gcnx_ctx* ctx, const int32_t* rowptr, const RowRec* rowrec, const int32_t* colidx, const float* vals, const float* h,
               int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int act,
               const int2* graphs, int ngraphs, const DuoFold* fold = nullptr, int mode = kDuoPlain, int out16 = 0) {
  constexpr int lds_bytes = DuoShape<THREADS, FT>::LDSF * 4;
  static bool attr_set = false;
  if (!attr_set) {
#define GCNX_DUO_ATTR(W, M)                                                                                                  \
    GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_duo_kernel<THREADS, FT, LPR, W, M>),               \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes))
    GCNX_DUO_ATTR(true, kDuoPlain); GCNX_DUO_ATTR(false, kDuoPlain); GCNX_DUO_ATTR(true, kDuoFold); GCNX_DUO_ATTR(false, kDuoFold);
    GCNX_DUO_ATTR(true, kDuoBitsOut); GCNX_DUO_ATTR(false, kDuoBitsOut); GCNX_DUO_ATTR(true, kDuoFoldBits); GCNX_DUO_ATTR(false, kDuoFoldBits);
#undef GCNX_DUO_ATTR
    if constexpr (THREADS == 1024) {     // the pooled layer without its output (bits + pool partials)
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_duo_kernel<THREADS, FT, LPR, true, kDuoBitsPool>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_duo_kernel<THREADS, FT, LPR, false, kDuoBitsPool>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    }
    if constexpr (THREADS == 1024) {     // the bf16-result forms (weighted, plain / folded from the bit image)
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_duo_kernel<THREADS, FT, LPR, true, kDuoPlain, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_duo_kernel<THREADS, FT, LPR, true, kDuoFoldBits, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    }
    attr_set = true;
  }
  int dbg = 0;
#ifdef GCNX_TUNING   // timing-only ablation bits (results are WRONG when set, bit 16 excepted): only in a tuning build (make TUNING=1)
  if (const char* e = getenv("GCNX_SPMM_DBG")) dbg = atoi(e);
#endif
  int full = (THREADS == 512 ? 2 : 1) * ctx->num_cus;   // resident workgroups
  if (THREADS == 1024 && ctx->knob_spmm_tile_wgs > 0 && ctx->knob_spmm_tile_wgs < full) full = ctx->knob_spmm_tile_wgs;
  // column slabs per unit share one index burst; keep >= 3 units per workgroup so the static deal stays balanced
  // (config 3, measured per shape: 4 slabs per unit = 3.9 units per workgroup beats 2 slabs by 5 % and, on the
  // 1024-thread shape, 8 slabs = 1.9 units by 10 %)
  const int slabs = f / FT;
  int sg = 1;
  // (r3, one workgroup per CU: >= 1.5 units per workgroup is enough -- the snake deal pairs the tall graphs of the first
  // round with the short ones of the second -- and at the shard sizes of an 8-GPU run the larger slab groups win:
  // 200 graphs: 4 slabs per unit 80 us against 85 (2) and 107 (8: fewer units than CUs); 400 graphs: 8 slabs 153 against 156)
  const long long need = THREADS == 512 ? 6LL * full : 3LL * full;            // in half units
  for (int c = 8; c > 1; c >>= 1)
    if (slabs % c == 0 && 2LL * ngraphs * (slabs / c) >= need) { sg = c; break; }
  if (ctx->knob_spmm_sg >= 1 && slabs % ctx->knob_spmm_sg == 0) sg = ctx->knob_spmm_sg;
  const int upg = slabs / sg;
  const long long nunits = (long long)ngraphs * upg;
  if (nunits >= 2000000000LL) return gcnx_fail(ctx, GCNX_ERR_INVALID, "gcnx_spmm_csr: too many work units");
  const int grid = (int)(nunits < full ? nunits : full);
  const DuoFold nofold{nullptr, nullptr, 0, 0, nullptr};
  const DuoFold fo = fold ? *fold : nofold;
#define GCNX_DUO_LAUNCH(W, M)                                                                                                \
  hipLaunchKernelGGL((spmm_duo_kernel<THREADS, FT, LPR, W, M>), dim3(grid), dim3(THREADS), lds_bytes, ctx->stream, rowptr, rowrec, \
                     colidx, vals, h, ldh, bias, out, ldo, graphs, upg, sg, act, (int)nunits, n, dbg, fo)
  if (mode == kDuoBitsPool) {
    if constexpr (THREADS == 1024) {
      if (vals) GCNX_DUO_LAUNCH(true, kDuoBitsPool); else GCNX_DUO_LAUNCH(false, kDuoBitsPool);
    }
  } else if (out16) {
    if constexpr (THREADS == 1024) {
      if (mode == kDuoFoldBits)
        hipLaunchKernelGGL((spmm_duo_kernel<THREADS, FT, LPR, true, kDuoFoldBits, true>), dim3(grid), dim3(THREADS), lds_bytes, ctx->stream, rowptr,
                           rowrec, colidx, vals, h, ldh, bias, out, ldo, graphs, upg, sg, act, (int)nunits, n, dbg, fo);
      else
        hipLaunchKernelGGL((spmm_duo_kernel<THREADS, FT, LPR, true, kDuoPlain, true>), dim3(grid), dim3(THREADS), lds_bytes, ctx->stream, rowptr,
                           rowrec, colidx, vals, h, ldh, bias, out, ldo, graphs, upg, sg, act, (int)nunits, n, dbg, fo);
    }
  } else if (vals) {
    switch (mode) {
      case kDuoFold: GCNX_DUO_LAUNCH(true, kDuoFold); break;
      case kDuoBitsOut: GCNX_DUO_LAUNCH(true, kDuoBitsOut); break;
      case kDuoFoldBits: GCNX_DUO_LAUNCH(true, kDuoFoldBits); break;
      default: GCNX_DUO_LAUNCH(true, kDuoPlain); break;
    }
  } else {
    switch (mode) {
      case kDuoFold: GCNX_DUO_LAUNCH(false, kDuoFold); break;
      case kDuoBitsOut: GCNX_DUO_LAUNCH(false, kDuoBitsOut); break;
      case kDuoFoldBits: GCNX_DUO_LAUNCH(false, kDuoFoldBits); break;
      default: GCNX_DUO_LAUNCH(false, kDuoPlain); break;
    }
  }
#undef GCNX_DUO_LAUNCH
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // namespace

// What the diagonal-block structure of a disjoint batch buys: which graphs fit an LDS tile (and
// at which slab width), largest first so the work queue ends on small items; 32-row chunks of
// the graphs that do not, for the rows kernel.  Built once per batch; owned by the caller.
// A column-block graph (spmm_cb_kernel): too tall for a tile, at least kCbMinRows rows, local rows in 16 bits (RowRec)
static inline bool cb_graph(int ng, int cap2) { return ng >= kCbMinRows && ng > cap2 && ng < 65536; }

struct RowOrder {                       // the tile kernels' degree order of one rowptr (device pointer): see RowRec
  const int32_t* rowptr = nullptr;
  RowRec* dev = nullptr;
  int n = 0;
  // hub rows (more than kHubDeg entries), those of graphs too tall for a tile first: [0, nhubs_tall) / [0, nsegs_tall)
  HubSeg* hub_segs = nullptr;
  HubRow* hub_rows = nullptr;
  int nhubs = 0, nsegs = 0, nhubs_tall = 0, nsegs_tall = 0;
  int nhubs_cb = 0, nsegs_cb = 0;       // ... and among those, first, the hub rows of the column-block graphs (>= kCbMinRows rows)
  int nnz = 0;
  // work items of spmm_cb_kernel, per column-block count (f / 64), built on first use from the host copy of the degrees
  struct CbItems { int nblocks = 0, nitems = 0; int4* dev = nullptr; } cb_items[2];
  int ncb_sets = 0;
  std::vector<int> cb_deg;              // degrees of the column-block graphs' rows in RowRec order (host; empty: none)
};
struct gcnx_spmm_plan {
  std::vector<int32_t> bp;              // host copy of block_ptr
  RowOrder orders[4];                   // row orders bound so far (a normalised / unweighted / transposed view may share the plan)
  int norders = 0, next_evict = 0;
  int chunk_rpc = kRowsPerChunk;        // rows per chunk of the row-chunk list
  int nblocks = 0;
  int n1 = 0, n2 = 0, nchunks = 0;      // tier-1 graphs, tier-2 graphs, row chunks of larger graphs
  long long tile_rows = 0;
  int cap1 = kDuoCap32, cap2 = kSoloCap32;   // tier limits the lists were built for
  int2* dev = nullptr;                  // [n1 | n2 | nchunks] int2 records
  int32_t* gids = nullptr;              // graph index of the n1 + n2 tile records (the folded backward's dPooled row)
  int32_t* tall_gids = nullptr;         // the graphs taller than a tile (their rows go through the row chunks), ascending
  int ntall = 0;
  PipeItem* items = nullptr;            // pipelined kernel: graphs, tallest first: [n16 graphs of 625..1024 rows | n32 of <= 624]
  int nitems = 0, n16 = 0;
  int2* pipe_chunks = nullptr;          // ... and the 32-row chunks of the graphs too tall for it (> 1248 rows), for the rows kernel
  int npipe_chunks = 0;
  // column-block graphs (>= kCbMinRows rows; spmm_cb_kernel): their row chunks come FIRST in the row-chunk list
  // [dev + n1 + n2, + nchunks) -- nchunks_cb of them -- so that every user of that list keeps working; `rest` lists the
  // rows of all OTHER graphs as chunks (what the row gather takes when no tile kernel runs and the cb graphs go their own way)
  int ncb_graphs = 0, nchunks_cb = 0;
  long long cb_rows = 0;
  int32_t* cb_gids = nullptr;           // graph index of the k-th column-block graph (the folded backward's dPooled row)
  int2* rest = nullptr;
  int nrest = 0, rest_rpc = kRowsPerChunk;
};

#ifdef GCNX_TUNING
// Tuning builds only (not in include/gcnx.h): where spmm_rows_kernel leaves its per-wave lifetimes; NULL switches them off.
extern "C" __attribute__((visibility("default"))) int gcnx_tuning_wave_stamps(gcnx_ctx* ctx, unsigned long long* dev_buf) {
  GCNX_CHECK_CTX(ctx);
  GCNX_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(g_wave_stamps), &dev_buf, sizeof(dev_buf), 0, hipMemcpyHostToDevice, ctx->stream));
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GCNX_OK;
}
#endif

// Every device allocation of a plan, freed in one place (also on every failure path of gcnx_spmm_plan_create).
static void plan_free(gcnx_spmm_plan* p) {
  if (!p) return;
  if (p->dev) (void)hipFree(p->dev);
  if (p->gids) (void)hipFree(p->gids);
  if (p->tall_gids) (void)hipFree(p->tall_gids);
  if (p->items) (void)hipFree(p->items);
  if (p->pipe_chunks) (void)hipFree(p->pipe_chunks);
  if (p->rest) (void)hipFree(p->rest);
  if (p->cb_gids) (void)hipFree(p->cb_gids);
  for (int i = 0; i < p->norders; ++i) {
    for (int k = 0; k < p->orders[i].ncb_sets; ++k) if (p->orders[i].cb_items[k].dev) (void)hipFree(p->orders[i].cb_items[k].dev);
    if (p->orders[i].dev) (void)hipFree(p->orders[i].dev);
    if (p->orders[i].hub_segs) (void)hipFree(p->orders[i].hub_segs);
    if (p->orders[i].hub_rows) (void)hipFree(p->orders[i].hub_rows);
  }
  delete p;
}

template <class T>
static hipError_t plan_upload(gcnx_ctx* ctx, T** dev, const std::vector<T>& host) {
  *dev = nullptr;
  if (host.empty()) return hipSuccess;
  hipError_t e = hipMalloc((void**)dev, host.size() * sizeof(T));
  if (e == hipSuccess) e = hipMemcpyAsync(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  return e;
}

// The tile kernels' row order for one rowptr: per graph that fits a tile, its rows by decreasing degree (ties in row
// order: a counting sort), as RowRec records at the graph's own positions.  Built on the host from a copy of rowptr
// (synchronises; once per (plan, rowptr)).
static int plan_build_order(gcnx_ctx* ctx, gcnx_spmm_plan* p, const int32_t* rowptr, int32_t n, RowOrder* slot) {
  try {
    std::vector<int32_t> rp((size_t)n + 1);
    hipError_t e = hipMemcpyAsync(rp.data(), rowptr, rp.size() * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return gcnx_fail(ctx, GCNX_ERR_HIP, "gcnx_spmm_plan_bind: %s", hipGetErrorString(e));
    std::vector<RowRec> rec((size_t)n);
    std::vector<int> count, start;
    for (int g = 0; g < p->nblocks; ++g) {
      const int r0 = p->bp[g], ng = p->bp[g + 1] - p->bp[g];
      if (ng <= 0) continue;
      if (r0 < 0 || r0 + ng > n) return gcnx_fail(ctx, GCNX_ERR_DATA, "gcnx_spmm_plan_bind: block %d leaves the %d rows of this operator", g, n);
      const bool cbg = cb_graph(ng, p->cap2);
      if (ng > p->cap2 && !cbg) {                  // neither a tile graph nor a column-block graph: identity (never read)
        for (int i = 0; i < ng; ++i) rec[(size_t)r0 + i] = RowRec{rp[r0 + i], (unsigned)0};
        continue;
      }
      // degree order inside windows of `win` consecutive rows (win >= ng: the whole graph; always for a column-block graph)
      const int span = ng <= p->cap1 ? 128 : 256;             // rows per row group of the tier that takes this graph
      const int win = ctx->knob_spmm_sort_win > 0 && !cbg ? ctx->knob_spmm_sort_win * span : ng;
      for (int w0 = 0; w0 < ng; w0 += win) {
        const int wn = std::min(win, ng - w0);
        int dmax = 0;
        for (int i = w0; i < w0 + wn; ++i) {
          const int d = rp[r0 + i + 1] - rp[r0 + i];
          if (d < 0) return gcnx_fail(ctx, GCNX_ERR_DATA, "gcnx_spmm_plan_bind: rowptr decreases at row %d", r0 + i);
          dmax = std::max(dmax, d);
        }
        if (dmax > 65535) return gcnx_fail(ctx, GCNX_ERR_DATA, "gcnx_spmm_plan_bind: a row of a tile graph has %d entries", dmax);
        // (measured for the column-block graphs and dropped: only the rows of more than 8 entries in degree order, the short
        // rows behind them in ROW order so that records, entries and result rows are walked front to back -- 937 us against
        // 755 in full degree order: what a wave gains in locality it loses in uneven row lengths)
        auto key = [&](int d) { return d; };
        count.assign((size_t)dmax + 2, 0);
        for (int i = w0; i < w0 + wn; ++i) ++count[key(rp[r0 + i + 1] - rp[r0 + i])];
        start.assign((size_t)dmax + 2, 0);           // start[d] = first position of degree d: longer rows first
        int pos = w0;
        for (int d = dmax; d >= 0; --d) { start[d] = pos; pos += count[d]; }
        for (int i = w0; i < w0 + wn; ++i) {
          const int d = rp[r0 + i + 1] - rp[r0 + i];
          rec[(size_t)r0 + start[key(d)]++] = RowRec{rp[r0 + i], (unsigned)i | ((unsigned)d << 16)};
        }
      }
    }
    // the column-block graphs' degrees in RowRec order (what their work lists are cut from, per feature width)
    std::vector<int> cb_deg;
    for (int g = 0; g < p->nblocks; ++g) {
      const int r0 = p->bp[g], ng = p->bp[g + 1] - p->bp[g];
      if (!cb_graph(ng, p->cap2)) continue;
      for (int i = 0; i < ng; ++i) cb_deg.push_back((int)(rec[(size_t)r0 + i].w >> 16));
    }
    // hub rows as segments: rows of graphs too tall for a tile first (the tile kernels walk their own long rows)
    std::vector<HubSeg> segs;
    std::vector<HubRow> hubs;
    int nh_tall = 0, ns_tall = 0, nh_cb = 0, ns_cb = 0;
    for (int pass = 0; pass < 3; ++pass) {         // column-block graphs, other graphs too tall for a tile, tile graphs
      for (int g = 0; g < p->nblocks; ++g) {
        const int r0 = p->bp[g], ng = p->bp[g + 1] - p->bp[g];
        const int cls = cb_graph(ng, p->cap2) ? 0 : (ng > p->cap2 ? 1 : 2);
        if (cls != pass) continue;
        for (int i = 0; i < ng; ++i) {
          const int a = rp[r0 + i], b = rp[r0 + i + 1];
          if (b - a <= kHubDeg) continue;
          const int ns = (b - a + kHubSeg - 1) / kHubSeg;
          hubs.push_back(HubRow{r0 + i, (int)segs.size(), ns, 0});
          for (int k = 0; k < ns; ++k) segs.push_back(HubSeg{r0 + i, a + k * kHubSeg, std::min(b, a + (k + 1) * kHubSeg), (int)segs.size()});
        }
      }
      if (pass == 0) { nh_cb = (int)hubs.size(); ns_cb = (int)segs.size(); }
      if (pass == 1) { nh_tall = (int)hubs.size(); ns_tall = (int)segs.size(); }
    }
    RowRec* dev = nullptr;
    HubSeg* dsegs = nullptr;
    HubRow* dhubs = nullptr;
    e = plan_upload(ctx, &dev, rec);
    if (e == hipSuccess) e = plan_upload(ctx, &dsegs, segs);
    if (e == hipSuccess) e = plan_upload(ctx, &dhubs, hubs);
    if (e != hipSuccess) {
      if (dev) (void)hipFree(dev);
      if (dsegs) (void)hipFree(dsegs);
      if (dhubs) (void)hipFree(dhubs);
      return gcnx_fail(ctx, GCNX_ERR_HIP, "gcnx_spmm_plan_bind: %s", hipGetErrorString(e));
    }
    // (a captured graph may hold the old arrays in its kernel arguments -- e.g. a step captured on this rowptr before a
    // re-bind: like a workspace block that had to grow they are retired, not freed, while any graph is alive)
    for (void* old : {(void*)slot->dev, (void*)slot->hub_segs, (void*)slot->hub_rows}) {
      if (!old) continue;
      if (ctx->live_graphs > 0) ctx->retired_ws.push_back(old);
      else (void)hipFree(old);
    }
    slot->rowptr = rowptr; slot->dev = dev; slot->n = n;
    slot->hub_segs = dsegs; slot->hub_rows = dhubs;
    slot->nhubs = (int)hubs.size(); slot->nsegs = (int)segs.size(); slot->nhubs_tall = nh_tall; slot->nsegs_tall = ns_tall;
    slot->nhubs_cb = nh_cb; slot->nsegs_cb = ns_cb;
    for (int k = 0; k < slot->ncb_sets; ++k) {       // work lists cut from the old degrees go (retired under live graphs)
      if (!slot->cb_items[k].dev) continue;
      if (ctx->live_graphs > 0) ctx->retired_ws.push_back(slot->cb_items[k].dev); else (void)hipFree(slot->cb_items[k].dev);
      slot->cb_items[k] = RowOrder::CbItems();
    }
    slot->ncb_sets = 0;
    slot->cb_deg.swap(cb_deg);
    slot->nnz = rp[n];
  } catch (const std::bad_alloc&) {
    return gcnx_fail(ctx, GCNX_ERR_NOMEM, "gcnx_spmm_plan_bind: out of host memory");
  }
  return GCNX_OK;
}

// The order bound to `rowptr`, built on first use (outside stream capture).  force: rebuild (the CSR behind the pointer changed).
static int plan_order(gcnx_ctx* ctx, const gcnx_spmm_plan* cplan, const int32_t* rowptr, int32_t n, bool force, const RowRec** out,
                      const RowOrder** order_out = nullptr) {
  gcnx_spmm_plan* p = const_cast<gcnx_spmm_plan*>(cplan);   // (a cache behind an opaque handle; a ctx is single-threaded by contract)
  RowOrder* slot = nullptr;
  for (int i = 0; i < p->norders; ++i)
    if (p->orders[i].rowptr == rowptr && p->orders[i].n == n) slot = &p->orders[i];
  if (slot && !force) { *out = slot->dev; if (order_out) *order_out = slot; return GCNX_OK; }
  if (ctx->capturing)
    return gcnx_fail(ctx, GCNX_ERR_INVALID, "gcnx_spmm_csr: the plan has no row order for this rowptr yet and building one synchronises: "
                     "call gcnx_spmm_plan_bind (or run the call once) before capturing");
  if (!slot) {
    if (p->norders < 4) slot = &p->orders[p->norders++];
    else { slot = &p->orders[p->next_evict]; p->next_evict = (p->next_evict + 1) % 4; (void)hipStreamSynchronize(ctx->stream); }
  }
  const int rc = plan_build_order(ctx, p, rowptr, n, slot);
  if (rc) return rc;
  *out = slot->dev;
  if (order_out) *order_out = slot;
  return GCNX_OK;
}

// The hub rows of a bound plan (all of them, or only those of graphs too tall for a tile): segments -> partial rows in the
// ctx workspace -> combine + epilogue.
static int launch_hubs(gcnx_ctx* ctx, const RowOrder* od, bool tall_only, const int32_t* colidx, const float* vals, const float* h,
                       int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int act, const FoldArgs* fold,
                       bool skip_cb = false /* the column-block graphs' hub rows are computed elsewhere */) {
  const int seg0 = skip_cb ? od->nsegs_cb : 0, hub0 = skip_cb ? od->nhubs_cb : 0;
  const int nsegs = (tall_only ? od->nsegs_tall : od->nsegs) - seg0, nhubs = (tall_only ? od->nhubs_tall : od->nhubs) - hub0;
  if (nsegs <= 0) return GCNX_OK;
  int rc = gcnx_ws_reserve(ctx, (size_t)(seg0 + nsegs) * f * sizeof(float));       // (slots are absolute list positions)
  if (rc) return rc;
  float* part = (float*)ctx->ws;
  const FoldArgs fo = fold ? *fold : FoldArgs{nullptr, nullptr, 0, 0, 0};
#define GCNX_HUB_SEG(W, F) hipLaunchKernelGGL((spmm_hub_seg_kernel<W, F>), dim3(nsegs), dim3(256), 0, ctx->stream, od->hub_segs + seg0, colidx, vals, h, ldh, part, n, f, od->nnz)
  if (vals) { if (fold) GCNX_HUB_SEG(true, true); else GCNX_HUB_SEG(true, false); }
  else { if (fold) GCNX_HUB_SEG(false, true); else GCNX_HUB_SEG(false, false); }
#undef GCNX_HUB_SEG
  GCNX_LAUNCH_OK(ctx);
  if (fold) hipLaunchKernelGGL((spmm_hub_combine_kernel<true>), dim3(nhubs), dim3(64), 0, ctx->stream, od->hub_rows + hub0, (const float*)part, bias, out, ldo, f, act, fo);
  else hipLaunchKernelGGL((spmm_hub_combine_kernel<false>), dim3(nhubs), dim3(64), 0, ctx->stream, od->hub_rows + hub0, (const float*)part, bias, out, ldo, f, act, fo);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

// Work items of spmm_cb_kernel for f = 64 * nblk columns: per column-block graph, per column block, its rows in degree order
// cut into kind-0 / kind-1 / kind-2 items (see the kernel) -- in that order, so that one XCD (a contiguous range of the list)
// walks a graph block by block, heaviest rows first.  Built on first use per block count (uploads: not inside a capture),
// two counts kept per bound row order.
static int plan_cb_items(gcnx_ctx* ctx, const gcnx_spmm_plan* p, const RowOrder* corder, int nblk, const int4** out, int* count) {
  RowOrder* od = const_cast<RowOrder*>(corder);
  for (int i = 0; i < od->ncb_sets; ++i)
    if (od->cb_items[i].nblocks == nblk) { *out = od->cb_items[i].dev; *count = od->cb_items[i].nitems; return GCNX_OK; }
  if (ctx->capturing)
    return gcnx_fail(ctx, GCNX_ERR_INVALID, "gcnx_spmm_csr: the plan has no column-block work list for this width yet and building one "
                     "uploads: run the call once before capturing");
  try {
    std::vector<int4> items;
    size_t at = 0;                                             // position in od->cb_deg
    int kcb = -1;                                              // index among the column-block graphs (item.z >> 16; plan->cb_gids)
    for (int g = 0; g < p->nblocks; ++g) {
      const int r0 = p->bp[g], ng = p->bp[g + 1] - p->bp[g];
      if (!cb_graph(ng, p->cap2)) continue;
      ++kcb;
      const int* deg = od->cb_deg.data() + at;                 // descending
      at += (size_t)ng;
      int n0 = 0, n1 = 0;                                      // rows of kind 0, of kind 0 or 1
      while (n0 < ng && deg[n0] > kCbHub) ++n0;
      n1 = n0;
      while (n1 < ng && deg[n1] > kCbShort) ++n1;
      int kinds = 15;
#ifdef GCNX_TUNING   // timing only: GCNX_SPMM_CB = 16 + mask launches only the kinds in mask (results are then incomplete)
      if (ctx->knob_spmm_cb >= 16) kinds = ctx->knob_spmm_cb & 15;
#endif
      std::vector<int4> heavy, light;
      for (int b = 0; b < nblk; ++b) {
        heavy.clear(); light.clear();
        if (kinds & 1) for (int q = 0; q < n0; ++q) heavy.push_back(make_int4(r0 + q, 1 | (0 << 16), b * kCbCols | (kcb << 16), r0));
        if (kinds & 2) for (int q = n0; q < n1; q += 4) heavy.push_back(make_int4(r0 + q, std::min(4, n1 - q) | (1 << 16), b * kCbCols | (kcb << 16), r0));
        if (kinds & 8) for (int q = n1; q < ng; q += 16 * kCbRpg) light.push_back(make_int4(r0 + q, std::min(16 * kCbRpg, ng - q) | (3 << 16), b * kCbCols | (kcb << 16), r0));
        // The long rows' items gather out of L2 (the block's source rows are re-read ~degree times), the short rows' items stream
        // from HBM: dealt evenly among each other (heaviest first on either side) the two kinds of traffic run side by side.
        if (ctx->knob_spmm_cb == 2 || heavy.empty() || light.empty()) {      // (2: heaviest first, as the first version had it)
          items.insert(items.end(), heavy.begin(), heavy.end());
          items.insert(items.end(), light.begin(), light.end());
        } else {
          size_t hi = 0, li = 0;
          const size_t nh = heavy.size(), nl = light.size();
          while (hi < nh || li < nl) {
            // next heavy item when its share of the list is behind the light items' share
            if (hi < nh && (li >= nl || hi * nl <= li * nh)) items.push_back(heavy[hi++]);
            else items.push_back(light[li++]);
          }
        }
      }
    }
    int4* dev = nullptr;
    const hipError_t e = plan_upload(ctx, &dev, items);
    if (e != hipSuccess) return gcnx_fail(ctx, GCNX_ERR_HIP, "gcnx_spmm_csr (column-block list): %s", hipGetErrorString(e));
    RowOrder::CbItems* set = od->ncb_sets < 2 ? &od->cb_items[od->ncb_sets++] : &od->cb_items[0];
    if (set->dev) {                                            // (a third width: the first set goes -- retired under live graphs)
      if (ctx->live_graphs > 0) ctx->retired_ws.push_back(set->dev); else (void)hipFree(set->dev);
    }
    set->nblocks = nblk; set->nitems = (int)items.size(); set->dev = dev;
    *out = dev; *count = set->nitems;
  } catch (const std::bad_alloc&) {
    return gcnx_fail(ctx, GCNX_ERR_NOMEM, "gcnx_spmm_csr: out of host memory");
  }
  return GCNX_OK;
}

// Whether the column-block graphs of `plan` go through spmm_cb_kernel for this call (else: the row gather + hub segments).
static bool cb_path_ok(const gcnx_ctx* ctx, const gcnx_spmm_plan* plan, int32_t n, int32_t f, int64_t ldh, int out16) {
  return plan && plan->ncb_graphs > 0 && plan->ncb_graphs < 32768 && ctx->knob_spmm_cb != 0 && !out16 && f >= kCbMinF && f % kCbCols == 0 &&
         f <= 32768 && (uint64_t)n * (uint64_t)ldh * 4u < 0xFFFFFF00ull;      // (an item packs column and graph index into 16 bits each)
}

static int launch_cb(gcnx_ctx* ctx, const gcnx_spmm_plan* plan, const RowOrder* order, const int32_t* colidx, const float* vals, const float* h,
                     int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int act, const FoldArgs* fold = nullptr) {
  const int4* items = nullptr;
  int nitems = 0;
  const int rc = plan_cb_items(ctx, plan, order, f / kCbCols, &items, &nitems);
  if (rc) return rc;
  if (nitems == 0) return GCNX_OK;
  int dbg = 0, pad_lds = 0;
#ifdef GCNX_TUNING   // timing-only ablation bits (results are WRONG when set): see scripts/cb_ablate.sh; GCNX_CB_LDS: bytes of unused
  if (const char* e = getenv("GCNX_CB_DBG")) dbg = atoi(e);        // dynamic LDS per workgroup (caps the workgroups per CU)
  if (const char* e = getenv("GCNX_CB_LDS")) pad_lds = atoi(e);
#endif
  const FoldArgs fo = fold ? *fold : FoldArgs{nullptr, nullptr, 0, 0, 0};
#define GCNX_CB_LAUNCH(W, F)                                                                                                       \
  hipLaunchKernelGGL((spmm_cb_kernel<W, F>), dim3(nitems), dim3(256), pad_lds, ctx->stream, (const RowRec*)order->dev, colidx, vals, h, ldh, \
                     bias, out, ldo, n, order->nnz, act, nitems, items, dbg, fo, (const int32_t*)plan->cb_gids)
  if (vals) { if (fold) GCNX_CB_LAUNCH(true, true); else GCNX_CB_LAUNCH(true, false); }
  else { if (fold) GCNX_CB_LAUNCH(false, true); else GCNX_CB_LAUNCH(false, false); }
#undef GCNX_CB_LAUNCH
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

extern "C" {

// Balanced deal of the tile graphs (r3).  spmm_duo_kernel deals units statically: round r of (virtual) workgroup v is list
// position r W + (r odd ? W - 1 - v : v).  Over the plain size-sorted list that snake leaves the busiest workgroup of
// config 3 with 1.10 x the mean row count (1 599 graphs on 256 workgroups: 6.2 each, and the sizes are not linear in the
// rank) -- the launch ends when THAT workgroup does.  Here the same positions are filled differently: every workgroup's
// unit COUNT stays what the snake gives it, the graphs go largest-first to the least-loaded workgroup that still has a
// free position (LPT), a few exchange passes between the busiest workgroup and the others follow, and each workgroup
// walks its graphs in descending size (single-buffered tiles first, then the double-buffered ones, so the prefetch chain
// is not broken).  Cost of a graph = rows (x `big` % for single-buffered tiles) + c0 rows for the index burst.  Results
// are unchanged (a unit is computed the same way wherever it runs).  Applies when one unit is one graph (upg == 1: enough
// graphs for 1.5 units per workgroup at <= 8 slabs); other shapes keep a valid, if not balanced, deal.
static void balance_tile_list(std::vector<int2>& t, int W, int c0, int big_pct, int dbl_cap) {
  const int n = (int)t.size();
  if (W <= 1 || n <= W) return;
  const int R = (n + W - 1) / W;
  auto pos_of = [&](int v, int r) { return r * W + ((r & 1) ? W - 1 - v : v); };
  std::vector<int> cap(W, 0);
  for (int v = 0; v < W; ++v)
    for (int r = 0; r < R; ++r) cap[v] += pos_of(v, r) < n ? 1 : 0;
  auto cost = [&](const int2& g) { return (long long)g.y * (g.y > dbl_cap ? big_pct : 100) + 100LL * c0; };
  std::vector<std::vector<int>> own(W);
  std::vector<long long> load(W, 0);
  typedef std::pair<long long, int> Key;                       // (load, workgroup): ties to the lower id -- deterministic
  std::priority_queue<Key, std::vector<Key>, std::greater<Key>> heap;
  for (int v = 0; v < W; ++v) heap.push(Key(0, v));
  for (int i = 0; i < n; ++i) {                                // t is sorted by size, descending
    const Key k = heap.top(); heap.pop();
    const int v = k.second;
    own[v].push_back(i); load[v] += cost(t[i]);
    if ((int)own[v].size() < cap[v]) heap.push(Key(load[v], v));
  }
  std::vector<int> light(W);
  for (int pass = 0; pass < 2 * W; ++pass) {                   // exchanges: one graph of the busiest workgroup against a smaller
    int vmax = 0;                                              // one of the 16 least loaded (host time: well under a millisecond)
    for (int v = 1; v < W; ++v) if (load[v] > load[vmax]) vmax = v;
    for (int v = 0; v < W; ++v) light[v] = v;
    const int nl = std::min(W, 16);
    std::partial_sort(light.begin(), light.begin() + nl, light.end(),
                      [&](int a, int b) { return load[a] != load[b] ? load[a] < load[b] : a < b; });
    long long best = load[vmax]; int bo = -1, bi = -1, bj = -1;
    for (int q = 0; q < nl; ++q) {
      const int o = light[q];
      if (o == vmax || load[o] >= load[vmax]) continue;
      for (int i : own[vmax]) for (int j : own[o]) {
        const long long d = cost(t[i]) - cost(t[j]);
        if (d <= 0) continue;
        const long long m = std::max(load[vmax] - d, load[o] + d);
        if (m < best) { best = m; bo = o; bi = i; bj = j; }
      }
    }
    if (bo < 0) break;
    const long long d = cost(t[bi]) - cost(t[bj]);
    *std::find(own[vmax].begin(), own[vmax].end(), bi) = bj;
    *std::find(own[bo].begin(), own[bo].end(), bj) = bi;
    load[vmax] -= d; load[bo] += d;
  }
  std::vector<int2> out(t.size());
  for (int v = 0; v < W; ++v) {
    std::sort(own[v].begin(), own[v].end());                   // list index ascending = size descending
    for (int r = 0; r < (int)own[v].size(); ++r) out[pos_of(v, r)] = t[own[v][r]];
  }
  t.swap(out);
}

int gcnx_spmm_plan_create(gcnx_ctx* ctx, const int32_t* block_ptr, int32_t nblocks, gcnx_spmm_plan** out) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, out != nullptr, "gcnx_spmm_plan_create: out is NULL");
  *out = nullptr;
  GCNX_REQUIRE(ctx, nblocks >= 0 && (nblocks == 0 || block_ptr), "gcnx_spmm_plan_create: bad block list");
  GCNX_REQUIRE(ctx, !ctx->capturing, "gcnx_spmm_plan_create synchronises and cannot be captured");
  gcnx_spmm_plan* p = nullptr;
  try {
    p = new gcnx_spmm_plan();
    std::vector<int32_t>& bp = p->bp;
    bp.assign((size_t)nblocks + 1, 0);
    if (nblocks > 0) {
      hipError_t e = hipMemcpyAsync(bp.data(), block_ptr, bp.size() * 4, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) { plan_free(p); return gcnx_fail(ctx, GCNX_ERR_HIP, "gcnx_spmm_plan_create: %s", hipGetErrorString(e)); }
    }
    std::vector<int2> t1, t2, ch;
    std::vector<std::pair<int, int>> by_row;       // (row0, graph index) of the tile graphs
    long long tile_rows = 0;
    const int cap1 = ctx->knob_spmm_cap1 >= 0 ? std::min(ctx->knob_spmm_cap1, kDuoCap32) : kDuoCap32, cap2 = kSoloCap32;
    // graphs taller than any tile go to the row gather as plan-listed chunks.  8-row chunks (r3): the gather re-reads a
    // graph's feature rows ~degree times and only the XCD's 4 MiB L2 can serve that; with 32-row chunks 7 workgroups per CU
    // x 32 CUs hold ~7 000 rows = 4-5 such graphs (2 MB of features each) in flight per XCD and the re-reads went to HBM
    // (r2 PMC: 496 MB fetched for 116 MB of rows); 8-row chunks keep it to about one graph.
    const int rpc = ctx->knob_spmm_tall_rpc == 32 ? kRowsPerChunk : kRowsPerChunkSmall;
    p->chunk_rpc = rpc;
    std::vector<int2> ch_cb, rest;
    std::vector<int32_t> cb_gid_list;
    long long rest_rows = 0;
    for (int g = 0; g < nblocks; ++g) {
      const int ng = bp[g + 1] - bp[g];
      if (ng < 0) { plan_free(p); return gcnx_fail(ctx, GCNX_ERR_DATA, "gcnx_spmm_plan_create: block_ptr is not non-decreasing at %d", g); }
      if (!cb_graph(ng, cap2)) rest_rows += ng;
    }
    p->rest_rpc = rest_rows < 16 * 1024 * kRowsPerChunk / 4 ? kRowsPerChunkSmall : kRowsPerChunk;   // (launch_rows' own rule)
    for (int g = 0; g < nblocks; ++g) {
      const int r0 = bp[g], ng = bp[g + 1] - bp[g];
      if (ng == 0) continue;
      const bool cbg = cb_graph(ng, cap2);                      // a column-block graph (spmm_cb_kernel)
      if (ng <= cap1) { t1.push_back(make_int2(r0, ng)); tile_rows += ng; by_row.emplace_back(r0, g); }
      else if (ng <= cap2) { t2.push_back(make_int2(r0, ng)); tile_rows += ng; by_row.emplace_back(r0, g); }
      else for (int r = r0; r < r0 + ng; r += rpc) (cbg ? ch_cb : ch).push_back(make_int2(r, std::min(r + rpc, r0 + ng)));
      if (cbg) { p->ncb_graphs++; p->cb_rows += ng; cb_gid_list.push_back(g); }
      else for (int r = r0; r < r0 + ng; r += p->rest_rpc) rest.push_back(make_int2(r, std::min(r + p->rest_rpc, r0 + ng)));
    }
    p->nchunks_cb = (int)ch_cb.size();
    ch.insert(ch.begin(), ch_cb.begin(), ch_cb.end());           // the column-block graphs' chunks first
    if (p->ncb_graphs == 0) rest.clear();                        // (only read next to spmm_cb_kernel)
    p->nrest = (int)rest.size();
    auto by_size = [](const int2& x, const int2& y) { return x.y != y.y ? x.y > y.y : x.x < y.x; };
    std::sort(t1.begin(), t1.end(), by_size);
    std::sort(t2.begin(), t2.end(), by_size);
    if (ctx->knob_spmm_bal > 0 && 2 * (long long)t2.size() >= 3LL * ctx->num_cus) {
      const int c0 = ctx->knob_spmm_bal == 1 ? 60 : ctx->knob_spmm_bal % 1000;
      const int big = ctx->knob_spmm_bal == 1 ? 110 : 100 + ctx->knob_spmm_bal / 1000;
      const int wgs = ctx->knob_spmm_tile_wgs > 0 && ctx->knob_spmm_tile_wgs < ctx->num_cus ? ctx->knob_spmm_tile_wgs : ctx->num_cus;
      balance_tile_list(t2, wgs, c0, big, 624);
    }
    std::vector<int32_t> gids;                     // by_row is sorted by row0 (graphs come in row order): look the records up
    gids.reserve(t1.size() + t2.size());
    for (const std::vector<int2>* tv : {&t1, &t2})
      for (const int2& rec : *tv)
        gids.push_back(std::lower_bound(by_row.begin(), by_row.end(), std::make_pair(rec.x, 0))->second);
    p->nblocks = nblocks;
    p->n1 = (int)t1.size(); p->n2 = (int)t2.size(); p->nchunks = (int)ch.size();
    p->tile_rows = tile_rows;
    p->cap1 = cap1; p->cap2 = cap2;
    std::vector<int2> all;
    all.reserve(t1.size() + t2.size() + ch.size());
    all.insert(all.end(), t1.begin(), t1.end());
    all.insert(all.end(), t2.begin(), t2.end());
    all.insert(all.end(), ch.begin(), ch.end());
    // work items of the pipelined kernel: every graph of up to 1248 rows (costliest first); taller ones as 32-row chunks
    // for the rows kernel
    std::vector<PipeItem> items;
    std::vector<int2> tall;
    for (int g = 0; g < nblocks; ++g) {
      const int r0 = bp[g], ng = bp[g + 1] - bp[g];
      if (ng <= 0) continue;
      if (ng <= kPipeCap16) items.push_back(PipeItem{r0, ng});
      else for (int r = r0; r < r0 + ng; r += kRowsPerChunk) tall.push_back(make_int2(r, std::min(r + kRowsPerChunk, r0 + ng)));
    }
    std::sort(items.begin(), items.end(), [&](const PipeItem& a, const PipeItem& b) { return a.ng != b.ng ? a.ng > b.ng : a.row0 < b.row0; });
    p->nitems = (int)items.size();
    for (const PipeItem& it : items) p->n16 += it.ng > kPipeCap32 ? 1 : 0;    // sorted by size: they come first
    p->npipe_chunks = (int)tall.size();
    std::vector<int32_t> tall_gids;
    for (int g = 0; g < nblocks; ++g) if (bp[g + 1] - bp[g] > cap2) tall_gids.push_back(g);
    p->ntall = (int)tall_gids.size();
    hipError_t e = plan_upload(ctx, &p->dev, all);
    if (e == hipSuccess) e = plan_upload(ctx, &p->gids, gids);
    if (e == hipSuccess) e = plan_upload(ctx, &p->tall_gids, tall_gids);
    if (e == hipSuccess) e = plan_upload(ctx, &p->pipe_chunks, tall);
    if (e == hipSuccess) e = plan_upload(ctx, &p->items, items);
    if (e == hipSuccess) e = plan_upload(ctx, &p->rest, rest);
    if (e == hipSuccess) e = plan_upload(ctx, &p->cb_gids, cb_gid_list);
    if (e != hipSuccess) {
      plan_free(p);
      return gcnx_fail(ctx, e == hipErrorOutOfMemory ? GCNX_ERR_NOMEM : GCNX_ERR_HIP, "gcnx_spmm_plan_create: %s", hipGetErrorString(e));
    }
    *out = p;
  } catch (const std::bad_alloc&) {
    plan_free(p);
    return gcnx_fail(ctx, GCNX_ERR_NOMEM, "gcnx_spmm_plan_create: out of host memory");
  }
  return GCNX_OK;
}

int gcnx_spmm_plan_destroy(gcnx_ctx* ctx, gcnx_spmm_plan* plan) {
  GCNX_CHECK_CTX(ctx);
  if (!plan) return GCNX_OK;
  (void)hipStreamSynchronize(ctx->stream);
  plan_free(plan);
  return GCNX_OK;
}

int gcnx_spmm_plan_bind(gcnx_ctx* ctx, gcnx_spmm_plan* plan, const int32_t* rowptr, int32_t n) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, plan && rowptr && n >= 0, "gcnx_spmm_plan_bind: NULL plan / rowptr");
  const RowRec* unused = nullptr;
  // inside a capture: an order already bound to this rowptr stands (nothing can be rebuilt here, and a captured sequence
  // that binds is a sequence that ran eagerly first); one that is missing fails in plan_order with its own message
  return plan_order(ctx, plan, rowptr, n, !ctx->capturing, &unused);
}

static int spmm_csr_impl(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                         int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int act,
                         const gcnx_spmm_plan* plan, uint32_t* relu_bits, int out16 = 0);

// The conditions of the bf16-result forms (gcnx_spmm_csr_bf16out / _pool_bwd_bf16out): weighted operator, every tile
// graph on the 1024-thread shape, taller graphs as 8-row chunks, no hub rows.
static bool out16_shape_ok(gcnx_ctx* ctx, const gcnx_spmm_plan* plan, const float* vals, int32_t f, int64_t ldo, const void* out) {
  return plan && vals && f % kSlab == 0 && f > 128 && ldo % 4 == 0 && aligned16(out) && plan->n1 == 0 && plan->n2 > 0 && plan->ncb_graphs == 0 &&
         (plan->nchunks == 0 || plan->chunk_rpc == kRowsPerChunkSmall) && ctx->knob_spmm_kernel != 1 && ctx->knob_spmm_kernel != 3 &&
         (long long)(2 * plan->n2) * (f / kSlab) >= 4LL * ctx->num_cus;
}

int gcnx_spmm_csr_bf16out(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                          int64_t ldh, const float* bias, void* out16, int64_t ldo, int32_t n, int32_t f, int act,
                          const gcnx_spmm_plan* plan) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, out16 != nullptr || n == 0 || f == 0, "gcnx_spmm_csr_bf16out: NULL output");
  if (!out16_shape_ok(ctx, plan, vals, f, ldo, out16)) return GCNX_ERR_UNSUPPORTED;       // (an answer, not a failure: no message)
  return spmm_csr_impl(ctx, rowptr, colidx, vals, h, ldh, bias, (float*)out16, ldo, n, f, act, plan, nullptr, 1);
}

int gcnx_spmm_csr(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                  int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int act,
                  const gcnx_spmm_plan* plan) {
  return spmm_csr_impl(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, plan, nullptr);
}

// pooled[g][c] (/ n_g for AVG) and cnt[g][c] of tile graph t = gids[t]: its 16 waves' partial rows, in wave order
__global__ __launch_bounds__(256) void pool_parts_reduce_kernel(const float* __restrict__ ppart, const float* __restrict__ cpart,
                                                                const int32_t* __restrict__ gids, const int32_t* __restrict__ gp,
                                                                int ntile, int32_t f, int avg, float* __restrict__ pooled, int64_t ldp,
                                                                float* __restrict__ cnt) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (int64_t)ntile * f) return;
  const int t = (int)(i / f), c = (int)(i - (int64_t)t * f);
  const float* pp = ppart + (size_t)t * 16 * f + c;
  const float* cp = cpart + (size_t)t * 16 * f + c;
  float4 a = *reinterpret_cast<const float4*>(pp), k = *reinterpret_cast<const float4*>(cp);
#pragma unroll
  for (int w = 1; w < 16; ++w) {
    const float4 v = *reinterpret_cast<const float4*>(pp + (size_t)w * f), q = *reinterpret_cast<const float4*>(cp + (size_t)w * f);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    k.x += q.x; k.y += q.y; k.z += q.z; k.w += q.w;
  }
  const int g = gids[t];
  if (avg) {
    const float inv = (float)max(gp[g + 1] - gp[g], 1);
    a.x /= inv; a.y /= inv; a.z /= inv; a.w /= inv;
  }
  *reinterpret_cast<float4*>(pooled + (int64_t)g * ldp + c) = a;
  *reinterpret_cast<float4*>(cnt + (int64_t)g * f + c) = k;
}

// The pooled GCNConv's forward for a training step on a tile plan WITHOUT its output (see kDuoBitsPool): relu_bits, the
// graphs' pooled rows and positive counts.  Graphs taller than a tile keep the old form -- their rows of `out` are written
// by the row chunks (the folded backward gathers them) and pooled by the pool kernel over the plan's list of such graphs.
// Rows of tile graphs in `out` are NOT written.
int gcnx_spmm_csr_relu_bits_pool(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                                 int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f,
                                 const gcnx_spmm_plan* plan, void* relu_bits, const int32_t* graph_ptr, int32_t b, int pool_mode,
                                 float* pooled, int64_t ldp, float* cnt) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "aggregation + pool (GCNConv under the global pool)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0 && b >= 0, "gcnx_spmm_csr_relu_bits_pool: negative size");
  GCNX_REQUIRE(ctx, pool_mode == GCNX_POOL_SUM || pool_mode == GCNX_POOL_AVG, "gcnx_spmm_csr_relu_bits_pool: SUM / AVG pooling only");
  if (n == 0 || f == 0 || b == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, rowptr && colidx && h && out && relu_bits && graph_ptr && pooled && cnt, "gcnx_spmm_csr_relu_bits_pool: NULL pointer");
  GCNX_REQUIRE(ctx, ldh >= f && ldo >= f && ldp >= f, "gcnx_spmm_csr_relu_bits_pool: leading dimension smaller than f=%d", f);
  const bool ok = plan && plan->nblocks == b && plan->n1 == 0 && plan->n2 > 0 && f % kSlab == 0 && f % 4 == 0 && ldh % 4 == 0 && ldo % 4 == 0 &&
                  ldp % 4 == 0 && aligned16(h) && aligned16(out) && aligned16(pooled) && aligned16(cnt) && (!bias || aligned16(bias)) &&
                  (reinterpret_cast<uintptr_t>(relu_bits) & 3) == 0 && ctx->knob_spmm_kernel != 1 && ctx->knob_spmm_kernel != 3 &&
                  (long long)(2 * plan->n2) * (f / kSlab) >= 4LL * ctx->num_cus;
  if (!ok) return GCNX_ERR_UNSUPPORTED;                   // (an answer: gcnx_spmm_csr_relu_bits + the pool then)
  const RowRec* rowrec = nullptr;
  const RowOrder* order = nullptr;
  { const int rb = plan_order(ctx, plan, rowptr, n, false, &rowrec, &order); if (rb) return rb; }
  if (order && order->nsegs_tall > 0) return GCNX_ERR_UNSUPPORTED;
  const size_t part_floats = (size_t)plan->n2 * 16 * f;
  int rc = gcnx_ws_reserve(ctx, 2 * part_floats * sizeof(float));
  if (rc) return rc;
  float* ppart = (float*)ctx->ws;
  float* cpart = ppart + part_floats;
  const DuoFold bo{nullptr, nullptr, 0, 0, (uint32_t*)relu_bits, ppart, cpart};
  rc = launch_duo<1024, 32, 4>(ctx, rowptr, rowrec, colidx, vals, h, ldh, bias, out, ldo, n, f, GCNX_ACT_RELU, plan->dev + plan->n1, plan->n2, &bo,
                               kDuoBitsPool);
  if (rc) return rc;
  hipLaunchKernelGGL(pool_parts_reduce_kernel, dim3(gcnx_cdiv((long long)plan->n2 * f / 4, 256)), dim3(256), 0, ctx->stream, (const float*)ppart,
                     (const float*)cpart, (const int32_t*)(plan->gids + plan->n1), graph_ptr, plan->n2, f, pool_mode == GCNX_POOL_AVG ? 1 : 0,
                     pooled, ldp, cnt);
  GCNX_LAUNCH_OK(ctx);
  if (plan->nchunks > 0) {      // graphs taller than a tile: rows of `out` from the row chunks, then the pool over those graphs
    dispatch_rows(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, GCNX_ACT_RELU, plan->dev + plan->n1 + plan->n2, plan->nchunks, nullptr,
                  plan->chunk_rpc, 0);
    GCNX_LAUNCH_OK(ctx);
    rc = gcnx_pool_graph_list(ctx, graph_ptr, plan->tall_gids, plan->ntall, out, ldo, f, pool_mode, pooled, ldp, cnt);
    if (rc) return rc;
  }
  return GCNX_OK;
}

int gcnx_spmm_csr_relu_bits(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                            int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f,
                            const gcnx_spmm_plan* plan, void* relu_bits) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, relu_bits && (reinterpret_cast<uintptr_t>(relu_bits) & 3) == 0, "gcnx_spmm_csr_relu_bits: relu_bits must be a 4-byte aligned buffer");
  return spmm_csr_impl(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, GCNX_ACT_RELU, plan, (uint32_t*)relu_bits);
}

static int spmm_csr_impl(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* h,
                         int64_t ldh, const float* bias, float* out, int64_t ldo, int32_t n, int32_t f, int act,
                         const gcnx_spmm_plan* plan, uint32_t* relu_bits, int out16) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "aggregation (GCNConv / GeneralConv SpMM)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0, "gcnx_spmm_csr: negative size");
  GCNX_REQUIRE(ctx, act == GCNX_ACT_NONE || act == GCNX_ACT_RELU, "gcnx_spmm_csr: activation %d not supported here", act);
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, rowptr && colidx && h && out, "gcnx_spmm_csr: NULL pointer");
  GCNX_REQUIRE(ctx, ldh >= f && ldo >= f, "gcnx_spmm_csr: leading dimension smaller than f=%d", f);
  GCNX_REQUIRE(ctx, h != out, "gcnx_spmm_csr: in-place aggregation is not possible");
  const bool vec = (f % 4 == 0) && (ldh % 4 == 0) && (ldo % 4 == 0) && aligned16(h) && aligned16(out) &&
                   (!bias || aligned16(bias));
  if (out16 && !vec) return GCNX_ERR_UNSUPPORTED;
  if (!vec) {
    hipLaunchKernelGGL(spmm_scalar_kernel, dim3(gcnx_cdiv(n, 4)), dim3(256), 0, ctx->stream, rowptr, colidx, vals,
                       h, ldh, bias, out, ldo, n, f, act);
    GCNX_LAUNCH_OK(ctx);
    return GCNX_OK;
  }
  const int force = ctx->knob_spmm_kernel;   // tuning knob GCNX_SPMM_KERNEL / gcnx_set_tuning: 1 rows, 2 tile (tiers), 3 pipe
  // The tile kernel is a throughput design (one item per CU at a time): it needs a few items
  // per CU to fill the chip, otherwise the rows kernel's finer decomposition wins.
  bool tiles = plan && f % kSlab == 0 && (long long)(plan->n1 + 2 * plan->n2) * (f / kSlab) >= 4LL * ctx->num_cus;
  if (force == 1) tiles = false;
  if (force >= 2 && plan && f % kSlab == 0) tiles = true;
  if (relu_bits && (!tiles || force == 3))
    return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_spmm_csr_relu_bits: the bit image is written by the tile kernels only "
                     "(needs a plan with enough tile units and f %% 32 == 0): use gcnx_spmm_csr");
  const RowRec* rowrec = nullptr;
  const RowOrder* order = nullptr;
  if (plan && force != 3) { const int rb = plan_order(ctx, plan, rowptr, n, false, &rowrec, &order); if (rb) return rb; }
  if (out16 && (!tiles || (order && order->nsegs_tall > 0))) return GCNX_ERR_UNSUPPORTED;     // (checked by the caller; hub rows: fp32 only)
  const bool cb = cb_path_ok(ctx, plan, n, f, ldh, out16) && order != nullptr;
  if (!tiles) {
    if (cb) {
      // graphs too large for an XCD's L2: column blocks (spmm_cb_kernel, hub rows included); every other graph's rows
      // as plan-listed chunks on the row gather, their hub rows as segments
      if (plan->nrest > 0) {
        const bool hubs = order->nsegs > order->nsegs_cb;
        dispatch_rows(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, plan->rest, plan->nrest, nullptr, plan->rest_rpc, hubs ? kHubDeg : 0);
        GCNX_LAUNCH_OK(ctx);
        if (hubs) { const int rh = launch_hubs(ctx, order, false, colidx, vals, h, ldh, bias, out, ldo, n, f, act, nullptr, true); if (rh) return rh; }
      }
      return launch_cb(ctx, plan, order, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
    }
    // (with a bound plan: the rows of more than kHubDeg entries go to the hub kernels)
    const bool hubs = order && order->nsegs > 0;
    dispatch_rows(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, nullptr, 0, nullptr, kRowsPerChunk, hubs ? kHubDeg : 0);
    GCNX_LAUNCH_OK(ctx);
    return hubs ? launch_hubs(ctx, order, false, colidx, vals, h, ldh, bias, out, ldo, n, f, act, nullptr) : GCNX_OK;
  }
  // The pipelined kernel: any graph size, one 1024-thread workgroup per CU with two source buffers.  Opt-in
  // (GCNX_SPMM_KERNEL=pipe / gcnx_set_tuning): correct on every case the tier kernels are tested on, but at config 3 it
  // measures 690-750 us against 646 for the tiers (LOG.md 4.1), so the tier kernels stay the default.
  auto launch_pipe = [&](const PipeItem* list, int count, int ft) -> int {
    static bool attr_set = false;
    if (!attr_set) {
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_pipe_kernel<true, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, kPipeLds));
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_pipe_kernel<false, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, kPipeLds));
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_pipe_kernel<true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, kPipeLds));
      GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_pipe_kernel<false, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, kPipeLds));
      attr_set = true;
    }
    const int slabs = f / kSlab, full = ctx->num_cus;
    int sg = 1;
    for (int c = 8; c > 1; c >>= 1)
      if (slabs % c == 0 && (long long)count * (slabs / c) >= 3LL * full) { sg = c; break; }
    if (ctx->knob_spmm_sg >= 1 && slabs % ctx->knob_spmm_sg == 0) sg = ctx->knob_spmm_sg;
    const int upg = slabs / sg;
    const long long nunits = (long long)count * upg;
    if (nunits >= 2000000000LL) return gcnx_fail(ctx, GCNX_ERR_INVALID, "gcnx_spmm_csr: too many work units");
    const int grid = (int)(nunits < full ? nunits : full);
    int pdbg = 0;
    unsigned long long* stamps = nullptr;
#ifdef GCNX_TUNING
    if (const char* e = getenv("GCNX_SPMM_DBG")) pdbg = atoi(e);
    static unsigned long long* stamp_buf = nullptr;
    if (getenv("GCNX_SPMM_STAMPS")) {
      if (!stamp_buf) (void)hipMalloc((void**)&stamp_buf, (size_t)full * 16 * 8 * sizeof(unsigned long long));
      (void)hipMemsetAsync(stamp_buf, 0, (size_t)full * 16 * 8 * sizeof(unsigned long long), ctx->stream);
      stamps = stamp_buf;
    }
#endif
#define GCNX_PIPE_LAUNCH(W, FT_)                                                                                            \
    hipLaunchKernelGGL((spmm_pipe_kernel<W, FT_>), dim3(grid), dim3(1024), kPipeLds, ctx->stream, rowptr, colidx, vals, h, ldh, \
                       bias, out, ldo, list, upg, sg, act, (int)nunits, n, f, pdbg, stamps)
    if (vals) { if (ft == 32) GCNX_PIPE_LAUNCH(true, 32); else GCNX_PIPE_LAUNCH(true, 16); }
    else { if (ft == 32) GCNX_PIPE_LAUNCH(false, 32); else GCNX_PIPE_LAUNCH(false, 16); }
#undef GCNX_PIPE_LAUNCH
    GCNX_LAUNCH_OK(ctx);
#ifdef GCNX_TUNING
    if (stamps) {      // segment times (100 MHz ticks) of wave 0 and wave 15, averaged over the workgroups, to stderr
      std::vector<unsigned long long> hst((size_t)full * 16 * 8);
      (void)hipMemcpyAsync(hst.data(), stamps, hst.size() * 8, hipMemcpyDeviceToHost, ctx->stream);
      (void)hipStreamSynchronize(ctx->stream);
      const char* names[8] = {"prologue", "index burst", "-", "reduce+epilogue+dma issue", "dma wait", "barrier", "-", "-"};
      for (int wv : {0, 15}) {
        double sum[8] = {0};
        for (int b = 0; b < grid; ++b) for (int k = 0; k < 8; ++k) sum[k] += (double)hst[((size_t)b * 16 + wv) * 8 + k];
        fprintf(stderr, "[pipe stamps ft %d] wave %2d:", ft, wv);
        for (int k = 0; k < 6; ++k) if (names[k][0] != '-') fprintf(stderr, "  %s %.1f us", names[k], sum[k] / grid * 0.01);
        fprintf(stderr, "\n");
      }
    }
#endif
    return GCNX_OK;
  };
  const bool pipe_ok = f <= kPipeMaxF && (uint64_t)n * (uint64_t)ldo * 4u < 0xFFFFFFF0ull;
  if (force == 3 && pipe_ok) {
    if (plan->n16 > 0) { int rc = launch_pipe(plan->items, plan->n16, 16); if (rc) return rc; }
    if (plan->nitems > plan->n16) { int rc = launch_pipe(plan->items + plan->n16, plan->nitems - plan->n16, 32); if (rc) return rc; }
    if (plan->npipe_chunks > 0) {   // graphs of more than 1248 rows: 32-row chunks on the rows kernel
      dispatch_rows(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, plan->pipe_chunks, plan->npipe_chunks);
      GCNX_LAUNCH_OK(ctx);
    }
    return GCNX_OK;
  }
  // tier 1: two 512-thread workgroups per CU; tier 2: one 1024-thread workgroup with the whole LDS
  const DuoFold bo{nullptr, nullptr, 0, 0, relu_bits};     // (graphs taller than a tile get no bits: their rows are folded from out)
  const int dmode = relu_bits ? kDuoBitsOut : kDuoPlain;
  // The three launches write disjoint rows.  GCNX_SPMM_CONC: as concurrent branches (two auxiliary streams), so that
  // one launch's tail is filled by the next one's workgroups instead of draining the chip between them.
  hipStream_t aux[2] = {nullptr, nullptr};
  hipStream_t const home = ctx->stream;
  const bool conc = ctx->knob_spmm_conc && (plan->n1 > 0) + (plan->n2 > 0) + (plan->nchunks > 0) >= 2;
  if (conc) { int rc = gcnx_aux_fork(ctx, aux); if (rc) return rc; }
  int rc = GCNX_OK;
  if (plan->n2 > 0) {        // (the 1024-thread tier first: its workgroups are the hardest to place)
    rc = launch_duo<1024, 32, 4>(ctx, rowptr, rowrec, colidx, vals, h, ldh, bias, out, ldo, n, f, act, plan->dev + plan->n1,
                                 plan->n2, &bo, dmode, out16);
  }
  if (!rc && plan->n1 > 0) {
    if (conc) ctx->stream = aux[0];
    rc = launch_duo<512, 32, 4>(ctx, rowptr, rowrec, colidx, vals, h, ldh, bias, out, ldo, n, f, act, plan->dev, plan->n1, &bo, dmode);
    ctx->stream = home;
  }
  const int ch0 = cb ? plan->nchunks_cb : 0;         // (column-block graphs' chunks come first in the list)
  if (!rc && plan->nchunks > ch0) {   // graphs taller than any tile: plan-listed 32-row chunks on the rows kernel
    if (conc) ctx->stream = aux[1];
    const bool hubs = order && order->nsegs_tall > (cb ? order->nsegs_cb : 0);
    dispatch_rows(ctx, rowptr, colidx, vals, h, ldh, bias, out, ldo, n, f, act, plan->dev + plan->n1 + plan->n2 + ch0,
                  plan->nchunks - ch0, nullptr, plan->chunk_rpc, hubs ? kHubDeg : 0, out16);
    if (hipGetLastError() != hipSuccess) rc = gcnx_fail(ctx, GCNX_ERR_HIP, "gcnx_spmm_csr: row-chunk launch failed");
    if (!rc && hubs) rc = launch_hubs(ctx, order, true, colidx, vals, h, ldh, bias, out, ldo, n, f, act, nullptr, cb);
    ctx->stream = home;
  }
  if (!rc && cb) {
    if (conc) ctx->stream = aux[1];
    rc = launch_cb(ctx, plan, order, colidx, vals, h, ldh, bias, out, ldo, n, f, act);
    ctx->stream = home;
  }
  if (conc) { const int rj = gcnx_aux_join(ctx); if (!rc) rc = rj; }
  return rc;
}

static int spmm_csr_pool_bwd_impl(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                                  const float* y, int64_t ldy, const int32_t* graph_ptr, int32_t b, const float* dpooled,
                                  int64_t lddp, float* out, int64_t ldo, int32_t n, int32_t f, int mode,
                                  const gcnx_spmm_plan* plan, const void* y_bits, int out16);

int gcnx_spmm_csr_pool_bwd(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                           const float* y, int64_t ldy, const int32_t* graph_ptr, int32_t b, const float* dpooled,
                           int64_t lddp, float* out, int64_t ldo, int32_t n, int32_t f, int mode,
                           const gcnx_spmm_plan* plan, const void* y_bits) {
  return spmm_csr_pool_bwd_impl(ctx, rowptr, colidx, vals, y, ldy, graph_ptr, b, dpooled, lddp, out, ldo, n, f, mode, plan, y_bits, 0);
}

int gcnx_spmm_csr_pool_bwd_bf16out(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                                   const float* y, int64_t ldy, const int32_t* graph_ptr, int32_t b, const float* dpooled,
                                   int64_t lddp, void* out16, int64_t ldo, int32_t n, int32_t f, int mode,
                                   const gcnx_spmm_plan* plan, const void* y_bits) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, out16 != nullptr || n == 0 || f == 0, "gcnx_spmm_csr_pool_bwd_bf16out: NULL output");
  if (!y_bits || !plan || plan->nblocks != b || !out16_shape_ok(ctx, plan, vals, f, ldo, out16)) return GCNX_ERR_UNSUPPORTED;
  return spmm_csr_pool_bwd_impl(ctx, rowptr, colidx, vals, y, ldy, graph_ptr, b, dpooled, lddp, (float*)out16, ldo, n, f, mode, plan, y_bits, 1);
}

static int spmm_csr_pool_bwd_impl(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals,
                                  const float* y, int64_t ldy, const int32_t* graph_ptr, int32_t b, const float* dpooled,
                                  int64_t lddp, float* out, int64_t ldo, int32_t n, int32_t f, int mode,
                                  const gcnx_spmm_plan* plan, const void* y_bits, int out16) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "aggregation bwd (pool' folded)");
  GCNX_REQUIRE(ctx, n >= 0 && f >= 0 && b >= 0, "gcnx_spmm_csr_pool_bwd: negative size");
  GCNX_REQUIRE(ctx, mode == GCNX_POOL_SUM || mode == GCNX_POOL_AVG,
               "gcnx_spmm_csr_pool_bwd: pool mode %d has no folded form (use gcnx_segment_pool_bwd + gcnx_spmm_csr)", mode);
  if (n == 0 || f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, b > 0, "gcnx_spmm_csr_pool_bwd: rows without graphs");
  GCNX_REQUIRE(ctx, rowptr && colidx && y && graph_ptr && dpooled && out, "gcnx_spmm_csr_pool_bwd: NULL pointer");
  GCNX_REQUIRE(ctx, ldy >= f && ldo >= f && lddp >= f, "gcnx_spmm_csr_pool_bwd: leading dimension smaller than f=%d", f);
  GCNX_REQUIRE(ctx, y != out, "gcnx_spmm_csr_pool_bwd: in-place aggregation is not possible");
  GCNX_REQUIRE(ctx, (f % 4 == 0) && (ldy % 4 == 0) && (ldo % 4 == 0) && (lddp % 4 == 0) && aligned16(y) && aligned16(out) &&
                        aligned16(dpooled),
               "gcnx_spmm_csr_pool_bwd: needs f and the leading dimensions in multiples of 4 floats and 16-byte aligned "
               "operands (use gcnx_segment_pool_bwd + gcnx_spmm_csr otherwise)");
  const FoldArgs fo{graph_ptr, dpooled, lddp, b, mode == GCNX_POOL_AVG ? 1 : 0};
  // with a plan (throughput regime): the tile kernels in their folded form, taller graphs' row chunks on the rows kernel
  const bool tiles = plan && plan->nblocks == b && f % kSlab == 0 && ctx->knob_spmm_kernel != 1 &&
                     ((long long)(plan->n1 + 2 * plan->n2) * (f / kSlab) >= 4LL * ctx->num_cus || ctx->knob_spmm_kernel >= 2);
  const RowRec* rowrec = nullptr;
  const RowOrder* order = nullptr;
  if (plan && plan->nblocks == b) { const int rb = plan_order(ctx, plan, rowptr, n, false, &rowrec, &order); if (rb) return rb; }
  if (out16 && (!tiles || (order && order->nsegs_tall > 0))) return GCNX_ERR_UNSUPPORTED;
  if (!tiles) {
    if (cb_path_ok(ctx, plan, n, f, ldy, out16) && plan->nblocks == b && order != nullptr) {
      // graphs too large for an XCD's L2 in column blocks, folded (r4: as the forward aggregation); the other graphs' rows as
      // plan-listed chunks on the row gather, their hub rows as segments
      if (plan->nrest > 0) {
        const bool hubs = order->nsegs > order->nsegs_cb;
        dispatch_rows(ctx, rowptr, colidx, vals, y, ldy, nullptr, out, ldo, n, f, GCNX_ACT_NONE, plan->rest, plan->nrest, &fo, plan->rest_rpc, hubs ? kHubDeg : 0);
        GCNX_LAUNCH_OK(ctx);
        if (hubs) { const int rh = launch_hubs(ctx, order, false, colidx, vals, y, ldy, nullptr, out, ldo, n, f, GCNX_ACT_NONE, &fo, true); if (rh) return rh; }
      }
      return launch_cb(ctx, plan, order, colidx, vals, y, ldy, nullptr, out, ldo, n, f, GCNX_ACT_NONE, &fo);
    }
    const bool hubs = order && order->nsegs > 0;
    dispatch_rows(ctx, rowptr, colidx, vals, y, ldy, nullptr, out, ldo, n, f, GCNX_ACT_NONE, nullptr, 0, &fo, kRowsPerChunk, hubs ? kHubDeg : 0);
    GCNX_LAUNCH_OK(ctx);
    return hubs ? launch_hubs(ctx, order, false, colidx, vals, y, ldy, nullptr, out, ldo, n, f, GCNX_ACT_NONE, &fo) : GCNX_OK;
  }
  // y_bits (the forward's gcnx_spmm_csr_relu_bits image of y): the tiers expand it instead of reading y
  const int dmode = y_bits ? kDuoFoldBits : kDuoFold;
  if (plan->n1 > 0) {
    const DuoFold df{plan->gids, dpooled, lddp, fo.avg, (uint32_t*)y_bits};
    int rc = launch_duo<512, 32, 4>(ctx, rowptr, rowrec, colidx, vals, y, ldy, nullptr, out, ldo, n, f, GCNX_ACT_NONE, plan->dev, plan->n1, &df,
                                    dmode);
    if (rc) return rc;
  }
  if (plan->n2 > 0) {
    const DuoFold df{plan->gids + plan->n1, dpooled, lddp, fo.avg, (uint32_t*)y_bits};
    int rc = launch_duo<1024, 32, 4>(ctx, rowptr, rowrec, colidx, vals, y, ldy, nullptr, out, ldo, n, f, GCNX_ACT_NONE, plan->dev + plan->n1,
                                     plan->n2, &df, dmode, out16);
    if (rc) return rc;
  }
  if (plan->nchunks > 0) {
    const bool hubs = order && order->nsegs_tall > 0;
    dispatch_rows(ctx, rowptr, colidx, vals, y, ldy, nullptr, out, ldo, n, f, GCNX_ACT_NONE, plan->dev + plan->n1 + plan->n2,
                  plan->nchunks, &fo, plan->chunk_rpc, hubs ? kHubDeg : 0, out16);
    GCNX_LAUNCH_OK(ctx);
    if (hubs) return launch_hubs(ctx, order, true, colidx, vals, y, ldy, nullptr, out, ldo, n, f, GCNX_ACT_NONE, &fo);
  }
  return GCNX_OK;
}

}  // extern "C"
