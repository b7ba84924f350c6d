// GCNConv as ONE launch in the small-feature regime (F <= 128: config 2's E. coli batches).
//
// Spektral's GCNConv.call (the layer behind gcn.py:334) computes A (X W): a dense product, then the aggregation --
// two launches here (gemm.hip, spmm.hip) with the [N, F] intermediate written and read back in between, each ~12-15 us
// at N = 22 576 where neither is anywhere near a throughput bound.  (A X) W is the same product; in that order a
// workgroup can own 32 rows end to end:
//     gather + weight the neighbours' rows of X  ->  S tile [32, K] in LDS  ->  MFMA with W  ->  bias / ReLU  ->  out
// and the backward of the layer above the pool runs the same way with the transposed operator:
//     gather [Y2 > 0] rows, scale by the graph's dPooled row (pool' and ReLU' folded: dZ2 is what is being gathered)
//     ->  T tile  ->  MFMA with W2^T  ->  [Y1 > 0] mask  ->  dZ1, plus the tile's column sums (db1 partials).
// S = A X is saved by the forward: the weight gradient of this order is dW = S^T dZ (gcnx_gemm_dw2).
//
// Shape of a workgroup: 512 threads, 32 rows.
//   gather   K / 4 lanes per row (float4 each), 64 / (K / 4) rows per wave instruction, every row group walks its two
//            rows together, four entries each per trip: eight 16-byte loads in flight per lane.  Range-checked buffer
//            loads: slots past a row's end fetch nothing.  The tile's CSR entries are staged in LDS first.
//   product  v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate): wave w owns output columns [16w, 16w + 16)
//            of both 16-row halves.  Its slice of W sits in REGISTERS (K / 4 per lane, loaded straight from L2 in the
//            B-operand layout: lane l holds k = 4 kk + (l >> 4), column l & 15) -- no LDS image of W, so a workgroup
//            needs only the 17 KiB tile and three fit a CU.  A operand: S[row = l & 15][k] from the row-major tile,
//            row stride K + 4 floats (4 row + k spreads the 64 lanes over the 64 banks).
//   epilogue in registers (bias / ReLU, or the ReLU mask and the tile's column sums): no barrier after the last MFMA.
//
// The global pool and the classifier head (GlobalSumPool / GlobalAvgPool -> Dense(softmax) -> CCE, gcn.py:332-337) ride in
// these launches too when the labels are binary: the forward of the pooled layer leaves per-(tile, graph) column sums and
// positive counts from its epilogue (gcnx_gcn_conv_fwd_pool), the backward adds up its tile's graphs' partial sums under
// the staging latency and evaluates dPooled itself (gcnx_head_args) -- neither a pool nor a head launch stands between
// the forward and the backward aggregation, and no workgroup ever waits for another.
#include "common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fbf16x8 __attribute__((ext_vector_type(8)));

constexpr int kFRows = 32;        // rows per workgroup
constexpr int kFCap = 1024;       // CSR entries of a tile staged in LDS (the rest is read from global memory)

struct FusedArgs {
  const int32_t* rowptr; const int32_t* colidx; const float* vals;
  const float* x; int64_t ldx;           // the gathered matrix [n, K]
  int32_t n;
  const float* w; int32_t ldw;           // forward: W [K, nc];  backward: W2 [nc, K]
  int32_t nc;                            // output columns (multiple of 16, <= 128)
  const float* bias; int act;            // forward epilogue
  float* s; int64_t lds;                 // forward: S = A X (may be NULL)
  float* wt_out;                         // forward: W^T [nc, K] written by workgroup 0 (may be NULL) -- the layout the
                                         // backward launch wants its weight operand in
  int w_t;                               // backward: w already holds W2^T [K, nc] (that by-product)
  float* out; int64_t ldo;
  // backward only
  const int32_t* node_graph; const int32_t* gp; const float* dp; int64_t lddp; int avg;
  const float* mask; int64_t ldmask;     // Y1 (ReLU output of the layer below)
  float* dz2; int64_t lddz2;             // dZ2 rows of this tile (may be NULL)
  float* colpart;                        // [tiles, nc] column sums of what was written to out (may be NULL)
  // forward with the global pool's partial sums in the epilogue (tp_part != NULL; node_graph set): row t + g of tp_part /
  // tp_cnt receives the column sums / positive counts of the rows of graph g in tile t -- (tile, graph) pairs form a
  // staircase, so t + g is a distinct row for every pair and a graph's partials are the rows t + g of its tiles
  float* tp_part; float* tp_cnt;
  // backward with the classifier head folded in (hd_part != NULL): dPooled is not read -- every workgroup evaluates
  // Dense(softmax) + CCE' of its tile's graphs from those partial sums (a few hundred flops each), so that neither the
  // pool nor the head is a launch between the forward and this one (the head's other outputs are leaves: gcnx_gemm_dw2
  // computes them from hd_psum / hd_csum, the per-graph totals the workgroup holding a graph's first row writes here)
  const float* hd_part; const float* hd_cnt; int hd_rows; int hd_b;   // tile partials [(tiles + b)][K]; rows of the arrays
  float* hd_psum; float* hd_csum;                                      // [b][K] each
  const float* hd_w; const float* hd_bias; const float* hd_y; int hd_c; float hd_denom; int hd_fl;
  int dbg;                               // tuning builds: phase-ablation bits (1 no gather, 2 no MFMA, 4 no weight load)
};

__device__ __forceinline__ float4 fbuf4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const f32x4v r = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  return make_float4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ float4 f4fma(float v, float4 h, float4 a) {
  const f32x2 w = {v, v};
  const f32x2 lo = __builtin_elementwise_fma(w, f32x2{h.x, h.y}, f32x2{a.x, a.y});
  const f32x2 hi = __builtin_elementwise_fma(w, f32x2{h.z, h.w}, f32x2{a.z, a.w});
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
// [a > 0] of a saved ReLU output (a >= 0): see f4_step in spmm.hip
__device__ __forceinline__ float4 f4step(float4 a) {
  const f32x2 big = {0x1p127f, 0x1p127f};
  f32x2 lo = {a.x, a.y}, hi = {a.z, a.w}, t0, t1;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t0) : "v"(lo), "v"(big));
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(t1) : "v"(hi), "v"(big));
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(lo) : "v"(t0), "v"(big));
  asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(hi) : "v"(t1), "v"(big));
  return make_float4(lo[0], lo[1], hi[0], hi[1]);
}

// The classifier head inside the backward launch: pooled sums of a graph (this lane's 4 of the K columns; LPR lanes hold
// a row) -> logits (lane partials + an xor tree inside the lane group) -> softmax -> CCE' -> dlogits, the arithmetic of
// head_kernel, times the pool's 1 / n_g.  Two class slots -- the reference's binary labels; a single class leaves the
// second slot at weight 0 and logit -inf (more classes: the host refuses and the step keeps the head's own launch).
// pool'(dPooled) of the graph is then dl.x * W3[:, 0] + dl.y * W3[:, 1], which every row group forms for its own columns.
__device__ __forceinline__ void f4acc(float4& a, const float4 q) { a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w; }

template <int LPR>
__device__ __forceinline__ float2 fused_head_dlogits(const FusedArgs& p, float4 pv, int rows, float4 w0, float4 w1, float b0,
                                                     float b1, float y0, float y1) {
  float sc = 1.0f;
  if (p.avg) { sc = 1.0f / (float)max(rows, 1); pv.x *= sc; pv.y *= sc; pv.z *= sc; pv.w *= sc; }
  const bool two = p.hd_c > 1;
  float z0 = fmaf(pv.w, w0.w, fmaf(pv.z, w0.z, fmaf(pv.y, w0.y, pv.x * w0.x)));
  float z1 = fmaf(pv.w, w1.w, fmaf(pv.z, w1.z, fmaf(pv.y, w1.y, pv.x * w1.x)));
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) { z0 += __shfl_xor(z0, off); z1 += __shfl_xor(z1, off); }
  z0 += b0;
  z1 = two ? z1 + b1 : -INFINITY;
  const float m = fmaxf(z0, z1), e0 = expf(z0 - m), e1 = expf(z1 - m), sum = e0 + e1;
  const float p0 = e0 / sum, p1 = e1 / sum;
  const bool k0 = p.hd_fl || (p0 > 1e-7f && p0 < 1.0f - 1e-7f), k1 = two && (p.hd_fl || (p1 > 1e-7f && p1 < 1.0f - 1e-7f));
  const float ymsum = (k0 ? y0 : 0.f) + (k1 ? y1 : 0.f);
  const float dl0 = (p0 * ymsum - (k0 ? y0 : 0.f)) / p.hd_denom, dl1 = two ? (p1 * ymsum - (k1 ? y1 : 0.f)) / p.hd_denom : 0.f;
  return make_float2(dl0 * sc, dl1 * sc);
}

// X3: the product phase on the bf16 MFMA with split operands (hi = bf16(x), lo = bf16(x - hi); hi*lo + lo*hi + hi*hi, fp32
// accumulate: GCNX_PREC_BF16X3, ~2^-17 per operand) instead of exact fp32 products -- three 16x16x32 MFMAs per 32 k where
// the fp32 path issues eight 16x16x4.
template <int K, bool WEIGHTED, bool BWD, bool X3 = false>
__global__ __launch_bounds__(512, 6) void gcn_conv_fused_kernel(FusedArgs p) {
  constexpr int LPR = K / 4;                 // lanes per gathered row
  constexpr int GW = 64 / LPR;               // row groups per wave
  constexpr int NG = 8 * GW;                 // row groups per workgroup
  constexpr int RPG = NG >= kFRows ? 1 : kFRows / NG;   // rows per group (2 at K = 128)
  constexpr int U = 4;                       // entries per row per trip
  static_assert(K == 32 || K == 64 || K == 128, "gather width");
  __shared__ __attribute__((aligned(16))) float tile[kFRows][K + 4];   // row stride K + 4: see the A-operand read
  // staged CSR entries: {byte offset of the gathered row = column * ldx * 4, weight}.  One multiply per entry here
  // instead of one per lane and entry in the gather loop, and one ds_read_b64 for both: the gather loop is bound by VALU
  // issue, not by memory (bit-image experiment: 2-byte loads in place of the 16-byte ones ran no faster), so every
  // instruction per (lane, entry) shows.  2 U slack entries: the loop reads up to U - 1 past a row's end unclamped.
  __shared__ __attribute__((aligned(8))) int2 s_ent[kFCap + 2 * U];
  __shared__ int32_t s_rp[kFRows + 1];
  __shared__ unsigned char s_mb[BWD ? kFRows * 32 : 1];   // backward: [Y1 > 0] of the tile, 4 columns per byte
  __shared__ float2 s_dl[BWD ? kFRows : 1];               // backward, head folded in: dlogits x pool scale of the tile's graphs
  __shared__ __attribute__((aligned(16))) float s_w3[BWD ? 2 : 1][K];   // ... and the two columns of W3
  __shared__ float s_b3[2];
  __shared__ int32_t s_g[BWD ? 1 : kFRows];               // forward with the pool's partial sums: graph of each row
  static_assert(!BWD || sizeof(float) * kFRows * (K + 4) >= sizeof(float4) * 2 * 8 * 2 * LPR, "the head's wave partials alias the tile");
#ifdef GCNX_TUNING
  const int dbg = p.dbg;
#else
  constexpr int dbg = 0;
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntiles = gridDim.x;
  const int t = gcnx_xcd_remap(blockIdx.x, ntiles);
  const int r0 = t * kFRows, nr = min(p.n - r0, kFRows);
  if (tid <= nr) s_rp[tid] = p.rowptr[r0 + tid];
  if (!BWD && p.tp_part && tid < kFRows)               // (clamped: a malformed id vector cannot write outside the two arrays)
    s_g[tid] = tid < nr ? min(max(p.node_graph[r0 + tid], 0), p.hd_b - 1) : -1;
  const int e0 = p.rowptr[r0], e1 = p.rowptr[r0 + nr];
  const int staged = min(e1 - e0, kFCap);
  const unsigned ld4 = (unsigned)p.ldx * 4u;
  for (int i = tid; i < staged + 2 * U; i += 512) {       // the slack entries carry weight 0: never NaN * 0
    int2 en = make_int2(0, 0);
    if (i < staged) {
      en.x = (int)((unsigned)p.colidx[e0 + i] * ld4);
      en.y = WEIGHTED ? __float_as_int(p.vals[e0 + i]) : 0x3f800000;
    }
    s_ent[i] = en;
  }
  // ---- gather ------------------------------------------------------------------------------------------------
  const int gid = wave * GW + lane / LPR, sub = lane % LPR;
  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)((unsigned)p.n * (unsigned)p.ldx * 4u), 0x00020000);
  // backward: the graph of each row (its dPooled vector is fetched after the gather: fewer live registers in the loop)
  int grow[RPG];
  if (BWD) {
#pragma unroll
    for (int j = 0; j < RPG; ++j) {
      const int r = gid + j * NG;
      grow[j] = (r < nr && gid < kFRows) ? p.node_graph[r0 + r] : -1;
    }
  }
  // backward: the ReLU mask rows of the tile (saved Y1), read here as whole 512-byte rows -- two loads per thread under
  // the staging latency -- and kept as bits in LDS; the epilogue's accumulator layout would read them as 64-byte pieces,
  // eight load instructions per wave in a loop whose cost is counted in instructions (2.2 us of the launch)
  float4 mrow[2];
  const int mlanes = p.nc >> 2;
  if (BWD) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = tid + 512 * q, row = i / mlanes, c4 = i - row * mlanes;
      mrow[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nr && !(dbg & 16)) mrow[q] = *reinterpret_cast<const float4*>(p.mask + (int64_t)(r0 + row) * p.ldmask + 4 * c4);
    }
  }
  // The classifier head of the tile's graphs (uniform branches), two graphs per round -- one round at E. coli sizes, up
  // to 16 where the graphs are single nodes.  All eight waves add up the two graphs' tile partials, strided (tile
  // tlo + wave, + 8, ...: two or three 16-byte loads per lane, range-checked buffer loads so that nothing branches around
  // them), under the staging latency; the round's barrier is the staging barrier; wave 0 then combines the eight in order
  // -- the same order wherever a graph is evaluated -- and leaves the graphs' dlogits in LDS while the others gather.
  // The workgroup that holds a graph's first row also writes its totals (the head's leaves read them in gcnx_gemm_dw2).
  float4 (*s_pv)[2][2][LPR] = reinterpret_cast<float4 (*)[2][2][LPR]>(&tile[0][0]);   // [wave][sum | count][graph][lane]
  const int hsel = lane / LPR, hsub = lane % LPR;
  int g_first = 0;
  if (BWD && p.hd_part) {
    g_first = min(max(__builtin_amdgcn_readfirstlane(p.node_graph[r0]), 0), p.hd_b - 1);       // (clamped, as in the forward)
    const int g_last = min(max(__builtin_amdgcn_readfirstlane(p.node_graph[r0 + nr - 1]), g_first), min(g_first + kFRows, p.hd_b) - 1);
    for (int i = tid; i < 2 * K; i += 512) {
      const int cls = i / K, k = i - cls * K;
      s_w3[cls][k] = cls < p.hd_c ? p.hd_w[(int64_t)k * p.hd_c + cls] : 0.f;
    }
    if (tid >= 2 * K && tid < 2 * K + 2) s_b3[tid - 2 * K] = (p.hd_bias && tid - 2 * K < p.hd_c) ? p.hd_bias[tid - 2 * K] : 0.f;
    const unsigned bytes = (unsigned)p.hd_rows * (unsigned)K * 4u;
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)p.hd_part, (short)0, (int)bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t cr = __builtin_amdgcn_make_buffer_rsrc((void*)p.hd_cnt, (short)0, (int)bytes, 0x00020000);
    for (int gb = g_first; gb <= g_last; gb += 2) {
      if (gb > g_first) __syncthreads();                 // wave 0 is done with the previous round's partials
      int hd_g = 0, hd_n = 0;
      bool hd_desig = false;
      float y0 = 0.f, y1 = 0.f;
      if (hsel < 2) {
        hd_g = min(gb + hsel, g_last);                   // (an odd graph out is evaluated twice: same values)
        const int hd_a = p.gp[hd_g];
        hd_n = p.gp[hd_g + 1] - hd_a;
        hd_desig = hd_a >= r0 && hd_n > 0;               // the graph's first row is in this tile
        const int tlo = hd_a / kFRows, thi = hd_n > 0 ? (hd_a + hd_n - 1) / kFRows : tlo - 1;
        float4 ps = make_float4(0.f, 0.f, 0.f, 0.f), pc = ps;
        for (int tb = tlo + wave; __builtin_amdgcn_ballot_w64(tb <= thi) != 0; tb += 32) {
          float4 q[4], c[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int tt = tb + 8 * u;
            const unsigned off = tt <= thi ? ((unsigned)(tt + hd_g) * (unsigned)K + (unsigned)hsub * 4u) * 4u : 0xFFFFFFF0u;
            q[u] = fbuf4(pr, off);
            c[u] = fbuf4(cr, hd_desig ? off : 0xFFFFFFF0u);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) { f4acc(ps, q[u]); f4acc(pc, c[u]); }
        }
        s_pv[wave][0][hsel][hsub] = ps;
        s_pv[wave][1][hsel][hsub] = pc;
        if (wave == 0) {
          y0 = p.hd_y[(int64_t)hd_g * p.hd_c];
          y1 = p.hd_c > 1 ? p.hd_y[(int64_t)hd_g * p.hd_c + 1] : 0.f;
        }
      }
      __syncthreads();
      if (wave == 0 && hsel < 2) {
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f), pc = pv;
#pragma unroll 2
        for (int w = 0; w < 8; ++w) { f4acc(pv, s_pv[w][0][hsel][hsub]); f4acc(pc, s_pv[w][1][hsel][hsub]); }
        if (hd_desig) {
          *reinterpret_cast<float4*>(p.hd_psum + (int64_t)hd_g * K + hsub * 4) = pv;
          *reinterpret_cast<float4*>(p.hd_csum + (int64_t)hd_g * K + hsub * 4) = pc;
        }
        const float2 dl = fused_head_dlogits<LPR>(p, pv, hd_n, *reinterpret_cast<const float4*>(&s_w3[0][hsub * 4]),
                                                  *reinterpret_cast<const float4*>(&s_w3[1][hsub * 4]),
                                                  s_b3[0], s_b3[1], y0, y1);
        if (hsub == 0) s_dl[hd_g - g_first] = dl;
      }
    }
  } else {
    __syncthreads();
  }
  if (BWD) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = tid + 512 * q, row = i / mlanes, c4 = i - row * mlanes;
      if (row < kFRows)
        s_mb[row * 32 + c4] = (unsigned char)((mrow[q].x > 0.f ? 1u : 0u) | (mrow[q].y > 0.f ? 2u : 0u) | (mrow[q].z > 0.f ? 4u : 0u) |
                                              (mrow[q].w > 0.f ? 8u : 0u));
    }
  }
  float4 acc[RPG];
  int ea[RPG], eb[RPG], ebf[RPG];
  int len = 0;
#pragma unroll
  for (int j = 0; j < RPG; ++j) {
    const int r = gid + j * NG;
    acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool live = r < nr && gid < kFRows;
    ea[j] = live ? s_rp[r] - e0 : 0;
    eb[j] = live ? s_rp[r + 1] - e0 : 0;
    ebf[j] = min(eb[j], kFCap);                      // the staged part of the row ...
    ea[j] = min(ea[j], kFCap);                       // ... (empty if the row starts past it)
    len = max(len, ebf[j] - ea[j]);
  }
  if (dbg & 1) len = 0;
  const unsigned sub16 = (unsigned)sub * 16u;
  // Branch-free on purpose -- a load inside a branch makes hipcc close the trip with s_waitcnt vmcnt(0), i.e. the eight
  // row loads would complete one after the other.  Slots past the row's end get an out-of-range offset (the buffer load
  // returns zeros without a fetch); their weight is whatever the next row's entry holds -- finite -- times zero.
  for (int tt = 0; __builtin_amdgcn_ballot_w64(tt < len) != 0; tt += U) {
    float4 hv[RPG][U];
    float wv[RPG][U];
#pragma unroll
    for (int j = 0; j < RPG; ++j) {
      // a row that is done while others of the wave are not stays at its end: the reads stay inside the slack
      const int eb_ = min(ea[j] + tt, ebf[j]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = eb_ + u;
        const int2 en = s_ent[e];
        wv[j][u] = __int_as_float(en.y);
        hv[j][u] = fbuf4(xr, e < ebf[j] ? (unsigned)en.x + sub16 : 0xFFFFFFF0u);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // all eight loads go out before the first is consumed (hipcc otherwise reuses the
                                         // destination registers of the first loads for the last and waits in between)
#pragma unroll
    for (int j = 0; j < RPG; ++j)
#pragma unroll
      for (int u = 0; u < U; ++u) acc[j] = f4fma(wv[j][u], (BWD && !(dbg & 64)) ? f4step(hv[j][u]) : hv[j][u], acc[j]);
  }
  if (e1 - e0 > kFCap) {       // uniform per workgroup, rare: entries beyond the staged ones, one at a time from global memory
#pragma unroll
    for (int j = 0; j < RPG; ++j)
      for (int e = max(eb[j] > 0 ? s_rp[gid + j * NG] - e0 : 0, kFCap); e < eb[j]; ++e) {   // (ea is clamped: the row's true start)
        const int c = p.colidx[e0 + e];
        const float v = WEIGHTED ? p.vals[e0 + e] : 1.0f;
        const float4 h = fbuf4(xr, (unsigned)c * ld4 + sub16);
        acc[j] = f4fma(v, BWD ? f4step(h) : h, acc[j]);
      }
  }
  float4 own[RPG], dscale[RPG];   // backward: pool'(dPooled) of the row's graph, and the row's own [Y2 > 0] row (dZ2 out)
  if (BWD) {
#pragma unroll
    for (int j = 0; j < RPG; ++j) {
      const int r = gid + j * NG;
      own[j] = dscale[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (grow[j] >= 0 && (dbg & 32)) dscale[j] = make_float4(1.f, 1.f, 1.f, 1.f);
      if (grow[j] >= 0 && !(dbg & 32) && !p.hd_part) {
        float4 d = *reinterpret_cast<const float4*>(p.dp + (int64_t)grow[j] * p.lddp + sub * 4);
        if (p.avg) {
          const float sc = 1.0f / (float)(p.gp[grow[j] + 1] - p.gp[grow[j]]);
          d.x *= sc; d.y *= sc; d.z *= sc; d.w *= sc;
        }
        dscale[j] = d;
      }
      if (grow[j] >= 0 && p.dz2) own[j] = *reinterpret_cast<const float4*>(p.x + (int64_t)(r0 + r) * p.ldx + sub * 4);
    }
    if (p.hd_part && !(dbg & 32)) {
      // wave 0 left the dlogits of the tile's graphs in LDS while the others were gathering (the barrier also ends the
      // life of the wave partials in the tile's LDS, which is written next): pool'(dPooled) = dlogits . W3^T per row
      __syncthreads();
      const float4 w0 = *reinterpret_cast<const float4*>(&s_w3[0][sub * 4]), w1 = *reinterpret_cast<const float4*>(&s_w3[1][sub * 4]);
#pragma unroll
      for (int j = 0; j < RPG; ++j) {
        if (grow[j] < 0) continue;
        const float2 dl = s_dl[(grow[j] - g_first) & (kFRows - 1)];
        dscale[j] = make_float4(fmaf(dl.y, w1.x, dl.x * w0.x), fmaf(dl.y, w1.y, dl.x * w0.y), fmaf(dl.y, w1.z, dl.x * w0.z),
                                fmaf(dl.y, w1.w, dl.x * w0.w));
      }
    }
  }
  // ---- this wave's slice of W, in the MFMA B layout; in flight while the tile is written -------------------------
  const int c16 = lane & 15, kq = lane >> 4;
  const bool wave_on = 16 * wave < p.nc;
  // register e of the slice holds k = 4 e + (l >> 4) (fp32 MFMA: 16x16x4 B layout) or k = 32 (e / 8) + 8 (l >> 4) + e % 8
  // (bf16 MFMA: 16x16x32 B layout, eight consecutive k per lane)
  auto k_of = [&](int e) { return X3 ? 32 * (e / 8) + 8 * kq + (e % 8) : 4 * e + kq; };
  float wreg[K / 4];
  if (wave_on && !(dbg & 4)) {
#pragma unroll
    for (int e = 0; e < K / 4; ++e)
      wreg[e] = (BWD && !p.w_t) ? p.w[(int64_t)(16 * wave + c16) * p.ldw + k_of(e)]     // W2 [nc, K]: strided
                                : p.w[(int64_t)k_of(e) * p.ldw + 16 * wave + c16];      // [K, nc]: 64-byte row pieces
    if (!BWD && p.wt_out && (int)blockIdx.x < K / 4) {   // W^T, dealt over the first K / 4 workgroups (uniform tests: wreg
#pragma unroll                                          // stays in registers)
      for (int e = 0; e < K / 4; ++e)
        if (e % ntiles == (int)blockIdx.x) p.wt_out[(int64_t)(16 * wave + c16) * K + k_of(e)] = wreg[e];
    }
  }
  fbf16x8 wh[X3 ? K / 32 : 1], wl[X3 ? K / 32 : 1];
  if (X3) {
#pragma unroll
    for (int st = 0; st < K / 32; ++st)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = wave_on ? wreg[st * 8 + j] : 0.f;
        const __bf16 h = (__bf16)v;
        wh[st][j] = h;
        wl[st][j] = (__bf16)(v - (float)h);
      }
  }
  if (gid < kFRows) {
#pragma unroll
    for (int j = 0; j < RPG; ++j) {
      const int r = gid + j * NG;
      float4 a = acc[j];
      if (BWD) {
        a.x *= dscale[j].x; a.y *= dscale[j].y; a.z *= dscale[j].z; a.w *= dscale[j].w;
        if (p.dz2 && r < nr && !(dbg & 8)) {
          const float4 m = f4step(own[j]);
          *reinterpret_cast<float4*>(p.dz2 + (int64_t)(r0 + r) * p.lddz2 + sub * 4) =
              make_float4(m.x * dscale[j].x, m.y * dscale[j].y, m.z * dscale[j].z, m.w * dscale[j].w);
        }
      } else if (p.s && r < nr) {
        *reinterpret_cast<float4*>(p.s + (int64_t)(r0 + r) * p.lds + sub * 4) = a;
      }
      *reinterpret_cast<float4*>(&tile[r][sub * 4]) = a;      // rows past the end hold zeros
    }
  }
  __syncthreads();
  // ---- product and epilogue, in registers: wave w owns output columns [16w, 16w + 16) of all 32 rows ---------------
  // C layout of the 16 x 16 tile: column = lane & 15, rows 4 (lane >> 4) + reg.  A store instruction therefore
  // writes 4 rows x 64 bytes -- narrower than a staged row-major epilogue would, but with no barrier and no LDS
  // round trip behind the last MFMA; the column sums need no other wave.
  if (!wave_on) return;
  const int col = 16 * wave + c16;
  float mk[8];
  if (BWD) {                                          // the ReLU mask bits of this lane's column (rows past the end: 0)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = 16 * (r >> 2) + 4 * kq + (r & 3);
      mk[r] = (dbg & 16) ? 1.f : (float)((s_mb[row * 32 + (col >> 2)] >> (col & 3)) & 1u);
    }
  }
  const float bcol = (!BWD && p.bias) ? p.bias[col] : 0.f;
  f32x4v c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
  if (X3) {
    if (!(dbg & 2))
#pragma unroll
    for (int st = 0; st < K / 32; ++st) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const float* ap = &tile[16 * hf + c16][32 * st + 8 * kq];
        const float4 x0 = *reinterpret_cast<const float4*>(ap), x1 = *reinterpret_cast<const float4*>(ap + 4);
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        fbf16x8 ah, al;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const __bf16 h = (__bf16)xv[j];
          ah[j] = h;
          al[j] = (__bf16)(xv[j] - (float)h);
        }
        f32x4v& cc = hf ? c1 : c0;
        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wl[st], cc, 0, 0, 0);     // small terms first
        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wh[st], cc, 0, 0, 0);
        cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wh[st], cc, 0, 0, 0);
      }
    }
  } else if (!(dbg & 2)) {
#pragma unroll
    for (int kk = 0; kk < K / 4; ++kk) {
      const float a0 = tile[c16][4 * kk + kq];
      const float a1 = tile[16 + c16][4 * kk + kq];
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wreg[kk], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, wreg[kk], c1, 0, 0, 0);
    }
  }
  float cs = 0.f;
  float vout[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int row = 16 * (r >> 2) + 4 * kq + (r & 3);
    float v = r < 4 ? c0[r & 3] : c1[r & 3];
    if (BWD) v = mk[r] > 0.f ? v : 0.f;               // rows past the end: mask 0 (and an all-zero gathered row)
    else { v += bcol; if (p.act == GCNX_ACT_RELU) v = fmaxf(v, 0.f); }
    if (row < nr) p.out[(int64_t)(r0 + row) * p.ldo + col] = v;
    cs += v;                                          // rows in ascending order within the lane
    vout[r] = v;
  }
  if (!BWD && p.tp_part) {
    // the global pool's partial sums of this tile, per graph present in it (one or two at E. coli sizes): the lane's
    // rows in ascending order, then the four row groups in a fixed tree -- one 64-byte store per graph and wave
    int gr[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) gr[r] = s_g[16 * (r >> 2) + 4 * kq + (r & 3)];      // rows past the end: -1
    const int g_lo = s_g[0], g_hi = s_g[nr - 1];
    for (int g = g_lo; g <= g_hi; ++g) {
      float sm = 0.f, cn = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const bool mine = gr[r] == g;
        sm += mine ? vout[r] : 0.f;
        cn += (mine && vout[r] > 0.f) ? 1.f : 0.f;
      }
      sm += __shfl_xor(sm, 16); cn += __shfl_xor(cn, 16);
      sm += __shfl_xor(sm, 32); cn += __shfl_xor(cn, 32);
      if (lane < 16) {
        p.tp_part[(int64_t)(t + g) * p.nc + col] = sm;
        p.tp_cnt[(int64_t)(t + g) * p.nc + col] = cn;
      }
    }
  }
  if (BWD && p.colpart) {                             // ... then the four row groups (lane >> 4) in a fixed tree
    cs += __shfl_xor(cs, 16);
    cs += __shfl_xor(cs, 32);
    if (lane < 16) p.colpart[(int64_t)t * p.nc + col] = cs;
  }
}

template <bool BWD>
int launch_fused(gcnx_ctx* ctx, const FusedArgs& a_in, int k, bool x3) {
  FusedArgs a = a_in;
#ifdef GCNX_TUNING
  if (const char* e = getenv("GCNX_FUSED_DBG")) a.dbg = atoi(e);
#endif
  const int tiles = gcnx_cdiv(a.n, kFRows);
  int dyn = 0;                          // extra dynamic LDS: caps the resident workgroups per CU (tuning builds)
#ifdef GCNX_TUNING
  if (const char* e = getenv("GCNX_FUSED_LDS")) dyn = atoi(e);
#endif
#define GCNX_FUSED_LAUNCH(K_)                                                                                         \
  do {                                                                                                                \
    if (x3) {                                                                                                         \
      if (a.vals) hipLaunchKernelGGL((gcn_conv_fused_kernel<K_, true, BWD, true>), dim3(tiles), dim3(512), dyn, ctx->stream, a); \
      else hipLaunchKernelGGL((gcn_conv_fused_kernel<K_, false, BWD, true>), dim3(tiles), dim3(512), dyn, ctx->stream, a);      \
    } else if (a.vals) hipLaunchKernelGGL((gcn_conv_fused_kernel<K_, true, BWD>), dim3(tiles), dim3(512), dyn, ctx->stream, a); \
    else hipLaunchKernelGGL((gcn_conv_fused_kernel<K_, false, BWD>), dim3(tiles), dim3(512), dyn, ctx->stream, a);      \
  } while (0)
  if (k == 128) GCNX_FUSED_LAUNCH(128);
  else if (k == 64) GCNX_FUSED_LAUNCH(64);
  else GCNX_FUSED_LAUNCH(32);
#undef GCNX_FUSED_LAUNCH
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

inline bool fal16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int64_t tiles_of(int64_t n) { return (n + kFRows - 1) / kFRows; }

bool fused_shape_ok(int64_t n, int32_t k, int32_t nc, int64_t ldx) {
  return n > 0 && (k == 32 || k == 64 || k == 128) && nc >= 16 && nc <= 128 && nc % 16 == 0 && ldx >= k && ldx % 4 == 0 &&
         (uint64_t)n * (uint64_t)ldx * 4u < 0xFFFFFFF0ull;
}

}  // namespace

extern "C" {

int gcnx_gcn_conv_fused_ok(int64_t n, int32_t fi, int32_t fo, int64_t ldx) { return fused_shape_ok(n, fi, fo, ldx) ? 1 : 0; }

int gcnx_gcn_conv_fwd_pool(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* x,
                           int64_t ldx, int32_t n, int32_t fi, const float* w, int32_t fo, const float* bias, int act, float* s,
                           int64_t lds, float* out, int64_t ldo, float* wt_out, int prec, const int32_t* node_graph, int32_t b,
                           float* tile_part, float* tile_cnt) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "GCNConv forward (one launch)");
  GCNX_REQUIRE(ctx, (!tile_part && !tile_cnt) || (tile_part && tile_cnt && node_graph && b > 0 && fal16(tile_part) && fal16(tile_cnt)),
               "gcnx_gcn_conv_fwd_pool: the pool's partial sums need node_graph, b > 0 and two 16-byte aligned outputs");
  GCNX_REQUIRE(ctx, n >= 0 && fi >= 0 && fo >= 0, "gcnx_gcn_conv_fwd: negative size");
  if (prec != GCNX_PREC_F32 && prec != GCNX_PREC_BF16X3)
    return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_gcn_conv_fwd: precision %d (GCNX_PREC_F32 or GCNX_PREC_BF16X3 here)", prec);
  GCNX_REQUIRE(ctx, act == GCNX_ACT_NONE || act == GCNX_ACT_RELU, "gcnx_gcn_conv_fwd: activation %d not supported here", act);
  if (n == 0 || fo == 0) return GCNX_OK;
  if (!fused_shape_ok(n, fi, fo, ldx))
    return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_gcn_conv_fwd: needs fi in {32, 64, 128}, fo a multiple of 16 up to 128 and "
                     "n * ldx * 4 < 2^32 (got n=%d fi=%d fo=%d): use gcnx_gemm + gcnx_spmm_csr", n, fi, fo);
  GCNX_REQUIRE(ctx, rowptr && colidx && x && w && out, "gcnx_gcn_conv_fwd: NULL pointer");
  GCNX_REQUIRE(ctx, ldo >= fo && ldo % 4 == 0 && fal16(x) && fal16(out) && (!bias || fal16(bias)) &&
                        (!s || (lds >= fi && lds % 4 == 0 && fal16(s))),
               "gcnx_gcn_conv_fwd: operands must be 16-byte aligned with leading dimensions in multiples of 4 floats");
  GCNX_REQUIRE(ctx, x != out && x != s, "gcnx_gcn_conv_fwd: in-place aggregation is not possible");
  FusedArgs a{};
  a.rowptr = rowptr; a.colidx = colidx; a.vals = vals; a.x = x; a.ldx = ldx; a.n = n; a.w = w; a.ldw = fo; a.nc = fo;
  a.bias = bias; a.act = act; a.s = s; a.lds = lds; a.out = out; a.ldo = ldo; a.wt_out = wt_out;
  a.node_graph = node_graph; a.tp_part = tile_part; a.tp_cnt = tile_cnt; a.hd_b = b;
  return launch_fused<false>(ctx, a, fi, prec == GCNX_PREC_BF16X3);
}

int gcnx_gcn_conv_fwd(gcnx_ctx* ctx, const int32_t* rowptr, const int32_t* colidx, const float* vals, const float* x,
                      int64_t ldx, int32_t n, int32_t fi, const float* w, int32_t fo, const float* bias, int act, float* s,
                      int64_t lds, float* out, int64_t ldo, float* wt_out, int prec) {
  return gcnx_gcn_conv_fwd_pool(ctx, rowptr, colidx, vals, x, ldx, n, fi, w, fo, bias, act, s, lds, out, ldo, wt_out, prec,
                                nullptr, 0, nullptr, nullptr);
}

int64_t gcnx_gcn_conv_bwd_scratch_floats(int64_t n, int32_t f1) { return n <= 0 || f1 <= 0 ? 0 : (int64_t)gcnx_cdiv(n, kFRows) * f1; }

int gcnx_gcn_conv_bwd_pool(gcnx_ctx* ctx, const int32_t* rowptr_t, const int32_t* colidx_t, const float* vals_t,
                           const float* y2, int64_t ldy2, const int32_t* node_graph, const int32_t* graph_ptr, int32_t b,
                           const float* dpooled, int64_t lddp, int mode, int32_t n, int32_t f2, const float* w2, int32_t f1,
                           int w2_transposed, const float* y1, int64_t ldy1, float* dz2, int64_t lddz2, float* dz1, int64_t lddz1, float* db1,
                           float* scratch, int64_t scratch_floats, gcnx_pending_reduce* pending, int prec,
                           const gcnx_head_args* head) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "GCNConv backward (one launch)");
  if (prec != GCNX_PREC_F32 && prec != GCNX_PREC_BF16X3)
    return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_gcn_conv_bwd_pool: precision %d (GCNX_PREC_F32 or GCNX_PREC_BF16X3 here)", prec);
  if (pending) *pending = gcnx_pending_reduce{nullptr, 0, 0, nullptr, nullptr, 0, 0, nullptr};
  GCNX_REQUIRE(ctx, n >= 0 && f1 >= 0 && f2 >= 0 && b >= 0, "gcnx_gcn_conv_bwd_pool: negative size");
  GCNX_REQUIRE(ctx, mode == GCNX_POOL_SUM || mode == GCNX_POOL_AVG,
               "gcnx_gcn_conv_bwd_pool: pool mode %d has no folded form", mode);
  if (n == 0 || f1 == 0) {
    if (db1 && f1 > 0) GCNX_HIP(ctx, hipMemsetAsync(db1, 0, (size_t)f1 * 4, ctx->stream));
    return GCNX_OK;
  }
  if (!fused_shape_ok(n, f2, f1, ldy2))
    return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_gcn_conv_bwd_pool: needs f2 in {32, 64, 128}, f1 a multiple of 16 up to 128 "
                     "and n * ldy2 * 4 < 2^32 (got n=%d f1=%d f2=%d)", n, f1, f2);
  GCNX_REQUIRE(ctx, b > 0 && rowptr_t && colidx_t && y2 && node_graph && graph_ptr && (dpooled || head) && w2 && y1 && dz1,
               "gcnx_gcn_conv_bwd_pool: NULL pointer");
  GCNX_REQUIRE(ctx, (head || (lddp >= f2 && lddp % 4 == 0 && fal16(dpooled))) && ldy1 >= f1 && ldy1 % 4 == 0 && lddz1 >= f1 &&
                        lddz1 % 4 == 0 && fal16(y2) && fal16(y1) && fal16(dz1) &&
                        (!dz2 || (fal16(dz2) && lddz2 >= f2 && lddz2 % 4 == 0)),
               "gcnx_gcn_conv_bwd_pool: operands must be 16-byte aligned with leading dimensions in multiples of 4 floats");
  if (head) {
    GCNX_REQUIRE(ctx, head->tile_part && head->tile_cnt && head->pool_sum && head->pool_cnt && head->w && head->y &&
                          head->tile_rows >= tiles_of(n) + b && head->tile_rows * f2 * 4 < 0xFFFFFFF0ll && head->b == b &&
                          head->h == f2 && head->c >= 1 && head->denom > 0.f && fal16(head->tile_part) && fal16(head->tile_cnt) &&
                          fal16(head->pool_sum) && fal16(head->pool_cnt) && head->pool_mode == mode &&
                          (head->cce_mode == GCNX_CCE_PROBS || head->cce_mode == GCNX_CCE_LOGITS),
                 "gcnx_gcn_conv_bwd_pool: inconsistent head arguments");
    if (head->c > 2)
      return gcnx_fail(ctx, GCNX_ERR_UNSUPPORTED, "gcnx_gcn_conv_bwd_pool: the in-kernel head serves one or two classes (got %d)", head->c);
  }
  GCNX_REQUIRE(ctx, y2 != dz1 && y2 != dz2 && y1 != dz1, "gcnx_gcn_conv_bwd_pool: outputs must not alias the saved activations");
  const int64_t tiles = gcnx_cdiv(n, kFRows);
  float* colpart = nullptr;
  bool defer = false;
  if (db1) {
    GCNX_REQUIRE(ctx, fal16(db1) && f1 % 4 == 0, "gcnx_gcn_conv_bwd_pool: db1 must be 16-byte aligned");
    defer = pending && scratch && fal16(scratch) && scratch_floats >= tiles * f1 && tiles <= 4096;
    if (defer) colpart = scratch;
    else {
      int rc = gcnx_ws_reserve(ctx, gcnx_colsum_partials_ws(tiles, f1));
      if (rc) return rc;
      colpart = (float*)ctx->ws;
    }
  }
  FusedArgs a{};
  a.rowptr = rowptr_t; a.colidx = colidx_t; a.vals = vals_t; a.x = y2; a.ldx = ldy2; a.n = n; a.w = w2; a.ldw = w2_transposed ? f1 : f2; a.w_t = w2_transposed ? 1 : 0; a.nc = f1;
  a.out = dz1; a.ldo = lddz1; a.node_graph = node_graph; a.gp = graph_ptr; a.dp = dpooled; a.lddp = lddp;
  a.avg = mode == GCNX_POOL_AVG ? 1 : 0; a.mask = y1; a.ldmask = ldy1; a.dz2 = dz2; a.lddz2 = lddz2; a.colpart = colpart;
  if (head) {
    a.hd_part = head->tile_part; a.hd_cnt = head->tile_cnt; a.hd_rows = (int)head->tile_rows; a.hd_b = head->b;
    a.hd_psum = head->pool_sum; a.hd_csum = head->pool_cnt; a.hd_w = head->w; a.hd_bias = head->bias; a.hd_y = head->y;
    a.hd_c = head->c; a.hd_denom = head->denom; a.hd_fl = head->cce_mode == GCNX_CCE_LOGITS ? 1 : 0;
  }
  int rc = launch_fused<true>(ctx, a, f2, prec == GCNX_PREC_BF16X3);
  if (rc) return rc;
  if (db1) {
    if (defer) *pending = gcnx_pending_reduce{colpart, tiles, f1, db1, nullptr, 0, 0, nullptr};
    else return gcnx_colsum_partials(ctx, tiles, f1, db1);
  }
  return GCNX_OK;
}

}  // extern "C"
