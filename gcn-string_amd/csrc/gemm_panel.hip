// Split-bf16 weight GEMMs for mid-size batches (r3): out[M, Nc] (+)= A[M, K] * B[K, Nc] with B = W (Dense forward, K1 of
// MatMul + BiasAdd under gcn.py:334) or W^T (MatMul grad wrt the input, gcn.py:337) -- the shapes of the
// reference's live model GeneralGNN (gcn.py:320) at an E. coli-sized batch: M = 22 576 rows, K up to 1280, 256-column panels.
//
// Why a third GEMM family.  At this size the fp32 MFMA kernels are compute-bound at 47-86 us per product (23 products per
// step: 98 GFLOP against a 157 TFLOP/s peak) and the tiled bf16 kernel of gemm.hip is latency-bound at 50 us whatever K
// is (352 workgroups, load -> LDS -> MFMA in sequence); the streaming kernels of gemm_stream.hip need 32 k+ rows and
// K = 256.  The products are HBM-bound once the arithmetic is bf16 MFMA: 46 MB (K = 256) .. 115 MB (K = 1024) -> 9 .. 23 us.
//
// Structure (the row-tile kernel of gemm.hip with bf16 planes):
//   * one 512-thread workgroup per CU (two waves per SIMD: 256 VGPRs each); wave w owns output columns [32 w, 32 w + 32) of
//     a 256-column panel (blockIdx.y) and keeps its slice of the weight operand in REGISTERS: per 256-wide K panel 2 x 8
//     MFMA B fragments of 16 x 32 (bf16 hi, + lo for bf16x3): 128 VGPRs, loaded as whole 1-KiB lines from a weight IMAGE in fragment order that ONE batched launch
//     prepares for every layer of a model per step (gcnx_wimage_prepare);
//   * the streamed operand goes global (fp32, range-checked buffer loads) -> registers -> bf16 hi / lo -> LDS in MFMA
//     A-fragment order (a wave's read is lane-linear: conflict-free ds_read_b128), 32-row tiles, double-buffered: the rows
//     of step s + 1 fly under the MFMAs of step s, one barrier per step;
//   * K > 256: the K panels are the OUTER loop and the accumulators of the workgroup's (up to three) row tiles stay in
//     registers across them, so the weight slice is loaded once per (workgroup, K panel) and the output is written once;
//   * epilogue from the accumulator layout: bias, accumulate (skip-connection gradients), and -- for the Dense -> BatchNorm
//     pairs of GeneralGNN -- the batch-norm statistics of what is written, as (count, mean, M2) of the workgroup's own
//     rows per column (a two-pass variance in registers); gcnx_bn_finalize_parts combines them with Chan's formula,
//     in workgroup order: the moments of tf.nn.moments without a pass over the Dense output.
#include <cstdint>

#include "common.h"

namespace {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x4 __attribute__((ext_vector_type(4)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));
typedef int pi32x4 __attribute__((ext_vector_type(4)));

constexpr int kPanel = 256;          // column panel
constexpr int kKPanel = 128;         // K panel: 4 MFMA k steps of 32 (the weight slice of 8 would not leave room for the accumulators)
constexpr int kKS = kKPanel / 32;
constexpr int kTileRows = 32;
constexpr int kTilesPerWg = 4;       // row tiles whose accumulators a workgroup keeps across the K panels

struct WimageJob {
  const float* w;
  __bf16* img;
  int fi, fo, transpose, np;
  long long elems;                   // of this job's image
};
constexpr int kMaxJobs = 16;
struct WimageJobs { WimageJob j[kMaxJobs]; int n; };

// image[(((pc * nkp + kp) * 16 + wave) * np + plane) * 4 + ks][lane][j] = B[kp * 128 + ks * 32 + 8 (lane >> 4) + j][pc * 256 + 16 wave + (lane & 15)]
// B = W [fi, fo] (transpose = 0: K = fi, columns = fo) or W^T (transpose = 1: K = fo, columns = fi); zero beyond K / columns.
__global__ __launch_bounds__(256) void wimage_prepare_kernel(WimageJobs jobs) {
  const WimageJob& jb = jobs.j[blockIdx.y];
  if (blockIdx.y >= jobs.n) return;
  const int K = jb.transpose ? jb.fo : jb.fi, ncols = jb.transpose ? jb.fi : jb.fo;
  const int nkp = (K + kKPanel - 1) / kKPanel;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < jb.elems; idx += (long long)gridDim.x * 256) {
    long long r = idx;
    const int j = (int)(r % 8); r /= 8;
    const int lane = (int)(r % 64); r /= 64;
    const int ks = (int)(r % kKS); r /= kKS;
    const int plane = (int)(r % jb.np); r /= jb.np;
    const int wave = (int)(r % 16); r /= 16;
    const int kp = (int)(r % nkp); r /= nkp;
    const int pc = (int)r;
    const int k = kp * kKPanel + ks * 32 + 8 * (lane >> 4) + j, col = pc * kPanel + 16 * wave + (lane & 15);
    float v = 0.f;
    if (k < K && col < ncols) v = jb.transpose ? jb.w[(long long)col * jb.fo + k] : jb.w[(long long)k * jb.fo + col];
    const __bf16 hi = (__bf16)v;
    jb.img[idx] = plane == 0 ? hi : (__bf16)(v - (float)hi);
  }
}

struct PanelEpi {
  const float* bias;      // [nc] or NULL
  int accumulate;         // out += product (skip-connection gradients)
  float* bn_parts;        // [nparts][3][nc]: rows counted, mean, M2 of the workgroup's rows per column, or NULL
};

template <int NP>
__global__ __launch_bounds__(512, 2) void gemm_panel_kernel(const float* __restrict__ a, int64_t lda, const __bf16* __restrict__ img,
                                                            float* __restrict__ c, int64_t ldc, int64_t M, int K, int nc,
                                                            PanelEpi ep, int ntiles) {
  constexpr int TPW = kTilesPerWg;
  constexpr int KS = kKS;
  constexpr int BUF = NP * KS * 2 * 1024;                // bytes of one A buffer: planes x k steps x row tiles x 1 KiB
  extern __shared__ __attribute__((aligned(16))) char plds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // 8 waves, two per SIMD
  const int pc = blockIdx.y, gx = gridDim.x, bx = blockIdx.x;
  const int col0 = pc * kPanel + 32 * wave + (lane & 15);             // the wave's two column tiles: col0, col0 + 16
  bool on[2];
  float bcol[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    on[ct] = pc * kPanel + 32 * wave + 16 * ct < nc;                  // (nc is a multiple of 16: whole column tiles)
    bcol[ct] = (ep.bias && on[ct]) ? ep.bias[col0 + 16 * ct] : 0.f;
  }
  const int nkp = (K + kKPanel - 1) / kKPanel;
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a, (short)0, (int)((uint64_t)M * (uint64_t)lda * 4u), 0x00020000);
  const unsigned lda4 = (unsigned)lda * 4u;
  // this thread's two float4 pieces of a 32 x 128 fp32 tile: piece i = tid + 512 q -> row (tid >> 5) + 16 q, k = 4 (tid & 31)
  const int prow0 = tid >> 5, pc4 = tid & 31;
  // ... and where their bf16 halves go in the A image: block (k step c4 / 8, row tile q), lane 16 ((c4 % 8) / 2) + row % 16,
  // element 4 (c4 & 1)
  // Bank swizzle: the 16 lanes of a ds_write_b64 group hold (k step parity, k quarter, half) = 16 pieces of ONE row: unswizzled
  // they land 256 bytes apart -- 4 of 32 banks, 8-way.  The row index within the 16-row tile is XORed with
  // (k quarter | k step parity << 2): the 8 (quarter, parity) pairs then write 8 different 16-byte slots, both halves of
  // each, all 32 banks once.  A reading lane l = 16 quarter + row applies the same XOR (still one slot per lane: a
  // permutation inside the 1-KiB block, so the ds_read_b128 stays conflict-free).
  const unsigned st_base = (unsigned)(((pc4 >> 3) * 2) * 1024 +
                                      (16 * ((pc4 & 7) >> 1) + (prow0 ^ (((pc4 & 7) >> 1) | (((pc4 >> 3) & 1) << 2)))) * 16 + (pc4 & 1) * 8);
  const unsigned rd_even = (unsigned)(((lane & 48) | ((lane & 15) ^ (lane >> 4))) * 16);
  const unsigned rd_odd = (unsigned)(((lane & 48) | ((lane & 15) ^ ((lane >> 4) | 4))) * 16);

  for (int g0 = 0; g0 * TPW * gx < ntiles; ++g0) {
    pf32x4 acc[TPW][2][2];                                 // [tile][row tile][column tile]
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) acc[i][rt][0] = acc[i][rt][1] = pf32x4{0.f, 0.f, 0.f, 0.f};
    const int nsteps = nkp * TPW;
    // Prefetch ring, TPW slots deep (slot of step s = s % TPW = its tile index: static in the unrolled loop): the rows of
    // step s + 3 are requested while step s computes -- 48 KiB in flight per CU.  One step ahead (16 KiB per CU) the
    // kernel ran at the latency bound 16 KiB / ~1.5 us x 256 CUs = 1.9 TB/s whatever K was.
    static_assert(TPW == 4, "the prefetch ring is indexed by the tile");
    float4 pa[TPW][2];
    auto fetch = [&](int s, float4 (&pa)[2]) {             // rows of step s (K panel s / TPW, tile s % TPW) -> registers
      const int kp = s / TPW, t = (g0 * TPW + s % TPW) * gx + bx;
      const int k = kp * kKPanel + 4 * pc4;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t row = (int64_t)t * kTileRows + prow0 + 16 * q;
        const bool ok = s < nsteps && t < ntiles && row < M && k < K;
        const unsigned off = ok ? (unsigned)row * lda4 + (unsigned)k * 4u : 0xFFFFFFF0u;
        const pf32x4 v = __builtin_bit_cast(pf32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, 0, 0));
        pa[q] = make_float4(v.x, v.y, v.z, v.w);
      }
    };
    auto stash = [&](int buf, const float4 (&pa)[2]) {     // registers -> bf16 planes in LDS, A-fragment order
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float f[4] = {pa[q].x, pa[q].y, pa[q].z, pa[q].w};
        pbf16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { hi[j] = (__bf16)f[j]; lo[j] = (__bf16)(f[j] - (float)hi[j]); }
        char* dst = plds + buf * BUF + st_base + q * 1024;  // (row + 16: the other row tile = the next 1-KiB block)
        *reinterpret_cast<pbf16x4*>(dst) = hi;
        if (NP == 2) *reinterpret_cast<pbf16x4*>(dst + KS * 2 * 1024) = lo;
      }
    };
    fetch(0, pa[0]); fetch(1, pa[1]); fetch(2, pa[2]);
    stash(0, pa[0]);
    __syncthreads();
    pbf16x8 wh[2][KS], wl[2][KS];
    for (int kp = 0; kp < nkp; ++kp) {
      const int ksn = min(KS, (K - kp * kKPanel + 31) / 32);  // k steps of this panel that hold data
      // the wave's weight slices of this K panel: per column tile 4 (+ 4) lines of 1 KiB from the image
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const __bf16* wi = img + ((((size_t)pc * nkp + kp) * 16 + 2 * wave + ct) * NP) * (KS * 64 * 8) + lane * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          wh[ct][ks] = *reinterpret_cast<const pbf16x8*>(wi + ks * 512);
          if (NP == 2) wl[ct][ks] = *reinterpret_cast<const pbf16x8*>(wi + KS * 512 + ks * 512);
        }
      }
#pragma unroll
      for (int i = 0; i < TPW; ++i) {                       // (unrolled: static accumulator indices)
        const int s = kp * TPW + i;
        fetch(s + 3, pa[(i + 3) % TPW]);                    // three steps ahead
        const int t = (g0 * TPW + i) * gx + bx;
        if (on[0] && t < ntiles) {
          const char* ab0 = plds + (s & 1) * BUF;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            if (ks < ksn) {
              pbf16x8 ah[2], al[2];
              const char* ab = ab0 + ((ks & 1) ? rd_odd : rd_even);
#pragma unroll
              for (int rt = 0; rt < 2; ++rt) {
                ah[rt] = *reinterpret_cast<const pbf16x8*>(ab + (ks * 2 + rt) * 1024);
                if (NP == 2) al[rt] = *reinterpret_cast<const pbf16x8*>(ab + (KS * 2 + ks * 2 + rt) * 1024);
              }
              // term-major: four independent accumulators between two MFMAs of the same one
              if (NP == 2) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                  for (int ct = 0; ct < 2; ++ct)
                    acc[i][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], wh[ct][ks], acc[i][rt][ct], 0, 0, 0);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                  for (int ct = 0; ct < 2; ++ct)
                    acc[i][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], wl[ct][ks], acc[i][rt][ct], 0, 0, 0);
              }
#pragma unroll
              for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                  acc[i][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], wh[ct][ks], acc[i][rt][ct], 0, 0, 0);
            }
          }
        }
        stash((s + 1) & 1, pa[(i + 1) % TPW]);
        __syncthreads();
      }
    }
    // epilogue: bias, accumulate, store; batch-norm partials of the rows this workgroup wrote
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int col = col0 + 16 * ct;
      float cnt = 0.f, sum = 0.f;
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int t = (g0 * TPW + i) * gx + bx;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = (int64_t)t * kTileRows + 16 * rt + 4 * (lane >> 4) + r;
            const bool ok = on[ct] && t < ntiles && row < M;
            float v = acc[i][rt][ct][r] + bcol[ct];
            if (ok && ep.accumulate) v += c[row * ldc + col];
            if (ok) { c[row * ldc + col] = v; cnt += 1.f; sum += v; }
            acc[i][rt][ct][r] = v;
          }
      }
      if (ep.bn_parts) {
        // (count, mean, M2) per column over this workgroup's rows: the four lane groups of a wave hold the rows of a
        // column; two-pass in registers, fixed shuffle order
        cnt += __shfl_xor(cnt, 16); cnt += __shfl_xor(cnt, 32);
        sum += __shfl_xor(sum, 16); sum += __shfl_xor(sum, 32);
        const float mean = cnt > 0.f ? sum / cnt : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int t = (g0 * TPW + i) * gx + bx;
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int64_t row = (int64_t)t * kTileRows + 16 * rt + 4 * (lane >> 4) + r;
              if (on[ct] && t < ntiles && row < M) { const float d = acc[i][rt][ct][r] - mean; m2 += d * d; }
            }
        }
        m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
        if (on[ct] && lane < 16) {
          float* p = ep.bn_parts + (size_t)(g0 * gx + bx) * 3 * nc;
          p[col] = cnt; p[nc + col] = mean; p[2 * nc + col] = m2;
        }
      }
    }
    __syncthreads();                                        // (the next group's first stash overwrites buffer 0)
  }
}

// K <= 256, many column panels (dH W^T of GeneralGNN's wide layers: K = 256, 512 .. 1024 output columns): the A operand of
// a workgroup's row tiles is converted ONCE and stays in LDS (3 tiles x 2 K panels x 16 KiB), the column panels are walked
// INSIDE the workgroup -- per panel: the wave's weight slices of both K panels (128 VGPRs), 3 x 2 x 4 k steps of MFMAs
// straight out of LDS, the accumulate epilogue.  No barrier after the prologue, so the two waves of a SIMD drift apart and
// cover each other's weight-load and epilogue stalls.  Against one launch of gemm_panel_kernel per column panel (grid.y):
// the rows are fetched, split and staged once instead of once per panel.
constexpr int kWideTiles = 3;
template <int NP>
__global__ __launch_bounds__(512, 2) void gemm_panel_wide_kernel(const float* __restrict__ a, int64_t lda, const __bf16* __restrict__ img,
                                                                 float* __restrict__ c, int64_t ldc, int64_t M, int K, int nc,
                                                                 PanelEpi ep, int ntiles) {
  constexpr int TPW = kWideTiles;
  constexpr int KS = kKS;
  constexpr int BUF = NP * KS * 2 * 1024;                // one (tile, K panel) image
  extern __shared__ __attribute__((aligned(16))) char plds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gx = gridDim.x, bx = blockIdx.x;
  const int nkp = (K + kKPanel - 1) / kKPanel;            // 1 or 2
  const int npc = (nc + kPanel - 1) / kPanel;
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a, (short)0, (int)((uint64_t)M * (uint64_t)lda * 4u), 0x00020000);
  const unsigned lda4 = (unsigned)lda * 4u;
  const int prow0 = tid >> 5, pc4 = tid & 31;
  const unsigned st_base = (unsigned)(((pc4 >> 3) * 2) * 1024 +
                                      (16 * ((pc4 & 7) >> 1) + (prow0 ^ (((pc4 & 7) >> 1) | (((pc4 >> 3) & 1) << 2)))) * 16 + (pc4 & 1) * 8);
  const unsigned rd_even = (unsigned)(((lane & 48) | ((lane & 15) ^ (lane >> 4))) * 16);
  const unsigned rd_odd = (unsigned)(((lane & 48) | ((lane & 15) ^ ((lane >> 4) | 4))) * 16);

  for (int g0 = 0; g0 * TPW * gx < ntiles; ++g0) {
    // prologue: every (tile, K panel) of this workgroup -> bf16 planes in LDS (all loads in flight, then the conversions)
    float4 pa[TPW][2][2];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) {
        const int t = (g0 * TPW + i) * gx + bx;
        const int k = kp * kKPanel + 4 * pc4;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int64_t row = (int64_t)t * kTileRows + prow0 + 16 * q;
          const bool ok = kp < nkp && t < ntiles && row < M && k < K;
          const unsigned off = ok ? (unsigned)row * lda4 + (unsigned)k * 4u : 0xFFFFFFF0u;
          const pf32x4 v = __builtin_bit_cast(pf32x4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, 0, 0));
          pa[i][kp][q] = make_float4(v.x, v.y, v.z, v.w);
        }
      }
    if (g0 > 0) __syncthreads();                            // (the previous group's reads of LDS are done)
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int kp = 0; kp < 2; ++kp)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float f[4] = {pa[i][kp][q].x, pa[i][kp][q].y, pa[i][kp][q].z, pa[i][kp][q].w};
          pbf16x4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) { hi[j] = (__bf16)f[j]; lo[j] = (__bf16)(f[j] - (float)hi[j]); }
          char* dst = plds + (i * 2 + kp) * BUF + st_base + q * 1024;
          *reinterpret_cast<pbf16x4*>(dst) = hi;
          if (NP == 2) *reinterpret_cast<pbf16x4*>(dst + KS * 2 * 1024) = lo;
        }
    __syncthreads();
    for (int pc = 0; pc < npc; ++pc) {
      const int col0 = pc * kPanel + 32 * wave + (lane & 15);
      bool on[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) on[ct] = pc * kPanel + 32 * wave + 16 * ct < nc;
      if (!on[0]) continue;                                 // (uniform per wave)
      pf32x4 acc[TPW][2][2];
#pragma unroll
      for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc[i][rt][0] = acc[i][rt][1] = pf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kp = 0; kp < 2; ++kp) {
        if (kp < nkp) {
          const int ksn = min(KS, (K - kp * kKPanel + 31) / 32);
          pbf16x8 wh[2][KS], wl[2][KS];
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const __bf16* wi = img + ((((size_t)pc * nkp + kp) * 16 + 2 * wave + ct) * NP) * (KS * 64 * 8) + lane * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              wh[ct][ks] = *reinterpret_cast<const pbf16x8*>(wi + ks * 512);
              if (NP == 2) wl[ct][ks] = *reinterpret_cast<const pbf16x8*>(wi + KS * 512 + ks * 512);
            }
          }
#pragma unroll
          for (int i = 0; i < TPW; ++i) {
            const int t = (g0 * TPW + i) * gx + bx;
            if (t < ntiles) {
              const char* ab0 = plds + (i * 2 + kp) * BUF;
#pragma unroll
              for (int ks = 0; ks < KS; ++ks) {
                if (ks < ksn) {
                  pbf16x8 ah[2], al[2];
                  const char* ab = ab0 + ((ks & 1) ? rd_odd : rd_even);
#pragma unroll
                  for (int rt = 0; rt < 2; ++rt) {
                    ah[rt] = *reinterpret_cast<const pbf16x8*>(ab + (ks * 2 + rt) * 1024);
                    if (NP == 2) al[rt] = *reinterpret_cast<const pbf16x8*>(ab + (KS * 2 + ks * 2 + rt) * 1024);
                  }
                  if (NP == 2) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                      for (int ct = 0; ct < 2; ++ct)
                        acc[i][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[rt], wh[ct][ks], acc[i][rt][ct], 0, 0, 0);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                      for (int ct = 0; ct < 2; ++ct)
                        acc[i][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], wl[ct][ks], acc[i][rt][ct], 0, 0, 0);
                  }
#pragma unroll
                  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                      acc[i][rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[rt], wh[ct][ks], acc[i][rt][ct], 0, 0, 0);
                }
              }
            }
          }
        }
      }
      // epilogue of the panel: bias, accumulate, store
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const int col = col0 + 16 * ct;
        const float bc = (ep.bias && on[ct]) ? ep.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
          const int t = (g0 * TPW + i) * gx + bx;
          float old[2][4];
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int64_t row = (int64_t)t * kTileRows + 16 * rt + 4 * (lane >> 4) + r;
              const bool ok = on[ct] && t < ntiles && row < M;
              old[rt][r] = (ok && ep.accumulate) ? c[row * ldc + col] : 0.f;
            }
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int64_t row = (int64_t)t * kTileRows + 16 * rt + 4 * (lane >> 4) + r;
              if (on[ct] && t < ntiles && row < M) c[row * ldc + col] = acc[i][rt][ct][r] + bc + old[rt][r];
            }
        }
      }
    }
  }
}

// Chan's combination of per-part (count, mean, M2): the batch moments (biased variance, tf.nn.moments) and the Keras
// moving-statistics update.  16 threads per column: thread j folds the parts j, j + 16, ... in order, then the 16
// partial results are merged pairwise in a fixed tree (the merge is associative up to rounding; the order is fixed, so
// the result is reproducible).  A single thread per column walking 256 dependent loads took 140 us.
__device__ __forceinline__ void chan_merge(float& n, float& m, float& m2, float nb, float mb, float m2b) {
  if (nb > 0.f) {
    const float tot = n + nb, d = mb - m;
    m += d * (nb / tot);
    m2 += m2b + d * d * (n * nb / tot);
    n = tot;
  }
}
__global__ __launch_bounds__(256) void bn_finalize_parts_kernel(const float* __restrict__ parts, int nparts, int32_t f, float momentum,
                                                                float eps, float* __restrict__ mean, float* __restrict__ inv,
                                                                float* __restrict__ moving_mean, float* __restrict__ moving_var) {
  const int j = threadIdx.x & 15;
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool col_ok = c < f;
  float n = 0.f, m = 0.f, m2 = 0.f;
  if (col_ok)
    for (int p0 = j; p0 < nparts; p0 += 16 * 8) {          // eight parts' loads in flight, merged in part order
      float nb[8], mb[8], m2b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = p0 + 16 * u;
        const float* q = parts + (size_t)(p < nparts ? p : 0) * 3 * f;
        nb[u] = p < nparts ? q[c] : 0.f; mb[u] = q[f + c]; m2b[u] = q[2 * f + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) chan_merge(n, m, m2, nb[u], mb[u], m2b[u]);
    }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) {                // lanes j and j ^ off hold disjoint part sets: the lower one absorbs
    const float nb = __shfl_xor(n, off), mb = __shfl_xor(m, off), m2b = __shfl_xor(m2, off);
    if ((j & off) == 0) chan_merge(n, m, m2, nb, mb, m2b);
  }
  if (!col_ok || j != 0) return;
  const float v = n > 0.f ? fmaxf(m2 / n, 0.f) : 0.f;
  if (moving_mean) {
    moving_mean[c] = momentum * moving_mean[c] + (1.f - momentum) * m;
    moving_var[c] = momentum * moving_var[c] + (1.f - momentum) * v;
  }
  mean[c] = m;
  inv[c] = 1.0f / sqrtf(v + eps);
}

inline bool pal16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

int64_t gcnx_wimage_elems(int32_t fi, int32_t fo, int transpose, int prec) {
  if (fi <= 0 || fo <= 0 || prec == GCNX_PREC_F32) return 0;
  const int K = transpose ? fo : fi, ncols = transpose ? fi : fo;
  const int np = prec == GCNX_PREC_BF16X3 ? 2 : 1;
  return (int64_t)((ncols + kPanel - 1) / kPanel) * ((K + kKPanel - 1) / kKPanel) * np * kPanel * kKPanel;
}

int gcnx_wimage_prepare(gcnx_ctx* ctx, int32_t njobs, const gcnx_wimage_job* jobs) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, "weight images (bf16 fragments)");
  GCNX_REQUIRE(ctx, njobs >= 0 && (njobs == 0 || jobs), "gcnx_wimage_prepare: bad job list");
  for (int base = 0; base < njobs; base += kMaxJobs) {
    WimageJobs js{};
    js.n = njobs - base < kMaxJobs ? njobs - base : kMaxJobs;
    long long most = 0;
    for (int i = 0; i < js.n; ++i) {
      const gcnx_wimage_job& j = jobs[base + i];
      GCNX_REQUIRE(ctx, j.w && j.img && j.fi > 0 && j.fo > 0, "gcnx_wimage_prepare: job %d: NULL pointer / empty matrix", base + i);
      GCNX_REQUIRE(ctx, j.prec == GCNX_PREC_BF16 || j.prec == GCNX_PREC_BF16X3, "gcnx_wimage_prepare: job %d: precision %d has no image", base + i, j.prec);
      GCNX_REQUIRE(ctx, pal16(j.img), "gcnx_wimage_prepare: job %d: the image must be 16-byte aligned", base + i);
      const long long e = gcnx_wimage_elems(j.fi, j.fo, j.transpose, j.prec);
      js.j[i] = WimageJob{j.w, (__bf16*)j.img, j.fi, j.fo, j.transpose ? 1 : 0, j.prec == GCNX_PREC_BF16X3 ? 2 : 1, e};
      most = e > most ? e : most;
    }
    const int gx = (int)((most / 8 + 255) / 256 < 2048 ? (most / 8 + 255) / 256 : 2048);
    hipLaunchKernelGGL(wimage_prepare_kernel, dim3(gx > 0 ? gx : 1, js.n), dim3(256), 0, ctx->stream, js);
    GCNX_LAUNCH_OK(ctx);
  }
  return GCNX_OK;
}

int gcnx_gemm_wimage(gcnx_ctx* ctx, const float* x, int64_t ldx, const void* img, int32_t fi, int32_t fo, int transpose,
                     const float* bias, float* out, int64_t ldo, int64_t n, int prec, int accumulate, float* bn_parts,
                     int32_t* n_parts) {
  GCNX_CHECK_CTX(ctx);
  GCNX_RANGE(ctx, transpose ? "weight GEMM (dX, image)" : "weight GEMM (X W, image)");
  GCNX_REQUIRE(ctx, n >= 0 && fi > 0 && fo > 0, "gcnx_gemm_wimage: bad size");
  GCNX_REQUIRE(ctx, prec == GCNX_PREC_BF16 || prec == GCNX_PREC_BF16X3, "gcnx_gemm_wimage: precision %d has no weight image", prec);
  const int K = transpose ? fo : fi, nc = transpose ? fi : fo;
  if (n_parts) *n_parts = 0;
  if (n == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, x && img && out, "gcnx_gemm_wimage: NULL pointer");
  GCNX_REQUIRE(ctx, ldx >= K && ldo >= nc, "gcnx_gemm_wimage: leading dimension too small");
  // what the kernel is built for (the caller falls back to gcnx_gemm / gcnx_gemm_dx otherwise; no message: an answer)
  if (nc % 16 != 0 || K % 4 != 0 || ldx % 4 != 0 || !pal16(x) || !pal16(img) || (uint64_t)n * (uint64_t)ldx * 4u >= 0xFFFFFF00ull)
    return GCNX_ERR_UNSUPPORTED;
  GCNX_REQUIRE(ctx, !bn_parts || n_parts, "gcnx_gemm_wimage: bn_parts needs n_parts");
  const int ntiles = gcnx_cdiv(n, kTileRows);
  const int gx = ntiles < ctx->num_cus ? ntiles : ctx->num_cus;
  const int groups = gcnx_cdiv(ntiles, kTilesPerWg * gx);
  GCNX_REQUIRE(ctx, !bn_parts || nc <= kPanel, "gcnx_gemm_wimage: batch-norm statistics need a single column panel (fo <= 256)");
  if (n_parts) *n_parts = groups * gx;
  const PanelEpi ep{bias, accumulate ? 1 : 0, bn_parts};
  if (K <= 2 * kKPanel && nc > kPanel && !bn_parts) {      // wide outputs of a short reduction: the A operand resident in LDS
    const int lds_np = prec == GCNX_PREC_BF16X3 ? 2 : 1;
    const int lds = kWideTiles * 2 * lds_np * kKS * 2 * 1024;
    if (prec == GCNX_PREC_BF16X3) {
      static bool set = false;
      if (!set) { GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_panel_wide_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); set = true; }
      hipLaunchKernelGGL((gemm_panel_wide_kernel<2>), dim3(gx), dim3(512), lds, ctx->stream, x, ldx, (const __bf16*)img, out, ldo, n, K, nc, ep, ntiles);
    } else {
      static bool set = false;
      if (!set) { GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_panel_wide_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); set = true; }
      hipLaunchKernelGGL((gemm_panel_wide_kernel<1>), dim3(gx), dim3(512), lds, ctx->stream, x, ldx, (const __bf16*)img, out, ldo, n, K, nc, ep, ntiles);
    }
    GCNX_LAUNCH_OK(ctx);
    return GCNX_OK;
  }
  const dim3 grid(gx, gcnx_cdiv(nc, kPanel));
  if (prec == GCNX_PREC_BF16X3) {
    constexpr int lds = 2 * 2 * kKS * 2 * 1024;
    static bool set = false;
    if (!set) { GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_panel_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); set = true; }
    hipLaunchKernelGGL((gemm_panel_kernel<2>), grid, dim3(512), lds, ctx->stream, x, ldx, (const __bf16*)img, out, ldo, n, K, nc, ep, ntiles);
  } else {
    constexpr int lds = 2 * 1 * kKS * 2 * 1024;
    hipLaunchKernelGGL((gemm_panel_kernel<1>), grid, dim3(512), lds, ctx->stream, x, ldx, (const __bf16*)img, out, ldo, n, K, nc, ep, ntiles);
  }
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

int64_t gcnx_gemm_wimage_parts(gcnx_ctx* ctx, int64_t n) {
  if (!ctx || n <= 0) return 0;
  const int ntiles = gcnx_cdiv(n, kTileRows);
  const int gx = ntiles < ctx->num_cus ? ntiles : ctx->num_cus;
  return (int64_t)gcnx_cdiv(ntiles, kTilesPerWg * gx) * gx;
}

int gcnx_bn_finalize_parts(gcnx_ctx* ctx, const float* parts, int32_t nparts, int32_t f, float momentum, float eps, float* mean,
                           float* inv, float* moving_mean, float* moving_var) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, f >= 0 && nparts >= 0, "gcnx_bn_finalize_parts: negative size");
  if (f == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, parts && mean && inv && nparts > 0, "gcnx_bn_finalize_parts: NULL pointer / no parts");
  GCNX_REQUIRE(ctx, (moving_mean == nullptr) == (moving_var == nullptr), "gcnx_bn_finalize_parts: pass both moving buffers or none");
  hipLaunchKernelGGL(bn_finalize_parts_kernel, dim3(gcnx_cdiv(f, 16)), dim3(256), 0, ctx->stream, parts, nparts, f, momentum, eps, mean, inv,
                     moving_mean, moving_var);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // extern "C"
