// Streaming bf16 / bf16x3 weight GEMM for tall activations (BASELINE config 3: N = 10^6 rows, F = 256):
//   out[N, Fo] = act(X[N, Fi] W[Fi, Fo] + b)      and      dX[N, Fi] = dH[N, Fo] W^T (* relu mask)
// -- MatMul / MatMul-grad of GCNConv.kernel reached from gcn.py:334 / 337 -- on v_mfma_f32_16x16x32_bf16.
//
// At these shapes the product is HBM-bound (2.05 GB in + out against 131 GFLOP: AI 64 flop/B, machine balance ~300),
// so the kernel is built around the activation stream, not around the tile:
//   * the weight operand -- bf16 hi (and lo) planes of W, at most 128 KiB -- is copied into LDS ONCE per workgroup
//     (LDS-DMA, lane-linear image: every MFMA fragment is one conflict-free ds_read_b128) and stays there while the
//     workgroup walks row blocks; one 512-thread workgroup per CU.
//   * the activation rows never touch LDS: a wave owns 32 rows, and each lane loads the 8 consecutive k of "its"
//     row straight into MFMA operand layout (two dwordx4 through a range-checked buffer resource: rows past the end
//     read zeros, no branches), FOUR K steps ahead of their use -- 128 KiB in flight per CU, which is what an HBM
//     miss under load takes to hide.  fp32 -> bf16 (hi, lo) happens in registers on the way into the MFMA.
//   * the MFMA is issued transposed (A = weight fragment, B = activation fragment), so a lane ends up holding FOUR
//     CONSECUTIVE output columns of one row: the epilogue stores float4s directly (bias / PReLU slopes from LDS, the
//     ReLU mask of the fused activation gradient as 64 bits per lane collected by the same prefetch ring).
//   * bf16x3 needs hi and lo planes (256 KiB for 256 x 256): a workgroup then owns HALF of the output columns; the two
//     workgroups of a pair sit on one XCD and walk the same row blocks, so the second read of a row block is an L2
//     (at worst Infinity-Cache) hit and HBM still sees every activation byte about once.
#include "common.h"

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef int si32x4 __attribute__((ext_vector_type(4)));

#ifndef GCNX_STREAM_STORE_AUX
#define GCNX_STREAM_STORE_AUX 0     // cache policy of the output stores (0 plain, 2 nt, 16 sc1)
#endif
#ifndef GCNX_STREAM_LOAD_AUX
#define GCNX_STREAM_LOAD_AUX 0
#endif
#ifndef GCNX_STREAM_ABL
#define GCNX_STREAM_ABL 0           // timing-only ablation (tuning builds): 1 = activation loads dropped (zero-record
#endif                              // descriptor: no fetch, same instruction stream), 2 = output stores dropped, 3 = both

namespace {

constexpr int kSRows = 32;                 // rows per wave
constexpr int kSWaves = 8;                 // waves per workgroup
// K steps of activation loads in flight per wave: 4 (128 KiB per CU); with a fused ReLU mask the mask pieces ride in
// the same ring, and two steps of activations + mask (96 KiB per CU) are what fits 256 registers without spills.
// MASK: 0 none, 1 the saved fp32 activation [M, ldmask] (its > 0 bits are collected on the way), 2 a BIT IMAGE of it
// written by the forward launch that produced the activation (ep.bits_out): 32 bytes per row instead of 1 KiB -- one
// 64-bit word per lane and row block, in this kernel's own lane order (word (row, q) = the nibbles of column tiles
// 0 .. 15 for the lane's 4-column group q), so it is private to the two launches -- which must have the same shape.
template <int MASK> struct StreamDepth { static constexpr int value = MASK == 1 ? 2 : 4; };

struct StreamEpi {
  const float* bias;      // [ncol] or null
  const float* alpha;     // PReLU slopes or null
  const float* mask;      // relu-mask source [M, ldmask] or null
  int64_t ldmask;
  int act;
  int accumulate;
  float* colpart;         // [pairs, ncol] column sums of what each workgroup wrote (BiasAddGrad partials) or null
  const unsigned long long* mbits_in;   // MASK == 2: the bit image [M][4]
  unsigned long long* bits_out;         // forward: write the bit image of (output > 0) here (RT == 1 only) or null
};

// Weight operand -> bf16 planes in MFMA-fragment order.  Element (col, k) of plane p of column half h lives at
//   ((((h * NP + p) * ksteps + k / 32) * (CW / 16) + (col % CW) / 16) * 4 + (k % 32) / 8) * 16 + col % 16) * 8 + k % 8
// so that the fragment of (k step, column tile) is 1 KiB, lane-linear: lane l = 16 q + c reads bytes [16 l, 16 l + 16).
// transpose: the product's k index runs over W's ROWS (X W: image col = W column, k = W row); otherwise over W's
// columns (dH W^T: image col = W row, k = W column).
__global__ __launch_bounds__(256) void stream_wprep_kernel(const float* __restrict__ w, int fi, int fo, int transpose, int np,
                                                           int cw, int ksteps, int ncol, __bf16* __restrict__ img) {
  const int K = transpose ? fi : fo;
  const int64_t per_half = (int64_t)np * ksteps * cw * 32;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int halves = (ncol + cw - 1) / cw;
  if (idx >= per_half * halves) return;
  const int h = (int)(idx / per_half);
  int64_t r = idx % per_half;
  const int p = (int)(r / ((int64_t)ksteps * cw * 32)); r %= (int64_t)ksteps * cw * 32;
  const int ks = (int)(r / (cw * 32)); r %= cw * 32;
  const int ct = (int)(r / 512); r %= 512;
  const int q = (int)(r / 128); r %= 128;
  const int c = (int)(r / 8), j = (int)(r % 8);
  const int col = h * cw + ct * 16 + c, k = ks * 32 + q * 8 + j;
  float v = 0.f;
  if (col < ncol && k < K) v = transpose ? w[(int64_t)k * fo + col] : w[(int64_t)col * fo + k];
  const __bf16 hi = (__bf16)v;
  img[idx] = p == 0 ? hi : (__bf16)(v - (float)hi);
}

// The one-plane 256 x 256 images of several weight operands in ONE launch (gcnx_gemm_stream_images): blockIdx.y = job.
struct StreamImageJobs { const float* w[4]; __bf16* img[4]; int transpose[4]; };
__global__ __launch_bounds__(256) void stream_wprep_multi_kernel(StreamImageJobs jobs) {
  const int j = blockIdx.y;
  const float* __restrict__ w = jobs.w[j];
  __bf16* __restrict__ img = jobs.img[j];
  const int transpose = jobs.transpose[j];
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // < 8 * 256 * 32: the layout of stream_wprep_kernel, np = 1
  int64_t r = idx;
  const int ks = (int)(r / (256 * 32)); r %= 256 * 32;
  const int ct = (int)(r / 512); r %= 512;
  const int q = (int)(r / 128); r %= 128;
  const int c = (int)(r / 8), jj = (int)(r % 8);
  const int col = ct * 16 + c, k = ks * 32 + q * 8 + jj;
  img[idx] = (__bf16)(transpose ? w[(int64_t)k * 256 + col] : w[(int64_t)col * 256 + k]);
}

__device__ __forceinline__ float4 sbuf4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  const sf32x4 r = __builtin_bit_cast(sf32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, GCNX_STREAM_LOAD_AUX));
  return make_float4(r.x, r.y, r.z, r.w);
}

__device__ __forceinline__ void split8(const float4& a, const float4& b, sbf16x8& hi, sbf16x8& lo, bool want_lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    hi[j] = (__bf16)v[j];
    if (want_lo) lo[j] = (__bf16)(v[j] - (float)hi[j]);
  }
}

// NP planes (1: bf16, 2: bf16x3), CW output columns per workgroup, RT 16-row tiles per wave, KSTEPS = K / 32.
//   bf16x3: NP = 2, CW = 128, RT = 2 (two column halves on XCD neighbours);  bf16: NP = 1, CW = 256, RT = 1 (one pass).
// The K loop is software-pipelined by hand, one block of two column tiles at a time (sched_barrier between blocks keeps
// hipcc from re-serialising it): the weight fragments of block b + 1 are read from LDS and a quarter of the NEXT K
// step's activations is converted fp32 -> bf16 (hi, lo) while the MFMAs of block b issue; the activation loads that
// replace the converted ring slot go out at the end of the step.
// IN16 / OUT16 (r3, plain bf16 only): the streamed operand is already stored as bf16 (`a` points at uint16 rows, lda in
// elements) / the result is stored as bf16 (round to nearest even -- exactly the rounding the NEXT weight GEMM would apply
// to an fp32 copy, so a chain of bf16-operand GEMMs computes the same bits with half the traffic).  A bf16 row piece of
// 8 k is ONE 16-byte load that is the MFMA fragment as it stands: no conversion, and the ring is eight K steps deep
// (the same 32 registers, 64 KiB in flight per CU).
template <int NP, int CW, int RT, int KSTEPS, int MASK, bool IN16 = false, bool OUT16 = false>
__global__ __launch_bounds__(512, 2) void gemm_stream_kernel(const float* __restrict__ a, int64_t lda, const __bf16* __restrict__ wimg,
                                                             float* __restrict__ c, int64_t ldc, int64_t M, int ncol, StreamEpi ep,
                                                             int n_rb, int halves) {
  static_assert(!IN16 || (NP == 1 && MASK != 1), "bf16 input: plain bf16 products, bit-image masks");
  constexpr int kSDepth = IN16 ? 8 : StreamDepth<MASK>::value;
  constexpr unsigned AB = IN16 ? 2u : 4u, CB = OUT16 ? 2u : 4u;       // bytes per streamed / stored element
  constexpr int CT = CW / 16;                            // column tiles per workgroup
  constexpr int NB = CT / 2;                             // blocks of two column tiles per K step
  constexpr int NF = 2 * RT;                             // float4 loads per lane and K step
  constexpr int ROWS = 16 * RT;                          // rows per wave
  constexpr int BLOCK = ROWS * kSWaves;                  // rows per workgroup step
  constexpr int IMG = NP * KSTEPS * CW * 32;             // bf16 elements of the LDS image
  static_assert(KSTEPS % kSDepth == 0 && NB % NF == 0 && RT * CT == 16, "pipeline shape");

  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  float* lbias = reinterpret_cast<float*>(lds + IMG);    // [CW] bias, [CW] alpha
  float* wsum = lbias + 2 * CW;                          // [waves][CW] column sums of the rows each wave wrote (ep.colpart)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 15, q = lane >> 4;

  // workgroup -> (column half, pair).  Physical ids b and b + 8 share an XCD (speed only): the two halves of a pair
  // are XCD neighbours and walk the same row blocks.
  int half = 0, pair = blockIdx.x, npairs = gridDim.x;
  if (halves == 2) {
    if ((gridDim.x & 15) == 0) {
      const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
      half = slot & 1;
      pair = (slot >> 1) * 8 + xcd;
    } else {
      half = blockIdx.x & 1;
      pair = blockIdx.x >> 1;
    }
    npairs = gridDim.x >> 1;
  }
  const int c0 = half * CW;

  // the weight image of this column half: one linear LDS-DMA copy, resident for the workgroup's lifetime
  {
    const __bf16* src = wimg + (size_t)half * IMG;
    constexpr int PIECES = IMG / 8;                      // 16-byte pieces
#pragma unroll 4
    for (int i = tid; i < PIECES; i += 512)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * 8),
                                       (__attribute__((address_space(3))) void*)(lds + (size_t)(i - lane) * 8), 16, 0, 0);
    for (int i = tid; i < CW; i += 512) {
      lbias[i] = (ep.bias && c0 + i < ncol) ? ep.bias[c0 + i] : 0.f;
      lbias[CW + i] = (ep.alpha && c0 + i < ncol) ? ep.alpha[c0 + i] : 0.f;
    }
    for (int i = tid; i < kSWaves * CW; i += 512) wsum[i] = 0.f;
  }
  __syncthreads();

  const int nunits = pair < n_rb ? (n_rb - pair + npairs - 1) / npairs : 0;
  const int total = nunits * KSTEPS;
  const __amdgpu_buffer_rsrc_t ars =
      __builtin_amdgcn_make_buffer_rsrc((void*)a, (short)0, (GCNX_STREAM_ABL & 1) ? 0 : (int)((uint64_t)M * (uint64_t)lda * AB), 0x00020000);
  const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(MASK == 1 ? (const void*)ep.mask : MASK == 2 ? (const void*)ep.mbits_in : (const void*)a), (short)0,
      MASK == 1 ? (int)((uint64_t)M * (uint64_t)ep.ldmask * 4u) : MASK == 2 ? (int)((uint64_t)M * (uint64_t)halves * 32u) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t crs =
      __builtin_amdgcn_make_buffer_rsrc((void*)c, (short)0, (GCNX_STREAM_ABL & 2) ? 0 : (int)((uint64_t)M * (uint64_t)ldc * CB), 0x00020000);
  const unsigned lda4 = (unsigned)lda * AB, ldm4 = (unsigned)ep.ldmask * 4u, ldc4 = (unsigned)ldc * CB;     // row strides in bytes
  const int wrow = wave * ROWS + rl;                     // this lane's row inside a block (tile 0; tile t = + 16 t)
  const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(ep.bits_out ? (void*)ep.bits_out : (void*)c), (short)0, ep.bits_out ? (int)((uint64_t)M * (uint64_t)halves * 32u) : 0, 0x00020000);
  // The image: word ((row * halves + half) * 4 + q) = the 16 nibbles (tile tt, column tile ct) of the lane whose FIRST row
  // (tt = 0) is `row` -- with two row tiles per wave only every other 16-row group has words, the array is indexed by
  // the row all the same (halves * 32 bytes per row).
  // MASK == 2: the 64 mask bits of this lane for row block `un_` (rows past M read zeros)
  auto load_bits = [&](int un_) -> unsigned long long {
    const int64_t row_ = (int64_t)(pair + un_ * npairs) * BLOCK + wrow;
    const unsigned off_ = (un_ < nunits && row_ < M) ? ((unsigned)row_ * (unsigned)halves + (unsigned)half) * 32u + (unsigned)q * 8u : 0xFFFFFFE0u;
    typedef unsigned int su32x2 __attribute__((ext_vector_type(2)));
    const su32x2 w_ = __builtin_bit_cast(su32x2, __builtin_amdgcn_raw_buffer_load_b64(mrs, off_, 0, 0));
    return (unsigned long long)w_[0] | ((unsigned long long)w_[1] << 32);
  };

  float4 pf[IN16 ? 1 : kSDepth][NF];                     // activation ring: [step % depth][tile * 2 + (0: k 0..3, 1: k 4..7)]
  si32x4 pq[IN16 ? kSDepth : 1][RT];                     // IN16: [step % depth][tile] = the 8 bf16 of the lane's row piece
  float4 mk[kSDepth][2];                                 // MASK: the mask pieces that ride with the same step

  // Loads of flat step t (unit t / KSTEPS, K step t % KSTEPS).  Range-checked: steps past the end and rows past M
  // get an out-of-range offset and read zeros without a branch.
#define GS_ISSUE(T, SLOT)                                                                                         \
  {                                                                                                                \
    const int t_ = (T);                                                                                            \
    const int un_ = t_ / KSTEPS, ks_ = t_ % KSTEPS;                                                                \
    const int64_t r0_ = (int64_t)(pair + un_ * npairs) * BLOCK + wrow;                                             \
    const bool live_ = t_ < total;                                                                                 \
    _Pragma("unroll") for (int tt = 0; tt < RT; ++tt) {                                                            \
      const int64_t row_ = r0_ + tt * 16;                                                                          \
      const unsigned off_ = (live_ && row_ < M) ? (unsigned)row_ * lda4 + (unsigned)(ks_ * 32 + q * 8) * AB : 0xFFFFFFE0u; \
      if constexpr (IN16) {                                                                                        \
        pq[SLOT][tt] = __builtin_bit_cast(si32x4, __builtin_amdgcn_raw_buffer_load_b128(ars, off_, 0, GCNX_STREAM_LOAD_AUX)); \
      } else {                                                                                                     \
        pf[SLOT][tt * 2] = sbuf4(ars, off_);                                                                       \
        pf[SLOT][tt * 2 + 1] = sbuf4(ars, off_ + 16u);                                                             \
      }                                                                                                            \
    }                                                                                                              \
    if (MASK == 1) {                                                                                               \
      _Pragma("unroll") for (int m = 0; m < 2; ++m) {                                                              \
        const int piece_ = ks_ * 2 + m;                   /* 0 .. 15 = tile * CT + column tile */                  \
        const int tt = piece_ / CT, ct = piece_ % CT;                                                              \
        const int64_t row_ = r0_ + tt * 16;                                                                        \
        const bool ok_ = live_ && row_ < M && piece_ < RT * CT && c0 + ct * 16 + q * 4 < ncol;                     \
        const unsigned off_ = ok_ ? (unsigned)row_ * ldm4 + (unsigned)(c0 + ct * 16 + q * 4) * 4u : 0xFFFFFFE0u;   \
        mk[SLOT][m] = sbuf4(mrs, off_);                                                                            \
      }                                                                                                            \
    }                                                                                                              \
  }
  // One float4 (4 consecutive k) of ring slot SLOT -> elements 4 (F & 1) .. + 3 of the bf16 fragments of tile F / 2.
#define GS_CONVERT(SLOT, F, XH, XL)                                                                               \
  if constexpr (IN16) {                                                                                            \
    if (((F) & 1) == 0) XH[(F) / 2] = __builtin_bit_cast(sbf16x8, pq[SLOT][(F) / 2]);                              \
  } else {                                                                                                         \
    const float4 v_ = pf[SLOT][F];                                                                                 \
    const float f_[4] = {v_.x, v_.y, v_.z, v_.w};                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                \
      const __bf16 h_ = (__bf16)f_[j];                                                                             \
      XH[(F) / 2][((F) & 1) * 4 + j] = h_;                                                                         \
      if (NP == 2) XL[(F) / 2][((F) & 1) * 4 + j] = (__bf16)(f_[j] - (float)h_);                                   \
    }                                                                                                              \
  }
  // Weight fragments (hi, lo) of block B of K step KS into WH / WL.
#define GS_WREAD(KS, B, WH, WL)                                                                                   \
  {                                                                                                                \
    const __bf16* wk_ = lds + (size_t)(KS) * CW * 32 + (size_t)(B) * 1024 + wofs;                                  \
    WH[0] = *reinterpret_cast<const sbf16x8*>(wk_);                                                                \
    WH[1] = *reinterpret_cast<const sbf16x8*>(wk_ + 512);                                                          \
    if (NP == 2) {                                                                                                 \
      WL[0] = *reinterpret_cast<const sbf16x8*>(wk_ + (size_t)KSTEPS * CW * 32);                                   \
      WL[1] = *reinterpret_cast<const sbf16x8*>(wk_ + (size_t)KSTEPS * CW * 32 + 512);                             \
    }                                                                                                              \
  }

  // Column sums of what is written (ep.colpart).  One-plane kernels with bit-image masks have the registers to keep them
  // per LANE over the workgroup's whole walk (CT float4s) and to combine lanes, waves and workgroups once at the end; the
  // others (fp32 mask pieces in the ring, two planes: at the register limit) combine the 16 lanes of a tile per unit.
  constexpr bool CSREG = NP == 1 && MASK != 1;
  float4 cs[CSREG ? CT : 1];
#pragma unroll
  for (int j = 0; j < (CSREG ? CT : 1); ++j) cs[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  sf32x4 acc[RT][CT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc[t][j] = sf32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long mbits = 0;                          // MASK: bit (piece * 4 + e) = mask element e of piece > 0
  unsigned mnext = 0;                                    // ... of the next unit's K step 0 (collected one step ahead)

#pragma unroll
  for (int s = 0; s < kSDepth; ++s) GS_ISSUE(s, s)

  // fragments in flight across block / step boundaries: activations of the current and the next K step, weights of
  // the current and the next block
  sbf16x8 xh[2][RT], xl[2][RT], wh[2][2], wl[2][2];
  unsigned wofs = (unsigned)lane * 8u;
#pragma unroll
  for (int f = 0; f < NF; ++f) GS_CONVERT(0, f, xh[0], xl[0])
  if (MASK == 1) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float4 v = mk[0][m];
      mbits |= (unsigned long long)((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (m * 4);
    }
  }
  unsigned long long mnext64 = 0;                        // MASK == 2: the next unit's word, in flight during this unit
  if (MASK == 2) mbits = load_bits(0);
  GS_ISSUE(kSDepth, 0)
  GS_WREAD(0, 0, wh[0], wl[0])

  for (int t0 = 0; t0 < total; t0 += KSTEPS) {
    const int64_t rb = pair + (int64_t)(t0 / KSTEPS) * npairs;
    // The image never changes, so without this the compiler hoists ALL fragment reads out of the unit loop -- up to
    // 128 KiB of "loop-invariant" registers, i.e. spills.  An opaque lane offset per unit keeps them per block.
    asm volatile("" : "+v"(wofs));
    if (MASK == 2) mnext64 = load_bits(t0 / KSTEPS + 1);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int xc = ks & 1, xn = xc ^ 1;                // activation fragment sets: current / next step
      const int nslot = (ks + 1) % kSDepth;              // ring slot of the next step's raw activations
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int wc = (ks * NB + b) & 1, wn = wc ^ 1;
        // (1) weight fragments of the next block (the first block of the next K step after the last one)
        if (b + 1 < NB) GS_WREAD(ks, b + 1, wh[wn], wl[wn])
        else GS_WREAD((ks + 1) % KSTEPS, 0, wh[wn], wl[wn])
        // (2) this block's MFMAs: column tiles 2 b, 2 b + 1 x RT row tiles (x 3 products), product-major so that an
        // accumulator is touched again only after 2 RT - 1 other MFMAs
        if (NP == 2) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int tt = 0; tt < RT; ++tt)
              acc[tt][2 * b + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[wc][h], xl[xc][tt], acc[tt][2 * b + h], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int tt = 0; tt < RT; ++tt)
              acc[tt][2 * b + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[wc][h], xh[xc][tt], acc[tt][2 * b + h], 0, 0, 0);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int tt = 0; tt < RT; ++tt)
            acc[tt][2 * b + h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[wc][h], xh[xc][tt], acc[tt][2 * b + h], 0, 0, 0);
        // (3) a share of the next step's conversion (its loads were issued kSDepth - 1 steps ago)
        if (b % (NB / NF) == 0) {
          const int f = b / (NB / NF);
          GS_CONVERT(nslot, f, xh[xn], xl[xn])
        }
        if (MASK == 1 && b == NB - 1) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const float4 v = mk[nslot][m];
            const unsigned b4 = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            // pieces of K step ks + 1 (of the NEXT unit after the last step: collected into the upper half, see below)
            if (ks + 1 < KSTEPS) mbits |= (unsigned long long)b4 << (((ks + 1) * 2 + m) * 4);
            else mnext |= b4 << (m * 4);
          }
        }
        // (4) the converted slot is free: the loads of step t0 + ks + 1 + depth go there
        if (b == NB - 1) GS_ISSUE(t0 + ks + 1 + kSDepth, nslot)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // epilogue of the unit: lane (rl, q) holds columns c0 + 16 ct + 4 q .. + 3 of rows rb * BLOCK + wave * ROWS + 16 tt + rl
    unsigned long long obits = 0;                        // forward with ep.bits_out: (output > 0), nibble ct
#pragma unroll
    for (int tt = 0; tt < RT; ++tt) {
      const int64_t row = rb * BLOCK + wrow + tt * 16;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = c0 + ct * 16 + q * 4;
        const float4 b4 = *reinterpret_cast<const float4*>(lbias + ct * 16 + q * 4);
        float4 v = make_float4(acc[tt][ct][0] + b4.x, acc[tt][ct][1] + b4.y, acc[tt][ct][2] + b4.z, acc[tt][ct][3] + b4.w);
        acc[tt][ct] = sf32x4{0.f, 0.f, 0.f, 0.f};
        if (ep.act == GCNX_ACT_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (ep.act == GCNX_ACT_PRELU) {
          const float4 al = *reinterpret_cast<const float4*>(lbias + CW + ct * 16 + q * 4);
          v.x = v.x > 0.f ? v.x : al.x * v.x; v.y = v.y > 0.f ? v.y : al.y * v.y;
          v.z = v.z > 0.f ? v.z : al.z * v.z; v.w = v.w > 0.f ? v.w : al.w * v.w;
        }
        if (MASK != 0) {
          const unsigned b = (unsigned)(mbits >> ((tt * CT + ct) * 4)) & 15u;
          v.x = (b & 1u) ? v.x : 0.f; v.y = (b & 2u) ? v.y : 0.f; v.z = (b & 4u) ? v.z : 0.f; v.w = (b & 8u) ? v.w : 0.f;
        }
        // range-checked store: rows past M / columns past ncol get an out-of-range offset and are dropped -- no branch
        // (a load or store inside a branch makes hipcc drain the whole prefetch ring with vmcnt(0))
        const unsigned off = (row < M && col < ncol) ? (unsigned)row * ldc4 + (unsigned)col * CB : 0xFFFFFFE0u;
        if constexpr (OUT16) {
          typedef __bf16 sbf16x4 __attribute__((ext_vector_type(4)));
          typedef int si32x2 __attribute__((ext_vector_type(2)));
          const sbf16x4 o16 = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(si32x2, o16), crs, off, 0, GCNX_STREAM_STORE_AUX);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(si32x4, sf32x4{v.x, v.y, v.z, v.w}), crs, off, 0, GCNX_STREAM_STORE_AUX);
        }
        if (MASK == 0)
          obits |= (unsigned long long)((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << ((tt * CT + ct) * 4);
        if (ep.colpart) {
          // column sums of what was written, without reading it back: the 16 rows of the tile are summed across the
          // lanes rl = 0 .. 15 (fixed tree), lane rl = 0 adds the result to its wave's LDS row -- the same lane in
          // the same order every time (units, tiles ascending): deterministic.  LDS only: nothing here waits on vmcnt.
          float4 sv = row < M ? v : make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (CSREG) {
            cs[ct].x += sv.x; cs[ct].y += sv.y; cs[ct].z += sv.z; cs[ct].w += sv.w;     // (units, tiles ascending: a fixed order)
          } else {
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) {
              sv.x += __shfl_xor(sv.x, sh); sv.y += __shfl_xor(sv.y, sh); sv.z += __shfl_xor(sv.z, sh); sv.w += __shfl_xor(sv.w, sh);
            }
            if (rl == 0) {
              float4* ws4 = reinterpret_cast<float4*>(wsum + wave * CW + ct * 16 + q * 4);
              float4 o = *ws4;
              o.x += sv.x; o.y += sv.y; o.z += sv.z; o.w += sv.w;
              *ws4 = o;
            }
          }
        }
      }
    }
    if (MASK == 0 && ep.bits_out) {                      // (uniform) one 8-byte store per lane and row block
      typedef unsigned int su32x2 __attribute__((ext_vector_type(2)));
      const int64_t row = rb * BLOCK + wrow;
      const su32x2 w2 = {(unsigned)obits, (unsigned)(obits >> 32)};
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) int, w2), brs,
                                            row < M ? ((unsigned)row * (unsigned)halves + (unsigned)half) * 32u + (unsigned)q * 8u : 0xFFFFFFE0u, 0, 0);
    }
    if (MASK == 2) mbits = mnext64;
    else mbits = mnext;                                  // the first two mask pieces of the next unit are already in
    mnext = 0;
  }
  if constexpr (CSREG) {
    if (ep.colpart) {                                    // the 16 lanes that share a column group (fixed tree), once per launch
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float4 sv = cs[ct];
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) {
          sv.x += __shfl_xor(sv.x, sh); sv.y += __shfl_xor(sv.y, sh); sv.z += __shfl_xor(sv.z, sh); sv.w += __shfl_xor(sv.w, sh);
        }
        if (rl == 0) *reinterpret_cast<float4*>(wsum + wave * CW + ct * 16 + q * 4) = sv;
      }
    }
  }
  if (ep.colpart) {                                      // the workgroup's column sums: its 8 waves in a fixed order
    __syncthreads();
    for (int i = tid; i < CW; i += 512) {
      float t = wsum[i];
#pragma unroll
      for (int w = 1; w < kSWaves; ++w) t += wsum[w * CW + i];
      if (c0 + i < ncol) ep.colpart[(int64_t)pair * ncol + c0 + i] = t;
    }
  }
#undef GS_ISSUE
#undef GS_CONVERT
#undef GS_WREAD
}

inline bool sal16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NP, int CW, int RT, int KSTEPS>
int launch_stream(gcnx_ctx* ctx, const float* a, int64_t lda, const __bf16* img, float* c, int64_t ldc, int64_t m, int ncol,
                  const StreamEpi& ep, int halves) {
  constexpr size_t lds_bytes = (size_t)NP * KSTEPS * CW * 32 * 2 + 2 * CW * 4 + (size_t)kSWaves * CW * 4;
  static_assert(lds_bytes <= 160 * 1024, "weight image must fit the CU's LDS");
  static bool attr_set = false;
  if (!attr_set) {
    GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<NP, CW, RT, KSTEPS, 0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<NP, CW, RT, KSTEPS, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<NP, CW, RT, KSTEPS, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  const int n_rb = gcnx_cdiv(m, 16 * RT * kSWaves);
  int grid = ctx->num_cus;                                 // one workgroup per CU (the image takes most of its LDS)
  if (halves == 2) grid &= ~1;
  const int npairs = grid / halves;
  if (npairs > n_rb) grid = n_rb * halves;
  if (ep.mbits_in)
    hipLaunchKernelGGL((gemm_stream_kernel<NP, CW, RT, KSTEPS, 2>), dim3(grid), dim3(512), lds_bytes, ctx->stream, a, lda, img, c,
                       ldc, m, ncol, ep, n_rb, halves);
  else if (ep.mask)
    hipLaunchKernelGGL((gemm_stream_kernel<NP, CW, RT, KSTEPS, 1>), dim3(grid), dim3(512), lds_bytes, ctx->stream, a, lda, img, c,
                       ldc, m, ncol, ep, n_rb, halves);
  else
    hipLaunchKernelGGL((gemm_stream_kernel<NP, CW, RT, KSTEPS, 0>), dim3(grid), dim3(512), lds_bytes, ctx->stream, a, lda, img, c,
                       ldc, m, ncol, ep, n_rb, halves);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}

}  // namespace

// X W (transpose = 1) / dH W^T (transpose = 0) on the streaming kernel.  Returns GCNX_ERR_UNSUPPORTED (without
// setting an error message) when the shape is not one it is built for; the caller then takes the tiled kernel.
int gcnx_gemm_stream_nn(gcnx_ctx* ctx, const float* a, int64_t lda, const float* w, int fi, int fo, int transpose, float* c,
                        int64_t ldc, int64_t m, int prec, const float* bias, const float* alpha, int act, const float* mask,
                        int64_t ldmask, int accumulate, float* colsum_out, const void* mask_bits, void* bits_out) {
  const int ncol = transpose ? fo : fi, K = transpose ? fi : fo;
  const int np = prec == GCNX_PREC_BF16X3 ? 2 : 1;
  // the bit image of a ReLU output (bits_out: written; mask_bits: read instead of `mask`): full-width outputs only (both
  // launches of a pair must split the columns the same way), 64 bytes per row reserved
  if ((mask_bits || bits_out) && (ncol != 256 || (reinterpret_cast<uintptr_t>(mask_bits) & 7) || (reinterpret_cast<uintptr_t>(bits_out) & 7) ||
                                  (uint64_t)m * 64u >= 0xFFFFFF00ull))
    return GCNX_ERR_UNSUPPORTED;
  if (accumulate) return GCNX_ERR_UNSUPPORTED;   // (a read-modify-write epilogue would drain the prefetch ring: tiled kernel)
  if (ctx->knob_gemm_stream == 0) return GCNX_ERR_UNSUPPORTED;
  const bool shape_ok = K == 256 && ncol % 4 == 0 && ncol <= 256 && ncol >= 64 && m >= 32 * 1024 &&
                        lda % 4 == 0 && ldc % 4 == 0 && sal16(a) && sal16(c) && (uint64_t)m * (uint64_t)lda * 4u < 0xFFFFFF00ull &&
                        (uint64_t)m * (uint64_t)ldc * 4u < 0xFFFFFF00ull &&
                        (!mask || (ldmask % 4 == 0 && sal16(mask) && (uint64_t)m * (uint64_t)ldmask * 4u < 0xFFFFFF00ull));
  if (!shape_ok) return GCNX_ERR_UNSUPPORTED;
  // bf16x3: hi + lo planes of 256 k x 128 columns = 128 KiB of LDS, wider outputs as two column halves on
  // XCD-neighbouring workgroups, 32 rows per wave.  bf16: one plane of 256 k x 256 columns = 128 KiB, every
  // activation byte read once, 16 rows per wave.  (K = 256 only: the fused ReLU mask rides two pieces per K step.)
  const int ksteps = 8;
  const int cw = np == 2 ? 128 : 256;
  const int halves = gcnx_cdiv(ncol, cw);
  const size_t img_elems = (size_t)halves * np * ksteps * cw * 32;
  // colsum_out (BiasAddGrad of the layer below, db = column sums of what is written): per-workgroup partial rows at the
  // START of the workspace (where gcnx_colsum_partials looks for them), the weight image behind them
  int grid_wgs = ctx->num_cus;
  if (halves == 2) grid_wgs &= ~1;
  // (one partial row per workgroup PAIR that runs: the launch caps the grid at the row blocks there are, and every
  // workgroup of it writes its row -- no rows to clear)
  const int64_t n_rb_ = gcnx_cdiv(m, (np == 2 ? 32 : 16) * kSWaves);
  const int64_t prow = colsum_out ? std::min<int64_t>(grid_wgs / halves, n_rb_) : 0;
  if (colsum_out && (ncol % 4 != 0 || !sal16(colsum_out))) return GCNX_ERR_UNSUPPORTED;
  const size_t part_bytes = colsum_out ? ((gcnx_colsum_partials_ws(prow, ncol) + 255) & ~(size_t)255) : 0;
  int rc = gcnx_ws_reserve(ctx, part_bytes + img_elems * sizeof(__bf16) + 256);
  if (rc) return rc;
  __bf16* img = (__bf16*)((char*)ctx->ws + part_bytes);
  hipLaunchKernelGGL(stream_wprep_kernel, dim3(gcnx_cdiv((long long)img_elems, 256)), dim3(256), 0, ctx->stream, w, fi, fo, transpose,
                     np, cw, ksteps, ncol, img);
  GCNX_LAUNCH_OK(ctx);
  const StreamEpi ep{bias, alpha, mask_bits ? nullptr : mask, ldmask, act, accumulate, colsum_out ? (float*)ctx->ws : nullptr,
                     (const unsigned long long*)mask_bits, (unsigned long long*)bits_out};
  rc = np == 2 ? launch_stream<2, 128, 2, 8>(ctx, a, lda, img, c, ldc, m, ncol, ep, halves)
               : launch_stream<1, 256, 1, 8>(ctx, a, lda, img, c, ldc, m, ncol, ep, halves);
  if (rc || !colsum_out) return rc;
  return gcnx_colsum_partials(ctx, prow, ncol, colsum_out);
}

namespace {
// the bf16-storage forms: <1, 256, 1, 8> with a bf16 streamed operand, fp32 or bf16 result; masks as bit images only
template <bool OUT16>
int launch_stream16(gcnx_ctx* ctx, const void* a, int64_t lda, const __bf16* img, void* c, int64_t ldc, int64_t m, int ncol,
                    const StreamEpi& ep) {
  constexpr size_t lds_bytes = (size_t)8 * 256 * 32 * 2 + 2 * 256 * 4 + (size_t)kSWaves * 256 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<1, 256, 1, 8, 0, true, OUT16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    GCNX_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_stream_kernel<1, 256, 1, 8, 2, true, OUT16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set = true;
  }
  const int n_rb = gcnx_cdiv(m, 16 * kSWaves);
  int grid = ctx->num_cus;
  if (grid > n_rb) grid = n_rb;
  if (ep.mbits_in)
    hipLaunchKernelGGL((gemm_stream_kernel<1, 256, 1, 8, 2, true, OUT16>), dim3(grid), dim3(512), lds_bytes, ctx->stream,
                       (const float*)a, lda, img, (float*)c, ldc, m, ncol, ep, n_rb, 1);
  else
    hipLaunchKernelGGL((gemm_stream_kernel<1, 256, 1, 8, 0, true, OUT16>), dim3(grid), dim3(512), lds_bytes, ctx->stream,
                       (const float*)a, lda, img, (float*)c, ldc, m, ncol, ep, n_rb, 1);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}
}  // namespace

// X W (transpose = 1) / dH W^T (transpose = 0) with the streamed operand stored as bf16 and the result stored as bf16
// (c_bf16) or fp32: plain bf16 products, K = 256, 256 output columns, tall inputs.  The entry point checks the arguments;
// this returns GCNX_ERR_UNSUPPORTED without a message for shapes outside the kernel.
int gcnx_gemm_stream_bf16(gcnx_ctx* ctx, const void* a16, int64_t lda, const float* w, int fi, int fo, int transpose, void* c,
                          int64_t ldc, int c_bf16, int64_t m, const float* bias, int act, float* colsum_out, const void* mask_bits,
                          void* bits_out, const void* wimg) {
  const int ncol = transpose ? fo : fi, K = transpose ? fi : fo;
  if (ctx->knob_gemm_stream == 0 || K != 256 || ncol != 256 || m < 32 * 1024) return GCNX_ERR_UNSUPPORTED;
  if (lda % 8 || ldc % 4 || !sal16(a16) || !sal16(c) || (uint64_t)m * (uint64_t)lda * 2u >= 0xFFFFFF00ull ||
      (uint64_t)m * (uint64_t)ldc * (c_bf16 ? 2u : 4u) >= 0xFFFFFF00ull || (uint64_t)m * 64u >= 0xFFFFFF00ull ||
      (reinterpret_cast<uintptr_t>(mask_bits) & 7) || (reinterpret_cast<uintptr_t>(bits_out) & 7))
    return GCNX_ERR_UNSUPPORTED;
  if (colsum_out && !sal16(colsum_out)) return GCNX_ERR_UNSUPPORTED;
  const size_t img_elems = (size_t)8 * 256 * 32;
  const int64_t prow = colsum_out ? std::min<int64_t>(ctx->num_cus, gcnx_cdiv(m, 16 * kSWaves)) : 0;   // (= the grid: every workgroup writes its row)
  const size_t part_bytes = colsum_out ? ((gcnx_colsum_partials_ws(prow, ncol) + 255) & ~(size_t)255) : 0;
  int rc = gcnx_ws_reserve(ctx, part_bytes + img_elems * sizeof(__bf16) + 256);
  if (rc) return rc;
  const __bf16* img = (const __bf16*)wimg;            // the caller's image of this operand (gcnx_gemm_stream_images) ...
  if (!img) {                                         // ... or one built here
    __bf16* own = (__bf16*)((char*)ctx->ws + part_bytes);
    hipLaunchKernelGGL(stream_wprep_kernel, dim3(gcnx_cdiv((long long)img_elems, 256)), dim3(256), 0, ctx->stream, w, fi, fo, transpose,
                       1, 256, 8, ncol, own);
    GCNX_LAUNCH_OK(ctx);
    img = own;
  }
  const StreamEpi ep{bias, nullptr, nullptr, 0, act, 0, colsum_out ? (float*)ctx->ws : nullptr,
                     (const unsigned long long*)mask_bits, (unsigned long long*)bits_out};
  rc = c_bf16 ? launch_stream16<true>(ctx, a16, lda, img, c, ldc, m, ncol, ep) : launch_stream16<false>(ctx, a16, lda, img, c, ldc, m, ncol, ep);
  if (rc || !colsum_out) return rc;
  return gcnx_colsum_partials(ctx, prow, ncol, colsum_out);
}

// ----------------------------------------------------------------------------------------------------------------
// dW[Fi, Fo] = X^T[Fi, N] dH[N, Fo]  (MatMul grad wrt the kernel, gcn.py:337): the reduction runs over the N rows
// (10^6 at config 3), both operands are streamed once and BOTH need the row index as the MFMA k index, i.e. a
// transpose of what memory holds.  One 512-thread workgroup per CU owns a contiguous row range (a split-K slice) and
// keeps the WHOLE 256 x 256 product in registers (8 waves x 128 accumulators), so every input byte is read once:
//   global (fp32, one coalesced 1-KiB row piece per wave instruction) -> registers -> bf16 hi / lo -> LDS rows of
//   544 bytes (row stride = 32 B mod 256 B: the transposing read below touches 8 rows x 32 B per half-wave: no conflict)
//   -> ds_read_b64_tr_b16: the 4 x 16 block comes back column-major, which IS the MFMA fragment with k = row.
// The loads of K step s + 1 are issued before the MFMAs of step s (64 KiB in flight per CU), converted and written to
// the other LDS stage after them; one barrier per step.  The slice's product goes to a slab; the existing
// deterministic reduction adds the slabs in slice order.
// ----------------------------------------------------------------------------------------------------------------
typedef short ss16x4 __attribute__((ext_vector_type(4)));
typedef short ss16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kDwRowB = 544;                    // LDS bytes per staged row (256 bf16 + 32 B pad)
constexpr int kDwPlane = 32 * kDwRowB;          // one (matrix, plane) image of a K step: 17 408 B
template <int NP> struct DwLds { static constexpr int stage = 2 * NP * kDwPlane, total = 2 * stage; };

__device__ __forceinline__ sbf16x8 tr_frag(const char* p) {
  // rows 8q .. 8q+3 and 8q+4 .. 8q+7 of the lane group's 16 columns, column-major
  const ss16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ss16x4*)(p));
  const ss16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ss16x4*)(p + 4 * kDwRowB));
  const ss16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(sbf16x8, v);
}

__device__ __forceinline__ void dw_store4(char* dst, const float4& v, bool lo_too) {
  sbf16x8 h, l;
  const float f[4] = {v.x, v.y, v.z, v.w};
  typedef __bf16 b4 __attribute__((ext_vector_type(4)));
  b4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    hi[j] = (__bf16)f[j];
    lo[j] = (__bf16)(f[j] - (float)hi[j]);
  }
  *reinterpret_cast<b4*>(dst) = hi;
  if (lo_too) *reinterpret_cast<b4*>(dst + kDwPlane) = lo;
  (void)h; (void)l;
}

// x [N, ldx] (columns 0..255), dh [N, lddh] (columns 0..255); out: slab + blockIdx.x * 65536 floats = dW[i][o].
// IN16 (r3, plain bf16): both operands are stored as bf16 -- a thread stages 8 columns of two rows per matrix with one
// 16-byte load each and writes them to the LDS rows as they are (no conversion; half the bytes of the fp32 form).
template <int NP, bool IN16 = false>
__global__ __launch_bounds__(512, 2) void gemm_dw_stream_kernel(const float* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ dh, int64_t lddh,
                                                                float* __restrict__ out, int64_t n, int64_t rows_per_wg) {
  static_assert(!IN16 || NP == 1, "bf16 operands: plain bf16 products");
  constexpr unsigned EB = IN16 ? 2u : 4u;
  extern __shared__ __attribute__((aligned(16))) char dlds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r_end = min(n, r_begin + rows_per_wg);
  const int nsteps = r_begin < r_end ? (int)((r_end - r_begin + 31) / 32) : 0;

  // blockIdx.y (r3): a 256-column panel of a wider x -- dW of a Dense layer with fi = 256 p inputs (GeneralGNN's concat
  // skips) as p products over the same row slices; panel q's slabs follow panel q - 1's
  x = reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + (size_t)blockIdx.y * 256 * EB);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, (short)0, (int)((uint64_t)n * (uint64_t)ldx * EB - (uint64_t)blockIdx.y * 256u * EB), 0x00020000);
  const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc((void*)dh, (short)0, (int)((uint64_t)n * (uint64_t)lddh * EB), 0x00020000);
  const unsigned ldx4 = (unsigned)ldx * EB, ldd4 = (unsigned)lddh * EB;              // row strides in bytes
  // staging: thread -> row (tid >> 6) + 8 j (j = 0..3), columns 4 (tid & 63) .. + 3;  IN16: row (tid >> 5) + 16 j (j = 0, 1),
  // columns 8 (tid & 31) .. + 7
  const int srow = IN16 ? tid >> 5 : tid >> 6, scol4 = IN16 ? (tid & 31) * 8 : (tid & 63) * 4;
  float4 sx[4], sd[4];
  si32x4 qx[2][2], qd[2][2];             // IN16: two register sets -- the loads run TWO steps ahead (64 KiB in flight per CU, as in
                                         // the fp32 form; with one step, 32 KiB, the kernel was latency-bound at 4.1 TB/s)
#define DW_ISSUE(STEP, SET)                                                                                       \
  if constexpr (IN16) {                                                                                            \
    const int64_t rb_ = r_begin + (int64_t)(STEP) * 32 + srow;                                                     \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                \
      const int64_t r_ = rb_ + 16 * j;                                                                             \
      const bool ok_ = (STEP) < nsteps && r_ < r_end;                                                              \
      qx[SET][j] = __builtin_bit_cast(si32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok_ ? (unsigned)r_ * ldx4 + (unsigned)scol4 * 2u : 0xFFFFFFE0u, 0, GCNX_STREAM_LOAD_AUX)); \
      qd[SET][j] = __builtin_bit_cast(si32x4, __builtin_amdgcn_raw_buffer_load_b128(drs, ok_ ? (unsigned)r_ * ldd4 + (unsigned)scol4 * 2u : 0xFFFFFFE0u, 0, GCNX_STREAM_LOAD_AUX)); \
    }                                                                                                              \
  } else {                                                                                                         \
    const int64_t rb_ = r_begin + (int64_t)(STEP) * 32 + srow;                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                \
      const int64_t r_ = rb_ + 8 * j;                                                                              \
      const bool ok_ = (STEP) < nsteps && r_ < r_end;                                                              \
      sx[j] = sbuf4(xrs, ok_ ? (unsigned)r_ * ldx4 + (unsigned)scol4 * 4u : 0xFFFFFFE0u);                          \
      sd[j] = sbuf4(drs, ok_ ? (unsigned)r_ * ldd4 + (unsigned)scol4 * 4u : 0xFFFFFFE0u);                          \
    }                                                                                                              \
  }
#define DW_WRITE(STAGE, SET)                                                                                       \
  if constexpr (IN16) {                                                                                            \
    char* sb_ = dlds + (STAGE) * DwLds<NP>::stage + srow * kDwRowB + scol4 * 2;                                    \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                \
      *reinterpret_cast<si32x4*>(sb_ + j * 16 * kDwRowB) = qx[SET][j];                                             \
      *reinterpret_cast<si32x4*>(sb_ + NP * kDwPlane + j * 16 * kDwRowB) = qd[SET][j];                             \
    }                                                                                                              \
  } else {                                                                                                         \
    char* sb_ = dlds + (STAGE) * DwLds<NP>::stage + srow * kDwRowB + scol4 * 2;                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                \
      dw_store4(sb_ + j * 8 * kDwRowB, sx[j], NP == 2);                                                            \
      dw_store4(sb_ + NP * kDwPlane + j * 8 * kDwRowB, sd[j], NP == 2);                                            \
    }                                                                                                              \
  }

  // wave tile: 64 output columns o (A operand, fragments kept) x 128 rows i of dW (B operand, iterated)
  const int o0 = (wave & 3) * 64, i0 = (wave >> 2) * 128;
  const int g = lane >> 4, u = lane & 15;
  // transposing read: lane 4 r' + p of group g points at row 8 g + r', columns 4 p .. 4 p + 3 of the 16-column block
  const int lane_off = (8 * g + (u >> 2)) * kDwRowB + (u & 3) * 8;

  sf32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = sf32x4{0.f, 0.f, 0.f, 0.f};

  DW_ISSUE(0, 0)
  DW_WRITE(0, 0)
  if constexpr (IN16) { DW_ISSUE(1, 1) }
  __syncthreads();
  // One K step (32 rows) on LDS stage ST: the loads that replace a free register set go out first, the other stage is
  // written behind the MFMAs.  fp32 operands: one set, one step ahead.  bf16 operands: set (step & 1), two steps ahead.
#define DW_STEP(S, ST)                                                                                            \
  {                                                                                                                \
    if constexpr (IN16) { DW_ISSUE((S) + 2, (ST)) } else { DW_ISSUE((S) + 1, 0) }                                  \
    dw_mfma_step((ST));                                                                                            \
    if ((S) + 1 < nsteps) { if constexpr (IN16) { DW_WRITE((ST) ^ 1, (ST) ^ 1) } else { DW_WRITE((ST) ^ 1, 0) } }  \
    __syncthreads();                                                                                               \
  }
  auto dw_mfma_step = [&](int st) {
    const char* xb = dlds + st * DwLds<NP>::stage + lane_off;           // X planes (hi, lo)
    const char* db = xb + NP * kDwPlane;                                // dH planes
    sbf16x8 ah[4], al[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      ah[a] = tr_frag(db + (o0 + 16 * a) * 2);
      if (NP == 2) al[a] = tr_frag(db + kDwPlane + (o0 + 16 * a) * 2);
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const sbf16x8 bh = tr_frag(xb + (i0 + 16 * b) * 2);
      if (NP == 2) {
        const sbf16x8 bl = tr_frag(xb + kDwPlane + (i0 + 16 * b) * 2);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bl, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[a], bh, acc[a][b], 0, 0, 0);
        }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[a], bh, acc[a][b], 0, 0, 0);
    }
  };
  if constexpr (IN16) {
    for (int s = 0; s < nsteps; s += 2) {                 // (the other stage was last read before the previous barrier)
      DW_STEP(s, 0)
      if (s + 1 < nsteps) DW_STEP(s + 1, 1)
    }
  } else {
    for (int s = 0; s < nsteps; ++s) DW_STEP(s, s & 1)    // (fp32 operands: the loop as it was -- one register set, a run-time stage)
  }
#undef DW_STEP
#undef DW_ISSUE
#undef DW_WRITE
  // C[row = o within tile (lane >> 4) * 4 + reg][col = i within tile (lane & 15)]: a lane holds dW[i][o .. o + 3]
  float* slab = out + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 65536;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int i = i0 + 16 * b + u, o = o0 + 16 * a + 4 * g;
      *reinterpret_cast<float4*>(slab + (size_t)i * 256 + o) = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
    }
}

}  // namespace

// Slabs [nslices][256 * 256] in `slabs` (caller reduces them in slice order); returns the number of slices, 0 when
// the shape is not one this kernel is built for.
// Slab reduction of the panel form: dw[p * 65536 + e] = sum over the slices, in slice order, of slabs[(p * nslices + s)][e].
__global__ __launch_bounds__(256) void dw_panel_reduce_kernel(const float* __restrict__ slabs, int nslices, float* __restrict__ dw) {
  // 64 float4 outputs x 4 slice groups per workgroup: each group adds its quarter of the slices in ascending order with eight
  // loads in flight, the four group sums are combined as (g0 + g1) + (g2 + g3) -- a fixed order (one thread walking all
  // ~117 slices four at a time was 29 dependent L2 round trips: 10.9 us for 7.7 MB)
  __shared__ float4 sm[4][64];
  const int el = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const size_t e = ((size_t)blockIdx.x * 64 + el) * 4;
  const float* p = slabs + (size_t)blockIdx.y * nslices * 65536 + e;
  const int per = (nslices + 3) / 4;
  const int s0 = grp * per, s1 = min(nslices, s0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int s = s0;
  for (; s + 8 <= s1; s += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(s + u) * 65536);
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; s < s1; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(p + (size_t)s * 65536);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  sm[grp][el] = acc;
  __syncthreads();
  if (grp == 0) {
    const float4 a = sm[0][el], b = sm[1][el], c = sm[2][el], d = sm[3][el];
    *reinterpret_cast<float4*>(dw + (size_t)blockIdx.y * 65536 + e) =
        make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
  }
}

// dW[fi, 256] = X^T dH for fi = 256 p (p <= 8) at mid-size batches (GeneralGNN: n = 22 576): the streaming kernel over
// (row slices) x (256-column panels of x), one workgroup per CU in all, then one reduction launch.  Returns 1 when it
// ran, 0 when the shape is not served (the caller takes gcnx_gemm_dw's tiled path), < 0 on a launch error.
int gcnx_gemm_dw_panels(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, float* dw, int64_t n,
                        int32_t fi, int32_t fo, int prec) {
  if (ctx->knob_gemm_stream == 0 || fo != 256 || fi % 256 != 0 || fi < 256 || fi > 2048 || n < 2048 || prec == GCNX_PREC_F32) return 0;
  if (ldx % 4 || lddh % 4 || !sal16(x) || !sal16(dh) || !sal16(dw) || (uint64_t)n * (uint64_t)ldx * 4u >= 0xFFFFFF00ull ||
      (uint64_t)n * (uint64_t)lddh * 4u >= 0xFFFFFF00ull)
    return 0;
  const int panels = fi / 256;
  const int64_t steps = (n + 31) / 32;
  int slices = ctx->num_cus / panels;                       // one workgroup per CU over all panels ...
  if (slices > steps / 6) slices = (int)(steps / 6);        // ... but no slice shorter than 6 steps (fill / drain of the pipeline)
  if (slices < 1) slices = 1;
  const int64_t rows_per = ((steps + slices - 1) / slices) * 32;
  slices = (int)((n + rows_per - 1) / rows_per);
  if (gcnx_ws_reserve(ctx, (size_t)panels * slices * 65536 * sizeof(float))) return -1;
  const int np = prec == GCNX_PREC_BF16X3 ? 2 : 1;
  const dim3 grid(slices, panels);
  if (np == 2) {
    static bool set2 = false;
    if (!set2) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dw_stream_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, DwLds<2>::total) != hipSuccess) return -1; set2 = true; }
    hipLaunchKernelGGL((gemm_dw_stream_kernel<2>), grid, dim3(512), DwLds<2>::total, ctx->stream, x, ldx, dh, lddh, (float*)ctx->ws, n, rows_per);
  } else {
    static bool set1 = false;
    if (!set1) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dw_stream_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, DwLds<1>::total) != hipSuccess) return -1; set1 = true; }
    hipLaunchKernelGGL((gemm_dw_stream_kernel<1>), grid, dim3(512), DwLds<1>::total, ctx->stream, x, ldx, dh, lddh, (float*)ctx->ws, n, rows_per);
  }
  if (hipGetLastError() != hipSuccess) return -1;
  hipLaunchKernelGGL(dw_panel_reduce_kernel, dim3(65536 / 4 / 64, panels), dim3(256), 0, ctx->stream, (const float*)ctx->ws, slices, dw);
  return hipGetLastError() == hipSuccess ? 1 : -1;
}

// The same with both operands stored as bf16 (uint16 rows, leading dimensions in elements): plain bf16 products.
int gcnx_gemm_dw_stream16(gcnx_ctx* ctx, const void* x16, int64_t ldx, const void* dh16, int64_t lddh, float* slabs, int64_t n,
                          int32_t fi, int32_t fo, int max_slices) {
  if (ctx->knob_gemm_stream == 0 || fi != 256 || fo != 256 || n < 32 * 1024) return 0;
  if (ldx % 8 || lddh % 8 || !sal16(x16) || !sal16(dh16) || (uint64_t)n * (uint64_t)ldx * 2u >= 0xFFFFFF00ull ||
      (uint64_t)n * (uint64_t)lddh * 2u >= 0xFFFFFF00ull)
    return 0;
  int slices = ctx->num_cus < max_slices ? ctx->num_cus : max_slices;
  const int64_t steps = (n + 31) / 32;
  if (slices > steps) slices = (int)steps;
  const int64_t rows_per = ((steps + slices - 1) / slices) * 32;
  slices = (int)((n + rows_per - 1) / rows_per);
  static bool set16 = false;
  if (!set16) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dw_stream_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, DwLds<1>::total) != hipSuccess) return 0; set16 = true; }
  hipLaunchKernelGGL((gemm_dw_stream_kernel<1, true>), dim3(slices), dim3(512), DwLds<1>::total, ctx->stream, (const float*)x16, ldx,
                     (const float*)dh16, lddh, slabs, n, rows_per);
  return hipGetLastError() == hipSuccess ? slices : -1;
}

int gcnx_gemm_dw_stream(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, float* slabs, int64_t n,
                        int32_t fi, int32_t fo, int prec, int max_slices) {
  if (ctx->knob_gemm_stream == 0 || fi != 256 || fo != 256 || n < 32 * 1024 || prec == GCNX_PREC_F32) return 0;
  if (ldx % 4 || lddh % 4 || !sal16(x) || !sal16(dh) || (uint64_t)n * (uint64_t)ldx * 4u >= 0xFFFFFF00ull ||
      (uint64_t)n * (uint64_t)lddh * 4u >= 0xFFFFFF00ull)
    return 0;
  int slices = ctx->num_cus < max_slices ? ctx->num_cus : max_slices;
  const int64_t steps = (n + 31) / 32;
  if (slices > steps) slices = (int)steps;
  const int64_t rows_per = ((steps + slices - 1) / slices) * 32;
  slices = (int)((n + rows_per - 1) / rows_per);
  const int np = prec == GCNX_PREC_BF16X3 ? 2 : 1;
  if (np == 2) {
    static bool set2 = false;
    if (!set2) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dw_stream_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, DwLds<2>::total) != hipSuccess) return 0; set2 = true; }
    hipLaunchKernelGGL((gemm_dw_stream_kernel<2>), dim3(slices), dim3(512), DwLds<2>::total, ctx->stream, x, ldx, dh, lddh, slabs, n, rows_per);
  } else {
    static bool set1 = false;
    if (!set1) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dw_stream_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, DwLds<1>::total) != hipSuccess) return 0; set1 = true; }
    hipLaunchKernelGGL((gemm_dw_stream_kernel<1>), dim3(slices), dim3(512), DwLds<1>::total, ctx->stream, x, ldx, dh, lddh, slabs, n, rows_per);
  }
  return hipGetLastError() == hipSuccess ? slices : -1;
}

// The images of up to four 256 x 256 weight operands for gcnx_gemm_fwd_bf16 (transpose = 1: X W) / gcnx_gemm_dx_bf16
// (transpose = 0: dH W^T), 128 KiB each, in one launch.
int gcnx_gemm_stream_images_impl(gcnx_ctx* ctx, int njobs, const float* const* w, const int* transpose, void* const* img) {
  StreamImageJobs jobs{};
  for (int j = 0; j < njobs; ++j) { jobs.w[j] = w[j]; jobs.img[j] = (__bf16*)img[j]; jobs.transpose[j] = transpose[j]; }
  hipLaunchKernelGGL(stream_wprep_multi_kernel, dim3(8 * 256 * 32 / 256, njobs), dim3(256), 0, ctx->stream, jobs);
  GCNX_LAUNCH_OK(ctx);
  return GCNX_OK;
}
