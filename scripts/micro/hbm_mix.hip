// Microbenchmark: what rate can the tile kernels' TRAFFIC reach on this part when nothing
// but the traffic is on the critical path?  Replaces store_pattern.hip's drain-per-phase loop (vmcnt(0) + barrier after
// every phase, one 1024-thread workgroup per CU), which had the very structure it was meant to bound.
//
//   * copy     : plain float4 grid-stride copy (the guide's 6.29 TB/s reference), same box, same process.
//   * mix<..>  : persistent workgroups stream (row block, column slab) tiles -- R rows x FTB bytes at a 1-KiB row pitch, the
//                aggregation's access shape -- by LDS-DMA into an LDS ring of NBUF buffers and write every tile back out
//                of LDS (ds_read_b128 -> global store, the quad pattern of spmm_duo_kernel or the DMA's own 8-lanes-per-
//                row pattern).  NBUF >= 2: the DMA of tile p + NBUF - 1 is issued before tile p is consumed and the wait
//                at the top of a phase is a COUNTED vmcnt that leaves the younger stores and DMAs in flight; one raw
//                s_barrier per phase, never vmcnt(0).  NBUF == 1 is the tier kernels' own structure (wait for the tile,
//                barrier, consume, barrier) without index burst and reduction.
//   * blocked  : the same tiles stored contiguously ([row block][slab][R][FTB]) instead of as column slabs of row-major
//                rows: tells what the 128-byte-pieces-at-1-KiB-pitch shape itself costs.
// Every variant's output is checked against the input (dst must equal src on the tiles it moved).
//   hipcc --offload-arch=gfx950 -O3 hbm_mix.hip -o hbm_mix && ./hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const char* g, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory");
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ s, float4* __restrict__ d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

constexpr int kPitch = 1024;   // bytes per feature row (256 fp32)

// STORE: 0 none (tile only read back from LDS), 1 quad pattern (4 lanes per row, chunks sub + 4 j), 2 DMA pattern
template <int THREADS, int NBUF, int R, int FTB, int STORE>
__global__ __launch_bounds__(THREADS) void mix(const char* __restrict__ src, char* __restrict__ dst, int rounds, int blocked,
                                               unsigned* __restrict__ sink, int SG) {
  constexpr int PPR = FTB / 16;                 // 16-byte pieces per tile row
  constexpr int D = R * PPR / THREADS;          // DMA instructions per thread and tile
  static_assert(R * PPR % THREADS == 0, "whole DMA instructions");
  constexpr int CPL = PPR / 4;                  // quad pattern: chunks per lane
  constexpr int PASSES = R / (THREADS / 4);
  static_assert(R % (THREADS / 4) == 0, "whole row passes");
  constexpr int S = STORE == 0 ? 0 : (STORE == 1 ? PASSES * CPL : D);   // store instructions per thread and tile
  constexpr int SLABS = kPitch / FTB;                                   // a unit = SG of a row block's slabs (run-time)
  const int UPG = SLABS / SG;
  constexpr int TILE = R * FTB;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const int tid = threadIdx.x;
  const int G = gridDim.x;
  const int vw = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
  const int nph = rounds * SG;
  auto tile_of = [&](int ph, size_t& base, int& pitch) {      // phase -> byte offset of the tile's first row, row pitch
    ph = min(ph, nph - 1);                                    // (phases past the end re-read the last tile: constant op counts)
    const int round = ph / SG, s = ph % SG;
    const int u = round * G + vw;
    const size_t rb = u / UPG;
    const int slab = (u % UPG) * SG + s;
    if (blocked) { base = (rb * SLABS + slab) * (size_t)TILE; pitch = FTB; }
    else { base = rb * (size_t)R * kPitch + (size_t)slab * FTB; pitch = kPitch; }
  };
  auto issue = [&](int ph, int buf) {
    size_t base; int pitch;
    tile_of(ph, base, pitch);
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int idx = tid + u * THREADS;
      const char* g = src + base + (size_t)(idx / PPR) * pitch + (idx % PPR) * 16;
      dma16(g, __builtin_amdgcn_readfirstlane(lds0 + buf * TILE + (u * THREADS + (tid & ~63)) * 16));
    }
  };
  unsigned acc = 0;
  if (NBUF > 1) {
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b) issue(b, b);
  }
  for (int ph = 0; ph < nph; ++ph) {
    const int buf = ph % NBUF;
    if (NBUF == 1) {
      issue(ph, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      // everything younger than the DMA of tile ph stays in flight: (NBUF - 1) phases of stores, (NBUF - 2) of DMAs
      if (ph == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * S + (NBUF - 2) * D) : "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (NBUF > 1) issue(ph + NBUF - 1, (ph + NBUF - 1) % NBUF);
    size_t base; int pitch;
    tile_of(ph, base, pitch);
    const char* t = lds + buf * TILE;
    if (STORE == 2 || STORE == 0) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int idx = tid + u * THREADS;
        const float4 v = *reinterpret_cast<const float4*>(t + idx * 16);
        if (STORE) *reinterpret_cast<float4*>(dst + base + (size_t)(idx / PPR) * pitch + (idx % PPR) * 16) = v;
        else acc += __float_as_uint(v.x) ^ __float_as_uint(v.w);
      }
    } else {
      const int sub = tid & 3;
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const int row = (tid >> 2) + p * (THREADS / 4);
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const int c = sub + 4 * j;
          const float4 v = *reinterpret_cast<const float4*>(t + row * FTB + c * 16);
          *reinterpret_cast<float4*>(dst + base + (size_t)row * pitch + c * 16) = v;
        }
      }
    }
    if (NBUF == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (STORE == 0 && acc == 0x12345u) sink[0] = acc;
}

struct Bufs { char *src, *dst; size_t bytes; unsigned* sink; };

template <int THREADS, int NBUF, int R, int FTB, int STORE>
void run(const Bufs& b, int wgpc, int blocked, const char* what, double copy_tbs, int sgo = 0) {
  constexpr int TILE = R * FTB;
  const int lds = NBUF * TILE;
  if ((size_t)lds * wgpc > 163840) { printf("%-78s skipped (LDS)\n", what); return; }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mix<THREADS, NBUF, R, FTB, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int G = 256 * wgpc;
  const int SG = sgo ? sgo : (kPitch / FTB) / 2;
  const int UPG = (kPitch / FTB) / SG;
  // about 1 GB in: rounds * G * SG tiles
  int rounds = (int)(1.0e9 / ((double)G * SG * TILE) + 0.5);
  if (rounds < 1) rounds = 1;
  const size_t rowblocks = ((size_t)rounds * G + UPG - 1) / UPG;
  const size_t need = rowblocks * R * kPitch;
  if (need > b.bytes) { printf("%-78s skipped (buffer)\n", what); return; }
  CK(hipMemsetAsync(b.dst, 0, need, 0));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((mix<THREADS, NBUF, R, FTB, STORE>), dim3(G), dim3(THREADS), lds, 0, b.src, b.dst, rounds, blocked, b.sink, SG);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 5;
  CK(hipEventRecord(e0));
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((mix<THREADS, NBUF, R, FTB, STORE>), dim3(G), dim3(THREADS), lds, 0, b.src, b.dst, rounds, blocked, b.sink, SG);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double moved = (double)rounds * G * SG * TILE * (STORE ? 2.0 : 1.0);
  // check: every float of the moved tiles equals the source (sampled rows)
  int bad = 0;
  if (STORE) {
    std::vector<float> hs(256), hd(256);
    for (size_t r = 0; r < rowblocks * R; r += 997) {
      CK(hipMemcpy(hs.data(), b.src + r * kPitch, 1024, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hd.data(), b.dst + r * kPitch, 1024, hipMemcpyDeviceToHost));
      // (an odd unit count leaves the last row block half moved; skip rows of that block)
      if (r / R >= (size_t)rounds * G / UPG) break;
      for (int i = 0; i < 256; ++i) bad += hs[i] != hd[i];
    }
  }
  printf("%-78s %8.1f us  %5.2f GB  %5.2f TB/s  (%.2f of 8, %.2f of copy)%s\n", what, us, moved / 1e9, moved / us / 1e6, moved / us / 1e6 / 8.0,
         moved / us / 1e6 / copy_tbs, bad ? "  *** MISMATCH ***" : "");
  fflush(stdout);
}

int main() {
  Bufs b;
  b.bytes = (size_t)1200 << 20;
  CK(hipMalloc(&b.src, b.bytes)); CK(hipMalloc(&b.dst, b.bytes)); CK(hipMalloc(&b.sink, 64));
  {
    std::vector<float> h(b.bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 16777213u);
    CK(hipMemcpy(b.src, h.data(), b.bytes, hipMemcpyHostToDevice));
  }
  // reference: float4 copy of 1 GiB
  double copy_tbs = 0;
  {
    const size_t n = ((size_t)1 << 30) / 16;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)b.src, (float4*)b.dst, n);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)b.src, (float4*)b.dst, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    copy_tbs = 2.0 * (double)((size_t)1 << 30) / (ms * 1e-3 / 10) / 1e12;
    printf("%-78s %8.1f us  %5.2f GB  %5.2f TB/s  (%.2f of 8)\n", "float4 copy, 2048 x 256 threads, 1 GiB in + 1 GiB out", ms * 100, 2.147, copy_tbs, copy_tbs / 8);
  }
  if (getenv("HBM_MIX_SG")) {   // slabs per unit: 8 workgroups on the 8 slabs of the same rows at once ... one workgroup walks all 8
    for (int sg : {1, 2, 4, 8}) {
      char t[160];
      snprintf(t, sizeof t, "sg %d: NBUF 1, 2 WG/CU x 512 thr, 640 x 128 B tiles, quad stores", sg);
      run<512, 1, 640, 128, 1>(b, 2, 0, t, copy_tbs, sg);
      snprintf(t, sizeof t, "sg %d: NBUF 2, 1 WG/CU x 1024 thr, 512 x 128 B tiles, quad stores", sg);
      run<1024, 2, 512, 128, 1>(b, 1, 0, t, copy_tbs, sg);
      snprintf(t, sizeof t, "sg %d: NBUF 1, 1 WG/CU x 1024 thr, 1280 x 128 B tiles, quad stores", sg);
      run<1024, 1, 1280, 128, 1>(b, 1, 0, t, copy_tbs, sg);
    }
    return 0;
  }
  // tier-1 shape, the tier kernels' own structure: one buffer, wait + barrier + consume + barrier
  run<512, 1, 640, 128, 1>(b, 2, 0, "NBUF 1, 2 WG/CU x 512 thr, 640 x 128 B tiles, quad stores (duo structure)", copy_tbs);
  run<512, 1, 640, 128, 2>(b, 2, 0, "NBUF 1, 2 WG/CU x 512 thr, 640 x 128 B tiles, 128-B-row stores", copy_tbs);
  run<512, 1, 640, 128, 0>(b, 2, 0, "NBUF 1, 2 WG/CU x 512 thr, 640 x 128 B tiles, no stores (read only)", copy_tbs);
  run<1024, 1, 1280, 128, 1>(b, 1, 0, "NBUF 1, 1 WG/CU x 1024 thr, 1280 x 128 B tiles, quad stores (tier-2 structure)", copy_tbs);
  run<256, 1, 320, 128, 1>(b, 4, 0, "NBUF 1, 4 WG/CU x 256 thr, 320 x 128 B tiles, quad stores", copy_tbs);
  // ring of 2: the next tile streams in while this one is written out
  run<1024, 2, 512, 128, 1>(b, 1, 0, "NBUF 2, 1 WG/CU x 1024 thr, 512 x 128 B tiles, quad stores (pipe structure)", copy_tbs);
  run<1024, 2, 512, 128, 2>(b, 1, 0, "NBUF 2, 1 WG/CU x 1024 thr, 512 x 128 B tiles, 128-B-row stores", copy_tbs);
  run<1024, 2, 512, 128, 0>(b, 1, 0, "NBUF 2, 1 WG/CU x 1024 thr, 512 x 128 B tiles, no stores (read only)", copy_tbs);
  run<512, 2, 256, 128, 1>(b, 2, 0, "NBUF 2, 2 WG/CU x 512 thr, 256 x 128 B tiles, quad stores", copy_tbs);
  run<512, 2, 640, 64, 1>(b, 2, 0, "NBUF 2, 2 WG/CU x 512 thr, 640 x 64 B tiles (16 columns), quad stores", copy_tbs);
  run<512, 2, 640, 64, 2>(b, 2, 0, "NBUF 2, 2 WG/CU x 512 thr, 640 x 64 B tiles (16 columns), 64-B-row stores", copy_tbs);
  run<512, 2, 640, 64, 0>(b, 2, 0, "NBUF 2, 2 WG/CU x 512 thr, 640 x 64 B tiles (16 columns), no stores (read only)", copy_tbs);
  run<256, 2, 320, 64, 1>(b, 4, 0, "NBUF 2, 4 WG/CU x 256 thr, 320 x 64 B tiles, quad stores", copy_tbs);
  // deeper rings of smaller tiles
  run<512, 3, 384, 64, 1>(b, 2, 0, "NBUF 3, 2 WG/CU x 512 thr, 384 x 64 B tiles, quad stores", copy_tbs);
  run<1024, 3, 256, 128, 1>(b, 1, 0, "NBUF 3, 1 WG/CU x 1024 thr, 256 x 128 B tiles, quad stores", copy_tbs);
  run<512, 3, 128, 128, 1>(b, 2, 0, "NBUF 3, 2 WG/CU x 512 thr, 128 x 128 B tiles, quad stores", copy_tbs);
  // wider slabs: 256-byte pieces
  run<1024, 2, 256, 256, 1>(b, 1, 0, "NBUF 2, 1 WG/CU x 1024 thr, 256 x 256 B tiles (64 columns), quad stores", copy_tbs);
  run<512, 1, 256, 256, 1>(b, 2, 0, "NBUF 1, 2 WG/CU x 512 thr, 256 x 256 B tiles (64 columns), quad stores", copy_tbs);
  // blocked layout: a tile is one contiguous chunk
  run<512, 1, 640, 128, 1>(b, 2, 1, "blocked: NBUF 1, 2 WG/CU x 512 thr, 640 x 128 B tiles, quad stores", copy_tbs);
  run<1024, 2, 512, 128, 1>(b, 1, 1, "blocked: NBUF 2, 1 WG/CU x 1024 thr, 512 x 128 B tiles, quad stores", copy_tbs);
  run<1024, 2, 512, 128, 2>(b, 1, 1, "blocked: NBUF 2, 1 WG/CU x 1024 thr, 512 x 128 B tiles, linear stores", copy_tbs);
  run<512, 2, 256, 128, 2>(b, 2, 1, "blocked: NBUF 2, 2 WG/CU x 512 thr, 256 x 128 B tiles, linear stores", copy_tbs);
  return 0;
}
