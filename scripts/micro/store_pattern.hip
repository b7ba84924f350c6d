// Microbenchmark: HBM write rate of the tile kernels' store pattern (a quad of lanes writes 64 contiguous bytes of a
// row, 16 rows per wave instruction, the row's other 64 bytes by the next instruction) against fully coalesced stores,
// alone and next to an LDS-DMA read stream of the same size.
//   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void dma16(const float* g, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory");
}
// rows x 256 floats matrix; a workgroup (1024 threads) handles blocks of 624 rows x 32 columns ("phases")
__global__ __launch_bounds__(1024) void k(const float* __restrict__ src, float* __restrict__ dst, int phases, int mode) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const float4 v = make_float4(tid, 1, 2, 3);
  for (int p = 0; p < phases; ++p) {
    const size_t blk = (size_t)blockIdx.x * phases + p;           // block of 624 rows; slab = p % 8
    const size_t row0 = (blk / 8) * 624;
    const int c0 = (int)(blk % 8) * 32;
    if (mode & 1) {                                                // DMA read of the block (624 x 128 B)
      for (int kk = 0; kk < 5; ++kk) {
        const int r = wave * 8 + kk * 128 + (lane >> 3);
        if (r < 624) dma16(src + (row0 + r) * 256 + c0 + (lane & 7) * 4, __builtin_amdgcn_readfirstlane(lds0 + (p & 1) * 79872 + (wave * 8 + kk * 128) * 128));
      }
    }
    if (mode & 2) {                                                // quad pattern: 3 row groups x 2 half rows
      for (int t = 0; t < 3; ++t) {
        const int r = wave * 16 + (lane >> 2) + t * 256;
        if (r < 624) for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(dst + (row0 + r) * 256 + c0 + (lane & 3) * 4 + j * 16) = v;
      }
    }
    if (mode & 4) {                                                // coalesced: 8 lanes per 128-byte row piece, 8 rows per instruction
      for (int kk = 0; kk < 5; ++kk) {
        const int r = wave * 8 + kk * 128 + (lane >> 3);
        if (r < 624) *reinterpret_cast<float4*>(dst + (row0 + r) * 256 + c0 + (lane & 7) * 4) = v;
      }
    }
    if (mode & 8) {                                                // whole rows: a wave writes 1 KiB = one full row of 256 floats (different layout of work)
      for (int kk = 0; kk < 5; ++kk) {
        const size_t r = (row0 * 32 + (size_t)c0 / 32 * 624 * 32 / 8 + (size_t)(wave + 16 * kk) * 256) ;   // just some distinct 1-KiB chunks
        if (wave + 16 * kk < 78) *reinterpret_cast<float4*>(dst + ((blk * 78 + wave + 16 * kk) * 256) % ((size_t)1 << 28) + lane * 4) = v;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}
int main() {
  const int cus = 256, phases = 48;
  const size_t rows = (size_t)cus * phases / 8 * 624 + 624;
  float *src, *dst;
  (void)hipMalloc(&src, rows * 1024); (void)hipMalloc(&dst, ((size_t)1 << 30) + rows * 1024);
  (void)hipMemset(src, 0, rows * 1024);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const double gb = (double)cus * phases * 624 * 128 / 1e9;
  const char* names[] = {"", "DMA read only", "quad-pattern stores only", "DMA + quad stores", "coalesced (8 lanes/row piece) stores only", "DMA + coalesced stores", "", "", "full-row (1 KiB/instr) stores only", "DMA + full-row stores"};
  for (int mode : {1, 2, 3, 4, 5, 8, 9}) {
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 160000, 0, src, dst, phases, mode);
    (void)hipEventRecord(e0);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 160000, 0, src, dst, phases, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 200;
    const double bytes = gb * ((mode & 1) ? 1 : 0) + gb * ((mode & 14) ? 1 : 0);
    printf("mode %d %-42s %8.1f us per launch   %.2f GB moved   %.2f TB/s\n", mode, names[mode], us, bytes, bytes / us * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3);
  }
  return 0;
}
