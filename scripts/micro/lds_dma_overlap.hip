// Microbenchmark: do LDS reads (a gather-reduce loop) and LDS-DMA landings overlap on one CU?
//   mode 1: reads only; mode 2: DMA only; mode 3: both, same waves; mode 4: both, DMA by wave 15 only;
//   mode 5: reads + plain global loads into registers (same HBM traffic, no LDS write)
// One 1024-thread workgroup per CU, two 78-KiB buffers.  hipcc --offload-arch=gfx950 -O3 lds_dma_overlap.hip -o lds_dma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int BUF = 79872;   // 624 rows x 128 B
__device__ __forceinline__ void dma16(const float* g, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_base) : "memory");
}
__global__ __launch_bounds__(1024) void k(const float* __restrict__ src, float* __restrict__ out, int phases, int mode, int reads) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  for (int i = tid; i < 2 * BUF / 4; i += 1024) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  float4 acc = make_float4(0, 0, 0, 0), racc = acc;
  const float* base = src + (size_t)blockIdx.x * phases * (BUF / 4);
  unsigned idx = tid * 2654435761u;
  for (int p = 0; p < phases; ++p) {
    const int buf = p & 1;
    const float* g = base + (size_t)p * (BUF / 4);
    if (mode == 2 || mode == 3) {
      for (int kk = 0; kk < 5; ++kk) { const int piece = wave * 64 + lane + kk * 1024; if (piece * 16 < BUF) dma16(g + piece * 4, __builtin_amdgcn_readfirstlane(lds0 + (buf ^ 1) * BUF + (wave * 64 + kk * 1024) * 16)); }
    } else if (mode == 4 && wave == 15) {
      for (int kk = 0; kk < 78; ++kk) dma16(g + (kk * 64 + lane) * 4, __builtin_amdgcn_readfirstlane(lds0 + (buf ^ 1) * BUF + kk * 1024));
    } else if (mode == 5) {
      for (int kk = 0; kk < 5; ++kk) { const int piece = tid + kk * 1024; if (piece * 16 < BUF) { float4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(g + piece * 4) : "memory"); racc.x += 0.f; (void)v; } }
    }
    if (mode != 2) {
      const char* cur = lds + buf * BUF;
#pragma unroll 8
      for (int r = 0; r < reads; ++r) {
        idx = idx * 1664525u + 1013904223u;
        const float4 v = *reinterpret_cast<const float4*>(cur + ((idx >> 8) % 624) * 128 + (lane & 7) * 16);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (acc.x + racc.x == -1.f) out[tid] = acc.y + acc.z + acc.w;
}
int main() {
  const int cus = 256, phases = 64;
  float *src, *out;
  const size_t bytes = (size_t)cus * phases * BUF;
  hipMalloc(&src, bytes); hipMalloc(&out, 4096); hipMemset(src, 0, bytes);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int reads : {64, 128, 256}) {
    for (int mode = 1; mode <= 5; ++mode) {
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 2 * BUF, 0, src, out, phases, mode, reads);
      hipEventRecord(e0);
      for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k, dim3(cus), dim3(1024), 2 * BUF, 0, src, out, phases, mode, reads);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("reads/phase %3d mode %d: %8.1f us per launch  (%.2f us per phase; DMA bytes/launch %.2f GB)\n", reads, mode, ms * 200, ms * 200 / phases, bytes / 1e9);
    }
  }
  return 0;
}
