// Product-free reproducer for the SIGSEGV inside hipGraphLaunch under `rocprofv3 --kernel-trace` (VERDICT r3 weak 11 / next 7,
// ADVICE r3 bench.py:247).  No libgcnx, no Python: one non-blocking stream, a captured graph of NODES small kernel nodes (+ one
// memset node, optionally a forked side branch, optionally by-value struct arguments), launched LAUNCHES times back to back
// without waiting for the GPU, then one synchronisation.  If THIS dies under the tracer and runs clean without it, the defect is
// the tool's dispatch tracking of deep graph-launch queues, not a lifetime bug of the library's captured graphs.
//   hipcc --offload-arch=gfx950 -O2 graph_trace_repro.hip -o graph_trace_repro
//   ./graph_trace_repro NODES LAUNCHES VARIANT [SYNC_EVERY]
//   VARIANT bit 0: memset node first; bit 1: fork / join a side stream in the middle; bit 2: 256-byte by-value struct argument;
//           bit 3: a second, different graph launched alternately (the bench's second model in one process)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

struct Big { float v[64]; };

__global__ void k_small(float* p, int n, float a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 0.999f + a;
}
__global__ void k_big(float* p, int n, Big b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 0.999f + b.v[i & 63];
}

static int build(hipStream_t s, hipStream_t side, hipEvent_t ef, hipEvent_t ej, float* buf, float* buf2, int n, int nodes, int variant,
                 hipGraphExec_t* out) {
  Big b;
  for (int i = 0; i < 64; ++i) b.v[i] = 1e-3f * i;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  if (variant & 1) CK(hipMemsetAsync(buf2, 0, n * sizeof(float), s));
  for (int k = 0; k < nodes; ++k) {
    if ((variant & 2) && k == nodes / 2) {
      CK(hipEventRecord(ef, s));
      CK(hipStreamWaitEvent(side, ef, 0));
      for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, side, buf2, n, 0.5f);
      CK(hipEventRecord(ej, side));
      CK(hipStreamWaitEvent(s, ej, 0));
    }
    if (variant & 4) hipLaunchKernelGGL(k_big, dim3((n + 255) / 256), dim3(256), 0, s, buf, n, b);
    else hipLaunchKernelGGL(k_small, dim3((n + 255) / 256), dim3(256), 0, s, buf, n, 1e-3f * k);
  }
  hipGraph_t g = nullptr;
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
  return 0;
}

int main(int argc, char** argv) {
  const int nodes = argc > 1 ? atoi(argv[1]) : 150, launches = argc > 2 ? atoi(argv[2]) : 200, variant = argc > 3 ? atoi(argv[3]) : 1;
  const int sync_every = argc > 4 ? atoi(argv[4]) : 0;
  const int n = 1 << 16;
  hipStream_t s, side;
  hipEvent_t ef, ej;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  float *buf, *buf2;
  CK(hipMalloc(&buf, n * sizeof(float)));
  CK(hipMalloc(&buf2, n * sizeof(float)));
  CK(hipMemset(buf, 0, n * sizeof(float)));
  hipGraphExec_t ex[2] = {nullptr, nullptr};
  if (build(s, side, ef, ej, buf, buf2, n, nodes, variant, &ex[0])) return 2;
  if (variant & 8) { if (build(s, side, ef, ej, buf, buf2, n, nodes / 2 + 3, variant & ~2, &ex[1])) return 2; }
  for (int l = 0; l < launches; ++l) {
    CK(hipGraphLaunch(ex[(variant & 8) ? (l & 1) : 0], s));
    if (sync_every && l % sync_every == sync_every - 1) CK(hipStreamSynchronize(s));
  }
  CK(hipStreamSynchronize(s));
  float h0 = 0.f;
  CK(hipMemcpy(&h0, buf, sizeof(float), hipMemcpyDeviceToHost));
  printf("ok nodes=%d launches=%d variant=%d sync_every=%d dispatches=%lld buf[0]=%g\n", nodes, launches, variant, sync_every,
         (long long)nodes * launches, h0);
  return 0;
}
