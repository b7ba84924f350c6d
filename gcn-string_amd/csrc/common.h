// Shared internals of libgcnx (not part of the ABI; the ABI is include/gcnx.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "gcnx.h"

struct gcnx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // Scratch owned by the ctx (split-K partials, column-sum partials, flags).  Grown on demand
  // outside stream capture; a capture that would need growth fails with a clear message.
  void* ws = nullptr;
  size_t ws_bytes = 0;
  int* flag = nullptr;       // device int[4]: [0] validation kernels, [1] gcnx_csr_inspect, [3] the head kernel's ticket
  bool capturing = false;
  // Captured graphs keep the workspace pointer they were recorded with in their kernel arguments.  While any graph
  // is alive a workspace that has to grow is therefore retired, not freed: old graphs keep replaying into their old
  // block, eager calls and later captures use the new one.  Retired blocks are released when the last graph is
  // destroyed (or with the ctx).
  int live_graphs = 0;
  std::vector<void*> retired_ws;
  // tuning knobs (diagnostics, not API), read once at ctx creation: GCNX_SPMM_KERNEL, GCNX_SPMM_SLAB, GCNX_SPMM_SG
  int knob_spmm_kernel = 0;  // 0 auto, 1 rows, 2 tile
  int knob_spmm_slab = 0, knob_spmm_sg = 0;
  const float* lr_dev = nullptr;  // gcnx_set_lr_source: the update launches read the learning rate from this device scalar
                                  // (NULL: from their `lr` argument) -- a captured step then serves every value of a schedule
  int knob_gemm_stream = 1;  // GCNX_GEMM_STREAM=0: bf16 GEMMs stay on the tiled kernel (A/B measurement)
  // Side stream for gradient "leaves" (weight / bias gradients that nothing downstream in the backward pass
  // consumes): gcnx_side_begin swaps stream and workspace, so every entry point launches there unchanged.
  hipStream_t main_stream = nullptr, side_stream = nullptr;
  void* ws_other = nullptr;          // the workspace of the stream that is NOT current
  size_t ws_other_bytes = 0;
  bool on_side = false, side_pending = false;
  static constexpr int kSideEvents = 8;
  hipEvent_t ev_fork[kSideEvents] = {}, ev_join[kSideEvents] = {};
  int ev_next = 0;
  int num_cus = 256;
  std::string arch;
  // roctx ranges around the kernel classes (GCNX_ROCTX=1: librocprofiler-sdk-roctx is loaded at ctx creation; shown by
  // rocprofv3 --marker-trace).  NULL = off: a range then costs one branch.
  int (*roctx_push)(const char*) = nullptr;
  int (*roctx_pop)() = nullptr;
  // Two auxiliary streams for launches of ONE call that write disjoint rows (the aggregation's tier kernels and its
  // row-chunk part): gcnx_aux_fork / gcnx_aux_join below.  Created on first use.
  hipStream_t aux_stream[2] = {};
  static constexpr int kAuxEvents = 8;
  hipEvent_t aux_ev[kAuxEvents][3] = {};
  int aux_next = 0;
  int knob_spmm_conc = 0;    // GCNX_SPMM_CONC=1: the plan path's three launches as concurrent branches
  int knob_spmm_tile_wgs = 0; // GCNX_SPMM_TILE_WGS=n: the 1024-thread tile launch on n workgroups (= n CUs) instead of one per CU (0), so that a
                             // concurrent branch (GCNX_SPMM_CONC=1: the tall graphs' row chunks) finds CUs to run on
  int knob_spmm_tall_rpc = 8; // GCNX_SPMM_TALL_RPC=32: 32-row chunks for graphs taller than a tile (r2 behaviour; default 8)
  int knob_spmm_sort_win = 0; // GCNX_SPMM_SORT_WIN: degree order inside windows of this many row groups (0: whole graph)
  int knob_spmm_bal = 1;     // GCNX_SPMM_BAL=0: tile graphs in plain size order (the snake deal of r2); c0 + 1000 * big%: cost model of the balanced deal
  int knob_spmm_cap1 = 0;    // GCNX_SPMM_CAP1: tallest graph of the two-workgroups-per-CU tile tier (-1: the kernel's capacity, 632;
                             // 0, the default since r3: every tile graph on the one-workgroup-per-CU tier -- measured faster at every size)
  int knob_spmm_cb = 1;      // graphs of >= 4096 rows in 64-column blocks (spmm_cb_kernel, r4): 1.10 x the compulsory HBM bytes at config 5 against
                             // 1.88 x for the row gather + hub segments, and 7-10 % faster since its index loads are few and wide.  GCNX_SPMM_CB=0:
                             // the r3 path; 2: the long rows' items first instead of dealt among the short rows' (no difference measured)
  int knob_pool_split = 0;   // GCNX_POOL_SPLIT=2..16: row slices per graph of the split global pool (0: the library's choice)
  // gcnx_h2d_async: a ring of pinned staging slots (allocated on first use), one event per slot -- a slot is reused only
  // after the copy out of it has completed
  static constexpr int kPinSlots = 32;
  static constexpr size_t kPinSlotBytes = 16384;
  char* pin_base = nullptr;
  hipEvent_t pin_ev[kPinSlots] = {};
  bool pin_busy[kPinSlots] = {};
  int pin_next = 0;
};

struct gcnx_event { hipEvent_t ev; };
struct gcnx_graph { hipGraph_t graph; hipGraphExec_t exec; };

extern thread_local std::string gcnx_tls_error;

int gcnx_fail(gcnx_ctx* ctx, int code, const char* fmt, ...);
int gcnx_ws_reserve(gcnx_ctx* ctx, size_t bytes);  // ensures ctx->ws has >= bytes
// runtime.hip: the two auxiliary streams, ordered after everything submitted to ctx->stream so far (fork), and
// ctx->stream ordered after everything submitted to them (join).  Both work inside stream capture.
extern "C" int gcnx_aux_fork(gcnx_ctx* ctx, hipStream_t out[2]);
extern "C" int gcnx_aux_join(gcnx_ctx* ctx);
// gemm_stream.hip: X W (transpose = 1) / dH W^T (transpose = 0) on the streaming bf16 kernel; GCNX_ERR_UNSUPPORTED
// (no message) when the shape is not one it is built for.
int gcnx_gemm_stream_nn(gcnx_ctx* ctx, const float* a, int64_t lda, const float* w, int fi, int fo, int transpose, float* c,
                        int64_t ldc, int64_t m, int prec, const float* bias, const float* alpha, int act, const float* mask,
                        int64_t ldmask, int accumulate, float* colsum_out /* column sums of c, or NULL */,
                        const void* mask_bits = nullptr /* bit image read instead of mask */, void* bits_out = nullptr /* written */);
// gemm_stream.hip: X^T dH for fi = fo = 256 on the streaming bf16 kernel: writes [slices][256 * 256] partial products to
// `slabs` (room for max_slices of them) and returns the number of slices; 0 = shape not handled, < 0 = launch error.
int gcnx_gemm_stream_bf16(gcnx_ctx* ctx, const void* a16, int64_t lda, const float* w, int fi, int fo, int transpose, void* c,
                          int64_t ldc, int c_bf16, int64_t m, const float* bias, int act, float* colsum_out, const void* mask_bits,
                          void* bits_out, const void* wimg);
int gcnx_gemm_stream_images_impl(gcnx_ctx* ctx, int njobs, const float* const* w, const int* transpose, void* const* img);
int gcnx_gemm_dw_stream16(gcnx_ctx* ctx, const void* x16, int64_t ldx, const void* dh16, int64_t lddh, float* slabs, int64_t n,
                          int32_t fi, int32_t fo, int max_slices);
int gcnx_gemm_dw_stream(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, float* slabs, int64_t n,
                        int32_t fi, int32_t fo, int prec, int max_slices);
// gemm_stream.hip: X^T dH for fi = 256 p, fo = 256 at mid-size batches (n >= 2048): streaming kernel over (row slices) x (column
// panels of x) + one reduction launch, result in dw.  1 = ran, 0 = shape not served, < 0 = launch error.
int gcnx_gemm_dw_panels(gcnx_ctx* ctx, const float* x, int64_t ldx, const float* dh, int64_t lddh, float* dw, int64_t n,
                        int32_t fi, int32_t fo, int prec);
#ifdef __HIPCC__
// Column sums of a few hundred partial rows [rows][f] (f % 4 == 0): workgroup bx owns 8 columns (two float4 lanes)
// x 128 row groups and folds the 128 partial sums in a fixed tree through LDS.  (colsum_kernel's 64 columns x 16 row
// groups would leave config 2's 642 rows to 2 workgroups walking 40 dependent trips each: 22 us against 6.)
// A device function so that it can share a launch with the split-K reduction (gemm.hip, gcnx_dense_bwd).
__device__ __forceinline__ void gcnx_colpart_reduce_body(const float* __restrict__ part, int64_t rows, int32_t f,
                                                         float* __restrict__ out, int bx, float4 (*s)[2]) {
  const int cl = threadIdx.x & 1, rg = threadIdx.x >> 1;
  const int c = bx * 8 + cl * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < f) {
#pragma unroll 4
    for (int64_t r = rg; r < rows; r += 128) {
      const float4 v = *reinterpret_cast<const float4*>(part + r * f + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  s[rg][cl] = acc;
  __syncthreads();
  for (int off = 64; off > 0; off >>= 1) {
    if (rg < off) {
      const float4 o = s[rg + off][cl];
      float4 m = s[rg][cl];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
      s[rg][cl] = m;
    }
    __syncthreads();
  }
  if (rg == 0 && c < f) *reinterpret_cast<float4*>(out + c) = s[0][cl];
}
#endif

// reduce.hip: out[f] = column sums of the `rows` partial rows [rows][f] that a producer kernel left at the START of
// the workspace (fixed order).  gcnx_colsum_partials_ws = bytes to reserve BEFORE the producer runs (partials plus
// this reduction's own second-stage scratch behind them).
size_t gcnx_colsum_partials_ws(int64_t rows, int32_t f);
int gcnx_colsum_partials(gcnx_ctx* ctx, int64_t rows, int32_t f, float* out);
int gcnx_pool_graph_list(gcnx_ctx* ctx, const int32_t* graph_ptr, const int32_t* glist, int32_t nlist, const float* x, int64_t ldx,
                         int32_t f, int mode, float* pooled, int64_t ldp, float* cnt);
// head.hip: the classifier head from the pool's partial sums as a launch of its own (what gcnx_gemm_dw2 falls back to when
// the head does not fit inside its launch)
int gcnx_head_from_parts(gcnx_ctx* ctx, const gcnx_head_args* leaf);
// reduce.hip: the split global pool (sum / avg).  gcnx_pool_split = slices per graph worth launching (1: none);
// gcnx_pool_partials writes the partial row sums [nsplit][b][f] (row stride f) to `part`.
// half_wgs_per_cu: first-stage workgroups to aim for, in halves per CU: 4 (= 2 per CU) for the stand-alone pool; 1
// (= one 1024-thread workgroup per two CUs) when the head's single workgroup per 32 graphs reads the partials -- its
// one CU pulls them at ~30 GB/s, so every slice costs the head ~1.5 us per 32 KB (config 2: 2 slices beat 4 by 2.5 us).
int gcnx_pool_split(const gcnx_ctx* ctx, int32_t b, int32_t f, int mode, int half_wgs_per_cu);
// cnt_part (may be NULL): per (slice, graph, column) the number of positive entries, same layout.
int gcnx_pool_partials(gcnx_ctx* ctx, const int32_t* graph_ptr, const float* x, int64_t ldx, int32_t b, int32_t f,
                       int mode, int nsplit, float* part, float* cnt_part, int wide);   // wide: 1024-thread workgroups

#define GCNX_CHECK_CTX(ctx) \
  do { if (!(ctx)) return gcnx_fail(nullptr, GCNX_ERR_INVALID, "%s: ctx is NULL", __func__); } while (0)

// A named range over the launches of one entry point (tracing: SURVEY section 5).  Place behind GCNX_CHECK_CTX.
struct GcnxRange {
  gcnx_ctx* c;
  GcnxRange(gcnx_ctx* ctx, const char* name) : c(ctx && ctx->roctx_push ? ctx : nullptr) { if (c) c->roctx_push(name); }
  ~GcnxRange() { if (c && c->roctx_pop) c->roctx_pop(); }
  GcnxRange(const GcnxRange&) = delete;
  GcnxRange& operator=(const GcnxRange&) = delete;
};
#define GCNX_RANGE(ctx, name) GcnxRange gcnx_range_((ctx), (name))

#define GCNX_HIP(ctx, expr)                                                                   \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return gcnx_fail((ctx), GCNX_ERR_HIP, "%s: %s -> %s", __func__, #expr, hipGetErrorString(e_)); \
  } while (0)

#define GCNX_REQUIRE(ctx, cond, ...)                                          \
  do { if (!(cond)) return gcnx_fail((ctx), GCNX_ERR_INVALID, __VA_ARGS__); } while (0)

// Launch-error check that is legal during stream capture (no sync).
#define GCNX_LAUNCH_OK(ctx) GCNX_HIP(ctx, hipGetLastError())

static inline int gcnx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Blocks b and b+8 share an XCD (observed round-robin dealing; speed only, never correctness).
// Bijective remap that gives every XCD one contiguous range of logical work items, so that
// neighbouring row chunks -- which gather the same feature rows -- share one L2.
__device__ __forceinline__ int gcnx_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, k = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}
