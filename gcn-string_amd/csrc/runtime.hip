// libgcnx runtime: context lifecycle, device memory, events, HIP-graph capture.
#include <cstdlib>
#include <new>
#include <utility>

#include "common.h"
#include <dlfcn.h>
#include <cstring>

thread_local std::string gcnx_tls_error;

int gcnx_fail(gcnx_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  gcnx_tls_error = buf;
  if (ctx) ctx->err = buf;
  return code;
}

int gcnx_ws_reserve(gcnx_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return GCNX_OK;
  if (ctx->capturing)
    return gcnx_fail(ctx, GCNX_ERR_INVALID,
                     "workspace of %zu bytes needed during stream capture (have %zu): run the "
                     "call sequence once eagerly before capturing", bytes, ctx->ws_bytes);
  size_t want = bytes + (bytes >> 2);
  if (ctx->ws) {
    GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->live_graphs > 0) {
      // a captured graph may hold this pointer (split-K slabs, head partials, BN partials): keep the block alive
      try { ctx->retired_ws.push_back(ctx->ws); } catch (...) { return gcnx_fail(ctx, GCNX_ERR_NOMEM, "out of host memory"); }
    } else {
      GCNX_HIP(ctx, hipFree(ctx->ws));
    }
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  hipError_t e = hipMalloc(&ctx->ws, want);
  if (e != hipSuccess) return gcnx_fail(ctx, GCNX_ERR_NOMEM, "workspace hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
  ctx->ws_bytes = want;
  return GCNX_OK;
}

extern "C" {

int gcnx_version(void) { return 400; }   // round 3: + plan_bind, wimage *, bf16 storage, stream images, relu_bits_pool, pooled head; 310: dropout, counter_add, add; 311: spmm_csr_minmax(_bwd); 400 (r4): spmm_csr_prod(_bwd)

int gcnx_device_count(int* n) {
  if (!n) return gcnx_fail(nullptr, GCNX_ERR_INVALID, "gcnx_device_count: n is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *n = 0; return gcnx_fail(nullptr, GCNX_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
  *n = c;
  return GCNX_OK;
}

int gcnx_ctx_create(int device, gcnx_ctx** out) {
  if (!out) return gcnx_fail(nullptr, GCNX_ERR_INVALID, "gcnx_ctx_create: out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return gcnx_fail(nullptr, GCNX_ERR_HIP, "gcnx_ctx_create: no HIP device (%s); libgcnx has no CPU fallback",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device < 0 || device >= count)
    return gcnx_fail(nullptr, GCNX_ERR_INVALID, "gcnx_ctx_create: device %d outside [0,%d)", device, count);
  gcnx_ctx* ctx = new (std::nothrow) gcnx_ctx();
  if (!ctx) return gcnx_fail(nullptr, GCNX_ERR_NOMEM, "gcnx_ctx_create: out of host memory");
  ctx->device = device;
  e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking);
  ctx->main_stream = ctx->stream;
  for (int i = 0; i < gcnx_ctx::kSideEvents && e == hipSuccess; ++i) {
    e = hipEventCreateWithFlags(&ctx->ev_fork[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->flag, 4 * sizeof(int));
  if (e == hipSuccess) e = hipMemset(ctx->flag, 0, 4 * sizeof(int));   // [3] is the head kernel's arrival ticket
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    int rc = gcnx_fail(nullptr, GCNX_ERR_HIP, "gcnx_ctx_create(device %d): %s", device, hipGetErrorString(e));
    delete ctx;
    return rc;
  }
  ctx->num_cus = prop.multiProcessorCount;
  ctx->arch = prop.gcnArchName;
  if (const char* k = getenv("GCNX_SPMM_KERNEL")) ctx->knob_spmm_kernel = k[0] == 'r' ? 1 : k[0] == 't' ? 2 : k[0] == 'p' ? 3 : 0;
  if (const char* k = getenv("GCNX_SPMM_SLAB")) ctx->knob_spmm_slab = atoi(k);
  if (const char* k = getenv("GCNX_SPMM_SG")) ctx->knob_spmm_sg = atoi(k);
  if (const char* k = getenv("GCNX_GEMM_STREAM")) ctx->knob_gemm_stream = atoi(k);
  if (const char* k = getenv("GCNX_SPMM_CONC")) ctx->knob_spmm_conc = atoi(k);
  if (const char* k = getenv("GCNX_SPMM_TILE_WGS")) ctx->knob_spmm_tile_wgs = atoi(k);
  if (const char* k = getenv("GCNX_POOL_SPLIT")) ctx->knob_pool_split = atoi(k);
  if (const char* k = getenv("GCNX_SPMM_TALL_RPC")) ctx->knob_spmm_tall_rpc = atoi(k);
  if (const char* k = getenv("GCNX_SPMM_SORT_WIN")) ctx->knob_spmm_sort_win = atoi(k);
  if (const char* k = getenv("GCNX_SPMM_CAP1")) ctx->knob_spmm_cap1 = atoi(k);
  if (const char* k = getenv("GCNX_SPMM_BAL")) ctx->knob_spmm_bal = atoi(k);
  if (const char* k = getenv("GCNX_SPMM_CB")) ctx->knob_spmm_cb = atoi(k);
  if (const char* k = getenv("GCNX_ROCTX")) {
    if (atoi(k)) {                        // tracing aid: named ranges around the kernel classes (rocprofv3 --marker-trace)
      void* lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
      if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
      if (lib) {
        ctx->roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
        ctx->roctx_pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
        if (!ctx->roctx_push || !ctx->roctx_pop) ctx->roctx_push = nullptr, ctx->roctx_pop = nullptr;
      }
    }
  }
  if (ctx->arch.rfind("gfx950", 0) != 0) {
    int rc = gcnx_fail(nullptr, GCNX_ERR_UNSUPPORTED, "gcnx_ctx_create: device %d is %s; libgcnx is built for gfx950 only",
                       device, ctx->arch.c_str());
    (void)hipFree(ctx->flag);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return rc;
  }
  *out = ctx;
  return GCNX_OK;
}

int gcnx_ctx_destroy(gcnx_ctx* ctx) {
  if (!ctx) return GCNX_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->main_stream);
  if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->ws_other) (void)hipFree(ctx->ws_other);
  for (void* p : ctx->retired_ws) (void)hipFree(p);
  if (ctx->flag) (void)hipFree(ctx->flag);
  for (int i = 0; i < 2; ++i) if (ctx->aux_stream[i]) { (void)hipStreamSynchronize(ctx->aux_stream[i]); (void)hipStreamDestroy(ctx->aux_stream[i]); }
  for (int i = 0; i < gcnx_ctx::kAuxEvents; ++i)
    for (int j = 0; j < 3; ++j) if (ctx->aux_ev[i][j]) (void)hipEventDestroy(ctx->aux_ev[i][j]);
  if (ctx->pin_base) {
    (void)hipHostFree(ctx->pin_base);
    for (int i = 0; i < gcnx_ctx::kPinSlots; ++i) if (ctx->pin_ev[i]) (void)hipEventDestroy(ctx->pin_ev[i]);
  }
  for (int i = 0; i < gcnx_ctx::kSideEvents; ++i) {
    if (ctx->ev_fork[i]) (void)hipEventDestroy(ctx->ev_fork[i]);
    if (ctx->ev_join[i]) (void)hipEventDestroy(ctx->ev_join[i]);
  }
  (void)hipStreamDestroy(ctx->main_stream);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  delete ctx;
  return GCNX_OK;
}

const char* gcnx_last_error(gcnx_ctx* ctx) {
  if (ctx && !ctx->err.empty()) return ctx->err.c_str();
  return gcnx_tls_error.c_str();
}

int gcnx_set_tuning(gcnx_ctx* ctx, const char* key, int value) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, key != nullptr, "gcnx_set_tuning: key is NULL");
  const std::string k = key;
  if (k == "spmm_kernel") ctx->knob_spmm_kernel = value;
  else if (k == "spmm_slab") ctx->knob_spmm_slab = value;
  else if (k == "spmm_sg") ctx->knob_spmm_sg = value;
  else if (k == "gemm_stream") ctx->knob_gemm_stream = value;
  else if (k == "spmm_conc") ctx->knob_spmm_conc = value;
  else if (k == "spmm_tile_wgs") ctx->knob_spmm_tile_wgs = value;
  else if (k == "pool_split") ctx->knob_pool_split = value;
  else if (k == "spmm_tall_rpc") ctx->knob_spmm_tall_rpc = value;
  else if (k == "spmm_sort_win") ctx->knob_spmm_sort_win = value;
  else if (k == "spmm_cap1") ctx->knob_spmm_cap1 = value;
  else if (k == "spmm_bal") ctx->knob_spmm_bal = value;
  else if (k == "spmm_cb") ctx->knob_spmm_cb = value;
  else return gcnx_fail(ctx, GCNX_ERR_INVALID, "gcnx_set_tuning: unknown key '%s'", key);
  return GCNX_OK;
}

int gcnx_device_info(gcnx_ctx* ctx, char* name, int len, int* cus, size_t* hbm_bytes) {
  GCNX_CHECK_CTX(ctx);
  if (name && len > 0) snprintf(name, (size_t)len, "%s", ctx->arch.c_str());
  if (cus) *cus = ctx->num_cus;
  if (hbm_bytes) {
    size_t fr = 0, tot = 0;
    GCNX_HIP(ctx, hipSetDevice(ctx->device));
    GCNX_HIP(ctx, hipMemGetInfo(&fr, &tot));
    *hbm_bytes = tot;
  }
  return GCNX_OK;
}

int gcnx_malloc(gcnx_ctx* ctx, size_t bytes, void** dptr) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, dptr != nullptr, "gcnx_malloc: dptr is NULL");
  *dptr = nullptr;
  if (bytes == 0) return GCNX_OK;
  GCNX_HIP(ctx, hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(dptr, bytes);
  if (e != hipSuccess) return gcnx_fail(ctx, GCNX_ERR_NOMEM, "gcnx_malloc(%zu): %s", bytes, hipGetErrorString(e));
  return GCNX_OK;
}

int gcnx_free(gcnx_ctx* ctx, void* dptr) {
  GCNX_CHECK_CTX(ctx);
  if (!dptr) return GCNX_OK;
  GCNX_HIP(ctx, hipSetDevice(ctx->device));
  GCNX_HIP(ctx, hipFree(dptr));
  return GCNX_OK;
}

int gcnx_memset(gcnx_ctx* ctx, void* dptr, int value, size_t bytes) {
  GCNX_CHECK_CTX(ctx);
  if (bytes == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, dptr != nullptr, "gcnx_memset: dptr is NULL");
  GCNX_HIP(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
  return GCNX_OK;
}

int gcnx_h2d(gcnx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  GCNX_CHECK_CTX(ctx);
  if (bytes == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, dst && src, "gcnx_h2d: NULL pointer");
  GCNX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GCNX_OK;
}

// Small host -> device copies that must not stall the host behind the GPU (the per-batch descriptor of the device-side
// collate: with the synchronous copy every streamed step waited for the previous one to finish before its launches
// could even be queued).  The bytes are taken from `src` before the call returns.
int gcnx_h2d_async(gcnx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  GCNX_CHECK_CTX(ctx);
  if (bytes == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, dst && src, "gcnx_h2d_async: NULL pointer");
  if (bytes > gcnx_ctx::kPinSlotBytes || ctx->capturing) return gcnx_h2d(ctx, dst, src, bytes);
  if (!ctx->pin_base) {
    GCNX_HIP(ctx, hipHostMalloc((void**)&ctx->pin_base, gcnx_ctx::kPinSlots * gcnx_ctx::kPinSlotBytes, hipHostMallocDefault));
    for (int i = 0; i < gcnx_ctx::kPinSlots; ++i) GCNX_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
  }
  const int slot = ctx->pin_next;
  ctx->pin_next = (slot + 1) % gcnx_ctx::kPinSlots;
  if (ctx->pin_busy[slot]) GCNX_HIP(ctx, hipEventSynchronize(ctx->pin_ev[slot]));   // 32 copies ago: almost never waits
  char* stage = ctx->pin_base + (size_t)slot * gcnx_ctx::kPinSlotBytes;
  memcpy(stage, src, bytes);
  GCNX_HIP(ctx, hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, ctx->stream));
  GCNX_HIP(ctx, hipEventRecord(ctx->pin_ev[slot], ctx->stream));
  ctx->pin_busy[slot] = true;
  return GCNX_OK;
}

int gcnx_d2h(gcnx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  GCNX_CHECK_CTX(ctx);
  if (bytes == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, dst && src, "gcnx_d2h: NULL pointer");
  if (ctx->side_pending && !ctx->on_side) { int rc = gcnx_side_join(ctx); if (rc) return rc; }   // results of side sections too
  GCNX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return GCNX_OK;
}

int gcnx_d2d(gcnx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  GCNX_CHECK_CTX(ctx);
  if (bytes == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, dst && src, "gcnx_d2d: NULL pointer");
  GCNX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return GCNX_OK;
}

int gcnx_sync(gcnx_ctx* ctx) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, !ctx->on_side, "gcnx_sync inside a side section");
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->main_stream));
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->side_stream));
  ctx->side_pending = false;
  return GCNX_OK;
}

// ---- side sections -------------------------------------------------------------------------------------------
// begin: everything launched until gcnx_side_end runs on the side stream, ordered after all work submitted to
// the main stream so far, concurrently with what the main stream is given next.  join: the main stream waits
// for the side sections ended so far.  Works inside stream capture (the side stream joins the capture through
// the fork event; the graph then has two branches).
static void gcnx_swap_streams(gcnx_ctx* ctx) {
  ctx->stream = ctx->on_side ? ctx->main_stream : ctx->side_stream;
  std::swap(ctx->ws, ctx->ws_other);
  std::swap(ctx->ws_bytes, ctx->ws_other_bytes);
  ctx->on_side = !ctx->on_side;
}

int gcnx_aux_fork(gcnx_ctx* ctx, hipStream_t out[2]) {
  if (!ctx->aux_stream[0]) {
    for (int i = 0; i < 2; ++i) GCNX_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream[i], hipStreamNonBlocking));
    for (int i = 0; i < gcnx_ctx::kAuxEvents; ++i)
      for (int j = 0; j < 3; ++j) GCNX_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_ev[i][j], hipEventDisableTiming));
  }
  hipEvent_t* ev = ctx->aux_ev[ctx->aux_next % gcnx_ctx::kAuxEvents];
  GCNX_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
  for (int i = 0; i < 2; ++i) {
    GCNX_HIP(ctx, hipStreamWaitEvent(ctx->aux_stream[i], ev[0], 0));
    out[i] = ctx->aux_stream[i];
  }
  return GCNX_OK;
}

int gcnx_aux_join(gcnx_ctx* ctx) {
  hipEvent_t* ev = ctx->aux_ev[ctx->aux_next % gcnx_ctx::kAuxEvents];
  ctx->aux_next++;
  for (int i = 0; i < 2; ++i) {
    GCNX_HIP(ctx, hipEventRecord(ev[1 + i], ctx->aux_stream[i]));
    GCNX_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev[1 + i], 0));
  }
  return GCNX_OK;
}

int gcnx_side_begin(gcnx_ctx* ctx) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, !ctx->on_side, "gcnx_side_begin: already inside a side section");
  hipEvent_t ev = ctx->ev_fork[ctx->ev_next % gcnx_ctx::kSideEvents];
  GCNX_HIP(ctx, hipEventRecord(ev, ctx->main_stream));
  GCNX_HIP(ctx, hipStreamWaitEvent(ctx->side_stream, ev, 0));
  gcnx_swap_streams(ctx);
  return GCNX_OK;
}

int gcnx_side_end(gcnx_ctx* ctx) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, ctx->on_side, "gcnx_side_end: not inside a side section");
  gcnx_swap_streams(ctx);
  ctx->side_pending = true;
  return GCNX_OK;
}

int gcnx_side_join(gcnx_ctx* ctx) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, !ctx->on_side, "gcnx_side_join inside a side section");
  if (!ctx->side_pending) return GCNX_OK;
  hipEvent_t ev = ctx->ev_join[ctx->ev_next % gcnx_ctx::kSideEvents];
  ctx->ev_next++;
  GCNX_HIP(ctx, hipEventRecord(ev, ctx->side_stream));
  GCNX_HIP(ctx, hipStreamWaitEvent(ctx->main_stream, ev, 0));
  ctx->side_pending = false;
  return GCNX_OK;
}

int gcnx_event_create(gcnx_ctx* ctx, gcnx_event** out) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, out != nullptr, "gcnx_event_create: out is NULL");
  gcnx_event* ev = new (std::nothrow) gcnx_event();
  if (!ev) return gcnx_fail(ctx, GCNX_ERR_NOMEM, "gcnx_event_create: out of host memory");
  hipError_t e = hipEventCreate(&ev->ev);
  if (e != hipSuccess) { delete ev; return gcnx_fail(ctx, GCNX_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e)); }
  *out = ev;
  return GCNX_OK;
}

int gcnx_event_record(gcnx_ctx* ctx, gcnx_event* ev) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, ev != nullptr, "gcnx_event_record: event is NULL");
  GCNX_HIP(ctx, hipEventRecord(ev->ev, ctx->stream));
  return GCNX_OK;
}

int gcnx_event_elapsed_ms(gcnx_ctx* ctx, gcnx_event* start, gcnx_event* stop, float* ms) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, start && stop && ms, "gcnx_event_elapsed_ms: NULL argument");
  GCNX_HIP(ctx, hipEventSynchronize(stop->ev));
  GCNX_HIP(ctx, hipEventElapsedTime(ms, start->ev, stop->ev));
  return GCNX_OK;
}

int gcnx_event_destroy(gcnx_ctx* ctx, gcnx_event* ev) {
  GCNX_CHECK_CTX(ctx);
  if (!ev) return GCNX_OK;
  (void)hipEventDestroy(ev->ev);
  delete ev;
  return GCNX_OK;
}

int gcnx_capture_begin(gcnx_ctx* ctx) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, !ctx->capturing, "gcnx_capture_begin: a capture is already active");
  GCNX_REQUIRE(ctx, !ctx->on_side, "gcnx_capture_begin inside a side section");
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->side_stream));
  GCNX_HIP(ctx, hipStreamSynchronize(ctx->stream));
  GCNX_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  ctx->capturing = true;
  return GCNX_OK;
}

int gcnx_capture_end(gcnx_ctx* ctx, gcnx_graph** out) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, ctx->capturing, "gcnx_capture_end: no capture active");
  GCNX_REQUIRE(ctx, out != nullptr, "gcnx_capture_end: out is NULL");
  GCNX_REQUIRE(ctx, !ctx->on_side, "gcnx_capture_end inside a side section");
  if (ctx->side_pending) { int rc = gcnx_side_join(ctx); if (rc) return rc; }   // a capture must end joined
  ctx->capturing = false;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(ctx->stream, &g);
  if (e != hipSuccess || !g)
    return gcnx_fail(ctx, GCNX_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    return gcnx_fail(ctx, GCNX_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  gcnx_graph* gg = new (std::nothrow) gcnx_graph();
  if (!gg) { (void)hipGraphExecDestroy(ex); (void)hipGraphDestroy(g); return gcnx_fail(ctx, GCNX_ERR_NOMEM, "out of host memory"); }
  gg->graph = g;
  gg->exec = ex;
  *out = gg;
  ctx->live_graphs++;
  return GCNX_OK;
}

int gcnx_graph_launch(gcnx_ctx* ctx, gcnx_graph* g) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, g != nullptr, "gcnx_graph_launch: graph is NULL");
  GCNX_HIP(ctx, hipGraphLaunch(g->exec, ctx->stream));
  return GCNX_OK;
}

int gcnx_graph_destroy(gcnx_ctx* ctx, gcnx_graph* g) {
  GCNX_CHECK_CTX(ctx);
  if (!g) return GCNX_OK;
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  if (--ctx->live_graphs <= 0) {           // nobody can replay into a retired workspace block any more
    ctx->live_graphs = 0;
    (void)hipStreamSynchronize(ctx->main_stream);
    (void)hipStreamSynchronize(ctx->side_stream);
    for (void* p : ctx->retired_ws) (void)hipFree(p);
    ctx->retired_ws.clear();
  }
  return GCNX_OK;
}

}  // extern "C"
