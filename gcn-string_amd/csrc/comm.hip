// Gradient all-reduce across the GPUs of one node: one process per GPU, RCCL over xGMI.
// The reference is single-device; this is new capability whose acceptance test is
// "N-GPU loss/gradients == 1-GPU loss/gradients on the same batch".
//
// librccl is opened lazily so that single-GPU use neither needs nor loads it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <new>

#include "common.h"

struct gcnx_comm {
  ncclComm_t comm;
  int nranks;
  int rank;
};

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int load_rccl() {
  if (g_rccl.handle) return GCNX_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return gcnx_fail(nullptr, GCNX_ERR_RCCL, "cannot dlopen librccl: %s", dlerror());
  RcclApi api;
  api.handle = h;
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
  api.AllReduce = (decltype(api.AllReduce))dlsym(h, "ncclAllReduce");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
    dlclose(h);
    return gcnx_fail(nullptr, GCNX_ERR_RCCL, "librccl lacks an expected nccl* symbol");
  }
  g_rccl = api;
  return GCNX_OK;
}

}  // namespace

static_assert(sizeof(ncclUniqueId) == GCNX_UNIQUE_ID_BYTES, "ncclUniqueId size changed");

extern "C" {

int gcnx_comm_unique_id(char id[GCNX_UNIQUE_ID_BYTES]) {
  if (!id) return gcnx_fail(nullptr, GCNX_ERR_INVALID, "gcnx_comm_unique_id: id is NULL");
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId uid;
  ncclResult_t r = g_rccl.GetUniqueId(&uid);
  if (r != ncclSuccess) return gcnx_fail(nullptr, GCNX_ERR_RCCL, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
  memcpy(id, &uid, GCNX_UNIQUE_ID_BYTES);
  return GCNX_OK;
}

int gcnx_comm_init_rank(gcnx_ctx* ctx, const char id[GCNX_UNIQUE_ID_BYTES], int nranks, int rank, gcnx_comm** out) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, id && out, "gcnx_comm_init_rank: NULL argument");
  GCNX_REQUIRE(ctx, nranks >= 1 && rank >= 0 && rank < nranks, "gcnx_comm_init_rank: rank %d of %d", rank, nranks);
  *out = nullptr;
  int rc = load_rccl();
  if (rc) { ctx->err = gcnx_tls_error; return rc; }
  GCNX_HIP(ctx, hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(&uid, id, GCNX_UNIQUE_ID_BYTES);
  ncclComm_t c = nullptr;
  ncclResult_t r = g_rccl.CommInitRank(&c, nranks, uid, rank);
  if (r != ncclSuccess) return gcnx_fail(ctx, GCNX_ERR_RCCL, "ncclCommInitRank(rank %d/%d): %s", rank, nranks, g_rccl.GetErrorString(r));
  gcnx_comm* cm = new (std::nothrow) gcnx_comm{c, nranks, rank};
  if (!cm) { g_rccl.CommDestroy(c); return gcnx_fail(ctx, GCNX_ERR_NOMEM, "out of host memory"); }
  *out = cm;
  return GCNX_OK;
}

int gcnx_comm_destroy(gcnx_comm* comm) {
  if (!comm) return GCNX_OK;
  if (g_rccl.CommDestroy) g_rccl.CommDestroy(comm->comm);
  delete comm;
  return GCNX_OK;
}

int gcnx_allreduce_f32(gcnx_ctx* ctx, gcnx_comm* comm, float* buf, int64_t n, int op) {
  GCNX_CHECK_CTX(ctx);
  GCNX_REQUIRE(ctx, comm != nullptr, "gcnx_allreduce_f32: comm is NULL");
  GCNX_REQUIRE(ctx, n >= 0, "gcnx_allreduce_f32: negative count");
  GCNX_REQUIRE(ctx, op == GCNX_RED_SUM || op == GCNX_RED_MAX, "gcnx_allreduce_f32: unknown op %d", op);
  if (n == 0) return GCNX_OK;
  GCNX_REQUIRE(ctx, buf != nullptr, "gcnx_allreduce_f32: buf is NULL");
  ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, op == GCNX_RED_SUM ? ncclSum : ncclMax,
                                    comm->comm, ctx->stream);
  if (r != ncclSuccess) return gcnx_fail(ctx, GCNX_ERR_RCCL, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
  return GCNX_OK;
}

}  // extern "C"
