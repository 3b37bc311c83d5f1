// rau_comm.hip -- native data-parallel gradient exchange for hosts without
// torch.distributed (the LuaJIT shim): RCCL all-reduce (average) of the three flat
// gradient buffers over xGMI.  The reference is single-GPU; this is the C-ABI form of
// rau_vqa_amd/dist.py.  librccl is bound with dlopen/dlsym on first use, so librau.so
// itself has no load-time dependency on it (and shares whichever RCCL the process already
// has, e.g. PyTorch's).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "rau_ctx.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (g_rccl.lib) return 0;
  void* lib = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(RAU_ERR_STATE, "RCCL not found (dlopen librccl.so: %s)", dlerror());
#define SYM(field, sym)                                                        \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, sym));    \
  if (!g_rccl.field) return fail(RAU_ERR_STATE, "RCCL symbol %s missing", sym)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(AllReduce, "ncclAllReduce");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(GetErrorString, "ncclGetErrorString");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
#undef SYM
  g_rccl.lib = lib;
  return 0;
}

#define NCCLC(expr)                                                                        \
  do {                                                                                     \
    ncclResult_t r_ = (expr);                                                              \
    if (r_ != ncclSuccess)                                                                 \
      return fail(RAU_ERR_DEVICE, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));      \
  } while (0)

}  // namespace

extern "C" {

int rau_comm_unique_id(void* id, size_t bytes) {
  NEED(id, "null argument");
  NEED(bytes >= sizeof(ncclUniqueId), "rau_comm_unique_id: buffer of %zu bytes, need %zu", bytes,
       sizeof(ncclUniqueId));
  if (int rc = rccl_load()) return rc;
  ncclUniqueId u;
  NCCLC(g_rccl.GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return RAU_OK;
}

int rau_comm_init(rau_ctx* ctx, int nranks, int rank, const void* id, size_t bytes) {
  NEED(ctx && id, "null argument");
  NEED(nranks >= 1 && rank >= 0 && rank < nranks, "rau_comm_init: rank %d of %d", rank, nranks);
  NEED(bytes >= sizeof(ncclUniqueId), "rau_comm_init: id of %zu bytes, need %zu", bytes,
       sizeof(ncclUniqueId));
  if (ctx->comm) return fail(RAU_ERR_STATE, "rau_comm_init: communicator already initialised");
  if (int rc = rccl_load()) return rc;
  HIPC(hipSetDevice(ctx->cfg.device_id));
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclComm_t comm = nullptr;
  NCCLC(g_rccl.CommInitRank(&comm, nranks, u, rank));
  ctx->comm = comm;
  ctx->comm_ranks = nranks;
  if (!ctx->st_comm) HIPC(hipStreamCreateWithFlags(&ctx->st_comm, hipStreamNonBlocking));
  if (!ctx->evC) HIPC(hipEventCreateWithFlags(&ctx->evC, hipEventDisableTiming));
  return RAU_OK;
}

// All-reduce (average) of the three flat gradient buffers, in place.  The mult bucket is
// final before the encoder BPTT has run (evD + evM3), so it is reduced on the side stream
// underneath the rest of rau_backward; rnn and embed follow at the end.  The ctx stream
// then waits for the reduced gradients: whatever is enqueued next (rau_noise_clip_adam)
// sees them.  No host synchronisation.
int rau_allreduce_grads(rau_ctx* ctx) {
  NEED(ctx, "null ctx");
  if (!ctx->comm) return fail(RAU_ERR_STATE, "rau_allreduce_grads: call rau_comm_init first");
  if (!ctx->bwd_done) return fail(RAU_ERR_STATE, "rau_allreduce_grads: no rau_backward to reduce");
  ncclComm_t comm = static_cast<ncclComm_t>(ctx->comm);
  hipStream_t sc = ctx->st_comm;
  if (ctx->graph_last) {   // the step ran as a graph: only its end is a waitable event
    HIPC(hipStreamWaitEvent(sc, ctx->evEnd, 0));
  } else {
    HIPC(hipStreamWaitEvent(sc, ctx->evD, 0));
    HIPC(hipStreamWaitEvent(sc, ctx->evM3, 0));
  }
  Group& gm = ctx->grp[RAU_GROUP_MULT];
  NCCLC(g_rccl.AllReduce(gm.g, gm.g, gm.n, ncclFloat, ncclAvg, comm, sc));
  HIPC(hipStreamWaitEvent(sc, ctx->evEnd, 0));
  NCCLC(g_rccl.GroupStart());
  for (int gi : {RAU_GROUP_RNN, RAU_GROUP_EMBED}) {
    Group& g = ctx->grp[gi];
    NCCLC(g_rccl.AllReduce(g.g, g.g, g.n, ncclFloat, ncclAvg, comm, sc));
  }
  NCCLC(g_rccl.GroupEnd());
  HIPC(hipEventRecord(ctx->evC, sc));
  HIPC(hipStreamWaitEvent(ctx->st, ctx->evC, 0));
  return RAU_OK;
}

int rau_comm_destroy(rau_ctx* ctx) {
  NEED(ctx, "null ctx");
  if (ctx->comm) {
    if (ctx->st_comm) hipStreamSynchronize(ctx->st_comm);
    NCCLC(g_rccl.CommDestroy(static_cast<ncclComm_t>(ctx->comm)));
    ctx->comm = nullptr;
    ctx->comm_ranks = 0;
  }
  return RAU_OK;
}

}  // extern "C"
