// RCCL behind the C ABI: the collectives of train.py's DDP wrapper (train.py:106 init_process_group, :152 DDP ctor broadcast, :257
// gradient all-reduce, :274 loss all-reduce, :297 barrier) as plain-pointer entry points on the library's own communicator.
// One communicator per process (= per GPU), created from a 128-byte unique id that rank 0 makes and the host side hands to every
// rank (torchrun's store, a file, MPI -- the library does not care).  Collectives are enqueued on the caller's stream, so they
// are ordered with the kernels that produce / consume the buffers exactly like every other entry point; the overlap with the
// backward pass is the host's doing (a side stream + events, as osu_diffusion_amd/training.py does).
//
// librccl is resolved at run time (dlopen of the copy the process already has -- PyTorch-ROCm ships one -- else the system's):
// libosud.so has no link-time dependency on it and loads on machines without RCCL; only osud_comm_* then fail, loudly.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

#include "common.h"

struct osud_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = -1;
};

namespace osud {
namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};

// Loaded once (std::call_once: the first collective may come from any thread); why a load failed is kept for the error message
// (dlerror() hands its string out once and then forgets it).
struct RcclLoad {
  Rccl r;
  std::string why;
  bool ok = false;
};

RcclLoad& rccl_load() {
  static RcclLoad L;
  static std::once_flag once;
  std::call_once(once, [] {
    Rccl& r = L.r;
    const char* override_path = getenv("OSUD_RCCL_LIB");
    const char* names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n) continue;
      r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
      const char* e = dlerror();
      L.why += std::string(L.why.empty() ? "" : "; ") + n + ": " + (e ? e : "dlopen failed");
    }
    if (!r.lib) return;
#define OSUD_SYM(field, name)                                          \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name)); \
    if (!r.field) {                                                    \
      const char* e = dlerror();                                       \
      L.why = std::string("symbol ") + name + ": " + (e ? e : "missing"); \
      return;                                                          \
    }
    OSUD_SYM(GetUniqueId, "ncclGetUniqueId")
    OSUD_SYM(CommInitRank, "ncclCommInitRank")
    OSUD_SYM(CommDestroy, "ncclCommDestroy")
    OSUD_SYM(AllReduce, "ncclAllReduce")
    OSUD_SYM(Broadcast, "ncclBroadcast")
    OSUD_SYM(ReduceScatter, "ncclReduceScatter")
    OSUD_SYM(AllGather, "ncclAllGather")
    OSUD_SYM(GetErrorString, "ncclGetErrorString")
    OSUD_SYM(GetVersion, "ncclGetVersion")
#undef OSUD_SYM
    L.ok = true;
  });
  return L;
}

Rccl* rccl() {
  RcclLoad& L = rccl_load();
  return L.ok ? &L.r : nullptr;
}

int need(Rccl** out) {
  *out = rccl();
  if (!*out) {
    set_error("osud_comm: librccl could not be loaded (OSUD_RCCL_LIB, librccl.so.1, librccl.so, /opt/rocm/lib/librccl.so.1): %s",
              rccl_load().why.c_str());
    return OSUD_ERR_UNSUPPORTED;
  }
  return OSUD_OK;
}

#define OSUD_NCCL(R, call)                                                                  \
  do {                                                                                      \
    ncclResult_t e__ = (call);                                                              \
    if (e__ != ncclSuccess) {                                                               \
      set_error("RCCL error %d (%s) in `%s`", (int)e__, (R)->GetErrorString(e__), #call);   \
      return OSUD_ERR_HIP;                                                                  \
    }                                                                                       \
  } while (0)

ncclDataType_t dtype_of(int wire) { return wire == OSUD_WIRE_BF16 ? ncclBfloat16 : ncclFloat32; }

}  // namespace
}  // namespace osud

using namespace osud;

extern "C" int osud_comm_unique_id(void* uid128) {
  OSUD_CHECK_ARG(uid128, "comm_unique_id: null argument");
  Rccl* R;
  OSUD_TRY(need(&R));
  static_assert(sizeof(ncclUniqueId) == OSUD_COMM_UID_BYTES, "unique id size");
  OSUD_NCCL(R, R->GetUniqueId(reinterpret_cast<ncclUniqueId*>(uid128)));
  return OSUD_OK;
}

extern "C" int osud_comm_init(int rank, int world, const void* uid128, osud_comm** out) {
  OSUD_CHECK_ARG(out && uid128 && world >= 1 && rank >= 0 && rank < world, "comm_init: bad rank %d / world %d", rank, world);
  Rccl* R;
  OSUD_TRY(need(&R));
  osud_comm* c = new osud_comm();
  c->rank = rank;
  c->world = world;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    return hip_fail(hipErrorNoDevice, "hipGetDevice", __FILE__, __LINE__);
  }
  ncclUniqueId id;
  memcpy(&id, uid128, sizeof id);
  const ncclResult_t e = R->CommInitRank(&c->comm, world, id, rank);
  if (e != ncclSuccess) {
    set_error("RCCL error %d (%s) in ncclCommInitRank(rank %d of %d)", (int)e, R->GetErrorString(e), rank, world);
    delete c;
    return OSUD_ERR_HIP;
  }
  *out = c;
  return OSUD_OK;
}

extern "C" void osud_comm_destroy(osud_comm* c) {
  if (!c) return;
  Rccl* R = rccl();
  if (R && c->comm) (void)R->CommDestroy(c->comm);
  delete c;
}

extern "C" int osud_comm_rank(const osud_comm* c) { return c ? c->rank : -1; }
extern "C" int osud_comm_world(const osud_comm* c) { return c ? c->world : -1; }
extern "C" int osud_comm_rccl_version(void) {
  Rccl* R = rccl();
  int v = 0;
  if (!R || R->GetVersion(&v) != ncclSuccess) return 0;
  return v;
}

// SUM all-reduce in place of n elements (fp32, or bf16 when the gradients were rounded for the wire); the 1/world of DDP's mean
// is folded into osud_adamw_ema_step's grad_scale.
extern "C" int osud_allreduce_grads(osud_comm* c, void* buf, size_t n, int wire, osud_stream stream) {
  OSUD_CHECK_ARG(c && buf && n > 0, "allreduce_grads: bad argument");
  Rccl* R;
  OSUD_TRY(need(&R));
  OSUD_NCCL(R, R->AllReduce(buf, buf, n, dtype_of(wire), ncclSum, c->comm, (hipStream_t)stream));
  return OSUD_OK;
}

extern "C" int osud_broadcast_params(osud_comm* c, float* buf, size_t n, int root, osud_stream stream) {
  OSUD_CHECK_ARG(c && buf && n > 0 && root >= 0 && root < c->world, "broadcast_params: bad argument");
  Rccl* R;
  OSUD_TRY(need(&R));
  OSUD_NCCL(R, R->Broadcast(buf, buf, n, ncclFloat32, root, c->comm, (hipStream_t)stream));
  return OSUD_OK;
}

// rank r receives the SUM of elements [r * n_per_rank, (r + 1) * n_per_rank) of every rank's `in` (world * n_per_rank elements)
extern "C" int osud_reduce_scatter_grads(osud_comm* c, const void* in, void* out_shard, size_t n_per_rank, int wire, osud_stream stream) {
  OSUD_CHECK_ARG(c && in && out_shard && n_per_rank > 0, "reduce_scatter_grads: bad argument");
  Rccl* R;
  OSUD_TRY(need(&R));
  OSUD_NCCL(R, R->ReduceScatter(in, out_shard, n_per_rank, dtype_of(wire), ncclSum, c->comm, (hipStream_t)stream));
  return OSUD_OK;
}

// every rank contributes n_per_rank fp32 elements; `full` receives them in rank order (shard may be full + rank * n_per_rank: in place)
extern "C" int osud_allgather_params(osud_comm* c, const float* shard, float* full, size_t n_per_rank, osud_stream stream) {
  OSUD_CHECK_ARG(c && shard && full && n_per_rank > 0, "allgather_params: bad argument");
  Rccl* R;
  OSUD_TRY(need(&R));
  OSUD_NCCL(R, R->AllGather(shard, full, n_per_rank, ncclFloat32, c->comm, (hipStream_t)stream));
  return OSUD_OK;
}
