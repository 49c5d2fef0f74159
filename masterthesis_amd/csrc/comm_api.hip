// Gradient exchange over RCCL / xGMI behind the C ABI: mt_comm_{unique_id, init, allreduce_async, wait, destroy}.
//
// Replaces the reference's only multi-GPU hook, the single-process nn.DataParallel wrap of every network
// (src/models/core/functions.py:98-101: replicate weights + scatter + gather + reduce-add on GPU 0 in every
// forward/backward), with one process per GPU and an all-reduce (average) of the flat fp32 gradient buffer of
// each network after its backward phase.  A communicator owns a side HIP stream: the
// collective starts as soon as the producing stream has finished the gradients (event edge) and the consumer
// (the Adam launch of THAT network) waits for its own buffer only, so exchanges overlap the Adam steps of the
// other networks and, for the discriminators, the generator phase that follows.
//
// RCCL is resolved at run time (dlopen of the librccl the process already holds -- PyTorch-ROCm loads one -- or
// the system one), so libmt_hip.so has no link-time dependency on it and loads on hosts without RCCL.
#include <dlfcn.h>
#include <string.h>
#include <vector>
#include "mt_common.h"

namespace {

typedef struct { char internal[128]; } rcclUniqueId;       // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rcclComm_t;
enum { RCCL_FLOAT32 = 7 };                                  // ncclFloat32
enum { RCCL_SUM = 0, RCCL_AVG = 4 };                        // ncclSum, ncclAvg

struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(rcclUniqueId*) = nullptr;
  int (*CommInitRank)(rcclComm_t*, int, rcclUniqueId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t) = nullptr;
  int (*CommDestroy)(rcclComm_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

RcclApi g_rccl;

int load_rccl() {
  if (g_rccl.lib) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)                               // the copy the process already mapped, if any
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) { mt_set_error("mt_comm: cannot load librccl (%s)", dlerror()); return 1; }
  g_rccl.GetUniqueId = (int (*)(rcclUniqueId*))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(rcclComm_t*, int, rcclUniqueId, int))dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, rcclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(rcclComm_t))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy) {
    mt_set_error("mt_comm: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
    return 1;
  }
  g_rccl.lib = h;
  return 0;
}

const char* rccl_err(int rc) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?"; }

}  // namespace

struct mt_comm {
  rcclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t side = nullptr;
  hipEvent_t ready = nullptr;              // producer stream -> side stream
  std::vector<hipEvent_t> done;            // one per in-flight handle, recycled round-robin
  unsigned next = 0;
};

static const int MT_COMM_HANDLES = 64;

extern "C" int mt_comm_unique_id(void* id128) {
  MT_CHECK(id128 != nullptr, "mt_comm_unique_id: null buffer");
  if (load_rccl()) return 1;
  rcclUniqueId id;
  const int rc = g_rccl.GetUniqueId(&id);
  MT_CHECK(rc == 0, "ncclGetUniqueId failed: %s", rccl_err(rc));
  memcpy(id128, id.internal, sizeof(id.internal));
  return 0;
}

extern "C" int mt_comm_init(mt_comm** out, int rank, int world, const void* id128, int device) {
  MT_CHECK(out != nullptr && id128 != nullptr, "mt_comm_init: null argument");
  MT_CHECK(world >= 1 && rank >= 0 && rank < world, "mt_comm_init: rank %d of %d", rank, world);
  if (load_rccl()) return 1;
  MT_CHECK(hipSetDevice(device) == hipSuccess, "mt_comm_init: hipSetDevice(%d) failed", device);
  mt_comm* c = new mt_comm();
  c->rank = rank; c->world = world; c->device = device;
  rcclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  const int rc = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (rc != 0) { delete c; mt_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, rccl_err(rc)); return 2; }
  bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&c->ready, hipEventDisableTiming) == hipSuccess;
  c->done.resize(MT_COMM_HANDLES, nullptr);
  for (int i = 0; ok && i < MT_COMM_HANDLES; i++) ok = hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) { mt_set_error("mt_comm_init: cannot create the side stream / events"); return 2; }
  *out = c;
  return 0;
}

// buf (fp32, `count` elements, in place) <- average over the ranks.  Enqueued on the communicator's side stream behind
// everything `producer` has been given so far; returns a handle (>= 0) for mt_comm_wait, or a negative value on error.
extern "C" int mt_comm_allreduce_async(mt_comm* c, float* buf, size_t count, mt_stream_t producer) {
  if (c == nullptr || buf == nullptr) { mt_set_error("mt_comm_allreduce_async: null argument"); return -1; }
  if (hipEventRecord(c->ready, (hipStream_t)producer) != hipSuccess ||
      hipStreamWaitEvent(c->side, c->ready, 0) != hipSuccess) { mt_set_error("mt_comm: event edge to the side stream failed"); return -2; }
  const int rc = g_rccl.AllReduce(buf, buf, count, RCCL_FLOAT32, RCCL_AVG, c->comm, c->side);
  if (rc != 0) { mt_set_error("ncclAllReduce failed: %s", rccl_err(rc)); return -3; }
  const int h = (int)(c->next++ % MT_COMM_HANDLES);
  if (hipEventRecord(c->done[h], c->side) != hipSuccess) { mt_set_error("mt_comm: event record failed"); return -4; }
  return h;
}

// `consumer` waits (on the device, not the host) until the all-reduce behind `handle` has finished
extern "C" int mt_comm_wait(mt_comm* c, int handle, mt_stream_t consumer) {
  MT_CHECK(c != nullptr && handle >= 0 && handle < MT_COMM_HANDLES, "mt_comm_wait: bad handle %d", handle);
  MT_CHECK(hipStreamWaitEvent((hipStream_t)consumer, c->done[handle], 0) == hipSuccess, "mt_comm_wait: hipStreamWaitEvent failed");
  return 0;
}

extern "C" int mt_comm_rank(const mt_comm* c) { return c ? c->rank : -1; }
extern "C" int mt_comm_world(const mt_comm* c) { return c ? c->world : -1; }

extern "C" int mt_comm_destroy(mt_comm* c) {
  if (c == nullptr) return 0;
  if (c->side) (void)hipStreamSynchronize(c->side);
  if (c->comm) g_rccl.CommDestroy(c->comm);
  for (hipEvent_t e : c->done) if (e) (void)hipEventDestroy(e);
  if (c->ready) (void)hipEventDestroy(c->ready);
  if (c->side) (void)hipStreamDestroy(c->side);
  delete c;
  return 0;
}
