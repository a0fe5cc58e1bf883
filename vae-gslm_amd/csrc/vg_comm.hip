// Gradient exchange over RCCL behind the C ABI: vg_comm_unique_id / vg_comm_init / vg_allreduce_bucket /
// vg_comm_destroy (include/vaegslm_hip.h).  One communicator per process (one process per GPU); the bucket
// all-reduce is in place on the caller's comm stream, so it orders like any other launch of this library.
//
// RCCL is bound at run time (dlopen + dlsym), not at link time: the library keeps loading on a box without RCCL,
// and inside a PyTorch process it shares the copy PyTorch already mapped instead of mapping a second one.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "vg_common.h"
#include "../../include/vaegslm_hip.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
  ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
  const char* (*error_string)(ncclResult_t) = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;

// order: VG_RCCL_LIB, a copy the process already mapped, the ROCm install, the loader's search path
bool bind_rccl() {
  if (g_rccl.handle) return true;
  void* h = nullptr;
  if (const char* e = getenv("VG_RCCL_LIB")) h = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    vg_host::set_error("vg_comm: RCCL not found (%s); set VG_RCCL_LIB", dlerror());
    return false;
  }
  Rccl r;
  r.handle = h;
  r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
  r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
  r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(h, "ncclAllReduce"));
  r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
  r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(h, "ncclGetErrorString"));
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy || !r.error_string) {
    vg_host::set_error("vg_comm: the RCCL library lacks an expected symbol");
    return false;
  }
  g_rccl = r;
  return true;
}

int fail(const char* what, ncclResult_t rc) {
  vg_host::set_error("%s: %s", what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
  return 1;
}

}  // namespace

extern "C" int vg_comm_unique_id(void* out, int nbytes) {
  VG_REQUIRE(out != nullptr && nbytes >= VG_COMM_ID_BYTES, "vg_comm_unique_id: need a %d-byte buffer", VG_COMM_ID_BYTES);
  static_assert(VG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "header constant out of date");
  std::lock_guard<std::mutex> lock(g_mu);
  if (!bind_rccl()) return 1;
  ncclUniqueId id;
  const ncclResult_t rc = g_rccl.get_unique_id(&id);
  if (rc != ncclSuccess) return fail("ncclGetUniqueId", rc);
  memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int vg_comm_init(int rank, int world, const void* unique_id, int nbytes) {
  VG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "vg_comm_init: rank %d of %d", rank, world);
  VG_REQUIRE(unique_id != nullptr && nbytes >= VG_COMM_ID_BYTES, "vg_comm_init: need the %d-byte id of rank 0",
             VG_COMM_ID_BYTES);
  std::lock_guard<std::mutex> lock(g_mu);
  VG_REQUIRE(g_comm == nullptr, "vg_comm_init: communicator already initialised (rank %d of %d)", g_rank, g_world);
  if (!bind_rccl()) return 1;
  ncclUniqueId id;
  memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  const ncclResult_t rc = g_rccl.comm_init_rank(&comm, world, id, rank);   // uses the calling thread's current device
  if (rc != ncclSuccess) return fail("ncclCommInitRank", rc);
  g_comm = comm;
  g_world = world;
  g_rank = rank;
  return 0;
}

extern "C" int vg_comm_world(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_comm ? g_world : 0;
}

extern "C" int vg_allreduce_bucket(void* buf, int64_t n, int dtype, int average, hipStream_t comm_stream) {
  VG_REQUIRE(buf != nullptr && n > 0, "vg_allreduce_bucket: empty bucket");
  VG_REQUIRE(dtype == VG_F32 || dtype == VG_BF16, "vg_allreduce_bucket: bad dtype %d", dtype);
  ncclComm_t comm;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    comm = g_comm;
  }
  VG_REQUIRE(comm != nullptr, "vg_allreduce_bucket: call vg_comm_init first");
  const ncclResult_t rc = g_rccl.all_reduce(buf, buf, (size_t)n, dtype == VG_F32 ? ncclFloat32 : ncclBfloat16,
                                            average ? ncclAvg : ncclSum, comm, comm_stream);
  if (rc != ncclSuccess) return fail("ncclAllReduce", rc);
  return 0;
}

extern "C" int vg_comm_destroy(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_comm == nullptr) return 0;
  const ncclResult_t rc = g_rccl.comm_destroy(g_comm);
  g_comm = nullptr;
  g_world = 0;
  g_rank = -1;
  if (rc != ncclSuccess) return fail("ncclCommDestroy", rc);
  return 0;
}
