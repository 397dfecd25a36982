// RCCL under the C ABI (include/zs3hip.h "collectives"): communicators and all-reduces issued by the library itself on the stream
// the caller names -- the compute stream for the SyncBN sums (pack -> all-reduce -> finalize on ONE stream: no event hand-over
// between a framework's collective stream and the kernel chain, 208 times per step), the weight-gradient side stream for the
// gradient buckets.  Being ordinary entry points with a stream argument, the collectives are recorded and replayed by a launch plan
// (csrc/plan.h) like any kernel launch: the N > 1 step runs from C too.
//
// RCCL is bound at run time (dlopen of the librccl the process already uses: the one PyTorch loaded -- the caller passes its path),
// so libzs3hip.so has no link-time dependency on a particular RCCL build and single-GPU users never load it.
// Host code only (compiled --cuda-host-only).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

#include "zs3hip.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

struct Comm {
  ncclComm_t comm;
  int nranks;
};

template <typename F>
bool bind(F& fn, const char* name) {
  fn = reinterpret_cast<F>(dlsym(g_rccl.handle, name));
  return fn != nullptr;
}

int fail(ncclResult_t rc, const char* what) {
  if (rc == ncclSuccess) return 0;
  fprintf(stderr, "[zs3] %s: %s\n", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
  return 1000 + (int)rc;      // (hipError_t values stay below 1000: the two families of positive return codes do not collide)
}

bool to_type(int dtype, ncclDataType_t& t) {
  switch (dtype) {
    case 0: t = ncclFloat32; return true;
    case 1: t = ncclFloat64; return true;
    case 2: t = ncclInt32; return true;
    case 3: t = ncclInt64; return true;
  }
  return false;
}

}  // namespace

extern "C" int zs3_comm_load(const char* librccl_path) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_rccl.handle) return 0;
  void* h = dlopen(librccl_path && librccl_path[0] ? librccl_path : "librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    fprintf(stderr, "[zs3] zs3_comm_load: %s\n", dlerror());
    return -1;
  }
  g_rccl.handle = h;
  const bool ok = bind(g_rccl.GetUniqueId, "ncclGetUniqueId") && bind(g_rccl.CommInitRank, "ncclCommInitRank") &&
                  bind(g_rccl.CommDestroy, "ncclCommDestroy") && bind(g_rccl.CommAbort, "ncclCommAbort") && bind(g_rccl.AllReduce, "ncclAllReduce") &&
                  bind(g_rccl.Broadcast, "ncclBroadcast") && bind(g_rccl.GetErrorString, "ncclGetErrorString");
  if (!ok) {
    g_rccl = Rccl();
    dlclose(h);
    return -2;
  }
  return 0;
}

extern "C" int zs3_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

extern "C" int zs3_comm_unique_id(void* id_out) {
  if (!g_rccl.handle || !id_out) return -1;
  ncclUniqueId id;
  const int rc = fail(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  if (rc == 0) std::memcpy(id_out, &id, sizeof id);
  return rc;
}

extern "C" long zs3_comm_create(const void* unique_id, int nranks, int rank) {
  if (!g_rccl.handle || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return 0;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  ncclComm_t comm = nullptr;
  if (fail(g_rccl.CommInitRank(&comm, nranks, id, rank), "ncclCommInitRank") != 0) return 0;
  return reinterpret_cast<long>(new Comm{comm, nranks});
}

extern "C" int zs3_comm_destroy(long comm) {
  if (!g_rccl.handle || !comm) return -1;
  Comm* c = reinterpret_cast<Comm*>(comm);
  const int rc = fail(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
  delete c;
  return rc;
}

// ncclCommAbort: ends the communicator's kernels in flight (a collective whose peers never arrive) and frees it
extern "C" int zs3_comm_abort(long comm) {
  if (!g_rccl.handle || !comm) return -1;
  Comm* c = reinterpret_cast<Comm*>(comm);
  const int rc = fail(g_rccl.CommAbort(c->comm), "ncclCommAbort");
  delete c;
  return rc;
}

extern "C" int zs3_comm_ranks(long comm) { return comm ? reinterpret_cast<Comm*>(comm)->nranks : -1; }

// In place over ONE rank both collectives are the identity: nothing is enqueued.  (RCCL's own one-rank path enqueues two fills and
// a copy per call -- measured with rocprofv3 on the one-rank selftest, 3 blit dispatches x 215 calls per step on the compute
// stream -- which says nothing about a real exchange; the call path above this line is the one every rank count shares.)
extern "C" int zs3_allreduce(long comm, void* buf, long count, int dtype, int op, void* stream) {
  ncclDataType_t t;
  if (!g_rccl.handle || !comm || !buf || count < 0 || !to_type(dtype, t) || (op != 0 && op != 1)) return -1;
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (count == 0 || c->nranks == 1) return 0;
  return fail(g_rccl.AllReduce(buf, buf, (size_t)count, t, op == 0 ? ncclSum : ncclMax, c->comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int zs3_broadcast(long comm, void* buf, long count, int dtype, int root, void* stream) {
  ncclDataType_t t;
  if (!g_rccl.handle || !comm || !buf || count < 0 || !to_type(dtype, t)) return -1;
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (root < 0 || root >= c->nranks) return -1;
  if (count == 0 || c->nranks == 1) return 0;
  return fail(g_rccl.Broadcast(buf, buf, (size_t)count, t, root, c->comm, (hipStream_t)stream), "ncclBroadcast");
}

// SyncBN exchange in one call: this rank's fp64 totals + sample count (zs3_bn_sync_pack) and their SUM all-reduce, both on `stream`
extern "C" int zs3_bn_sync_exchange(long comm, const float* partial, int chunks, int C, double count, double* totals,
                                    void* stream) {
  const int rc = zs3_bn_sync_pack(partial, chunks, C, count, totals, stream);
  if (rc != 0) return rc;
  return zs3_allreduce(comm, totals, 2L * C + 1, 1, 0, stream);
}

// the backward exchange: zs3_bn_sync_pack_bwd (totals + this rank's dgamma / dbeta) and the SUM all-reduce of the totals
extern "C" int zs3_bn_sync_exchange_bwd(long comm, const float* partial, int chunks, int C, double count, double* totals,
                                        float* dgamma, float* dbeta, void* stream) {
  const int rc = zs3_bn_sync_pack_bwd(partial, chunks, C, count, totals, dgamma, dbeta, stream);
  if (rc != 0) return rc;
  return zs3_allreduce(comm, totals, 2L * C + 1, 1, 0, stream);
}
