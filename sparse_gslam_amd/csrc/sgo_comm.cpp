// sgo_comm.cpp -- RCCL binding of the multi-GPU modes (one process per GPU).
//
// librccl is dlopen'ed on first use so that the single-GPU product path carries no RCCL dependency.  Two collectives:
//   ncclAllGather  row-owner mode's exchanges (one fixed-size packet per rank: partial dot products + boundary rows of a
//                  vector, or 72-byte records of the smoothed prolongator) and the ranks' owned slices of a vector
//                  (the Gauss-Newton step; the all-reduce mode's product vectors: one contributor per row, nothing to sum);
//   ncclAllReduce  (sum) the coarse right-hand side of every multigrid cycle, the level-1 Galerkin blocks once per GN
//                  iteration, the two chi2 sums, and -- rank-emulation hook only -- zero-filled product vectors.
// A communicator of ONE rank still calls both (the single-GPU test of this path).  A caller may bring its own transport
// instead (sgo_comm_init_host + optional sgo_comm_host_allgather: MPI, gloo, ...): the same collectives, staged through
// pinned host memory.
#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "sgo_comm.h"

namespace sgo {
namespace {

struct NcclUniqueId { char internal[128]; };
using ncclComm_t = void*;
constexpr int kNcclSum = 0;
constexpr int kNcclInt32 = 2;
constexpr int kNcclFloat64 = 8;

struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};

Api& api() {
  static Api a;
  return a;
}

bool load(std::string* err) {
  static std::mutex mu;   // contexts on different threads may initialise their communicators at once
  std::lock_guard<std::mutex> lock(mu);
  Api& a = api();
  if (a.handle) return true;
  const char* env = std::getenv("SGO_RCCL_LIB");
  const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (a.handle) break;
  }
  if (!a.handle) {
    *err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?");
    return false;
  }
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
  a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
  a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather) {
    *err = "librccl lacks a required symbol";
    dlclose(a.handle);
    a.handle = nullptr;
    return false;
  }
  return true;
}

std::string nccl_err(int rc) {
  Api& a = api();
  return a.GetErrorString ? a.GetErrorString(rc) : ("nccl error " + std::to_string(rc));
}

}  // namespace

bool comm_unique_id(void* out128, std::string* err) {
  if (!load(err)) return false;
  NcclUniqueId id;
  int rc = api().GetUniqueId(&id);
  if (rc != 0) {
    *err = "ncclGetUniqueId: " + nccl_err(rc);
    return false;
  }
  std::memcpy(out128, id.internal, 128);
  return true;
}

bool Comm::init(int nranks_, int rank_, const void* id128, std::string* err) {
  if (!load(err)) return false;
  destroy();
  NcclUniqueId id;
  std::memcpy(id.internal, id128, 128);
  ncclComm_t c = nullptr;
  int rc = api().CommInitRank(&c, nranks_, id, rank_);
  if (rc != 0) {
    *err = "ncclCommInitRank: " + nccl_err(rc);
    return false;
  }
  handle = c;
  nranks = nranks_;
  rank = rank_;
  return true;
}

bool Comm::init_host(int nranks_, int rank_, HostAllreduce fn, void* user, HostAllgather gather) {
  destroy();
  host_fn = fn;
  host_gather_fn = gather;
  host_user = user;
  nranks = nranks_;
  rank = rank_;
  return true;
}

void Comm::destroy() {
  if (handle) api().CommDestroy((ncclComm_t)handle);
  handle = nullptr;
  if (stage) hipHostFree(stage);
  stage = nullptr;
  stage_cap = 0;
  host_fn = nullptr;
  host_gather_fn = nullptr;
  host_user = nullptr;
  nranks = 1;
  rank = 0;
}

bool Comm::stage_reserve(size_t count, std::string* err) {
  if (count <= stage_cap) return true;
  if (stage) hipHostFree(stage);
  stage = nullptr;
  stage_cap = 0;
  const size_t cap = count + count / 2 + 64;
  if (hipHostMalloc((void**)&stage, cap * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    *err = "host transport: cannot allocate the pinned staging buffer";
    return false;
  }
  stage_cap = cap;
  return true;
}

bool Comm::allgather_f64(const double* send, double* recv, size_t count, hipStream_t s, std::string* err) {
  if (count == 0) return true;
  if (host_fn) {   // caller's transport, staged through pinned host memory
    const size_t total = count * (size_t)nranks;
    if (!stage_reserve(total + count, err)) return false;
    double* own = stage + total;   // this rank's contribution (behind the gathered image)
    if (hipMemcpyAsync(own, send, count * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      *err = "host transport: device -> host copy failed";
      return false;
    }
    int rc;
    if (host_gather_fn) {
      rc = host_gather_fn(own, count, stage, host_user);
    } else {
      // an all-reduce of zero-padded slots IS an all-gather: every element has one non-zero contributor, the sum is exact
      std::memset(stage, 0, total * sizeof(double));
      std::memcpy(stage + (size_t)rank * count, own, count * sizeof(double));
      rc = host_fn(stage, total, host_user);
    }
    if (rc != 0) {
      *err = "host transport: the all-gather callback returned " + std::to_string(rc);
      return false;
    }
    if (hipMemcpyAsync(recv, stage, total * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) {
      *err = "host transport: host -> device copy failed";
      return false;
    }
    return true;
  }
  if (!handle) {   // no communicator: the rank-emulation test hook, or one rank
    if (nranks == 1 && hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) {
      *err = "all-gather: device copy failed";
      return false;
    }
    return true;
  }
  int rc = api().AllGather(send, recv, count, kNcclFloat64, (ncclComm_t)handle, s);
  if (rc != 0) {
    *err = "ncclAllGather(f64): " + nccl_err(rc);
    return false;
  }
  return true;
}

bool Comm::allreduce_f64(double* buf, size_t count, hipStream_t s, std::string* err) {
  if (host_fn) {   // caller's transport: device -> pinned host -> callback (sum over ranks) -> device
    if (!stage_reserve(count, err)) return false;
    if (hipMemcpyAsync(stage, buf, count * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
      *err = "host transport: device -> host copy failed";
      return false;
    }
    const int rc = host_fn(stage, count, host_user);
    if (rc != 0) {
      *err = "host transport: the all-reduce callback returned " + std::to_string(rc);
      return false;
    }
    // the next collective's device -> host copy is ordered behind this one on the stream: the buffer is free by then
    if (hipMemcpyAsync(buf, stage, count * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) {
      *err = "host transport: host -> device copy failed";
      return false;
    }
    return true;
  }
  if (!handle) return true;   // no communicator (single GPU, or the rank-emulation test hook)
  int rc = api().AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, (ncclComm_t)handle, s);
  if (rc != 0) {
    *err = "ncclAllReduce(f64): " + nccl_err(rc);
    return false;
  }
  return true;
}

bool Comm::allreduce_i32(int* buf, size_t count, hipStream_t s, std::string* err) {
  if (host_fn) {
    *err = "host transport: integer all-reduce is not provided";
    return false;
  }
  if (!handle) return true;
  int rc = api().AllReduce(buf, buf, count, kNcclInt32, kNcclSum, (ncclComm_t)handle, s);
  if (rc != 0) {
    *err = "ncclAllReduce(i32): " + nccl_err(rc);
    return false;
  }
  return true;
}

}  // namespace sgo
