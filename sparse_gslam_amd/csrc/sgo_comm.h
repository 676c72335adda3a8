// sgo_comm.h -- thin RCCL communicator wrapper (see sgo_comm.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace sgo {

bool comm_unique_id(void* out128, std::string* err);

// in-place sum over ranks of `count` doubles in HOST memory; returns 0 on success (sgo_host_allreduce_fn, sgo.h)
using HostAllreduce = int (*)(double* buf, size_t count, void* user);
// recv[r * count .. (r + 1) * count) = rank r's send[0 .. count), HOST memory; returns 0 on success (sgo_host_allgather_fn)
using HostAllgather = int (*)(const double* send, size_t count, double* recv, void* user);

struct Comm {
  void* handle = nullptr;          // RCCL communicator
  HostAllreduce host_fn = nullptr; // caller-supplied transport (sgo_comm_init_host): collectives staged through pinned memory
  HostAllgather host_gather_fn = nullptr;   // optional: without it an all-gather is an all-reduce of zero-padded slots (exact)
  void* host_user = nullptr;
  double* stage = nullptr;
  size_t stage_cap = 0;
  int nranks = 1;
  int rank = 0;
  bool active() const { return handle != nullptr || host_fn != nullptr; }
  bool init(int nranks, int rank, const void* id128, std::string* err);
  bool init_host(int nranks, int rank, HostAllreduce fn, void* user, HostAllgather gather = nullptr);
  void destroy();
  // in-place sum over ranks on stream s (no-op when nranks == 1)
  bool allreduce_f64(double* buf, size_t count, hipStream_t s, std::string* err);
  bool allreduce_i32(int* buf, size_t count, hipStream_t s, std::string* err);
  // recv[r * count ..) = rank r's send[0 .. count) on stream s (device memory; send and recv must not overlap);
  // a plain device copy when nranks == 1 without a communicator
  bool allgather_f64(const double* send, double* recv, size_t count, hipStream_t s, std::string* err);
  bool stage_reserve(size_t count, std::string* err);
};

}  // namespace sgo
