// sgo_comm.h -- thin RCCL communicator wrapper (see sgo_comm.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>

namespace sgo {

bool comm_unique_id(void* out128, std::string* err);

struct Comm {
  void* handle = nullptr;
  int nranks = 1;
  int rank = 0;
  bool init(int nranks, int rank, const void* id128, std::string* err);
  void destroy();
  // in-place sum over ranks on stream s (no-op when nranks == 1)
  bool allreduce_f64(double* buf, size_t count, hipStream_t s, std::string* err);
  bool allreduce_i32(int* buf, size_t count, hipStream_t s, std::string* err);
};

}  // namespace sgo
