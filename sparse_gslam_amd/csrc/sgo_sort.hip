// sgo_sort.hip -- see sgo_sort.h
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "sgo_sort.h"

namespace sgo {

size_t sort_u64_temp_bytes(size_t n, int bits) {
  size_t bytes = 0;
  const uint64_t* in = nullptr;
  uint64_t* out = nullptr;
  if (rocprim::radix_sort_keys(nullptr, bytes, in, out, n, 0, (unsigned)bits, (hipStream_t)0) != hipSuccess) return 0;
  return bytes ? bytes : 8;
}

bool sort_u64(void* tmp, size_t tmp_bytes, const uint64_t* in, uint64_t* out, size_t n, int bits, hipStream_t s) {
  return rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, n, 0, (unsigned)bits, s) == hipSuccess;
}

size_t sort_u64_u32_temp_bytes(size_t n, int bits) {
  size_t bytes = 0;
  const uint64_t* ki = nullptr;
  uint64_t* ko = nullptr;
  const uint32_t* vi = nullptr;
  uint32_t* vo = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, bytes, ki, ko, vi, vo, n, 0, (unsigned)bits, (hipStream_t)0) != hipSuccess) return 0;
  return bytes ? bytes : 8;
}

bool sort_u64_u32(void* tmp, size_t tmp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                  size_t n, int bits, hipStream_t s) {
  return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, (unsigned)bits, s) == hipSuccess;
}

}  // namespace sgo
