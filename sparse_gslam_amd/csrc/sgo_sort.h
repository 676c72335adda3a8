// sgo_sort.h -- device radix sort of 64-bit keys (rocPRIM), in a translation unit of its own: the multigrid set-up
// transposes the pattern of the folded transfer operator (sgo_amg.hip) with it instead of a host counting sort.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace sgo {

// Bytes of temporary storage sort_u64 needs for n keys (0 on failure).
size_t sort_u64_temp_bytes(size_t n, int bits);
// out = in sorted ascending by the low `bits` bits (stable); tmp: sort_u64_temp_bytes(n, bits) bytes.  Stream-ordered,
// no synchronisation.  Returns false when rocPRIM reports an error.
bool sort_u64(void* tmp, size_t tmp_bytes, const uint64_t* in, uint64_t* out, size_t n, int bits, hipStream_t s);

// The same with a 32-bit value riding along with every key (stable: values of equal keys keep their input order).
size_t sort_u64_u32_temp_bytes(size_t n, int bits);
bool sort_u64_u32(void* tmp, size_t tmp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                  size_t n, int bits, hipStream_t s);

}  // namespace sgo
