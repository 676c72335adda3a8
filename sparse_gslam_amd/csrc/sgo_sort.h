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

}  // namespace sgo
