// rt_sort.hip -- device radix sort of (Morton key, ray index) pairs for the hit-point ordering of
// secondary rays.  Plain library plumbing (rocPRIM), kept in its own translation unit.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "rt_internal.h"

int rt_sort_pairs(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t n,
                  void* tmp, size_t* tmp_bytes, void* stream) {
  return (int)rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 32u,
                                        (hipStream_t)stream);
}
