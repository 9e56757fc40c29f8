// lh_sort.hip -- 32-bit key/value radix sort used by the index build (Morton order) and the voxel grid.
// Thin wrapper over rocPRIM's device radix sort (a plain library sort, kept in its own TU because the
// rocPRIM headers dominate compile time).
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "lh_kernels.hpp"

namespace lh {

size_t sort_temp_bytes(int n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                            (uint32_t*)nullptr, (size_t)n, 0, 32, (hipStream_t)0);
  return bytes;
}

void sort_pairs_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                    uint32_t* vals_out, int n, int end_bit, hipStream_t s) {
  (void)rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, (unsigned)end_bit, s);
}

size_t sort64_temp_bytes(int n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                  (uint32_t*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
  return bytes;
}
void sort_pairs_u64(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                    uint32_t* vals_out, int n, int end_bit, hipStream_t s) {
  (void)rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, (unsigned)end_bit, s);
}

size_t sort_keys64_temp_bytes(int n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
  return bytes;
}
void sort_keys_u64(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, int n, hipStream_t s) {
  (void)rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, (size_t)n, 0, 64, s);
}

size_t scan_temp_bytes(int n) {
  size_t bytes = 0;
  (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, rocprim::plus<uint32_t>(),
                                (hipStream_t)0);
  return bytes;
}
void inclusive_scan_u32(void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, int n, hipStream_t s) {
  (void)rocprim::inclusive_scan(temp, temp_bytes, in, out, (size_t)n, rocprim::plus<uint32_t>(), s);
}

}  // namespace lh
