// fh_internal.h -- what the translation units of the host side share besides the C ABI: the thread-local error message
// behind fh_last_error, allocations that give parked handles back before they fail, and the k-mer word -> ASCII conversion.
// Defined in fh_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace fh {

int api_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
hipError_t api_dev_malloc(void **p, size_t bytes);
hipError_t api_host_malloc(void **p, size_t bytes);
// the m-form k-mer word (first base most significant; mhi: the first k - 32 bases of a k-mer longer than 32) as k ASCII bytes
void api_kmer_ascii(uint64_t m, uint64_t mhi, int k, uint8_t *out);
// fh_batch.hip: free the parked batch handles (fh_release_cached)
void batch_release_cached();

} // namespace fh
