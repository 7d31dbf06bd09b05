// sort_pairs.hip — 64-bit key sort for the builder's (target, new) pair list.
// A plain library radix sort (hipCUB, header-only, ships with ROCm) in its own translation
// unit: it is a set-up step of the insert path, not part of the search hot path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

// tmp == nullptr: only report the temp-storage size in *tmp_bytes.
extern "C" int pgemb_sort_u64(void *tmp, size_t *tmp_bytes, const uint64_t *in, uint64_t *out, int n, void *stream)
{
	hipError_t e = hipcub::DeviceRadixSort::SortKeys(tmp, *tmp_bytes, in, out, n, 0, 64, (hipStream_t) stream);
	return e == hipSuccess ? 0 : -1;
}
