// The builder's 64-bit key sort (pg_embedding_amd/csrc/sort_pairs.hip wraps hipCUB) for the SIMT emulator build: std::sort.
// Test infrastructure only (tests/emu/hip/hip_runtime.h).
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>

extern "C" int pgemb_sort_u64(void *tmp, size_t *tmp_bytes, const uint64_t *in, uint64_t *out, int n, void *stream)
{
	(void) stream;
	if (!tmp) { *tmp_bytes = 256; return 0; }
	if (out != in) memcpy(out, in, (size_t) n * 8);
	std::sort(out, out + n);
	return 0;
}
