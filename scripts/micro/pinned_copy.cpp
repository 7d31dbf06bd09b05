// How fast does the HOST write / read pinned memory of the two kinds (hipHostMallocDefault vs hipHostMallocCoherent)?  The stream ring is
// written by reader threads and read by answer threads; if "coherent" host memory is mapped uncached on the CPU side, a 3 KB query copy
// costs microseconds instead of a fraction of one.   hipcc -O2 scripts/micro/pinned_copy.cpp -o /tmp/pinned_copy && /tmp/pinned_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	const size_t slots = 4096, bytes = 3072;
	char *src = (char *) malloc(slots * bytes);
	memset(src, 1, slots * bytes);
	struct { const char *name; unsigned flags; } kinds[] = { { "hipHostMallocDefault", hipHostMallocDefault }, { "hipHostMallocCoherent", hipHostMallocCoherent },
		{ "hipHostMallocNonCoherent", hipHostMallocNonCoherent } };
	for (auto &k : kinds)
	{
		char *p = nullptr;
		if (hipHostMalloc((void **) &p, slots * bytes, k.flags) != hipSuccess) { printf("%s: allocation failed\n", k.name); continue; }
		memset(p, 0, slots * bytes);
		double best_w = 1e9, best_r = 1e9;
		volatile unsigned long sink = 0;
		for (int rep = 0; rep < 5; rep++)
		{
			double t0 = now();
			for (size_t i = 0; i < slots; i++) memcpy(p + i * bytes, src + i * bytes, bytes);
			double t1 = now();
			unsigned long acc = 0;
			for (size_t i = 0; i < slots; i++) { char tmp[3072]; memcpy(tmp, p + i * bytes, bytes); acc += (unsigned char) tmp[i & 1023]; }
			double t2 = now();
			sink += acc;
			if (t1 - t0 < best_w) best_w = t1 - t0;
			if (t2 - t1 < best_r) best_r = t2 - t1;
		}
		printf("%-26s host write %.3f us per 3 KB, host read %.3f us per 3 KB\n", k.name, best_w / slots * 1e6, best_r / slots * 1e6);
		(void) hipHostFree(p);
	}
	return 0;
}
