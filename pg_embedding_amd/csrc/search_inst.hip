// search_inst.hip — the search kernels of ONE load shape (-DSEARCH_INST_SHAPE=1..6), a translation unit each: build.py compiles
// the six in parallel with hnsw_gpu.hip and links the objects into libhnsw_gpu.so.
#include <hip/hip_runtime.h>
#include "search_kernels.h"

namespace pgemb {

#if SEARCH_INST_SHAPE == 1
search_kernel_t pick_kernel_shape2x4(int func, int rreg, bool team) { return pick_search_kernel_s<Shape2x4>(func, rreg, team); }
#elif SEARCH_INST_SHAPE == 2
search_kernel_t pick_kernel_shape4x2(int func, int rreg, bool team) { return pick_search_kernel_s<Shape4x2>(func, rreg, team); }
#elif SEARCH_INST_SHAPE == 3
search_kernel_t pick_kernel_shape8x2(int func, int rreg, bool team) { return pick_search_kernel_s<Shape8x2>(func, rreg, team); }
#elif SEARCH_INST_SHAPE == 4
search_kernel_t pick_kernel_shape12x2(int func, int rreg, bool team) { return pick_search_kernel_s<Shape12x2>(func, rreg, team); }
#elif SEARCH_INST_SHAPE == 5
// LEAN = a launch that asked for no pop sequence, no evaluation trace and no clock stamps: those optional outputs (and the abort
// poll inside the walk; the one between queries stays) cost the issue-bound narrow-row kernel 30 spilled SGPRs in and around its
// hop loop (profiles/r4a_c2_regression.txt)
search_kernel_t pick_kernel_shape2x2(int func, int rreg, bool lean)
{
	if (lean)
		switch (func)
		{
			case F_L2: return rreg == -2 ? hnsw_search_kernel_beam<F_L2, Shape2x2, 2, false, true> : hnsw_search_kernel_beam<F_L2, Shape2x2, 4, false, true>;
			default:   return rreg == -2 ? hnsw_search_kernel_beam<F_MANHATTAN, Shape2x2, 2, false, true> : hnsw_search_kernel_beam<F_MANHATTAN, Shape2x2, 4, false, true>;
		}
	switch (func)
	{
		case F_L2: return rreg == -2 ? hnsw_search_kernel_beam<F_L2, Shape2x2, 2, false> : hnsw_search_kernel_beam<F_L2, Shape2x2, 4, false>;
		default:   return rreg == -2 ? hnsw_search_kernel_beam<F_MANHATTAN, Shape2x2, 2, false> : hnsw_search_kernel_beam<F_MANHATTAN, Shape2x2, 4, false>;
	}
}
#elif SEARCH_INST_SHAPE == 6 && defined(HNSW_EXPERIMENT)
search_kernel_t pick_kernel_shape12x1(int func, int rreg, bool team) { return pick_search_kernel_s<Shape12x1>(func, rreg, team); }
#endif

}  // namespace pgemb
