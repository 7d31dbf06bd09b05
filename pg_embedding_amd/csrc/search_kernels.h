// search_kernels.h — which instantiation of the search kernels a launch runs.  The kernels of one load shape are compiled in a
// translation unit of their own (csrc/search_inst.hip with -DSEARCH_INST_SHAPE=...), so that the library builds in parallel; the
// host code (hnsw_gpu.hip) only sees the pick functions declared at the end.
#pragma once
#include "device_dist.h"
#include "device_search.h"
#include "device_search_wide.h"

namespace pgemb {

typedef void (*search_kernel_t)(const SearchArgs);

// rreg: 0 = generic form, sets in LDS; 1 = generic form, sets in HBM (any ef); 2 / 4 = two-set register form for ef <= 128 / 256
//       (experiment builds only),
//       -2 / -4 / -8 / -16 = beam form (counting acceptance) with that many set registers, ef <= 64 / 128 / 256 / 512
template <typename SH, int RREG>
inline search_kernel_t pick_search_kernel_f(int func, bool team)
{
	if (RREG < 0)
	{
		constexpr int U = RREG < 0 ? -RREG : 2;
		if (team)
			switch (func)
			{
				case F_L2:     return hnsw_search_kernel_beam<F_L2, SH, U, true>;
				case F_COSINE: return hnsw_search_kernel_beam<F_COSINE, SH, U, true>;
				default:       return hnsw_search_kernel_beam<F_MANHATTAN, SH, U, true>;
			}
		switch (func)
		{
			case F_L2:     return hnsw_search_kernel_beam<F_L2, SH, U, false>;
			case F_COSINE: return hnsw_search_kernel_beam<F_COSINE, SH, U, false>;
			case F_L2_REF:        if (U == 4) return hnsw_search_kernel_beam<F_L2_REF, SH, 4, false>; return nullptr;          // (debug arithmetic:
			case F_MANHATTAN_REF: if (U == 4) return hnsw_search_kernel_beam<F_MANHATTAN_REF, SH, 4, false>; return nullptr;   //  one set size only)
			case F_COSINE_REF:    if (U == 4) return hnsw_search_kernel_beam<F_COSINE_REF, SH, 4, false>; return nullptr;
			default:       return hnsw_search_kernel_beam<F_MANHATTAN, SH, U, false>;
		}
	}
	if (RREG == 3)          // wide-beam form (any ef), device_search_wide.h
		switch (func)
		{
			case F_L2:     return hnsw_search_kernel_wide<F_L2, SH>;
			case F_COSINE: return hnsw_search_kernel_wide<F_COSINE, SH>;
			default:       return hnsw_search_kernel_wide<F_MANHATTAN, SH>;
		}
	if (RREG == 1)          // generic form, sets in HBM
		switch (func)
		{
			case F_L2:     return hnsw_search_kernel_lds<F_L2, SH, true>;
			case F_COSINE: return hnsw_search_kernel_lds<F_COSINE, SH, true>;
			default:       return hnsw_search_kernel_lds<F_MANHATTAN, SH, true>;
		}
	if (RREG == 0)
		switch (func)
		{
			case F_L2:     return hnsw_search_kernel_lds<F_L2, SH, false>;
			case F_COSINE: return hnsw_search_kernel_lds<F_COSINE, SH, false>;
			default:       return hnsw_search_kernel_lds<F_MANHATTAN, SH, false>;
		}
#ifdef HNSW_EXPERIMENT
	constexpr int R = (RREG <= 1 || RREG == 3) ? 2 : RREG;
	switch (func)
	{
		case F_L2:     return hnsw_search_kernel_reg<F_L2, SH, R>;
		case F_COSINE: return hnsw_search_kernel_reg<F_COSINE, SH, R>;
		default:       return hnsw_search_kernel_reg<F_MANHATTAN, SH, R>;
	}
#else
	return nullptr;          // (the two-set register form exists in experiment builds only; the host never asks for it otherwise)
#endif
}

template <typename SH>
inline search_kernel_t pick_search_kernel_s(int func, int rreg, bool team)
{
	switch (rreg)
	{
#ifdef HNSW_EXPERIMENT
		case 2:  return pick_search_kernel_f<SH, 2>(func, false);
		case 4:  return pick_search_kernel_f<SH, 4>(func, false);
#endif
		case -2: return pick_search_kernel_f<SH, -2>(func, team);
		case -4: return pick_search_kernel_f<SH, -4>(func, team);
		case -8: return pick_search_kernel_f<SH, -8>(func, team);
		case -16: return pick_search_kernel_f<SH, -16>(func, team);
		case 1:  return pick_search_kernel_f<SH, 1>(func, false);
		case 3:  return pick_search_kernel_f<SH, 3>(func, false);
		default: return pick_search_kernel_f<SH, 0>(func, false);
	}
}


// one function per load shape, each defined in its own translation unit (search_inst.hip)
search_kernel_t pick_kernel_shape2x4(int func, int rreg, bool team);
search_kernel_t pick_kernel_shape4x2(int func, int rreg, bool team);
search_kernel_t pick_kernel_shape8x2(int func, int rreg, bool team);
search_kernel_t pick_kernel_shape12x2(int func, int rreg, bool team);
// the hot narrow-row form (rows of <= 128 floats, beam form with 2 / 4 set registers, L2 / Manhattan, one wave per query)
search_kernel_t pick_kernel_shape2x2(int func, int rreg, bool lean);
#ifdef HNSW_EXPERIMENT
search_kernel_t pick_kernel_shape12x1(int func, int rreg, bool team);
#endif

}  // namespace pgemb
