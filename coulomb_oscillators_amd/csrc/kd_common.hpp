// kd_common.hpp -- device helpers shared by the kd-tree build kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace kdc {

// order-preserving 32-bit image of a float (fmm_cart3_kdtree.cuh:175-185) and its inverse
__device__ inline uint32_t ordered_bits(float f)
{
	uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float unordered_bits(uint32_t o)
{
	return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}

__device__ inline float axis_of(const float4 &p, int a) { return a == 0 ? p.x : (a == 1 ? p.y : p.z); }

// argmax of the box extents with the reference's tie rule (fmm_cart3_kdtree.cuh:92,129)
__device__ inline int longest_axis(float dx, float dy, float dz)
{
	return (dx > dy) ? ((dx > dz) ? 0 : 2) : ((dy > dz) ? 1 : 2);
}

// first particle of node j at a level with m = 2^l nodes: ceil(n j / m) (fmm_cart3_kdtree.cuh:117)
// (m is a power of two wherever this is called: a shift, not the ~100-instruction 64-bit division sequence)
__host__ __device__ inline long long range_start(long long n, long long j, long long m) { return (j == 0) ? 0 : ((n * j - 1) >> (63 - __builtin_clzll((unsigned long long)m))) + 1; }

// floor(a / b) for 0 <= a < 2^52, b > 0 through one double-precision division and a correction step
__device__ inline long long div_floor_small(long long a, long long b)
{
	const long long q = (long long)((double)a / (double)b);
	const long long r = a - q * b;
	return r < 0 ? q - 1 : (r >= b ? q + 1 : q);
}

// wave64 inclusive prefix sum / minimum across the lanes on DPP row shifts and row broadcasts (gfx9): six vector instructions
// instead of six LDS round trips (__shfl_up / __shfl_xor are ds_bpermute).  All 64 lanes must be active.
// halves: the two 32-lane halves are scanned on their own (the last step is left out)
__device__ inline uint32_t wave_scan_add(uint32_t x, bool halves = false)
{
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);   // row_shr:1
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);   // row_shr:2
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);   // row_shr:4
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);   // row_shr:8: scanned inside the rows of 16
	x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
	const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
	return halves ? x : x + y;
}
__device__ inline uint32_t wave_min_u32(uint32_t x)   // the minimum over the wave, in every lane
{
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x111, 0xF, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x112, 0xF, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x114, 0xF, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x118, 0xF, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x142, 0xA, 0xF, false));
	x = min(x, (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x143, 0xC, 0xF, false));
	return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

} // namespace kdc
