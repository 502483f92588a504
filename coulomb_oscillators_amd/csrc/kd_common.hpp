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
__host__ __device__ inline long long range_start(long long n, long long j, long long m) { return (j == 0) ? 0 : (n * j - 1) / m + 1; }

} // namespace kdc
