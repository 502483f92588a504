// k_m2l.hip -- register-resident M2L for the kd-tree FMM (reference: fmm_c2c3_kdtree_krnl host
// branch, fmm_cart3_kdtree.cuh:613-671; operator fmm_cart_base3.cuh:1181-1208).
//
// Mapping: the directed (target, source) list is sorted by target, so a wave owns kTargets consecutive
// target nodes = one contiguous slice of the list.  Each LANE evaluates one whole interaction with the
// generated straight-line body (m2l_gen.inc: ~650 fma at p = 6, everything in VGPRs), the 64
// contributions are transposed through LDS, and lane c walks the rows in list order accumulating
// component c of the current target; a target's local expansion is stored exactly once, by one lane
// per component, in a fixed order: no atomics, bit-reproducible.
#include "nbco_internal.hpp"

namespace {

__device__ __forceinline__ float nb_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double nb_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float nb_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double nb_sqrt(double x) { return sqrt(x); }

template <int P, typename T> struct M2LBody;   // generic in the scalar type: float, or double for the fp64 far field

#include "m2l_gen.inc"

// target nodes per wave: a target has 9 list entries on average (BASELINE ball, p = 6), so 6 targets fill one 64-lane batch; with 8
// most waves ran a second, nearly empty batch (M2L 96 -> 78 us, and the near-field kernel it overlaps finishes earlier)
constexpr int kTargets = 6;

template <int P, typename T>
__global__ __launch_bounds__(64) void m2l_lane_kernel(const float4 *__restrict__ csz, const T *__restrict__ mpole,
                                                      T *__restrict__ local, const uint64_t *__restrict__ keys,
                                                      const int *__restrict__ start, int shift, int ntot, float eps2, int mstride)
{
	constexpr int offM = P * (P + 1) * (P + 2) / 6, offL = (P + 1) * (P + 1), NOUT = offL - 1;
	constexpr int CPL = (NOUT + 63) / 64;   // components per lane in the reduction walk
	__shared__ T buf[64][NOUT + 1 + (NOUT % 2)];   // odd row stride: conflict-free column writes
	__shared__ int tg[64];
	const int lane = threadIdx.x;
	const int t0 = blockIdx.x * kTargets, t1 = min(t0 + kTargets, ntot);
	const int i0 = start[t0], i1 = start[t1];
	const uint64_t mask = (1ull << shift) - 1;
	T acc[CPL];
#pragma unroll
	for (int q = 0; q < CPL; ++q) acc[q] = T(0);
	int cur = -1;
	for (int base = i0; base < i1; base += 64)
	{
		const int i = base + lane;
		int tgt = -1;
		T L[NOUT];
		if (i < i1)
		{
			const uint64_t key = keys[i];
			tgt = (int)(key >> shift);
			const int src = (int)(key & mask);
			const float4 ct = csz[tgt], cs = csz[src];
			const T dx = (T)ct.x - (T)cs.x, dy = (T)ct.y - (T)cs.y, dz = (T)ct.z - (T)cs.z;
			const T r = nb_sqrt(dx * dx + dy * dy + dz * dz + (T)eps2);
			const T rinv = T(1) / r;
			M2LBody<P, T>::run(mpole + (size_t)src * mstride, dx * rinv, dy * rinv, dz * rinv, rinv, L);
		}
		else
		{
#pragma unroll
			for (int c = 0; c < NOUT; ++c) L[c] = T(0);
		}
#pragma unroll
		for (int c = 0; c < NOUT; ++c) buf[lane][c] = L[c];
		tg[lane] = tgt;
		__syncthreads();
		const int rows = min(64, i1 - base);
		// Lane c adds component c of the rows in list order.  The rows are read eight at a time (independent LDS reads, one wait)
		// and then added one after the other: the chain is eight additions long per wait instead of one LDS round trip per row
		// (the walk used to take as long as the 650-instruction body in front of it).  Same additions in the same order.
		constexpr int RB = 8;
		for (int r0 = 0; r0 < rows; r0 += RB)
		{
			T v[RB][CPL];
			int tt[RB];
#pragma unroll
			for (int k = 0; k < RB; ++k)
			{
				const int r = min(r0 + k, 63);
				tt[k] = tg[r];
#pragma unroll
				for (int q = 0; q < CPL; ++q)
				{
					const int c = lane + 64 * q;
					v[k][q] = c < NOUT ? buf[r][c] : T(0);
				}
			}
#pragma unroll
			for (int k = 0; k < RB; ++k)
			{
				if (r0 + k >= rows) break;
				const int t = tt[k];
				if (t != cur)
				{
					if (cur >= 0)
					{
#pragma unroll
						for (int q = 0; q < CPL; ++q)
						{
							const int c = lane + 64 * q;
							if (c < NOUT) local[(size_t)cur * offL + 1 + c] = acc[q];
						}
					}
#pragma unroll
					for (int q = 0; q < CPL; ++q) acc[q] = T(0);
					cur = t;
				}
#pragma unroll
				for (int q = 0; q < CPL; ++q) acc[q] += v[k][q];
			}
		}
		__syncthreads();
	}
	if (cur >= 0)
	{
#pragma unroll
		for (int q = 0; q < CPL; ++q)
		{
			const int c = lane + 64 * q;
			if (c < NOUT) local[(size_t)cur * offL + 1 + c] = acc[q];
		}
	}
}

template <int P, typename T>
static void launch(nbco_ctx *c, const float4 *csz, const T *mpole, T *local, const uint64_t *keys, const int *start, int shift, int ntot, int mstride)
{
	const int grid = (ntot + kTargets - 1) / kTargets;
	hipLaunchKernelGGL((m2l_lane_kernel<P, T>), dim3(grid), dim3(64), 0, c->stream, csz, mpole, local, keys, start, shift, ntot, c->o.eps2,
	                   mstride > 0 ? mstride : P * (P + 1) * (P + 2) / 6);
}

template <typename T>
static int dispatch(nbco_ctx *c, int P, const float4 *csz, const T *mpole, T *local, const uint64_t *keys, const int *start, int shift, int ntot, int mstride)
{
	switch (P)
	{
	case 1: launch<1, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	case 2: launch<2, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	case 3: launch<3, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	case 4: launch<4, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	case 5: launch<5, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	case 6: launch<6, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	case 7: launch<7, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	case 8: launch<8, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	case 9: launch<9, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	case 10: launch<10, T>(c, csz, mpole, local, keys, start, shift, ntot, mstride); break;
	default: return c->fail(NBCO_ERR_UNSUPPORTED, "launch_m2l_lanes: order not generated");
	}
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

} // namespace

// targets without sources are never visited: `local` must be zero-filled by the caller.
int launch_m2l_lanes(nbco_ctx *c, int P, const float4 *csz, const float *mpole, float *local, const uint64_t *keys, const int *start,
                     int shift, int ntot, int mstride)
{
	return dispatch<float>(c, P, csz, mpole, local, keys, start, shift, ntot, mstride);
}
// fp64 far field (octree evaluator with opts.far_fp64): centres stay fp32, everything else is double
int launch_m2l_lanes_f64(nbco_ctx *c, int P, const float4 *csz, const double *mpole, double *local, const uint64_t *keys, const int *start,
                         int shift, int ntot, int mstride)
{
	return dispatch<double>(c, P, csz, mpole, local, keys, start, shift, ntot, mstride);
}
