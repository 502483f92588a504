// k_axpy.hip -- HBM-bound particle kernels: step (axpy), elastic terms, rescale, permutations,
// float3 -> float4 packing, and the fused kick/drift pieces the integrators use.
// Reference behaviour: kernel.cuh:85-311, appel.cuh:506-527.  The reference launches these with at
// most 10 blocks x 128 threads (constants.cuh:36-37); here every kernel is a grid-stride loop over
// 16-byte vectors sized to fill 256 CUs.
#include "nbco_internal.hpp"

namespace {

constexpr int kBlock = 256;

static inline int stream_grid(long long items, int per_thread = 1)
{
	long long blocks = (items + (long long)kBlock * per_thread - 1) / ((long long)kBlock * per_thread);
	if (blocks < 1) blocks = 1;
	if (blocks > 2048) blocks = 2048;   // 256 CUs x 8 blocks, grid-stride beyond
	return (int)blocks;
}

static inline bool aligned16(const void *p) { return (((uintptr_t)p) & 15u) == 0; }

// b[i] = fma(ds, a[i], b[i]) over the flat float view of the xyz triplets (kernel.cuh:85-98)
__global__ __launch_bounds__(kBlock) void step_vec4(float4 *__restrict__ b, const float4 *__restrict__ a, float ds, long long n4)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long long)gridDim.x * kBlock)
	{
		float4 t = a[i], s = b[i];
		s.x = fmaf(ds, t.x, s.x); s.y = fmaf(ds, t.y, s.y); s.z = fmaf(ds, t.z, s.z); s.w = fmaf(ds, t.w, s.w);
		b[i] = s;
	}
}
__global__ __launch_bounds__(kBlock) void step_scalar(float *__restrict__ b, const float *__restrict__ a, float ds, long long beg, long long n)
{
	for (long long i = beg + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
		b[i] = fmaf(ds, a[i], b[i]);
}

// a[i] = fma(-k, p[i], a[i]) (kernel.cuh:119-143) or a[i] = -k*p[i] (kernel.cuh:175-196);
// k is a device pointer to 3 floats or null (k = 1)
template <bool ASSIGN>
__global__ __launch_bounds__(kBlock) void elastic_kernel(const float *__restrict__ p, float *__restrict__ a, long long n3,
                                                         const float *__restrict__ k)
{
	float k3[3] = {1.f, 1.f, 1.f};
	if (k) { k3[0] = k[0]; k3[1] = k[1]; k3[2] = k[2]; }
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n3; i += (long long)gridDim.x * kBlock)
	{
		int c = (int)(i % 3);
		float kc = c == 0 ? k3[0] : (c == 1 ? k3[1] : k3[2]);
		if (ASSIGN) a[i] = -kc * p[i];
		else a[i] = fmaf(-kc, p[i], a[i]);
	}
}
// 12 floats (4 particles) per thread, three 16-byte accesses per array
template <bool ASSIGN>
__global__ __launch_bounds__(kBlock) void elastic_vec(const float4 *__restrict__ p, float4 *__restrict__ a, long long ngroups,
                                                      const float *__restrict__ k)
{
	float kx = 1.f, ky = 1.f, kz = 1.f;
	if (k) { kx = k[0]; ky = k[1]; kz = k[2]; }
	for (long long g = (long long)blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += (long long)gridDim.x * kBlock)
	{
		float4 p0 = p[3 * g], p1 = p[3 * g + 1], p2 = p[3 * g + 2];
		float4 a0, a1, a2;
		if (ASSIGN)
		{
			a0 = make_float4(-kx * p0.x, -ky * p0.y, -kz * p0.z, -kx * p0.w);
			a1 = make_float4(-ky * p1.x, -kz * p1.y, -kx * p1.z, -ky * p1.w);
			a2 = make_float4(-kz * p2.x, -kx * p2.y, -ky * p2.z, -kz * p2.w);
		}
		else
		{
			a0 = a[3 * g]; a1 = a[3 * g + 1]; a2 = a[3 * g + 2];
			a0 = make_float4(fmaf(-kx, p0.x, a0.x), fmaf(-ky, p0.y, a0.y), fmaf(-kz, p0.z, a0.z), fmaf(-kx, p0.w, a0.w));
			a1 = make_float4(fmaf(-ky, p1.x, a1.x), fmaf(-kz, p1.y, a1.y), fmaf(-kx, p1.z, a1.z), fmaf(-ky, p1.w, a1.w));
			a2 = make_float4(fmaf(-kz, p2.x, a2.x), fmaf(-kx, p2.y, a2.y), fmaf(-ky, p2.z, a2.z), fmaf(-kz, p2.w, a2.w));
		}
		a[3 * g] = a0; a[3 * g + 1] = a1; a[3 * g + 2] = a2;
	}
}

// a *= param[0] (appel.cuh:506-518)
__global__ __launch_bounds__(kBlock) void rescale_kernel(float *__restrict__ a, long long n3, const float *__restrict__ param)
{
	float s = param[0];
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n3; i += (long long)gridDim.x * kBlock)
		a[i] *= s;
}

// dst[i] = src[map[i]] / dst[map[i]] = src[i] on xyz triplets (kernel.cuh:228-278)
template <bool INVERSE>
__global__ __launch_bounds__(kBlock) void gather3_kernel(float *__restrict__ dst, const float *__restrict__ src,
                                                         const int *__restrict__ map, long long n)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		long long s = INVERSE ? i : (long long)map[i], d = INVERSE ? (long long)map[i] : i;
		float x = src[3 * s], y = src[3 * s + 1], z = src[3 * s + 2];
		dst[3 * d] = x; dst[3 * d + 1] = y; dst[3 * d + 2] = z;
	}
}

__global__ __launch_bounds__(kBlock) void pack4_kernel(float4 *__restrict__ dst, const float *__restrict__ src, long long n)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
		dst[i] = make_float4(src[3 * i], src[3 * i + 1], src[3 * i + 2], 0.f);
}

// fused K(ks) D(ds): v += a*ks; x += v*ds  (the first two step() calls of leapfrog, integrator.cuh:85-89)
__global__ __launch_bounds__(kBlock) void kick_drift_kernel(float *__restrict__ x, float *__restrict__ v, const float *__restrict__ a,
                                                            float ks, float ds, long long n3)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n3; i += (long long)gridDim.x * kBlock)
	{
		float vi = fmaf(ks, a[i], v[i]);
		v[i] = vi;
		x[i] = fmaf(ds, vi, x[i]);
	}
}

// fused tail of a force evaluation + kick: a = a*param[0] - k o x (rescale + add_elastic), v += a*ks
// (v_in != v: the evaluator left the tree-ordered velocities in a scratch buffer and this kick brings them home)
__global__ __launch_bounds__(kBlock) void finish_kick_kernel(const float *__restrict__ x, const float *v_in, float *v, float *__restrict__ a,
                                                             const float *__restrict__ param, float ks, long long n3, int elastic,
                                                             int rescale)
{
	float s = rescale ? param[0] : 1.f;
	float k3[3] = {param[3], param[4], param[5]};
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n3; i += (long long)gridDim.x * kBlock)
	{
		int c = (int)(i % 3);
		float kc = c == 0 ? k3[0] : (c == 1 ? k3[1] : k3[2]);
		float ai = a[i] * s;
		if (elastic) ai = fmaf(-kc, x[i], ai);
		a[i] = ai;
		v[i] = fmaf(ks, ai, v_in[i]);
	}
}

} // namespace

int launch_step(nbco_ctx *c, float *b, const float *a, float ds, long long n3)
{
	if (n3 <= 0) return NBCO_OK;
	if (aligned16(b) && aligned16(a))
	{
		long long n4 = n3 / 4;
		if (n4 > 0)
			hipLaunchKernelGGL(step_vec4, dim3(stream_grid(n4)), dim3(kBlock), 0, c->stream, (float4 *)b, (const float4 *)a, ds, n4);
		if (n4 * 4 < n3)
			hipLaunchKernelGGL(step_scalar, dim3(1), dim3(kBlock), 0, c->stream, b, a, ds, n4 * 4, n3);
	}
	else
		hipLaunchKernelGGL(step_scalar, dim3(stream_grid(n3)), dim3(kBlock), 0, c->stream, b, a, ds, 0LL, n3);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

int launch_add_elastic(nbco_ctx *c, const float *p, float *a, long long n, const float *k, bool assign)
{
	if (n <= 0) return NBCO_OK;
	long long groups = (aligned16(p) && aligned16(a)) ? n / 4 : 0;
	if (groups > 0)
	{
		if (assign) hipLaunchKernelGGL(elastic_vec<true>, dim3(stream_grid(groups)), dim3(kBlock), 0, c->stream, (const float4 *)p, (float4 *)a, groups, k);
		else hipLaunchKernelGGL(elastic_vec<false>, dim3(stream_grid(groups)), dim3(kBlock), 0, c->stream, (const float4 *)p, (float4 *)a, groups, k);
	}
	long long done = groups * 12, n3 = 3 * n;
	if (done < n3)
	{
		// offset `done` is a multiple of 3, so the component phase is preserved
		if (assign) hipLaunchKernelGGL(elastic_kernel<true>, dim3(stream_grid(n3 - done)), dim3(kBlock), 0, c->stream, p + done, a + done, n3 - done, k);
		else hipLaunchKernelGGL(elastic_kernel<false>, dim3(stream_grid(n3 - done)), dim3(kBlock), 0, c->stream, p + done, a + done, n3 - done, k);
	}
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

int launch_rescale(nbco_ctx *c, float *a, long long n3, const float *param)
{
	if (n3 <= 0) return NBCO_OK;
	hipLaunchKernelGGL(rescale_kernel, dim3(stream_grid(n3)), dim3(kBlock), 0, c->stream, a, n3, param);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

int launch_gather3(nbco_ctx *c, float *dst, const float *src, const int *map, long long n, bool inverse)
{
	if (n <= 0) return NBCO_OK;
	if (inverse) hipLaunchKernelGGL(gather3_kernel<true>, dim3(stream_grid(n)), dim3(kBlock), 0, c->stream, dst, src, map, n);
	else hipLaunchKernelGGL(gather3_kernel<false>, dim3(stream_grid(n)), dim3(kBlock), 0, c->stream, dst, src, map, n);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

int launch_copy(nbco_ctx *c, float *dst, const float *src, long long n3)
{
	if (n3 <= 0) return NBCO_OK;
	NBCO_HIP(hipMemcpyAsync(dst, src, (size_t)n3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
	return NBCO_OK;
}

int launch_pack4(nbco_ctx *c, float4 *dst, const float *src3, long long n)
{
	if (n <= 0) return NBCO_OK;
	hipLaunchKernelGGL(pack4_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, c->stream, dst, src3, n);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

int launch_kick_drift(nbco_ctx *c, float *x, float *v, const float *a, float ks, float ds, long long n3)
{
	if (n3 <= 0) return NBCO_OK;
	hipLaunchKernelGGL(kick_drift_kernel, dim3(stream_grid(n3)), dim3(kBlock), 0, c->stream, x, v, a, ks, ds, n3);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

int launch_finish_kick(nbco_ctx *c, const float *x, const float *v_in, float *v, float *a, const float *param, float ks, long long n, bool elastic)
{
	if (n <= 0) return NBCO_OK;
	hipLaunchKernelGGL(finish_kick_kernel, dim3(stream_grid(3 * n)), dim3(kBlock), 0, c->stream, x, v_in, v, a, param, ks, 3 * n, elastic ? 1 : 0, 0);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}
