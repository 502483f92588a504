// nbco_cpu.hpp -- host-only force / step functions for `nbco3 -cpu` (BASELINE config 1: direct O(N^2), leapfrog, C++20 threads).
//
// Plumbing, not the product: the MI355X engine has no CPU fallback and nothing here is used by libnbco_hip.so.  These are the
// reference's CPU function-pointer conventions (evaluator f(p, a, n, param), step(b, a, ds, n): direct.cuh:246-256,
// kernel.cuh:106-117, :151-173) over std::jthread: every worker owns a contiguous range of particles, as the reference's
// CPU_THREADS workers do.  The evaluator is the compensated direct sum (direct3_cpu, the project's accuracy reference); the
// reference's `-cpu` switch runs its kd-tree FMM on the host instead -- that twin is out of scope here, so `-cpu`
// is meant for small N.
#pragma once
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

namespace nbco_cpu {

struct V3 { float x, y, z; };

inline int &threads() { static int t = 8; return t; }   // constants.cuh:51 CPU_THREADS

template <class F> void for_ranges(int n, F &&body)
{
	const int workers = std::max(1, std::min(threads(), n));
	const int per = (n - 1) / workers + 1;
	std::vector<std::jthread> pool;
	pool.reserve((size_t)workers);
	for (int w = 0; w < workers; ++w)
	{
		const int lo = per * w, hi = std::min(per * (w + 1), n);
		if (lo < hi) pool.emplace_back([=, &body] { body(lo, hi); });
	}
}   // (jthreads join here)

// a_i = k sum_j d_ij / (|d_ij|^2 + eps2)^(3/2), d_ij = p_i - p_j, j = i included (it contributes exactly zero), every term
// folded into a compensated (Kahan) running sum, all in fp32 (direct.cuh:192-226)
inline void direct3(const V3 *p, V3 *a, int n, const float *param, float eps2)
{
	const float k = param ? param[0] : 1.f;
	for_ranges(n, [=](int lo, int hi) {
		for (int i = lo; i < hi; ++i)
		{
			float sx = 0.f, sy = 0.f, sz = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
			const V3 pi = p[i];
			for (int j = 0; j < n; ++j)
			{
				const float dx = pi.x - p[j].x, dy = pi.y - p[j].y, dz = pi.z - p[j].z;
				const float inv2 = 1.f / (dx * dx + dy * dy + dz * dz + eps2);
				const float w = std::sqrt(inv2);
				const float yx = dx * inv2 * w - cx, yy = dy * inv2 * w - cy, yz = dz * inv2 * w - cz;
				const float tx = sx + yx, ty = sy + yy, tz = sz + yz;
				cx = (tx - sx) - yx; cy = (ty - sy) - yy; cz = (tz - sz) - yz;
				sx = tx; sy = ty; sz = tz;
			}
			a[i] = V3{k * sx, k * sy, k * sz};
		}
	});
}

inline void step(V3 *b, const V3 *a, float ds, int n)   // b += a * ds
{
	for_ranges(n, [=](int lo, int hi) {
		for (int i = lo; i < hi; ++i) { b[i].x += a[i].x * ds; b[i].y += a[i].y * ds; b[i].z += a[i].z * ds; }
	});
}

inline void add_elastic(const V3 *p, V3 *a, int n, const float *k3)   // a -= k o p
{
	for_ranges(n, [=](int lo, int hi) {
		for (int i = lo; i < hi; ++i) { a[i].x -= p[i].x * k3[0]; a[i].y -= p[i].y * k3[1]; a[i].z -= p[i].z * k3[2]; }
	});
}

// coulombOscillatorDirect_cpu (main3.cu:53-57): direct sum + trap
inline void force(V3 *buf, int n, const float *par, float eps2)
{
	direct3(buf, buf + 2 * (size_t)n, n, par, eps2);
	add_elastic(buf, buf + 2 * (size_t)n, n, par + 3);
}

// one step of the symplectic schemes of integrator.cuh:32-167 (coefficients in long double, narrowed at the step call)
enum Scheme { Euler, Leapfrog, ForestRuth, Pefrl };
inline void integrate(Scheme s, V3 *buf, int n, const float *par, float eps2, long double dt)
{
	V3 *x = buf, *v = buf + n, *a = buf + 2 * (size_t)n;
	auto K = [&](long double c) { step(v, a, (float)c, n); };
	auto D = [&](long double c) { step(x, v, (float)c, n); };
	auto F = [&] { force(buf, n, par, eps2); };
	switch (s)
	{
	case Euler: K(dt); D(dt); F(); break;
	case Leapfrog: K(dt * 0.5L); D(dt); F(); K(dt * 0.5L); break;
	case ForestRuth:
	{
		constexpr long double th = 1.3512071919596576340476878089715L;
		D(dt * th / 2); F(); K(dt * th); D(dt * (1 - th) / 2); F(); K(dt * (1 - 2 * th)); D(dt * (1 - th) / 2); F(); K(dt * th); D(dt * th / 2);
		break;
	}
	case Pefrl:
	{
		constexpr long double xi = +0.1786178958448091E+00L, la = -0.2123418310626054E+00L, ch = -0.6626458266981849E-01L;
		D(dt * xi); F(); K(dt * (1 - 2 * la) / 2); D(dt * ch); F(); K(dt * la); D(dt * (1 - 2 * (ch + xi))); F(); K(dt * la); D(dt * ch); F();
		K(dt * (1 - 2 * la) / 2); D(dt * xi);
		break;
	}
	}
}

} // namespace nbco_cpu
