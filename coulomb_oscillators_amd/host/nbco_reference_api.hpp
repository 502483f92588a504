// nbco_reference_api.hpp -- C++20 adapters with the reference's exact function-pointer conventions
// over the C ABI of include/nbco.h, so that call sites read like the reference's:
//
//     leapfrog(coulombOscillatorFMMKD3, d_buf, n, d_par, dt, step, 1);      // main3.cu:846
//
// evaluator   void (*)(VEC *p, VEC *a, int n, const SCAL *param)            direct.cuh:233, fmm_cart3_kdtree.cuh:1478
// step        void (*)(VEC *b, const VEC *a, SCAL ds, int n)                kernel.cuh:100
// integrator  void (*)(f, SCAL *buf, int n, const SCAL *param, long double dt, step_func, long double scale)   integrator.cuh:32-167
//
// All pointers are DEVICE pointers.  Errors follow gpuErrchk (kernel.cuh:52-65): message + exit.
#pragma once
#include <cstdio>
#include <cstdlib>
#include "../../include/nbco.h"

namespace nbco_ref {

using SCAL = float;
struct VEC { SCAL x, y, z; };
using evaluator_t = void (*)(VEC *, VEC *, int, const SCAL *);
using step_t = void (*)(VEC *, const VEC *, SCAL, int);

inline nbco_ctx *&ctx() { static nbco_ctx *c = nullptr; return c; }

inline void check(int rc, const char *what)
{
	if (rc != NBCO_OK)
	{
		std::fprintf(stderr, "GPUassert: %s (%s)\n", ctx() ? nbco_last_error(ctx()) : "no context", what);
		std::exit(rc);
	}
}

inline void init(const nbco_opts &o)
{
	if (ctx()) check(nbco_set_opts(ctx(), &o), "nbco_set_opts");
	else check(nbco_create(&ctx(), &o), "nbco_create");
}

// evaluators
inline void direct(VEC *p, VEC *a, int n, const SCAL *param) { check(nbco_direct(ctx(), &p->x, &a->x, n, param), "direct"); }
inline void direct3(VEC *p, VEC *a, int n, const SCAL *param) { check(nbco_direct3(ctx(), &p->x, &a->x, n, param), "direct3"); }
inline void fmm_cart3_kdtree(VEC *p, VEC *a, int n, const SCAL *param) { check(nbco_fmm_kdtree(ctx(), &p->x, &a->x, n, param), "fmm_cart3_kdtree"); }
inline void fmm_cart3_traceless(VEC *p, VEC *a, int n, const SCAL *param) { check(nbco_fmm_traceless(ctx(), &p->x, &a->x, n, param), "fmm_cart3_traceless"); }   // fmm_cart3_traceless.cuh:282
inline void fmm_cart3(VEC *p, VEC *a, int n, const SCAL *param) { check(nbco_fmm_symmetric(ctx(), &p->x, &a->x, n, param), "fmm_cart3"); }   // fmm_cart3_symmetric.cuh:413
// basic kernels
inline void step(VEC *b, const VEC *a, SCAL ds, int n) { check(nbco_step(ctx(), &b->x, &a->x, ds, n), "step"); }
inline void add_elastic(VEC *p, VEC *a, int n, const SCAL *param) { check(nbco_add_elastic(ctx(), &p->x, &a->x, n, param), "add_elastic"); }
// main3.cu:47-63
inline void coulombOscillatorDirect(VEC *p, VEC *a, int n, const SCAL *param) { direct3(p, a, n, param); add_elastic(p, a, n, param + 3); }
inline void coulombOscillatorFMMKD3(VEC *p, VEC *a, int n, const SCAL *param) { fmm_cart3_kdtree(p, a, n, param); add_elastic(p, a, n, param + 3); }

// integrator.cuh:22-28
inline void compute_force(evaluator_t f, SCAL *d_buf, int n, const SCAL *param)
{
	VEC *b = reinterpret_cast<VEC *>(d_buf);
	f(b, b + 2 * (size_t)n, n, param);
}

// integrator.cuh:32-167 -- compositions of step_func and f; coefficients in long double, narrowed at the call
inline void symplectic_euler(evaluator_t f, SCAL *d_buf, int n, const SCAL *param, long double dt, step_t step_func = step, long double scale = 1)
{
	VEC *x = reinterpret_cast<VEC *>(d_buf), *v = x + n, *a = x + 2 * (size_t)n;
	step_func(v, a, (SCAL)(dt * scale), n);
	step_func(x, v, (SCAL)dt, n);
	f(x, a, n, param);
}
inline void pre_symplectic_euler(evaluator_t f, SCAL *d_buf, int n, const SCAL *param, long double dt, step_t step_func = step, long double scale = 1)
{
	VEC *x = reinterpret_cast<VEC *>(d_buf), *v = x + n, *a = x + 2 * (size_t)n;
	f(x, a, n, param);
	step_func(v, a, (SCAL)(dt * scale), n);
	step_func(x, v, (SCAL)dt, n);
}
inline void leapfrog(evaluator_t f, SCAL *d_buf, int n, const SCAL *param, long double dt, step_t step_func = step, long double scale = 1)
{
	VEC *x = reinterpret_cast<VEC *>(d_buf), *v = x + n, *a = x + 2 * (size_t)n;
	long double ds = dt * scale * 0.5L;
	step_func(v, a, (SCAL)ds, n);
	step_func(x, v, (SCAL)dt, n);
	f(x, a, n, param);
	step_func(v, a, (SCAL)ds, n);
}
inline void forestruth(evaluator_t f, SCAL *d_buf, int n, const SCAL *param, long double dt, step_t step_func = step, long double scale = 1)
{
	VEC *x = reinterpret_cast<VEC *>(d_buf), *v = x + n, *a = x + 2 * (size_t)n;
	constexpr long double th = 1.3512071919596576340476878089715L;
	long double ds = dt * scale;
	step_func(x, v, (SCAL)(dt * th / 2), n); f(x, a, n, param);
	step_func(v, a, (SCAL)(ds * th), n); step_func(x, v, (SCAL)(dt * (1 - th) / 2), n); f(x, a, n, param);
	step_func(v, a, (SCAL)(ds * (1 - 2 * th)), n); step_func(x, v, (SCAL)(dt * (1 - th) / 2), n); f(x, a, n, param);
	step_func(v, a, (SCAL)(ds * th), n); step_func(x, v, (SCAL)(dt * th / 2), n);
}
inline void pefrl(evaluator_t f, SCAL *d_buf, int n, const SCAL *param, long double dt, step_t step_func = step, long double scale = 1)
{
	VEC *x = reinterpret_cast<VEC *>(d_buf), *v = x + n, *a = x + 2 * (size_t)n;
	constexpr long double xi = +0.1786178958448091E+00L, la = -0.2123418310626054E+00L, ch = -0.6626458266981849E-01L;
	long double ds = dt * scale;
	step_func(x, v, (SCAL)(dt * xi), n); f(x, a, n, param);
	step_func(v, a, (SCAL)(ds * (1 - 2 * la) / 2), n); step_func(x, v, (SCAL)(dt * ch), n); f(x, a, n, param);
	step_func(v, a, (SCAL)(ds * la), n); step_func(x, v, (SCAL)(dt * (1 - 2 * (ch + xi))), n); f(x, a, n, param);
	step_func(v, a, (SCAL)(ds * la), n); step_func(x, v, (SCAL)(dt * ch), n); f(x, a, n, param);
	step_func(v, a, (SCAL)(ds * (1 - 2 * la) / 2), n); step_func(x, v, (SCAL)(dt * xi), n);
}

using integrator_t = void (*)(evaluator_t, SCAL *, int, const SCAL *, long double, step_t, long double);

} // namespace nbco_ref
