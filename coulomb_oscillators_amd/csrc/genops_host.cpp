// genops_host.cpp -- the GENERATED far-field operator bodies (fmm_ops_gen.inc from gen_ops.py, m2l_gen.inc from gen_m2l.py:
// the very text the gfx950 kernels of k_farfield.hip, k_m2l.hip and k_fmm_oct.hip compile) built for the host, so that a
// machine without a GPU can run what the device runs and compare it with the oracle's operators (tests/test_genops_host.py).
// g++ only; the qualifiers of the generated code are defined away, nothing else differs.  Not linked into libnbco_hip.so.
#include <cmath>
#include <cstddef>
#define __device__
#define __forceinline__ inline

namespace {
#include "fmm_ops.hpp"
template <int P, typename T> struct M2LBody;
#include "m2l_gen.inc"

template <int P, typename T> int p2m(const T *pts, int npts, const T *c, T *M)
{
	constexpr int offM = P * (P + 1) * (P + 2) / 6;
	T A[NBCO_OFFM(P)] = {};
	for (int j = 0; j < npts; ++j) p2m_accum<P, T>(pts[3 * j] - c[0], pts[3 * j + 1] - c[1], pts[3 * j + 2] - c[2], A);
	if (offM > 0) M[0] = (T)npts;
	if (offM > 1) { M[1] = 0; M[2] = 0; M[3] = 0; }
	p2m_store<P, T>(A, M);
	return 0;
}
template <int P, typename T> int m2m(const T *Mc, const T *d, T *Mp)
{
	T A[NBCO_OFFM(P)] = {};
	if (P >= 3) m2m_accum<P, T>(Mc, d[0], d[1], d[2], A);
	m2m_store<P, T>(A, Mp);
	return 0;
}
template <int P, typename T> int m2l(const T *M, const T *d, T eps2, T *L)
{
	constexpr int NOUT = (P + 1) * (P + 1) - 1;
	const T r = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + eps2), rinv = T(1) / r;
	T out[NOUT];
	M2LBody<P, T>::run(M, d[0] * rinv, d[1] * rinv, d[2] * rinv, rinv, out);
	L[0] = 0;
	for (int c = 0; c < NOUT; ++c) L[1 + c] = out[c];
	return 0;
}
template <int P, typename T> int l2l(const T *Lp, const T *d, T *O)
{
	constexpr int offL = (P + 1) * (P + 1);
	T in[offL], out[offL];
	for (int q = 0; q < offL; ++q) in[q] = Lp[q];
	l2l_body<P, T>(in, d[0], d[1], d[2], out);
	for (int q = 0; q < offL; ++q) O[q] = out[q];
	return 0;
}
template <int P, typename T> int l2p(const T *Lp, const T *d, T *f)
{
	constexpr int offL = (P + 1) * (P + 1);
	T in[offL];
	for (int q = 0; q < offL; ++q) in[q] = Lp[q];
	l2p_body<P, T>(in, d[0], d[1], d[2], f[0], f[1], f[2]);
	return 0;
}
template <int P, typename T> int p2m_tl(const T *pts, int npts, const T *c, T *A)
{
	T acc[NBCO_OFFL(P)] = {};
	for (int j = 0; j < npts; ++j) p2m_tl_accum<P, T>(pts[3 * j] - c[0], pts[3 * j + 1] - c[1], pts[3 * j + 2] - c[2], acc);
	for (int q = 0; q < NBCO_OFFL(P); ++q) A[q] = acc[q];
	return 0;
}
template <int P, typename T> int m2m_tl(const T *Mc, const T *d, T *A)
{
	T acc[NBCO_OFFL(P)] = {};
	m2m_tl_accum<P, T>(Mc, d[0], d[1], d[2], acc);
	for (int q = 0; q < NBCO_OFFL(P); ++q) A[q] = acc[q];
	return 0;
}
} // namespace

#define DISPATCH(FN, ...)                                                                          \
	switch (order)                                                                                 \
	{                                                                                              \
	case 1: return FN<1, T>(__VA_ARGS__); case 2: return FN<2, T>(__VA_ARGS__); case 3: return FN<3, T>(__VA_ARGS__);   \
	case 4: return FN<4, T>(__VA_ARGS__); case 5: return FN<5, T>(__VA_ARGS__); case 6: return FN<6, T>(__VA_ARGS__);   \
	case 7: return FN<7, T>(__VA_ARGS__); case 8: return FN<8, T>(__VA_ARGS__); case 9: return FN<9, T>(__VA_ARGS__);   \
	case 10: return FN<10, T>(__VA_ARGS__);                                                                              \
	default: return -1;                                                                            \
	}

template <typename T> static int d_p2m(int order, const T *pts, int npts, const T *c, T *M) { DISPATCH(p2m, pts, npts, c, M) }
template <typename T> static int d_m2m(int order, const T *Mc, const T *d, T *Mp) { DISPATCH(m2m, Mc, d, Mp) }
template <typename T> static int d_m2l(int order, const T *M, const T *d, T eps2, T *L) { DISPATCH(m2l, M, d, eps2, L) }
template <typename T> static int d_l2l(int order, const T *Lp, const T *d, T *O) { DISPATCH(l2l, Lp, d, O) }
template <typename T> static int d_l2p(int order, const T *Lp, const T *d, T *f) { DISPATCH(l2p, Lp, d, f) }
template <typename T> static int d_p2m_tl(int order, const T *pts, int npts, const T *c, T *A) { DISPATCH(p2m_tl, pts, npts, c, A) }
template <typename T> static int d_m2m_tl(int order, const T *Mc, const T *d, T *A) { DISPATCH(m2m_tl, Mc, d, A) }

extern "C" {
// kd-tree flavour (symmetric multipoles orders 0..p-1 in the reference's normalisation, traceless locals orders 1..p).
// M of p2m: full tuple (order 0 = count, dipole 0).  m2m adds nothing: Mp receives orders 2..p-1 of the child's tuple shifted by
// d = parent centre - child centre.  m2l: L[0] = 0, L[1..] = the contribution of one source at d = target - source centre.
int nbco_genop_p2m_f32(int order, const float *pts, int npts, const float *c, float *M) { return d_p2m<float>(order, pts, npts, c, M); }
int nbco_genop_p2m_f64(int order, const double *pts, int npts, const double *c, double *M) { return d_p2m<double>(order, pts, npts, c, M); }
int nbco_genop_m2m_f32(int order, const float *Mc, const float *d, float *Mp) { return d_m2m<float>(order, Mc, d, Mp); }
int nbco_genop_m2m_f64(int order, const double *Mc, const double *d, double *Mp) { return d_m2m<double>(order, Mc, d, Mp); }
int nbco_genop_m2l_f32(int order, const float *M, const float *d, float eps2, float *L) { return d_m2l<float>(order, M, d, eps2, L); }
int nbco_genop_m2l_f64(int order, const double *M, const double *d, double eps2, double *L) { return d_m2l<double>(order, M, d, eps2, L); }
int nbco_genop_l2l_f32(int order, const float *Lp, const float *d, float *O) { return d_l2l<float>(order, Lp, d, O); }
int nbco_genop_l2l_f64(int order, const double *Lp, const double *d, double *O) { return d_l2l<double>(order, Lp, d, O); }
int nbco_genop_l2p_f32(int order, const float *Lp, const float *d, float *f) { return d_l2p<float>(order, Lp, d, f); }
int nbco_genop_l2p_f64(int order, const double *Lp, const double *d, double *f) { return d_l2p<double>(order, Lp, d, f); }
// octree flavour: traceless multipoles orders 0..p as ACCUMULATORS (the kernels of k_fmm_oct.hip scale / store them)
int nbco_genop_p2m_tl_f32(int order, const float *pts, int npts, const float *c, float *A) { return d_p2m_tl<float>(order, pts, npts, c, A); }
int nbco_genop_p2m_tl_f64(int order, const double *pts, int npts, const double *c, double *A) { return d_p2m_tl<double>(order, pts, npts, c, A); }
int nbco_genop_m2m_tl_f32(int order, const float *Mc, const float *d, float *A) { return d_m2m_tl<float>(order, Mc, d, A); }
int nbco_genop_m2m_tl_f64(int order, const double *Mc, const double *d, double *A) { return d_m2m_tl<double>(order, Mc, d, A); }
}
