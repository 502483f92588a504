// fmm_ops.hpp -- declarations of the generated straight-line tensor operators (fmm_ops_gen.inc, see gen_ops.py).
// Included inside an anonymous namespace by the kernel translation units that use them.  The bodies are generic
// in the scalar type T (float, or double for the fp64 far field); T is deduced from the arguments.
#pragma once

#define NBCO_OFFM(P) ((P) * ((P) + 1) * ((P) + 2) / 6 > 0 ? (P) * ((P) + 1) * ((P) + 2) / 6 : 1)
#define NBCO_OFFL(P) (((P) + 1) * ((P) + 1))

__device__ __forceinline__ float nb_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double nb_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <int P, typename T> struct FmmOps;      // kd-tree flavour, specialised per order in the generated file
template <int P, typename T> struct FmmOctOps;   // octree flavour

#include "fmm_ops_gen.inc"

// kd-tree flavour: symmetric multipoles orders 0..P-1, traceless locals orders 1..P
template <int P, typename T> __device__ __forceinline__ void p2m_accum(T dx, T dy, T dz, T (&A)[NBCO_OFFM(P)]) { FmmOps<P, T>::p2m_accum(dx, dy, dz, A); }
template <int P, typename T> __device__ __forceinline__ void p2m_store(const T (&A)[NBCO_OFFM(P)], T *__restrict__ M) { FmmOps<P, T>::p2m_store(A, M); }
template <int P, typename T> __device__ __forceinline__ void m2m_accum(const T *__restrict__ Mc, T dx, T dy, T dz, T (&A)[NBCO_OFFM(P)])
{
	FmmOps<P, T>::m2m_accum(Mc, dx, dy, dz, A);
}
template <int P, typename T> __device__ __forceinline__ void m2m_store(const T (&A)[NBCO_OFFM(P)], T *__restrict__ M) { FmmOps<P, T>::m2m_store(A, M); }
template <int P, typename T> __device__ __forceinline__ void l2l_body(const T (&Lp)[NBCO_OFFL(P)], T dx, T dy, T dz, T (&O)[NBCO_OFFL(P)])
{
	FmmOps<P, T>::l2l_body(Lp, dx, dy, dz, O);
}
template <int P, typename T> __device__ __forceinline__ void l2p_body(const T (&Lp)[NBCO_OFFL(P)], T dx, T dy, T dz, T &fx, T &fy, T &fz)
{
	FmmOps<P, T>::l2p_body(Lp, dx, dy, dz, fx, fy, fz);
}
// octree flavour: traceless multipoles orders 0..P (tuple of (P+1)^2 entries, dipole identically 0)
template <int P, typename T> __device__ __forceinline__ void p2m_tl_accum(T dx, T dy, T dz, T (&A)[NBCO_OFFL(P)]) { FmmOctOps<P, T>::p2m_tl_accum(dx, dy, dz, A); }
template <int P, typename T> __device__ __forceinline__ void m2m_tl_accum(const T *__restrict__ Mc, T dx, T dy, T dz, T (&A)[NBCO_OFFL(P)])
{
	FmmOctOps<P, T>::m2m_tl_accum(Mc, dx, dy, dz, A);
}
