// fmm_ops.hpp -- declarations of the generated straight-line tensor operators (fmm_ops_gen.inc, see gen_ops.py).
// Included inside an anonymous namespace by the kernel translation units that use them.
#pragma once

#define NBCO_OFFM(P) ((P) * ((P) + 1) * ((P) + 2) / 6 > 0 ? (P) * ((P) + 1) * ((P) + 2) / 6 : 1)
#define NBCO_OFFL(P) (((P) + 1) * ((P) + 1))

// kd-tree flavour: symmetric multipoles orders 0..P-1, traceless locals orders 1..P
template <int P> __device__ __forceinline__ void p2m_accum(float dx, float dy, float dz, float (&A)[NBCO_OFFM(P)]);
template <int P> __device__ __forceinline__ void p2m_store(const float (&A)[NBCO_OFFM(P)], float *__restrict__ M);
template <int P> __device__ __forceinline__ void m2m_accum(const float *__restrict__ Mc, float dx, float dy, float dz, float (&A)[NBCO_OFFM(P)]);
template <int P> __device__ __forceinline__ void m2m_store(const float (&A)[NBCO_OFFM(P)], float *__restrict__ M);
template <int P> __device__ __forceinline__ void l2l_body(const float (&Lp)[NBCO_OFFL(P)], float dx, float dy, float dz, float (&O)[NBCO_OFFL(P)]);
template <int P> __device__ __forceinline__ void l2p_body(const float (&Lp)[NBCO_OFFL(P)], float dx, float dy, float dz, float &fx, float &fy, float &fz);
// octree flavour: traceless multipoles orders 0..P (tuple of (P+1)^2 floats, dipole identically 0)
template <int P> __device__ __forceinline__ void p2m_tl_accum(float dx, float dy, float dz, float (&A)[NBCO_OFFL(P)]);
template <int P> __device__ __forceinline__ void m2m_tl_accum(const float *__restrict__ Mc, float dx, float dy, float dz, float (&A)[NBCO_OFFL(P)]);

#include "fmm_ops_gen.inc"
