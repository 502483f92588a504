// fmm_tables.hpp -- host-side construction of the coefficient / index tables that drive the
// cartesian-tensor FMM operators on the GPU.  The reference recomputes trinomials and index
// arithmetic per term inside its loops (fmm_cart_base3.cuh:270-426, mymath.cuh:215-231); here every
// operator is flattened once per expansion order into "out[o] += coef * A[ia] * B[ib]" term lists.
//
// Layouts (fmm_cart_base3.cuh:170-241):
//   symmetric rank-n tensor: component (x,y,z), x+y+z=n, at sym_idx(x,z,n); tuples of orders
//   0..P concatenated at sym_off(n) ("full" layout, nfull = sym_off(P+1) entries);
//   traceless rank-n tensor: only z in {0,1} stored, at tl_idx(x,z,n), tuples at tl_off(n)=n^2.
#pragma once
#include <cstdint>
#include <vector>

namespace fmmtab {

constexpr int sym_elems(int n) { return (n + 1) * (n + 2) / 2; }
constexpr int sym_off(int p) { return p * (p + 1) * (p + 2) / 6; }
constexpr int tl_off(int p) { return p * p; }
constexpr int sym_idx(int x, int z, int n) { return (n * (n + 1) - (n - z) * (n - z + 1)) / 2 + n - x; }
constexpr int tl_idx(int x, int z, int n) { return (z + 1) * n - x; }

struct Tables
{
	int P = 0, offM = 0, offL = 0, nfull = 0, ntl = 0;
	// per full-layout component: order | x<<8 | y<<16 | z<<24
	std::vector<uint32_t> sym_xyz;
	// monomial recurrence: mono[i] = mono[parent] * d[axis]; entry = parent | axis<<16 (i >= 1)
	std::vector<uint32_t> mono_rec;
	// P2M coefficient per multipole component: (-1)^q / q!  (0 for orders 0 and 1)
	std::vector<float> p2m_coef;
	// M2M: for each output multipole component o in [0, offM): terms coef * D[didx] * Mchild[midx]
	std::vector<int> m2m_start;
	std::vector<uint32_t> m2m_idx;   // didx | midx<<16
	std::vector<float> m2m_coef;
	// traceless -> full index map for a local tuple (orders 0..P): tl2full[tl_off(n)+i] = sym_off(n)+i
	std::vector<int> tl2full;
	// order of each traceless-layout entry
	std::vector<int> tl_order;
	// gradient polynomial G^_m (dimensionless, unit vector u): for each traceless entry t (orders 1..P)
	// terms coef * ux^ex * uy^ey * uz^ez
	std::vector<int> gp_start;       // [offL + 1]
	std::vector<uint32_t> gp_exp;    // ex | ey<<8 | ez<<16
	std::vector<float> gp_coef;
	// traceless refinement schedule: pass z = 2..P, entries dst = -(a + b) in the full layout
	std::vector<int> rf_start;       // [P + 2], indexed by z
	std::vector<uint32_t> rf_dst, rf_a, rf_b;
	// M2L contraction: per traceless output entry o (orders 1..P): coef * Mscaled[midx] * G^[gidx]
	std::vector<int> m2l_start;      // [offL + 1]
	std::vector<uint32_t> m2l_idx;   // midx | gidx<<16
	std::vector<float> m2l_coef;
	// order of each multipole component (for the r^-k pre-scaling)
	std::vector<int> m_order;        // [offM]
	// L2L: per traceless output entry o: coef * Lfull[lidx] * D[didx]
	std::vector<int> l2l_start;      // [offL + 1]
	std::vector<uint32_t> l2l_idx;   // lidx | didx<<16
	std::vector<float> l2l_coef;
	// L2P: per monomial k in [0, offM): coef and the three Lfull indices (x, y, z field components)
	std::vector<float> l2p_coef;
	std::vector<uint32_t> l2p_idx;   // lx | ly<<10 | lz<<20
};

Tables build(int P);

// flat device image: ints first, then floats; offsets in elements
struct Packed
{
	std::vector<int32_t> ints;
	std::vector<float> floats;
	// offsets into ints
	int o_sym_xyz, o_mono_rec, o_m2m_start, o_m2m_idx, o_tl2full, o_tl_order, o_gp_start, o_gp_exp, o_rf_start, o_rf_dst,
	    o_rf_a, o_rf_b, o_m2l_start, o_m2l_idx, o_m_order, o_l2l_start, o_l2l_idx, o_l2p_idx;
	// offsets into floats
	int f_p2m_coef, f_m2m_coef, f_gp_coef, f_m2l_coef, f_l2l_coef, f_l2p_coef;
};
Packed pack(const Tables &t);

} // namespace fmmtab
