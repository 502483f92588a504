// fmm_tables.cpp -- see fmm_tables.hpp.  Pure host code (no HIP), so the tables can be inspected
// and unit-tested on a machine without a GPU (nbco_debug_table in nbco_api.hip).
#include "fmm_tables.hpp"
#include <cmath>

namespace fmmtab {

namespace {

long double fact(int n) { long double f = 1; for (int i = 2; i <= n; ++i) f *= i; return f; }
long double odfact(int n) { long double f = 1; for (int i = n; i > 1; i -= 2) f *= i; return f; }   // n!! (n odd, or -1 -> 1)
long double binom(int n, int k) { return fact(n) / (fact(k) * fact(n - k)); }
long double trinom(int n, int kx, int kz) { return fact(n) / (fact(kx) * fact(kz) * fact(n - kx - kz)); }
// fmm_cart_base.cuh:33-40: a! / (2^k k! (a-2k)!)
long double coeff2(int a, int k) { return fact(a) / (std::pow(2.0L, k) * fact(k) * fact(a - 2 * k)); }

} // namespace

Tables build(int P)
{
	Tables t;
	t.P = P;
	t.offM = sym_off(P);
	t.offL = tl_off(P + 1);
	t.nfull = sym_off(P + 1);
	t.ntl = t.offL;

	// component table and monomial recurrence over the full layout (orders 0..P)
	t.sym_xyz.assign(t.nfull, 0);
	t.mono_rec.assign(t.nfull, 0);
	for (int n = 0; n <= P; ++n)
		for (int z = 0; z <= n; ++z)
			for (int x = n - z; x >= 0; --x)
			{
				int y = n - x - z, i = sym_off(n) + sym_idx(x, z, n);
				t.sym_xyz[i] = (uint32_t)n | ((uint32_t)x << 8) | ((uint32_t)y << 16) | ((uint32_t)z << 24);
				if (n > 0)
				{
					int axis = x > 0 ? 0 : (y > 0 ? 1 : 2);
					int px = x - (axis == 0), pz = z - (axis == 2);
					int parent = sym_off(n - 1) + sym_idx(px, pz, n - 1);
					t.mono_rec[i] = (uint32_t)parent | ((uint32_t)axis << 16);
				}
			}

	// P2M (fmm_cart_base3.cuh:908-918; kd driver uses orders 2..P-1, fmm_cart3_kdtree.cuh:246-247)
	t.p2m_coef.assign(t.offM > 0 ? t.offM : 1, 0.f);
	t.m_order.assign(t.offM > 0 ? t.offM : 1, 0);
	for (int q = 0; q <= P - 1; ++q)
		for (int i = 0; i < sym_elems(q); ++i)
		{
			t.m_order[sym_off(q) + i] = q;
			if (q >= 2) t.p2m_coef[sym_off(q) + i] = (float)(((q & 1) ? -1.0L : 1.0L) / fact(q));
		}

	// M2M (fmm_cart_base3.cuh:1042-1076): M'_n[x,y,z] += (1/n!) sum_m (n-m)! sum_{k1+k2+k3=m}
	// C(x,k1) C(y,k2) C(z,k3) d^(k1,k2,k3) M_{n-m}[x-k1,y-k2,z-k3]; outputs of orders 2..P-1.
	// Terms that read the (identically zero) dipole of the child are dropped.
	t.m2m_start.assign(t.offM + 1, 0);
	for (int n = 0; n <= P - 1; ++n)
		for (int z = 0; z <= n; ++z)
			for (int x = n - z; x >= 0; --x)
			{
				int y = n - x - z, o = sym_off(n) + sym_idx(x, z, n);
				t.m2m_start[o] = (int)t.m2m_idx.size();
				if (n < 2) continue;
				for (int m = 0; m <= n; ++m)
				{
					if (n - m == 1) continue;
					for (int k1 = 0; k1 <= std::min(x, m); ++k1)
						for (int k3 = std::max(0, m - k1 - y); k3 <= std::min(z, m - k1); ++k3)
						{
							int k2 = m - k1 - k3;
							long double c = fact(n - m) / fact(n) * binom(x, k1) * binom(y, k2) * binom(z, k3);
							int didx = sym_off(m) + sym_idx(k1, k3, m);
							int midx = sym_off(n - m) + sym_idx(x - k1, z - k3, n - m);
							t.m2m_idx.push_back((uint32_t)didx | ((uint32_t)midx << 16));
							t.m2m_coef.push_back((float)c);
						}
				}
			}
	t.m2m_start[t.offM] = (int)t.m2m_idx.size();

	// traceless <-> full maps
	t.tl2full.assign(t.offL, 0);
	t.tl_order.assign(t.offL, 0);
	for (int n = 0; n <= P; ++n)
		for (int i = 0; i < 2 * n + 1; ++i)
		{
			t.tl2full[tl_off(n) + i] = sym_off(n) + i;   // the z in {0,1} rows lead the symmetric layout
			t.tl_order[tl_off(n) + i] = n;
		}

	// gradient polynomial (fmm_cart_base3.cuh:698-729 without the r^(-m-1) factor):
	// G^_m[x,y,z] = (-1)^m uz^z sum_{k1<=x/2} sum_{k2<=y/2} (-1)^(k1+k2) (2m-2(k1+k2)-1)!! c2(x,k1) c2(y,k2)
	//               ux^(x-2k1) uy^(y-2k2),  z in {0,1}
	t.gp_start.assign(t.offL + 1, 0);
	for (int m = 0; m <= P; ++m)
		for (int z = 0; z <= std::min(1, m); ++z)
			for (int x = m - z; x >= 0; --x)
			{
				int y = m - x - z, e = tl_off(m) + tl_idx(x, z, m);
				t.gp_start[e] = (int)t.gp_exp.size();
				if (m == 0) continue;
				for (int k1 = 0; k1 <= x / 2; ++k1)
					for (int k2 = 0; k2 <= y / 2; ++k2)
					{
						int j = k1 + k2;
						long double c = ((m & 1) ? -1.0L : 1.0L) * ((j & 1) ? -1.0L : 1.0L) * odfact(2 * (m - j) - 1)
						                * coeff2(x, k1) * coeff2(y, k2);
						t.gp_exp.push_back((uint32_t)(x - 2 * k1) | ((uint32_t)(y - 2 * k2) << 8) | ((uint32_t)z << 16));
						t.gp_coef.push_back((float)c);
					}
			}
	t.gp_start[t.offL] = (int)t.gp_exp.size();

	// refinement A[x,y,z] = -A[x+2,y,z-2] - A[x,y+2,z-2] (fmm_cart_base3.cuh:611-623), pass per z
	t.rf_start.assign(P + 2, 0);
	for (int z = 0; z <= P; ++z)
	{
		t.rf_start[z] = (int)t.rf_dst.size();
		if (z < 2) continue;
		for (int n = z; n <= P; ++n)
			for (int x = n - z; x >= 0; --x)
			{
				t.rf_dst.push_back(sym_off(n) + sym_idx(x, z, n));
				t.rf_a.push_back(sym_off(n) + sym_idx(x + 2, z - 2, n));
				t.rf_b.push_back(sym_off(n) + sym_idx(x, z - 2, n));
			}
	}
	t.rf_start[P + 1] = (int)t.rf_dst.size();

	// M2L (fmm_cart_base3.cuh:1181-1208 with minm=1, maxm=P, no_dipole; contraction :378-426):
	// L_n[x,y,z] += (1/n!) sum_{m=n..P, k=m-n != 1} sum_{|kappa|=k} k!/(kx!ky!kz!) M_k[kappa] G_m[(x,y,z)+kappa]
	t.m2l_start.assign(t.offL + 1, 0);
	for (int n = 0; n <= P; ++n)
		for (int z = 0; z <= std::min(1, n); ++z)
			for (int x = n - z; x >= 0; --x)
			{
				int o = tl_off(n) + tl_idx(x, z, n);
				t.m2l_start[o] = (int)t.m2l_idx.size();
				if (n == 0) continue;
				for (int m = n; m <= P; ++m)
				{
					int k = m - n;
					if (k == 1 || k > P - 1) continue;
					for (int kz = 0; kz <= k; ++kz)
						for (int kx = 0; kx <= k - kz; ++kx)
						{
							int midx = sym_off(k) + sym_idx(kx, kz, k);
							int gidx = sym_off(m) + sym_idx(x + kx, z + kz, m);
							t.m2l_idx.push_back((uint32_t)midx | ((uint32_t)gidx << 16));
							t.m2l_coef.push_back((float)(trinom(k, kx, kz) / fact(n)));
						}
				}
			}
	t.m2l_start[t.offL] = (int)t.m2l_idx.size();

	// L2L (fmm_cart_base3.cuh:1348-1363): L'_n[x,y,z] += sum_{m=n..P} C(m,m-n) sum_{|kappa|=m-n}
	// (m-n)!/(kx!ky!kz!) L_m[(x,y,z)+kappa] d^kappa
	t.l2l_start.assign(t.offL + 1, 0);
	for (int n = 0; n <= P; ++n)
		for (int z = 0; z <= std::min(1, n); ++z)
			for (int x = n - z; x >= 0; --x)
			{
				int o = tl_off(n) + tl_idx(x, z, n);
				t.l2l_start[o] = (int)t.l2l_idx.size();
				if (n == 0) continue;
				for (int m = n; m <= P; ++m)
				{
					int k = m - n;
					for (int kz = 0; kz <= k; ++kz)
						for (int kx = 0; kx <= k - kz; ++kx)
						{
							int lidx = sym_off(m) + sym_idx(x + kx, z + kz, m);
							int didx = sym_off(k) + sym_idx(kx, kz, k);
							t.l2l_idx.push_back((uint32_t)lidx | ((uint32_t)didx << 16));
							t.l2l_coef.push_back((float)(binom(m, k) * trinom(k, kx, kz)));
						}
				}
			}
	t.l2l_start[t.offL] = (int)t.l2l_idx.size();

	// L2P (fmm_cart_base3.cuh:1511-1529): a_c -= sum_{n=1..P} n sum_{|kappa|=n-1} (n-1)!/(kx!ky!kz!)
	// L_n[e_c + kappa] d^kappa.  One entry per monomial kappa of order q = n-1 in 0..P-1.
	int nm = sym_off(P);
	t.l2p_coef.assign(nm > 0 ? nm : 1, 0.f);
	t.l2p_idx.assign(nm > 0 ? nm : 1, 0);
	for (int q = 0; q <= P - 1; ++q)
		for (int kz = 0; kz <= q; ++kz)
			for (int kx = q - kz; kx >= 0; --kx)
			{
				int k = sym_off(q) + sym_idx(kx, kz, q), n = q + 1;
				uint32_t lx = sym_off(n) + sym_idx(kx + 1, kz, n);
				uint32_t ly = sym_off(n) + sym_idx(kx, kz, n);
				uint32_t lz = sym_off(n) + sym_idx(kx, kz + 1, n);
				t.l2p_idx[k] = lx | (ly << 10) | (lz << 20);
				t.l2p_coef[k] = (float)(n * trinom(q, kx, kz));
			}
	return t;
}

Packed pack(const Tables &t)
{
	Packed p;
	auto addi = [&](const auto &v) { int o = (int)p.ints.size(); for (auto e : v) p.ints.push_back((int32_t)e); return o; };
	auto addf = [&](const std::vector<float> &v) { int o = (int)p.floats.size(); p.floats.insert(p.floats.end(), v.begin(), v.end()); return o; };
	p.o_sym_xyz = addi(t.sym_xyz);
	p.o_mono_rec = addi(t.mono_rec);
	p.o_m2m_start = addi(t.m2m_start);
	p.o_m2m_idx = addi(t.m2m_idx);
	p.o_tl2full = addi(t.tl2full);
	p.o_tl_order = addi(t.tl_order);
	p.o_gp_start = addi(t.gp_start);
	p.o_gp_exp = addi(t.gp_exp);
	p.o_rf_start = addi(t.rf_start);
	p.o_rf_dst = addi(t.rf_dst);
	p.o_rf_a = addi(t.rf_a);
	p.o_rf_b = addi(t.rf_b);
	p.o_m2l_start = addi(t.m2l_start);
	p.o_m2l_idx = addi(t.m2l_idx);
	p.o_m_order = addi(t.m_order);
	p.o_l2l_start = addi(t.l2l_start);
	p.o_l2l_idx = addi(t.l2l_idx);
	p.o_l2p_idx = addi(t.l2p_idx);
	p.f_p2m_coef = addf(t.p2m_coef);
	p.f_m2m_coef = addf(t.m2m_coef);
	p.f_gp_coef = addf(t.gp_coef);
	p.f_m2l_coef = addf(t.m2l_coef);
	p.f_l2l_coef = addf(t.l2l_coef);
	p.f_l2p_coef = addf(t.l2p_coef);
	return p;
}

} // namespace fmmtab
