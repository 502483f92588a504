// farfield_wide.hpp -- part of k_farfield.hip (included there, inside its anonymous namespace; no include guard on purpose)
// P2M / M2M / L2L for the HIGH orders: one WORKGROUP per leaf / node, one LANE per tensor component.
//
// The generated bodies (fmm_ops_gen.inc) keep a node's whole tuple in the registers of one thread.  At order 10 that is 220
// multipole + 220 monomial values per thread -- 880 VGPRs in double -- and the compiler spills kilobytes per lane: the upward
// shift of a 1M-particle tree took 3.8 ms in fp64 (`profiles/r03f_far_fp64_p10_before.txt`) for 164 M multiply-adds, and the
// one-workgroup top levels ran one node per THREAD.  Here the operators are what gen_ops.py's header says they are in the
// normalised forms D~[K] = d^K / K!, M~[X] = M[X] |X|! / X!, L~[X] = L[X] |X|! -- coefficient-free correlations --
//     M2M   M~'[X]   = sum_{K <= X} D~[K] M~[X - K]
//     L2L   L~'_n[X] = sum_{m >= n} sum_{|K| = m - n} D~[K] L~_m[X + K]
//     P2M   M~_q[X]  = (-1)^q sum_particles D~[X]
// with the source tuple in LDS as a dense 3-D array indexed [x][y][z] -- the partner of monomial K for output X is then at
// at(X) -/+ at(K), one subtraction instead of an index table -- and every lane owning ONE output component.  Same
// arithmetic per node whatever the launch shape, so the sharded evaluation stays bit-identical to the single GPU.  The sums
// run in a different order than the generated chains: results agree to rounding (tests: the oracle parity tests of orders 9,
// 10 and of the fp64 far field, which run through these kernels).
//
// Reference operators: fmm_cart_base3.cuh P2M :908-918, M2M :1042-1076, L2L :1348-1363.

constexpr int kWide = 256;

// which (order, scalar) pairs take this path: where the generated bodies spill (orders 9, 10; order 8 in double)
template <int P, typename T>
constexpr bool use_wide() { return P >= 9 || (sizeof(T) == 8 && P >= 8); }

__device__ inline double wide_fact(int n)
{
	double r = 1.0;
	for (int j = 2; j <= n; ++j) r *= (double)j;
	return r;
}
// d^k / k! the way the generated monomials form it: a chain of products with d / j
template <typename T>
__device__ inline T wide_mono(T d, int k)
{
	T r = T(1);
	for (int j = 1; j <= k; ++j) r *= d * (T)(1.0 / (double)j);
	return r;
}
// component i of the full symmetric layout (orders 0, 1, 2, .. in storage order: z ascending, x descending) -> order and exponents
__device__ inline void wide_decode_full(int i, int &n, int &x, int &y, int &z)
{
	n = 0;
	while ((n + 1) * (n + 2) * (n + 3) / 6 <= i) ++n;
	int r = i - n * (n + 1) * (n + 2) / 6;
	z = 0;
	while (r >= n - z + 1) { r -= n - z + 1; ++z; }
	x = n - z - r;
	y = n - x - z;
}
// component q >= 1 of the traceless layout (order n at n^2, 2n + 1 components with z in {0, 1})
__device__ inline void wide_decode_tl(int q, int &n, int &x, int &y, int &z)
{
	n = 1;
	while ((n + 1) * (n + 1) <= q) ++n;
	const int r = q - n * n;
	if (r <= n) { z = 0; x = n - r; }
	else { z = 1; x = 2 * n - r; }
	y = n - x - z;
}

// ---- P2M: one workgroup per leaf, particles in batches of 64 ---------------------------------------------------------------
template <int P, typename T>
__global__ __launch_bounds__(kWide) void p2m_wide_kernel(const float4 *__restrict__ pos, const float *__restrict__ center, const int *__restrict__ mult,
                                                         const int *__restrict__ index, T *__restrict__ mpole, int beg)
{
	constexpr int offM = P * (P + 1) * (P + 2) / 6, JB = 64, JP = JB + 1;   // (rows one element apart in the banks)
	__shared__ T pw[3][P][JP];   // pw[a][k][j] = d_a^k / k! of particle j of the batch
	const int t = threadIdx.x, leaf = beg + blockIdx.x, mlt = mult[leaf], ind = index[leaf];
	int n = 0, x = 0, y = 0, z = 0;
	const bool comp = t < offM;
	if (comp) wide_decode_full(t, n, x, y, z);
	T acc = T(0);
	for (int j0 = 0; j0 < mlt; j0 += JB)
	{
		const int nb = min(JB, mlt - j0);
		if (t < 3 * JB)
		{
			const int a = t / JB, j = t - a * JB;
			if (j < nb)
			{
				const float4 p = pos[ind + j0 + j];
				const T d = (T)(a == 0 ? p.x : (a == 1 ? p.y : p.z)) - (T)center[3 * leaf + a];
				T r = T(1);
				pw[a][0][j] = r;
				for (int k = 1; k < P; ++k) { r *= d * (T)(1.0 / (double)k); pw[a][k][j] = r; }
			}
		}
		__syncthreads();
		if (comp && n >= 2)
		{
			const T *px = pw[0][x], *py = pw[1][y], *pz = pw[2][z];
			for (int j = 0; j < nb; ++j) acc += px[j] * py[j] * pz[j];
		}
		__syncthreads();
	}
	if (!comp) return;
	T *M = mpole + (size_t)leaf * offM;
	if (n == 0) M[0] = (T)mlt;
	else if (n == 1) M[t] = T(0);
	else M[t] = acc * (T)(((n & 1) ? -1.0 : 1.0) * wide_fact(x) * wide_fact(y) * wide_fact(z) / wide_fact(n));
}

// ---- M2M: one workgroup per parent of level l ---------------------------------------------------------------------------------
// Lane X runs ONE flat loop over the monomials K in storage order and keeps the terms with K <= X (component-wise: a borrow
// test on the packed exponents); in the dense [x][y][z] array the partner M~[X - K] sits at at(X) - at(K).  A wave executes
// the union of its lanes' terms whichever way the loops are written -- up to all 220 monomials for the lanes of order 9 -- so
// the flat form costs no extra iterations over a nest with per-lane bounds, and unlike the nest it unrolls: the LDS reads of
// several terms are in flight together (30 -> 8 us for the one workgroup of a top level).
template <typename T>
__device__ inline int wide_wave_max(int v)
{
	for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
	return v;
}
// FLAT = false: the nest over k1 <= x, k2 <= y, k3 <= z with per-lane bounds -- fewer instructions per term, and with thousands of
// workgroups in flight nobody waits for a single one's LDS latency: the levels with >= 1024 nodes (104 + 60 us against 152 + 86
// for the two lowest levels of a 16 384-leaf tree).  Same terms, another order: a level always runs one form.
template <int P, typename T, bool FLAT>
__global__ __launch_bounds__(kWide) void m2m_wide_kernel(float *center, T *mpole, int *mult, int l, int write_geom)
{
	constexpr int offM = P * (P + 1) * (P + 2) / 6, S = P;   // exponents 0 .. P - 1
	__shared__ T Dl[offM], M3[S * S * S], D3[FLAT ? 1 : S * S * S];
	__shared__ int2 KT[offM];   // {at(K), packed exponents of K}
	__shared__ float cs[3];
	__shared__ int ms;
	__builtin_amdgcn_s_setprio(3);   // (a handful of workgroups beside a chip full of near-field waves: do not queue behind them)
	const int t = threadIdx.x, k = (1 << l) - 1 + (int)blockIdx.x;
	int n = 0, x = 0, y = 0, z = 0;
	const bool comp = t < offM;
	if (comp) wide_decode_full(t, n, x, y, z);
	const int at = (x * S + y) * S + z, xp = (x | (y << 8) | (z << 16)) | 0x808080;
	if (comp) KT[t] = make_int2(at, x | (y << 8) | (z << 16));
	if (t == 0)
	{
		float c[3];
		int mlt;
		parent_centre<false>(center, mult, k, c, mlt);
		cs[0] = c[0]; cs[1] = c[1]; cs[2] = c[2];
		ms = mlt;
	}
	__syncthreads();
	// K <= X needs |K| <= |X|: the wave stops at the end of its highest order
	const int nmax = wide_wave_max<T>(comp ? n : 0), kend = (nmax + 1) * (nmax + 2) * (nmax + 3) / 6;
	T acc = T(0);
	for (int ch = 0; ch < 2; ++ch)
	{
		const int child = 2 * k + 1 + ch;
		if (comp)
		{
			const T dx = (T)cs[0] - (T)center[3 * child], dy = (T)cs[1] - (T)center[3 * child + 1], dz = (T)cs[2] - (T)center[3 * child + 2];
			const T dk = wide_mono(dx, x) * wide_mono(dy, y) * wide_mono(dz, z);
			if (FLAT) Dl[t] = dk; else D3[at] = dk;
			// (the first-order multipoles about a centre of charge vanish; the array holds zeros there)
			M3[at] = n == 1 ? T(0) : mpole[(size_t)child * offM + t] * (T)(wide_fact(n) / (wide_fact(x) * wide_fact(y) * wide_fact(z)));
		}
		__syncthreads();
		if (FLAT)
		{
#pragma unroll 4
			for (int kk = 0; kk < kend; ++kk)
			{
				const int2 kt = KT[kk];
				const bool ok = ((xp - kt.y) & 0x808080) == 0x808080;
				const T m = M3[ok ? at - kt.x : 0];
				acc = nb_fma(Dl[kk], ok ? m : T(0), acc);
			}
		}
		else if (comp && n >= 2)
			for (int k1 = 0; k1 <= x; ++k1)
				for (int k2 = 0; k2 <= y; ++k2)
				{
					const T *d = D3 + (k1 * S + k2) * S, *m = M3 + ((x - k1) * S + (y - k2)) * S + z;
					for (int k3 = 0; k3 <= z; ++k3) acc = nb_fma(d[k3], m[-k3], acc);
				}
		__syncthreads();
	}
	if (t == 0 && write_geom)
	{
		center[3 * k] = cs[0]; center[3 * k + 1] = cs[1]; center[3 * k + 2] = cs[2];
		mult[k] = ms;
	}
	if (!comp) return;
	T *M = mpole + (size_t)k * offM;
	if (n == 0) M[0] = (T)ms;
	else if (n == 1) M[t] = T(0);
	else M[t] = acc * (T)(wide_fact(x) * wide_fact(y) * wide_fact(z) / wide_fact(n));
}

// ---- L2L: one workgroup per child of level lchild (nodes first .. first + gridDim.x - 1 of that level) ----------------------
// Output (n, X) takes the monomials of orders 0 .. P - n: a prefix of the storage order; L~[X + K] sits at at(X) + at(K).
template <int P, typename T>
__global__ __launch_bounds__(kWide) void l2l_wide_kernel(const float *__restrict__ center, T *local, int lchild, int first)
{
	constexpr int offL = (P + 1) * (P + 1), offD = P * (P + 1) * (P + 2) / 6, S = P + 1;   // exponents 0 .. P
	__shared__ T Dl[offD], F3[S * S * S];
	__shared__ int KA[offD];   // at(K)
	__builtin_amdgcn_s_setprio(3);
	const int t = threadIdx.x, c = (1 << lchild) - 1 + first + (int)blockIdx.x, p = (c - 1) >> 1;
	const T dx = (T)center[3 * c] - (T)center[3 * p], dy = (T)center[3 * c + 1] - (T)center[3 * p + 1], dz = (T)center[3 * c + 2] - (T)center[3 * p + 2];
	// monomials of orders 0 .. P - 1
	if (t < offD)
	{
		int n, x, y, z;
		wide_decode_full(t, n, x, y, z);
		Dl[t] = wide_mono(dx, x) * wide_mono(dy, y) * wide_mono(dz, z);
		KA[t] = (x * S + y) * S + z;
	}
	// the parent's tuple in the full layout: the stored components (z = 0, 1) scaled by n!, the others from the vanishing trace
	int n = 0, x = 0, y = 0, z = 0;
	const bool comp = t >= 1 && t < offL;
	if (comp)
	{
		wide_decode_tl(t, n, x, y, z);
		F3[(x * S + y) * S + z] = local[(size_t)p * offL + t] * (T)wide_fact(n);
	}
	__syncthreads();
	for (int zz = 2; zz <= P; ++zz)
	{
		// components (order m >= zz, exponent of x in 0 .. m - zz) with this z: (P - zz + 1)(P - zz + 2) / 2 of them
		int r = t, m = zz;
		while (m <= P && r >= m - zz + 1) { r -= m - zz + 1; ++m; }
		if (m <= P)
		{
			const int fx = r, fy = m - r - zz;
			F3[(fx * S + fy) * S + zz] = -(F3[((fx + 2) * S + fy) * S + zz - 2] + F3[(fx * S + fy + 2) * S + zz - 2]);
		}
		__syncthreads();
	}
	const int at = (x * S + y) * S + z, room = P - n;
	const int mine = comp ? (room + 1) * (room + 2) * (room + 3) / 6 : 0, kend = wide_wave_max<T>(mine);
	T acc = T(0);
#pragma unroll 4
	for (int kk = 0; kk < kend; ++kk)
	{
		const bool ok = kk < mine;
		const T f = F3[ok ? at + KA[kk] : 0];
		acc = nb_fma(Dl[kk], ok ? f : T(0), acc);
	}
	if (comp) local[(size_t)c * offL + t] += acc * (T)(1.0 / wide_fact(n));
}
