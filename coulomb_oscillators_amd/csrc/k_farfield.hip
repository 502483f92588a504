// k_farfield.hip -- register-resident P2M / M2M / L2L / L2P for the kd-tree FMM, orders 1..10 (the highest orders spill part of their tensors to scratch).
// Reference drivers: fmm_multipoleLeaves3_kdtree (fmm_cart3_kdtree.cuh:231-250), fmm_buildTree3_kdtree2
// (:328-368), fmm_pushl3_kdtree (:1134-1194), fmm_pushLeaves3_kdtree (:1227-1275).  One THREAD per leaf /
// node / particle runs a generated straight-line body (fmm_ops_gen.inc, see gen_ops.py): all tensor
// components live in VGPRs, no LDS, no tables, no index arithmetic.
//
// The top of the tree has too few nodes to fill a launch, and every level would cost a kernel boundary:
// levels with at most kTopNodes nodes are processed by ONE workgroup that walks the levels with
// __syncthreads(); data it wrote at the previous level is re-read past the L1 (agent-scope loads).
#include "nbco_internal.hpp"

namespace {

#include "fmm_ops.hpp"

constexpr int kBlock = 256;
constexpr int kTopNodes = 256;   // levels with <= this many nodes are fused into one workgroup

__device__ inline float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- P2M: one thread per leaf ----------------------------------------------------------------------
// T = scalar type of the expansions and of the far-field arithmetic: float, or double with opts.far_fp64 (positions, centres and
// the whole tree geometry stay fp32 in both: the differences are formed in T from the fp32 values)
template <int P, typename T>
__global__ __launch_bounds__(kBlock) void p2m_gen_kernel(const float4 *__restrict__ pos, const float *__restrict__ center,
                                                         const int *__restrict__ mult, const int *__restrict__ index, T *__restrict__ mpole,
                                                         int beg, int nleaf)
{
	constexpr int offM = P * (P + 1) * (P + 2) / 6;
	const int i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= nleaf) return;
	const int leaf = beg + i, mlt = mult[leaf], ind = index[leaf];
	const T cx = (T)center[3 * leaf], cy = (T)center[3 * leaf + 1], cz = (T)center[3 * leaf + 2];
	T A[offM > 0 ? offM : 1];
#pragma unroll
	for (int k = 0; k < (offM > 0 ? offM : 1); ++k) A[k] = T(0);
	for (int j = 0; j < mlt; ++j)
	{
		const float4 p = pos[ind + j];
		p2m_accum<P>((T)p.x - cx, (T)p.y - cy, (T)p.z - cz, A);
	}
	T *M = mpole + (size_t)leaf * offM;
	if (offM > 0) M[0] = (T)mlt;
	if (offM > 1) { M[1] = T(0); M[2] = T(0); M[3] = T(0); }
	p2m_store<P>(A, M);
}

// ---- M2M ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// centre of charge of a parent, rounded like the oracle (fmm_cart3_kdtree.cuh:339-348)
template <bool AGENT>
__device__ inline void parent_centre(const float *center, const int *mult, int k, float c[3], int &mlt)
{
	const int c0 = 2 * k + 1, c1 = 2 * k + 2;
	const int m0 = AGENT ? ld_agent(&mult[c0]) : mult[c0], m1 = AGENT ? ld_agent(&mult[c1]) : mult[c1];
	mlt = m0 + m1;
	const float f0 = (float)m0, f1 = (float)m1, ft = (float)mlt;
	for (int a = 0; a < 3; ++a)
	{
		const float a0 = AGENT ? ld_agent(&center[3 * c0 + a]) : center[3 * c0 + a];
		const float a1 = AGENT ? ld_agent(&center[3 * c1 + a]) : center[3 * c1 + a];
		float s = f0 * a0;
		s = s + f1 * a1;
		c[a] = s / ft;
	}
}
// same rounding with the children's data passed in
__device__ inline void centre_of(int m0, int m1, const float *c0, const float *c1, float c[3])
{
	const float f0 = (float)m0, f1 = (float)m1, ft = (float)(m0 + m1);
	for (int a = 0; a < 3; ++a)
	{
		float s = f0 * c0[a];
		s = s + f1 * c1[a];
		c[a] = s / ft;
	}
}
// centre + squared box diagonal (kd_size, fmm_cart3_kdtree.cuh:395-399), the record the traversal reads
__device__ inline float4 node_csz(const float *__restrict__ lbound, const float *__restrict__ rbound, int i, float cx, float cy, float cz)
{
	const float dx = rbound[3 * i] - lbound[3 * i], dy = rbound[3 * i + 1] - lbound[3 * i + 1], dz = rbound[3 * i + 2] - lbound[3 * i + 2];
	const float sz = dx * dx + dy * dy + dz * dz;
	return make_float4(cx, cy, cz, sz);
}
#pragma clang fp contract(on)

// ---- centres pass: centre of charge + multiplicity of every internal node (fmm_cart3_kdtree.cuh:339-348) -------------
// One workgroup per subtree of <= kBlock leaves walks its levels in LDS; a second launch (one workgroup) does the levels
// above the subtree roots.
__global__ __launch_bounds__(kBlock) void kd_centres_kernel(float *center, int *mult, int L, int lr, const float *__restrict__ lbound,
                                                            const float *__restrict__ rbound, float4 *__restrict__ csz, int leaves_too)
{
	__shared__ float Cl[kBlock][3];
	__shared__ int Nl[kBlock];
	const int t = threadIdx.x, b = blockIdx.x;
	const int nl = 1 << (L - lr);   // leaves of this subtree
	if (t < nl)
	{
		const int leaf = (1 << L) - 1 + b * nl + t;
		Cl[t][0] = center[3 * leaf]; Cl[t][1] = center[3 * leaf + 1]; Cl[t][2] = center[3 * leaf + 2];
		Nl[t] = mult[leaf];
		if (leaves_too) csz[leaf] = node_csz(lbound, rbound, leaf, Cl[t][0], Cl[t][1], Cl[t][2]);
	}
	__syncthreads();
	for (int l = L - 1; l >= lr; --l)
	{
		const int cnt = 1 << (l - lr);
		float c[3] = {0.f, 0.f, 0.f};
		int m0 = 0, m1 = 0;
		if (t < cnt)
		{
			m0 = Nl[2 * t]; m1 = Nl[2 * t + 1];
			centre_of(m0, m1, Cl[2 * t], Cl[2 * t + 1], c);
		}
		__syncthreads();
		if (t < cnt)
		{
			const int node = (1 << l) - 1 + b * cnt + t;
			Cl[t][0] = c[0]; Cl[t][1] = c[1]; Cl[t][2] = c[2];
			Nl[t] = m0 + m1;
			center[3 * node] = c[0]; center[3 * node + 1] = c[1]; center[3 * node + 2] = c[2];
			mult[node] = m0 + m1;
			csz[node] = node_csz(lbound, rbound, node, c[0], c[1], c[2]);
		}
		__syncthreads();
	}
}
__global__ __launch_bounds__(kBlock) void kd_centres_top_kernel(float *center, int *mult, int ltop, const float *__restrict__ lbound,
                                                                const float *__restrict__ rbound, float4 *__restrict__ csz)
{
	__shared__ float Cl[kBlock][3];
	__shared__ int Nl[kBlock];
	const int t = threadIdx.x;
	for (int l = ltop; l >= 0; --l)
	{
		const int cnt = 1 << l;
		float c[3] = {0.f, 0.f, 0.f};
		int mlt = 0;
		if (t < cnt)
		{
			if (l == ltop) parent_centre<false>(center, mult, (1 << l) - 1 + t, c, mlt);   // children in HBM
			else
			{
				const int m0 = Nl[2 * t], m1 = Nl[2 * t + 1];
				centre_of(m0, m1, Cl[2 * t], Cl[2 * t + 1], c);
				mlt = m0 + m1;
			}
		}
		__syncthreads();
		if (t < cnt)
		{
			const int node = (1 << l) - 1 + t;
			Cl[t][0] = c[0]; Cl[t][1] = c[1]; Cl[t][2] = c[2];
			Nl[t] = mlt;
			center[3 * node] = c[0]; center[3 * node + 1] = c[1]; center[3 * node + 2] = c[2];
			mult[node] = mlt;
			csz[node] = node_csz(lbound, rbound, node, c[0], c[1], c[2]);
		}
		__syncthreads();
	}
}

// Levels ltop .. lroot of the subtrees hanging off level lroot, one workgroup per subtree (lroot = 0: the top of the tree in
// one workgroup).  The expansions, centres and multiplicities of the level just built stay in LDS (node i of a subtree's
// level sits in slot i), so a level costs LDS latency instead of a launch and HBM round trips.
template <int P, typename T>
__global__ __launch_bounds__(kTopNodes) void m2m_top_kernel(float *center, T *mpole, int *mult, int ltop, int lroot, int write_geom)
{
	constexpr int offM = P * (P + 1) * (P + 2) / 6, offS = offM > 0 ? offM : 1;
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	const int width = 1 << (ltop - lroot);            // nodes of this subtree at level ltop
	T *Ml = reinterpret_cast<T *>(lds_raw);           // [width][offS]
	float *Cl = reinterpret_cast<float *>(Ml + (size_t)width * offS);   // [width][3]
	int *Nl = (int *)(Cl + (size_t)width * 3);        // [width]
	const int t = threadIdx.x;
	for (int l = ltop; l >= lroot; --l)
	{
		const bool on = t < (1 << (l - lroot));
		const int k = (1 << l) - 1 + ((int)blockIdx.x << (l - lroot)) + t;
		T A[offS];
		float c[3] = {0.f, 0.f, 0.f};
		int mlt = 0;
		if (on)
		{
#pragma unroll
			for (int q = 0; q < offS; ++q) A[q] = T(0);
			if (l == ltop)
			{
				// children are in HBM (written by the previous launch)
				parent_centre<false>(center, mult, k, c, mlt);
				if (P >= 3)
					for (int ch = 0; ch < 2; ++ch)
					{
						const int child = 2 * k + 1 + ch;
						m2m_accum<P>(mpole + (size_t)child * offM, (T)c[0] - (T)center[3 * child], (T)c[1] - (T)center[3 * child + 1],
						             (T)c[2] - (T)center[3 * child + 2], A);
					}
			}
			else
			{
				const int s0 = 2 * t, s1 = 2 * t + 1;   // LDS slots of the children
				const int m0 = Nl[s0], m1 = Nl[s1];
				mlt = m0 + m1;
				centre_of(m0, m1, Cl + 3 * s0, Cl + 3 * s1, c);
				if (P >= 3)
					for (int ch = 0; ch < 2; ++ch)
					{
						const int sl = 2 * t + ch;
						m2m_accum<P>(Ml + (size_t)sl * offS, (T)c[0] - (T)Cl[3 * sl], (T)c[1] - (T)Cl[3 * sl + 1], (T)c[2] - (T)Cl[3 * sl + 2], A);
					}
			}
		}
		__syncthreads();   // all reads of the child slots are done
		if (on)
		{
			T *M = mpole + (size_t)k * offM;
			T *Ms = Ml + (size_t)t * offS;
			if (offM > 0) M[0] = (T)mlt;
			if (offM > 1) { M[1] = T(0); M[2] = T(0); M[3] = T(0); }
			m2m_store<P>(A, M);
			if (offM > 0) Ms[0] = (T)mlt;
			if (offM > 1) { Ms[1] = T(0); Ms[2] = T(0); Ms[3] = T(0); }
			m2m_store<P>(A, Ms);
			if (write_geom)
			{
				center[3 * k] = c[0]; center[3 * k + 1] = c[1]; center[3 * k + 2] = c[2];
				mult[k] = mlt;
			}
			Cl[3 * t] = c[0]; Cl[3 * t + 1] = c[1]; Cl[3 * t + 2] = c[2];
			Nl[t] = mlt;
		}
		__syncthreads();
	}
}

// ---- L2L ---------------------------------------------------------------------------------------------
template <int P, typename T, bool AGENT>
__device__ inline void l2l_node(const float *center, T *local, int c)
{
	constexpr int offL = (P + 1) * (P + 1);
	const int p = (c - 1) >> 1;
	T Lp[offL], O[offL];
#pragma unroll
	for (int q = 0; q < offL; ++q) Lp[q] = local[(size_t)p * offL + q];
	l2l_body<P>(Lp, (T)center[3 * c] - (T)center[3 * p], (T)center[3 * c + 1] - (T)center[3 * p + 1], (T)center[3 * c + 2] - (T)center[3 * p + 2], O);
	T *Lc = local + (size_t)c * offL;
#pragma unroll
	for (int q = 1; q < offL; ++q) Lc[q] += O[q];
}

template <int P, typename T>
__global__ __launch_bounds__(kBlock) void l2l_gen_kernel(const float *__restrict__ center, T *local, int lchild, int first, int count)
{
	const int i = blockIdx.x * kBlock + threadIdx.x;
	if (i >= count) return;
	l2l_node<P, T, false>(center, local, (1 << lchild) - 1 + first + i);
}

// (child levels 2 .. ltop ran in ONE workgroup of this shape, l2l_top_kernel, until round 3: see run_downward_gen)
// child levels lroot + 1 .. L of the subtrees hanging off level lroot, one workgroup per subtree: the level just
// finished stays in LDS as the next level's parents, so the whole lower part of the downward pass is one launch
// (per-level launches cost 10-25 us each while the near-field kernel fills the chip on the other stream).
// Same arithmetic per node as l2l_node: bit-identical locals.
// (Requesting a node's own tuple one level ahead takes the HBM round trip out of the level chain and this kernel from 36 to 26 us
// in a kernel trace -- and the step gains nothing: with 217 registers per lane instead of 128 its workgroups wait longer for room
// beside the near-field kernel's waves, which hold 480 of a SIMD's 512 registers.  The one-workgroup l2l_top_kernel went from 35
// to 89 us with the same change.  Kept simple.)
template <int P, typename T>
__global__ __launch_bounds__(256) void l2l_sub_kernel(const float *__restrict__ center, T *local, int lroot, int L, int first)
{
	constexpr int offL = (P + 1) * (P + 1);
	extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
	T *lds = reinterpret_cast<T *>(lds_raw);   // [2^(L - lroot - 1)][offL]
	const int t = threadIdx.x, r = first + blockIdx.x;
	for (int q = t; q < offL; q += blockDim.x) lds[q] = local[(size_t)((1 << lroot) - 1 + r) * offL + q];
	__syncthreads();
	for (int lc = lroot + 1; lc <= L; ++lc)
	{
		const bool on = t < (1 << (lc - lroot));
		const int c = (1 << lc) - 1 + (r << (lc - lroot)) + t, p = (c - 1) >> 1;
		T O[offL];
		if (on)
		{
			T Lp[offL];
			const T *src = lds + (size_t)(t >> 1) * offL;
#pragma unroll
			for (int q = 0; q < offL; ++q) Lp[q] = src[q];
			l2l_body<P>(Lp, (T)center[3 * c] - (T)center[3 * p], (T)center[3 * c + 1] - (T)center[3 * p + 1], (T)center[3 * c + 2] - (T)center[3 * p + 2], O);
			T *Lc = local + (size_t)c * offL;
#pragma unroll
			for (int q = 1; q < offL; ++q) { O[q] += Lc[q]; Lc[q] = O[q]; }
			O[0] = T(0);
		}
		if (lc == L) break;
		__syncthreads();
		if (on)
		{
			T *dst = lds + (size_t)t * offL;
#pragma unroll
			for (int q = 0; q < offL; ++q) dst[q] = O[q];
		}
		__syncthreads();
	}
}

// ---- L2P + near field + rescale + (un)sort -------------------------------------------------------------
// one thread per particle in tree order; its leaf is floor(2^L i / n) (the inverse of evalBox's ranges)
template <int P, typename T>
__global__ __launch_bounds__(kBlock) void l2p_gen_kernel(const float4 *__restrict__ pos, const float *__restrict__ center,
                                                         const T *__restrict__ local, const float4 *__restrict__ near,
                                                         const int *__restrict__ chunk_off, const int *__restrict__ index, int mlt_max,
                                                         const int *__restrict__ unsort, int scatter, const float *__restrict__ param,
                                                         float *__restrict__ a_out, int have_near, long long n, int L, long long own0,
                                                         long long own_n, const int2 *__restrict__ sec_range, const float4 *__restrict__ react,
                                                         int react_cap, int react_stride)
{
	constexpr int offL = (P + 1) * (P + 1);
	const long long io = (long long)blockIdx.x * kBlock + threadIdx.x;   // index among the domain's own particles
	if (io >= own_n) return;
	const long long i = own0 + io;
	const int lf = (int)(((1LL << L) * i) / n), leaf = (1 << L) - 1 + lf;
	const float4 p = pos[i];
	T Lp[offL];
#pragma unroll
	for (int q = 0; q < offL; ++q) Lp[q] = local[(size_t)leaf * offL + q];
	T fx, fy, fz;
	l2p_body<P>(Lp, (T)p.x - (T)center[3 * leaf], (T)p.y - (T)center[3 * leaf + 1], (T)p.z - (T)center[3 * leaf + 2], fx, fy, fz);
	if (have_near)
	{
		const int j = (int)(i - index[leaf]);
		float nx = 0.f, ny = 0.f, nz = 0.f;
		int ck0 = chunk_off[lf], ck1 = chunk_off[lf + 1];
		if (!NBCO_CHECKED_OK(ck0 >= 0 && ck0 <= ck1 && j >= 0 && j < mlt_max, NBCO_CHK_L2P)) ck1 = ck0;
		// (eight partial sums requested per wait: late in a run a leaf stretched by an ejected particle has a thousand work units, and
		// one load per round trip made this loop -- not the pair kernel -- the tail of the step.  Same additions in the same order.)
#pragma unroll 8
		for (int ck = ck0; ck < ck1; ++ck)
		{
			const float4 nr = near[(size_t)ck * mlt_max + j];
			nx += nr.x; ny += nr.y; nz += nr.z;
		}
		if (sec_range)
		{
			// mutual near field: the sums of the leaf pairs that the other leaf's wave evaluated lie side by side at the sorted
			// positions of this leaf's entries [sr.x, sr.y) (independent loads, added in list order)
			int2 sr = sec_range[lf];
			if (!NBCO_CHECKED_OK(sr.x >= 0 && sr.x <= sr.y, NBCO_CHK_L2P)) sr.y = sr.x;
			if (sr.y > react_cap) sr.y = max(sr.x, react_cap);   // (the host repeats an evaluation whose list outgrew the records)
#pragma unroll 8
			for (int e = sr.x; e < sr.y; ++e)
			{
				const float4 rr = react[(size_t)e * react_stride + j];
				nx += rr.x; ny += rr.y; nz += rr.z;
			}
		}
		fx += (T)nx; fy += (T)ny; fz += (T)nz;   // (fp64 far field: the fp32 near-field sums join it in double, one narrowing at the end)
	}
	const T scale = param ? (T)param[0] : T(1);
	const long long o = scatter ? (long long)unsort[io] : io;
	a_out[3 * o] = (float)(fx * scale); a_out[3 * o + 1] = (float)(fy * scale); a_out[3 * o + 2] = (float)(fz * scale);
}

#include "farfield_wide.hpp"   // orders 9, 10 (8 in double): a workgroup per node, a lane per component

static int grid_for(long long n) { return (int)((n + kBlock - 1) / kBlock); }

template <int P, typename T>
static int run_upward_gen(nbco_ctx *c, const float4 *pos, float *center, T *mpole, int *mult, const int *index, int L, int write_geom)
{
	const int nleaf = 1 << L, beg = nleaf - 1;
	hipLaunchKernelGGL((p2m_gen_kernel<P, T>), dim3(grid_for(nleaf)), dim3(kBlock), 0, c->stream, pos, (const float *)center, (const int *)mult, index,
	                   mpole, beg, nleaf);
	constexpr int offM = P * (P + 1) * (P + 2) / 6, offS = offM > 0 ? offM : 1;
	// a workgroup walks as many levels of its subtree as fit its LDS (up to 256 nodes at the bottom level: 8 levels at
	// p <= 5, 7 at p = 6): two or three launches for the whole upward shift instead of one per level
	int depth = 8;
	while (depth > 0 && (size_t)(1 << depth) * (offS * sizeof(T) + 16) > 60 * 1024) --depth;
	for (int l = L - 1; l >= 0;)
	{
		const int lroot = std::max(l - depth, 0), width = 1 << (l - lroot);
		hipLaunchKernelGGL((m2m_top_kernel<P, T>), dim3(1 << lroot), dim3(std::max(64, width)), (size_t)width * (offS * sizeof(T) + 16), c->stream, center, mpole,
		                   mult, l, lroot, write_geom);
		l = lroot - 1;
	}
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

// levels ltop .. 0 of a tree whose level ltop + 1 is already in place (the levels above the kd-domains)
template <int P, typename T>
static int run_m2m_top_gen(nbco_ctx *c, float *center, T *mpole, int *mult, int ltop, int write_geom)
{
	constexpr int offM = P * (P + 1) * (P + 2) / 6, offS = offM > 0 ? offM : 1;
	if ((1 << ltop) > kTopNodes || (size_t)(1 << ltop) * (offS * sizeof(T) + 16) > 60 * 1024)
		return c->fail(NBCO_ERR_UNSUPPORTED, "launch_m2m_top_gen: too many top levels");
	hipLaunchKernelGGL((m2m_top_kernel<P, T>), dim3(1), dim3(kTopNodes), (size_t)(1 << ltop) * (offS * sizeof(T) + 16), c->stream, center, mpole, mult, ltop, 0,
	                   write_geom);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

template <int P, typename T>
static int run_downward_gen(nbco_ctx *c, const float *center, T *local, int L, int dom_d, int dom_g)
{
	constexpr int offL = (P + 1) * (P + 1);
	int top = kTopNodes;
	while (top > 4 && (size_t)top * offL * sizeof(T) > 60 * 1024) top >>= 1;
	int ltop = 1;
	while (ltop + 1 <= L && (1 << (ltop + 1)) <= top) ++ltop;
	{
		// The levels with at most 256 nodes, one launch per level, a one-wave workgroup per node with a lane per component
		// (farfield_wide.hpp; ~30 registers and 3 KB of LDS).  They used to be ONE workgroup of 256 threads x 128 registers with 50 KB
		// of LDS walking the levels (l2l_top_kernel): beside the near-field kernel, which holds 480 of a SIMD's 512 registers and
		// most of the LDS, such a workgroup waits until two near-field workgroups of one CU retire together -- 30-90 us in the
		// benchmark's first steps, and 1.0 ms of a 3.0 ms step once the lists have grown (`profiles/r03q_late_timeline_l2l_top_starved.txt`): the whole
		// far-field chain behind it, and with it L2P, waited for the near-field kernel's tail.  A thin workgroup always fits.
	}
	// The lowest levels in one launch, a workgroup per subtree of at most 128 leaves (two waves, the parents' tuples in LDS); the
	// levels between run as thin workgroups as well.  (Subtrees of 256 leaves -- trees of 2^16 leaves and more -- meant
	// workgroups of four 128-register waves: 37 us at N = 1M became 175-200 us at N = 2M beside the near-field kernel.)
	// The arithmetic of a level depends on the level alone (wide form up to wtop, generated body below), whatever the launch shape.
	constexpr int offD = P * (P + 1) * (P + 2) / 6;
	const int threads = 64 * ((std::max(offD, offL) + 63) / 64);
	const int wtop = ltop >= 2 ? std::max(ltop, L - 7) : 1;
	int lroot = L;
	if (ltop >= 2)
	{
		lroot = wtop;
		while (lroot < L && (size_t)(1 << (L - lroot - 1)) * offL * sizeof(T) > 60 * 1024) ++lroot;
		if (lroot < dom_d) lroot = L;   // never with <= 8 domains; keep the per-level path for that case
	}
	for (int lc = 2; lc <= lroot; ++lc)
	{
		// below the domain level only the own subtree's nodes are needed
		const int first = lc >= dom_d ? dom_g << (lc - dom_d) : 0, count = lc >= dom_d ? 1 << (lc - dom_d) : 1 << lc;
		if (lc <= wtop) hipLaunchKernelGGL((l2l_wide_kernel<P, T>), dim3(count), dim3(threads), 0, c->stream, center, local, lc, first);
		else hipLaunchKernelGGL((l2l_gen_kernel<P, T>), dim3(grid_for(count)), dim3(kBlock), 0, c->stream, center, local, lc, first, count);
	}
	if (lroot < L)
	{
		const int first = dom_g << (lroot - dom_d), count = 1 << (lroot - dom_d);
		hipLaunchKernelGGL((l2l_sub_kernel<P, T>), dim3(count), dim3(std::max(64, 1 << (L - lroot))), (size_t)(1 << (L - lroot - 1)) * offL * sizeof(T),
		                   c->stream, center, local, lroot, L, first);
	}
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

// ---- the wide forms (farfield_wide.hpp): a workgroup per leaf / node, one launch per level (the levels are short: every
// node's components work side by side); the generated bodies of these orders are not even instantiated
// (the form depends on the level's HEIGHT above the leaves only -- the four lowest internal levels take the nest -- so a node is
// shifted by the same arithmetic in every launch shape: single GPU, a domain's subtree, the levels above the domains)
template <int P, typename T>
static void launch_m2m_wide(nbco_ctx *c, float *center, T *mpole, int *mult, int l, int height, int write_geom)
{
	if (height <= 4) hipLaunchKernelGGL((m2m_wide_kernel<P, T, false>), dim3(1 << l), dim3(kWide), 0, c->stream, center, mpole, mult, l, write_geom);
	else hipLaunchKernelGGL((m2m_wide_kernel<P, T, true>), dim3(1 << l), dim3(kWide), 0, c->stream, center, mpole, mult, l, write_geom);
}
template <int P, typename T>
static int run_upward(nbco_ctx *c, const float4 *pos, float *center, T *mpole, int *mult, const int *index, int L, int write_geom)
{
	if constexpr (use_wide<P, T>())
	{
		const int nleaf = 1 << L;
		hipLaunchKernelGGL((p2m_wide_kernel<P, T>), dim3(nleaf), dim3(kWide), 0, c->stream, pos, (const float *)center, (const int *)mult, index, mpole,
		                   nleaf - 1);
		for (int l = L - 1; l >= 0; --l) launch_m2m_wide<P, T>(c, center, mpole, mult, l, L - l, write_geom);
		NBCO_HIP(hipGetLastError());
		return NBCO_OK;
	}
	else return run_upward_gen<P, T>(c, pos, center, mpole, mult, index, L, write_geom);
}
template <int P, typename T>
static int run_m2m_top(nbco_ctx *c, float *center, T *mpole, int *mult, int ltop, int L, int write_geom)
{
	if constexpr (use_wide<P, T>())
	{
		for (int l = ltop; l >= 0; --l) launch_m2m_wide<P, T>(c, center, mpole, mult, l, L - l, write_geom);
		NBCO_HIP(hipGetLastError());
		return NBCO_OK;
	}
	else return run_m2m_top_gen<P, T>(c, center, mpole, mult, ltop, write_geom);
}
template <int P, typename T>
static int run_downward(nbco_ctx *c, const float *center, T *local, int L, int dom_d, int dom_g)
{
	if constexpr (use_wide<P, T>())
	{
		for (int lc = 2; lc <= L; ++lc)
		{
			// below the domain level only the own subtree's nodes are needed
			const int first = lc >= dom_d ? dom_g << (lc - dom_d) : 0, count = lc >= dom_d ? 1 << (lc - dom_d) : 1 << lc;
			hipLaunchKernelGGL((l2l_wide_kernel<P, T>), dim3(count), dim3(kWide), 0, c->stream, center, local, lc, first);
		}
		NBCO_HIP(hipGetLastError());
		return NBCO_OK;
	}
	else return run_downward_gen<P, T>(c, center, local, L, dom_d, dom_g);
}

template <int P, typename T>
static int run_l2p(nbco_ctx *c, const float4 *pos, const float *center, const T *local, const float4 *near, const int *chunk_off,
                   const int *index, int mlt_max, const int *unsort, int scatter, const float *param, float *a, int have_near, long long n, int L,
                   long long own0, long long own_n, const int2 *sec_range, const float4 *react, long long react_cap, int react_stride)
{
	hipLaunchKernelGGL((l2p_gen_kernel<P, T>), dim3(grid_for(own_n)), dim3(kBlock), 0, c->stream, pos, center, local, near, chunk_off, index, mlt_max,
	                   unsort, scatter, param, a, have_near, n, L, own0, own_n, sec_range, react, (int)std::min<long long>(react_cap, 0x7fffffff), react_stride);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

} // namespace

#define NBCO_DISPATCH_P(P, CALL)                                                       \
	switch (P)                                                                         \
	{                                                                                  \
	case 1: return CALL(1); case 2: return CALL(2); case 3: return CALL(3); case 4: return CALL(4); \
	case 5: return CALL(5); case 6: return CALL(6); case 7: return CALL(7); case 8: return CALL(8); \
	case 9: return CALL(9); case 10: return CALL(10);                                                \
	default: return c->fail(NBCO_ERR_UNSUPPORTED, "generated far-field operators exist for orders 1..10"); \
	}

// centres + multiplicities of all internal nodes from the leaves' (2 launches up to 16 levels)
int launch_kd_centres(nbco_ctx *c, float *center, int *mult, int L, const float *lbound, const float *rbound, float4 *csz)
{
	if (L == 0)
	{
		// a one-node tree: only the traversal record of the root is missing
		hipLaunchKernelGGL(kd_centres_kernel, dim3(1), dim3(kBlock), 0, c->stream, center, mult, 0, 0, lbound, rbound, csz, 1);
		NBCO_HIP(hipGetLastError());
		return NBCO_OK;
	}
	// a workgroup walks 8 levels (256 leaves of its subtree in LDS), the last launch (one workgroup) the <= 8 levels left:
	// trees deeper than 16 levels (N > 2M at p = 6) take one more subtree stage per 8 levels.  Every node's traversal
	// record (centre + squared box diagonal) is written along the way.
	int bottom = L;
	for (; bottom > 16; bottom -= 8)
		hipLaunchKernelGGL(kd_centres_kernel, dim3(1 << (bottom - 8)), dim3(kBlock), 0, c->stream, center, mult, bottom, bottom - 8, lbound, rbound, csz,
		                   bottom == L ? 1 : 0);
	const int lr = bottom > 8 ? bottom - 8 : 0;
	hipLaunchKernelGGL(kd_centres_kernel, dim3(1 << lr), dim3(kBlock), 0, c->stream, center, mult, bottom, lr, lbound, rbound, csz, bottom == L ? 1 : 0);
	if (lr > 0) hipLaunchKernelGGL(kd_centres_top_kernel, dim3(1), dim3(kBlock), 0, c->stream, center, mult, lr - 1, lbound, rbound, csz);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

// (f64 != 0: mpole / local point at double tuples -- opts.far_fp64; orders above 8 in double are not instantiated for the upward /
// downward shifts' fused kernels beyond what fits their LDS: the launchers size themselves from sizeof(T))
int launch_upward_gen(nbco_ctx *c, int P, const float4 *pos, float *center, void *mpole, int *mult, const int *index, int L, int write_geom, int f64)
{
	if (f64)
	{
#define CALL(PP) run_upward<PP, double>(c, pos, center, (double *)mpole, mult, index, L, write_geom)
		NBCO_DISPATCH_P(P, CALL)
#undef CALL
	}
#define CALL(PP) run_upward<PP, float>(c, pos, center, (float *)mpole, mult, index, L, write_geom)
	NBCO_DISPATCH_P(P, CALL)
#undef CALL
}

// centres, multiplicities and traversal records of levels ltop .. 0 from level ltop + 1 (the levels above the kd-domains; the
// boxes of those levels come from the partition step)
int launch_kd_centres_top(nbco_ctx *c, float *center, int *mult, int ltop, const float *lbound, const float *rbound, float4 *csz)
{
	if ((1 << ltop) > kBlock) return c->fail(NBCO_ERR_UNSUPPORTED, "launch_kd_centres_top: too many top levels");
	hipLaunchKernelGGL(kd_centres_top_kernel, dim3(1), dim3(kBlock), 0, c->stream, center, mult, ltop, lbound, rbound, csz);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

int launch_m2m_top_gen(nbco_ctx *c, int P, float *center, void *mpole, int *mult, int ltop, int L, int write_geom, int f64)
{
	if (f64)
	{
#define CALL(PP) run_m2m_top<PP, double>(c, center, (double *)mpole, mult, ltop, L, write_geom)
		NBCO_DISPATCH_P(P, CALL)
#undef CALL
	}
#define CALL(PP) run_m2m_top<PP, float>(c, center, (float *)mpole, mult, ltop, L, write_geom)
	NBCO_DISPATCH_P(P, CALL)
#undef CALL
}

int launch_downward_gen(nbco_ctx *c, int P, const float *center, void *local, int L, int dom_d, int dom_g, int f64)
{
	if (f64)
	{
#define CALL(PP) run_downward<PP, double>(c, center, (double *)local, L, dom_d, dom_g)
		NBCO_DISPATCH_P(P, CALL)
#undef CALL
	}
#define CALL(PP) run_downward<PP, float>(c, center, (float *)local, L, dom_d, dom_g)
	NBCO_DISPATCH_P(P, CALL)
#undef CALL
}

int launch_l2p_gen(nbco_ctx *c, int P, const float4 *pos, const float *center, const void *local, const float4 *near, const int *chunk_off,
                   const int *index, int mlt_max, const int *unsort, int scatter, const float *param, float *a, int have_near, long long n, int L,
                   long long own0, long long own_n, const int2 *sec_range, const float4 *react, long long react_cap, int react_stride, int f64)
{
	if (f64)
	{
#define CALL(PP) run_l2p<PP, double>(c, pos, center, (const double *)local, near, chunk_off, index, mlt_max, unsort, scatter, param, a, have_near, n, L, own0, own_n, sec_range, react, react_cap, react_stride)
		NBCO_DISPATCH_P(P, CALL)
#undef CALL
	}
#define CALL(PP) run_l2p<PP, float>(c, pos, center, (const float *)local, near, chunk_off, index, mlt_max, unsort, scatter, param, a, have_near, n, L, own0, own_n, sec_range, react, react_cap, react_stride)
	NBCO_DISPATCH_P(P, CALL)
#undef CALL
}

NBCO_CHECKED_COLLECT(nbco_checked_collect_far)
