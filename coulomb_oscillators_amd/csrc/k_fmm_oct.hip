// k_fmm_oct.hip -- uniform-octree FMM with traceless multipoles and locals.
// Reference behaviour: fmm_cart3_traceless.cuh (driver :282-435 GPU, :437-571 CPU), shared pieces
// fmm_cart3_symmetric.cuh:293-411 (L2L / L2P drivers), appel.cuh:44-70 (keys), :141-212 (indexLeaves,
// multLeaves), :226-258 (centerLeaves), :320-366 (stencil P2P), operators of fmm_cart_base3.cuh.
// Semantics follow the reference's CPU driver (as the oracle does): tree rebuilt on every call, positions,
// velocities and accelerations are left in cell order, M2L implemented as intended for every order.
//
// gfx950 design (one evaluation, no float atomics anywhere, bit-reproducible):
//   keys      integer cell key = row-major flatten of the clipped integer cell coordinates -- evaluated without
//             FMA contraction, bit-exact with the oracle; one stable rocPRIM radix sort over 3L bits
//   leaves    one wave per cell: multiplicity, centroid, traceless P2M with the generated straight-line body
//   M2M       one thread per parent cell, 8 children, generated body (all tensors in VGPRs)
//   M2L       the (parent's-neighbours minus own-neighbours) stencil of every non-empty cell is written out as a
//             directed (target, source) list -- already sorted by construction -- and the traceless multipoles are
//             expanded into the symmetric layout, so the register-resident M2L of the kd-tree path (k_m2l.hip) is
//             reused as it is: contracting with the traceless gradient only sees the traceless part of M
//   P2P       a cell's (2r+1)^2 neighbour columns are contiguous particle runs; they are cut into source
//             descriptors of <= 64 particles, the cell's targets into groups of <= TPL, and (group, descriptor
//             range) chunks go through the same chunked pair kernel as the kd-tree path (k_p2p.hpp)
//   L2L/L2P   generated bodies shared with the kd-tree path; L2P is fused with the near-field sum and the rescale
#include "nbco_internal.hpp"
#include "k_p2p.hpp"
#include <rocprim/rocprim.hpp>
#include <cmath>
#include <algorithm>

namespace {

#include "fmm_ops.hpp"

constexpr int kBlock = 256;
constexpr int kSrcPiece = 64;   // particles per P2P source descriptor
constexpr int kOctChunk = 64;   // descriptors per P2P chunk

__host__ __device__ inline int oct_beg(int l) { return ((1 << (3 * l)) - 1) / 7; }
__host__ __device__ inline int oct_cnt(int l) { return 1 << (3 * l); }

// T = scalar type of the expansions and of all far-field arithmetic: float, or double with opts.far_fp64 (the range of
// fp32 ends near r^-11 19!! ~ 1e40 at order 10); positions, centres and the near field are fp32 either way
template <typename T>
struct OctView
{
	float4 *csz;              // [ntot] centre (w unused)
	T *mpole, *local, *msym;  // [ntot][offL] traceless tuples, [ntot][offM] symmetric copy of orders 0..P-1 for M2L
	int *mult, *index;            // [ntot], [ntot + 1] (index[ntot] = n)
	int L, ntot, side;
	long long n;
};

static int grid1d(long long n, int cap = 1 << 20) { return (int)std::min<long long>((n + kBlock - 1) / kBlock, cap); }

// ---- keys -----------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
// scal = {min x, min y, min z, rdelta}; fmm_cart3_traceless.cuh:358-362 / :466-470
__global__ void oct_scalars_kernel(const float *__restrict__ minmax6, int side, float eps, float *__restrict__ scal)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const float Dx = minmax6[3] - minmax6[0], Dy = minmax6[4] - minmax6[1], Dz = minmax6[5] - minmax6[2];
	float delta = fmaxf(fmaxf(Dx, Dy), Dz) / (float)side;
	if (delta < eps) delta = eps;
	scal[0] = minmax6[0]; scal[1] = minmax6[1]; scal[2] = minmax6[2];
	scal[3] = 1.f / delta;
}
// appel.cuh:44-55 with to_ivec / clip / flatten of mymath.cuh:326-398
__global__ __launch_bounds__(kBlock) void oct_keys_kernel(const float4 *__restrict__ pos, long long n, const float *__restrict__ scal, int side,
                                                          uint32_t *__restrict__ keys, uint32_t *__restrict__ idx)
{
	const float mx = scal[0], my = scal[1], mz = scal[2], rd = scal[3];
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		const float4 p = pos[i];
		const float qx = (p.x - mx) * rd, qy = (p.y - my) * rd, qz = (p.z - mz) * rd;
		int ix = (int)qx, iy = (int)qy, iz = (int)qz;
		ix = ix < 0 ? 0 : (ix > side - 1 ? side - 1 : ix);
		iy = iy < 0 ? 0 : (iy > side - 1 ? side - 1 : iy);
		iz = iz < 0 ? 0 : (iz > side - 1 ? side - 1 : iz);
		keys[i] = (uint32_t)((ix * side + iy) * side + iz);
		idx[i] = (uint32_t)i;
	}
}
#pragma clang fp contract(on)

__global__ __launch_bounds__(kBlock) void oct_gather4_kernel(const float4 *__restrict__ src, const uint32_t *__restrict__ idx, float4 *__restrict__ dst,
                                                             long long n)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) dst[i] = src[idx[i]];
}
__global__ __launch_bounds__(kBlock) void oct_unpack4_kernel(const float4 *__restrict__ src, float *__restrict__ dst, long long n)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		const float4 q = src[i];
		dst[3 * i] = q.x; dst[3 * i + 1] = q.y; dst[3 * i + 2] = q.z;
	}
}

// indexLeaves (appel.cuh:141-167): first particle whose key is >= the cell number (empty cells point at the next
// non-empty one); index[ntot] = n closes the last cell
template <typename T>
__global__ __launch_bounds__(kBlock) void oct_index_kernel(OctView<T> t, const uint32_t *__restrict__ keys)
{
	const int m = oct_cnt(t.L), beg = oct_beg(t.L);
	for (int c = blockIdx.x * kBlock + threadIdx.x; c <= m; c += gridDim.x * kBlock)
	{
		long long lo = 0, hi = t.n;
		while (lo < hi)
		{
			const long long mid = (lo + hi) >> 1;
			if (keys[mid] < (uint32_t)c) lo = mid + 1; else hi = mid;
		}
		t.index[beg + c] = (int)lo;
	}
}

// ---- multi-GPU: slabs of the sorted cell keys ---------------------------------------------------------------
// The cell keys are row-major (x slowest), so a contiguous range of leaf cells in key order is a slab of x-layers.  Rank r of G
// serves the cells [c_r, c_r+1), c_r = the first cell whose first particle is >= n r / G: its particles are the contiguous range
// [index[c_r], index[c_r+1]) of the cell order.  Every rank builds the whole tree and the whole upward pass (O(N), a few per
// cent of a step); M2L / L2L run for the nodes whose x-range meets the slab's layers, P2P and L2P for the slab's cells only.
// slab (device, 6 ints): {c0, c1, x0, x1, p0, p1} -- leaf cells, leaf x-layers (inclusive), particles; a null pointer = everything.
template <typename T>
__global__ void oct_slab_kernel(OctView<T> t, int world, int rank, int *__restrict__ slab, long long *__restrict__ pbounds)
{
	const int r = threadIdx.x;
	if (blockIdx.x != 0 || r > world) return;
	const int m = oct_cnt(t.L), beg = oct_beg(t.L);
	const long long b = (t.n * r) / world;
	int lo = 0, hi = m;   // smallest c in [0, m] with index[beg + c] >= b   (index[beg + m] = n)
	while (lo < hi)
	{
		const int mid = (lo + hi) >> 1;
		if (t.index[beg + mid] < b) lo = mid + 1; else hi = mid;
	}
	__shared__ int cb[65];
	cb[r] = lo;
	pbounds[r] = t.index[beg + lo];
	__syncthreads();
	if (r == rank)
	{
		const int c0 = cb[r], c1 = cb[r + 1], ss = t.side * t.side;
		slab[0] = c0; slab[1] = c1;
		slab[2] = c0 / ss; slab[3] = c1 > c0 ? (c1 - 1) / ss : c0 / ss - 1;
		slab[4] = t.index[beg + c0]; slab[5] = t.index[beg + c1];
	}
}
// does the node (level l, x index i at that level) lie above or inside one of the slab's x-layers?
__device__ inline bool oct_in_slab(const int *__restrict__ slab, int L, int l, int i)
{
	if (!slab) return true;
	const int lo = i << (L - l), hi = ((i + 1) << (L - l)) - 1;
	return lo <= slab[3] && hi >= slab[2];
}

// ---- leaves: multLeaves + centerLeaves + P2M, one wave per cell ------------------------------------------------
// SYM: symmetric multipoles of orders 0..P about the centroid (fmm_multipoleLeaves3, fmm_cart3_symmetric.cuh:71-99): the kd-tree
// flavour's generated P2M of order P + 1, whose tuple holds orders 0..P
template <int P, typename T, bool SYM>
__global__ __launch_bounds__(kBlock) void oct_leaf_kernel(OctView<T> t, const float4 *__restrict__ pos)
{
	constexpr int offL = SYM ? NBCO_OFFM(P + 1) : NBCO_OFFL(P);
	const int m = oct_cnt(t.L), beg = oct_beg(t.L);
	const int c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (c >= m) return;
	const int node = beg + c, i0 = t.index[node], mlt = t.index[node + 1] - i0;
	float sx = 0.f, sy = 0.f, sz = 0.f;
	for (int j = lane; j < mlt; j += 64)
	{
		const float4 q = pos[i0 + j];
		sx += q.x; sy += q.y; sz += q.z;
	}
	for (int o = 32; o > 0; o >>= 1) { sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o); }
	float cx = 0.f, cy = 0.f, cz = 0.f;
	if (mlt > 0) { const float d = (float)mlt; cx = sx / d; cy = sy / d; cz = sz / d; }
	T A[offL];
#pragma unroll
	for (int q = 0; q < offL; ++q) A[q] = T(0);
	for (int j = lane; j < mlt; j += 64)
	{
		const float4 q = pos[i0 + j];
		if constexpr (SYM) p2m_accum<P + 1>((T)q.x - (T)cx, (T)q.y - (T)cy, (T)q.z - (T)cz, A);
		else p2m_tl_accum<P>((T)q.x - (T)cx, (T)q.y - (T)cy, (T)q.z - (T)cz, A);
	}
#pragma unroll
	for (int q = 4; q < offL; ++q)
		for (int o = 32; o > 0; o >>= 1) A[q] += __shfl_xor(A[q], o);
	T *M = t.mpole + (size_t)node * offL;
	if (lane == 0)
	{
		t.mult[node] = mlt;
		t.csz[node] = make_float4(cx, cy, cz, 0.f);
		M[0] = (T)mlt; M[1] = T(0); M[2] = T(0); M[3] = T(0);
		if constexpr (SYM) p2m_store<P + 1>(A, M);   // normalised accumulators -> the reference's normalisation, orders 2..P
	}
	// lane q stores component q (every lane holds the full sums after the butterfly)
	if constexpr (!SYM)
	{
#pragma unroll
		for (int q = 4; q < offL; ++q)
			if (lane == (q & 63)) M[q] = A[q];
	}
}

// children of cell (i, j, k) of level l (fmm_cart3_traceless.cuh:128-139): level-major, row-major cells
__device__ inline void oct_children(int l, int c0, int inds[8])
{
	const int sl = 1 << l, sp = 1 << (l + 1), begp = oct_beg(l + 1);
	const int i = c0 / (sl * sl), jk = c0 - i * sl * sl, j = jk / sl, k = jk - j * sl;
	const int b = begp + 2 * (i * sp * sp + j * sp + k);
	inds[0] = b; inds[1] = b + 1; inds[2] = b + sp; inds[3] = b + sp + 1;
	inds[4] = b + sp * sp; inds[5] = b + sp * sp + 1; inds[6] = b + sp * sp + sp; inds[7] = b + sp * sp + sp + 1;
}

// ---- M2M, one thread per parent (fmm_cart3_traceless.cuh:110-168) ------------------------------------------------
template <int P, typename T, bool SYM>
__global__ __launch_bounds__(kBlock) void oct_m2m_kernel(OctView<T> t, int l)
{
	constexpr int offL = SYM ? NBCO_OFFM(P + 1) : NBCO_OFFL(P);
	const int c0 = blockIdx.x * kBlock + threadIdx.x;
	if (c0 >= oct_cnt(l)) return;
	const int node = oct_beg(l) + c0;
	int inds[8];
	oct_children(l, c0, inds);
	int mlt = 0;
	float cx = 0.f, cy = 0.f, cz = 0.f;
	for (int q = 0; q < 8; ++q)
	{
		const int mq = t.mult[inds[q]];
		if (mq == 0) continue;
		const float4 cc = t.csz[inds[q]];
		const float f = (float)mq;
		cx += f * cc.x; cy += f * cc.y; cz += f * cc.z;
		mlt += mq;
	}
	T *M = t.mpole + (size_t)node * offL;
	if (mlt > 0)
	{
		const float d = (float)mlt;
		cx /= d; cy /= d; cz /= d;
		T A[offL];
#pragma unroll
		for (int q = 0; q < offL; ++q) A[q] = T(0);
		for (int q = 0; q < 8; ++q)
		{
			if (t.mult[inds[q]] == 0) continue;
			const float4 cc = t.csz[inds[q]];
			if constexpr (SYM)
			{
				// fmm_buildTree3 (fmm_cart3_symmetric.cuh:121-179): m2m_acc3 over the children, orders 2..P
				if (P + 1 >= 3) m2m_accum<P + 1>((const T *)(t.mpole + (size_t)inds[q] * offL), (T)cx - (T)cc.x, (T)cy - (T)cc.y, (T)cz - (T)cc.z, A);
			}
			else m2m_tl_accum<P>((const T *)(t.mpole + (size_t)inds[q] * offL), (T)cx - (T)cc.x, (T)cy - (T)cc.y, (T)cz - (T)cc.z, A);
		}
		M[0] = (T)mlt; M[1] = T(0); M[2] = T(0); M[3] = T(0);
		if constexpr (SYM) m2m_store<P + 1>(A, M);
		else
		{
#pragma unroll
			for (int q = 4; q < offL; ++q) M[q] = A[q];
		}
	}
	else
		for (int q = 0; q < offL; ++q) M[q] = T(0);
	t.mult[node] = mlt;
	t.csz[node] = make_float4(cx, cy, cz, 0.f);
}

// symmetric-layout copy of the traceless multipoles, orders 0..P-1 (the first 2n+1 entries of both layouts coincide;
// the rest follows from tracelessness, fmm_cart_base3.cuh:611-623)
template <typename T>
__global__ __launch_bounds__(kBlock) void oct_expand_kernel(OctView<T> t, int P, int first)
{
	const int offL = (P + 1) * (P + 1), offM = P * (P + 1) * (P + 2) / 6;
	for (int node = first + blockIdx.x * kBlock + threadIdx.x; node < t.ntot; node += gridDim.x * kBlock)
	{
		if (t.mult[node] == 0) continue;
		const T *Tl = t.mpole + (size_t)node * offL;
		T *S = t.msym + (size_t)node * offM;
		for (int n = 0; n < P; ++n)
		{
			T *Sn = S + n * (n + 1) * (n + 2) / 6;
			const T *Tn = Tl + n * n;
			for (int q = 0; q < 2 * n + 1; ++q) Sn[q] = Tn[q];
			for (int z = 2; z <= n; ++z)
				for (int x = n - z; x >= 0; --x)
				{
					auto si = [n](int xx, int zz) { return (n * (n + 1) - (n - zz) * (n - zz + 1)) / 2 + n - xx; };
					Sn[si(x, z)] = -Sn[si(x + 2, z - 2)] - Sn[si(x, z - 2)];
				}
		}
	}
}

// ---- M2L stencil lists (fmm_cart3_traceless.cuh:196-254) -------------------------------------------------------------
__device__ inline int oct_level_of(int node)
{
	int l = 0;
	while (oct_beg(l + 1) <= node) ++l;
	return l;
}

// FILL = false: cnt[node] = number of non-empty sources; FILL = true: keys[start[node] ..] = target << shift | source
template <bool FILL, typename T>
__global__ __launch_bounds__(kBlock) void oct_m2l_list_kernel(OctView<T> t, int radius, int first, int *__restrict__ cnt, const int *__restrict__ start,
                                                              int shift, uint64_t *__restrict__ keys, const int *__restrict__ slab)
{
	for (int node = blockIdx.x * kBlock + threadIdx.x; node < t.ntot; node += gridDim.x * kBlock)
	{
		int count = 0;
		if (node >= first && t.mult[node] > 0)
		{
			const int l = oct_level_of(node), sl = 1 << l, lb = oct_beg(l), c = node - lb;
			const int i = c / (sl * sl), jk = c - i * sl * sl, j = jk / sl, k = jk - j * sl;
			if (!oct_in_slab(slab, t.L, l, i)) { if (!FILL) cnt[node] = 0; continue; }
			const int im = (i / 2) * 2, jm = (j / 2) * 2, km = (k / 2) * 2;
			const int f0 = max(im - 2 * radius, 0), f1 = min(im + 2 * radius + 1, sl - 1);
			const int g0 = max(jm - 2 * radius, 0), g1 = min(jm + 2 * radius + 1, sl - 1);
			const int h0 = max(km - 2 * radius, 0), h1 = min(km + 2 * radius + 1, sl - 1);
			uint64_t *out = FILL ? keys + start[node] : nullptr;
			for (int f = f0; f <= f1; ++f)
				for (int g = g0; g <= g1; ++g)
					for (int h = h0; h <= h1; ++h)
					{
						if (!(f > i + radius || f < i - radius || g > j + radius || g < j - radius || h > k + radius || h < k - radius)) continue;
						const int src = lb + (f * sl + g) * sl + h;
						if (t.mult[src] == 0) continue;   // empty sources carry zero multipoles
						if (FILL) out[count] = ((uint64_t)node << shift) | (uint64_t)src;
						++count;
					}
		}
		if (!FILL) cnt[node] = count;
	}
	if (!FILL && blockIdx.x == 0 && threadIdx.x == 0) cnt[t.ntot] = 0;
}

// ---- L2L: one thread per child cell (fmm_cart3_symmetric.cuh:293-334) ---------------------------------------------
template <int P, typename T>
__global__ __launch_bounds__(kBlock) void oct_l2l_kernel(OctView<T> t, int lchild, const int *__restrict__ slab)
{
	constexpr int offL = NBCO_OFFL(P);
	const int c = blockIdx.x * kBlock + threadIdx.x;
	if (c >= oct_cnt(lchild)) return;
	const int node = oct_beg(lchild) + c;
	if (t.mult[node] == 0) return;
	const int sl = 1 << lchild, sp = sl >> 1;
	const int i = c / (sl * sl), jk = c - i * sl * sl, j = jk / sl, k = jk - j * sl;
	if (!oct_in_slab(slab, t.L, lchild, i)) return;
	const int parent = oct_beg(lchild - 1) + ((i >> 1) * sp + (j >> 1)) * sp + (k >> 1);
	T Lp[offL], O[offL];
#pragma unroll
	for (int q = 0; q < offL; ++q) Lp[q] = t.local[(size_t)parent * offL + q];
	const float4 cc = t.csz[node], cp = t.csz[parent];
	l2l_body<P>(Lp, (T)cc.x - (T)cp.x, (T)cc.y - (T)cp.y, (T)cc.z - (T)cp.z, O);
	T *Lc = t.local + (size_t)node * offL;
#pragma unroll
	for (int q = 1; q < offL; ++q) Lc[q] += O[q];
}

// ---- P2P work units (appel.cuh:320-366: (2r+1)^2 neighbour columns, contiguous z-runs merged) ----------------------
// per leaf cell: number of target groups, of source descriptors and of chunks
template <typename T>
__global__ __launch_bounds__(kBlock) void oct_p2p_count_kernel(OctView<T> t, int radius, int tpl, int *__restrict__ ngroup, int *__restrict__ ndesc,
                                                               int *__restrict__ nchunk, const int *__restrict__ slab)
{
	const int m = oct_cnt(t.L), beg = oct_beg(t.L), side = t.side;
	for (int c = blockIdx.x * kBlock + threadIdx.x; c <= m; c += gridDim.x * kBlock)
	{
		int ng = 0, nd = 0;
		if (c < m && t.mult[beg + c] > 0 && (!slab || (c >= slab[0] && c < slab[1])))
		{
			ng = (t.mult[beg + c] + tpl - 1) / tpl;
			const int i = c / (side * side), jk = c - i * side * side, j = jk / side, k = jk - j * side;
			const int x0 = max(i - radius, 0), x1 = min(i + radius, side - 1), y0 = max(j - radius, 0), y1 = min(j + radius, side - 1);
			const int z0 = max(k - radius, 0), z1 = min(k + radius, side - 1);
			for (int x = x0; x <= x1; ++x)
				for (int y = y0; y <= y1; ++y)
				{
					const int cT = beg + (x * side + y) * side + z0;
					const int cntp = t.index[cT + (z1 - z0) + 1] - t.index[cT];
					nd += (cntp + kSrcPiece - 1) / kSrcPiece;
				}
		}
		ngroup[c] = ng; ndesc[c] = nd; nchunk[c] = ng * ((nd + kOctChunk - 1) / kOctChunk);
	}
}

// per leaf cell: target groups (first particle, count), the cell of every group, and the cell's source descriptors
template <typename T>
__global__ __launch_bounds__(kBlock) void oct_p2p_fill_kernel(OctView<T> t, int radius, int tpl, const int *__restrict__ group_off,
                                                              const int *__restrict__ desc_off, int *__restrict__ grp_index,
                                                              int *__restrict__ grp_mult, int *__restrict__ grp_cell, int2 *__restrict__ desc,
                                                              const int *__restrict__ slab)
{
	const int m = oct_cnt(t.L), beg = oct_beg(t.L), side = t.side;
	for (int c = blockIdx.x * kBlock + threadIdx.x; c < m; c += gridDim.x * kBlock)
	{
		const int mlt = t.mult[beg + c];
		if (mlt == 0 || (slab && (c < slab[0] || c >= slab[1]))) continue;
		const int i0 = t.index[beg + c], g0 = group_off[c], ng = group_off[c + 1] - g0;
		for (int g = 0; g < ng; ++g)
		{
			grp_index[g0 + g] = i0 + g * tpl;
			grp_mult[g0 + g] = min(tpl, mlt - g * tpl);
			grp_cell[g0 + g] = c;
		}
		const int i = c / (side * side), jk = c - i * side * side, j = jk / side, k = jk - j * side;
		const int x0 = max(i - radius, 0), x1 = min(i + radius, side - 1), y0 = max(j - radius, 0), y1 = min(j + radius, side - 1);
		const int z0 = max(k - radius, 0), z1 = min(k + radius, side - 1);
		int2 *out = desc + desc_off[c];
		for (int x = x0; x <= x1; ++x)
			for (int y = y0; y <= y1; ++y)
			{
				const int cT = beg + (x * side + y) * side + z0;
				const int s0 = t.index[cT], cntp = t.index[cT + (z1 - z0) + 1] - s0;
				for (int o = 0; o < cntp; o += kSrcPiece) *out++ = make_int2(s0 + o, min(kSrcPiece, cntp - o));
			}
	}
}

// per target group: its chunks = slices of its cell's descriptor range
__global__ __launch_bounds__(kBlock) void oct_p2p_chunk_kernel(const int *__restrict__ ngroups_total, const int *__restrict__ grp_cell,
                                                               const int *__restrict__ group_off, const int *__restrict__ desc_off,
                                                               const int *__restrict__ chunk_off, const int *__restrict__ grp_index,
                                                               const int *__restrict__ grp_mult, int4 *__restrict__ chunk)
{
	const int ng = *ngroups_total;
	for (int g = blockIdx.x * kBlock + threadIdx.x; g < ng; g += gridDim.x * kBlock)
	{
		const int c = grp_cell[g], d0 = desc_off[c], d1 = desc_off[c + 1];
		const int per = (d1 - d0 + kOctChunk - 1) / kOctChunk;
		const int o = chunk_off[c] + (g - group_off[c]) * per;
		const int ind = grp_index[g], mlt = grp_mult[g];   // chunk record: {first target particle, first desc, end desc, targets}
		for (int q = 0; q < per; ++q) chunk[o + q] = make_int4(ind, d0 + q * kOctChunk, min(d0 + (q + 1) * kOctChunk, d1), mlt);
	}
}

// ---- L2P + near field + rescale: one thread per particle in cell order (fmm_cart3_symmetric.cuh:362-385) ---------
template <int P, typename T>
__global__ __launch_bounds__(kBlock) void oct_l2p_kernel(OctView<T> t, const float4 *__restrict__ pos, const uint32_t *__restrict__ keys,
                                                         const float4 *__restrict__ near, const int *__restrict__ group_off,
                                                         const int *__restrict__ desc_off, const int *__restrict__ chunk_off, int tpl,
                                                         int have_near, const float *__restrict__ param, float *__restrict__ a_out,
                                                         const int *__restrict__ slab)
{
	constexpr int offL = NBCO_OFFL(P);
	const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
	if (i >= t.n || (slab && (i < slab[4] || i >= slab[5]))) return;
	const int c = (int)keys[i], leaf = oct_beg(t.L) + c;
	const float4 p = pos[i], cc = t.csz[leaf];
	T Lp[offL];
#pragma unroll
	for (int q = 0; q < offL; ++q) Lp[q] = t.local[(size_t)leaf * offL + q];
	T fx, fy, fz;
	l2p_body<P>(Lp, (T)p.x - (T)cc.x, (T)p.y - (T)cc.y, (T)p.z - (T)cc.z, fx, fy, fz);
	if (have_near)
	{
		const int rel = (int)(i - t.index[leaf]), g = rel / tpl, j = rel - g * tpl;
		const int per = (desc_off[c + 1] - desc_off[c] + kOctChunk - 1) / kOctChunk;
		const int ck0 = chunk_off[c] + g * per;
		float nx = 0.f, ny = 0.f, nz = 0.f;
#pragma unroll 8
		for (int ck = ck0; ck < ck0 + per; ++ck)
		{
			const float4 nr = near[(size_t)ck * tpl + j];
			nx += nr.x; ny += nr.y; nz += nr.z;
		}
		fx += (T)nx; fy += (T)ny; fz += (T)nz;
	}
	const T scale = param ? (T)param[0] : T(1);
	a_out[3 * i] = (float)(fx * scale); a_out[3 * i + 1] = (float)(fy * scale); a_out[3 * i + 2] = (float)(fz * scale);
}

static int oct_levels(long long n, int p, float dens_inhom)   // fmm_cart3_traceless.cuh:304 / :452-455
{
	const float s = (float)(p * p);
	const int L = (int)std::ceil(std::log2(dens_inhom * (float)n / s) / 3);
	return std::max(L, 2);
}

static int scan_ints(nbco_ctx *c, int *in, int *out, size_t count)
{
	size_t bytes = 0;
	NBCO_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, count, rocprim::plus<int>(), c->stream));
	NBCO_TRY(c->reserve(c->sort_tmp, bytes));
	bytes = c->sort_tmp.bytes;
	NBCO_HIP(rocprim::exclusive_scan(c->sort_tmp.ptr, bytes, in, out, 0, count, rocprim::plus<int>(), c->stream));
	return NBCO_OK;
}

static int m2l_lanes(nbco_ctx *c, int P, const float4 *csz, const float *mpole, float *local, const uint64_t *keys, const int *start, int shift, int ntot, int mstride)
{
	return launch_m2l_lanes(c, P, csz, mpole, local, keys, start, shift, ntot, mstride);
}
static int m2l_lanes(nbco_ctx *c, int P, const float4 *csz, const double *mpole, double *local, const uint64_t *keys, const int *start, int shift, int ntot, int mstride)
{
	return launch_m2l_lanes_f64(c, P, csz, mpole, local, keys, start, shift, ntot, mstride);
}

// world > 1: this rank's slab of the cell order only (oct_slab_kernel); pbounds_host[world + 1] = the particle boundaries of all slabs
template <int P, typename T, bool SYM>
static int oct_eval(nbco_ctx *c, float *p, float *a, long long n, const float *param, int world = 1, int rank = 0, long long *pbounds_host = nullptr)
{
	// SYM: fmm_cart3 (fmm_cart3_symmetric.cuh:413-580) -- multipole tuples in the symmetric layout, orders 0..P (offMP reals);
	// the M2L kernel reads their first offM reals (orders 0..P-1: an order-P multipole only meets the order-0 local, which is not
	// formed, m2l_acc3 with minm = 1), everything from the M2L list on is shared with the traceless evaluator
	constexpr int offL = NBCO_OFFL(P), offM = P * (P + 1) * (P + 2) / 6;
	constexpr int offMP = SYM ? NBCO_OFFM(P + 1) : offL;   // reals per stored multipole tuple
	hipStream_t st = c->stream;
	const int radius = (int)c->o.tree_radius;   // fmm_cart3_traceless.cuh:439
	if (radius < 1) return c->fail(NBCO_ERR_ARG, "nbco_fmm_traceless: tree_radius must be >= 1");
	const int L = oct_levels(n, P, c->o.dens_inhom);
	if (L > 8) return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_fmm_traceless: more than 8 octree levels");
	const int side = 1 << L, m = oct_cnt(L), beg = oct_beg(L), ntot = oct_beg(L + 1);
	const int first = oct_beg(2);   // levels 0 and 1 carry no expansions

	// ---- storage -----------------------------------------------------------------------------------------
	{
		size_t bytes = (size_t)ntot * (sizeof(float4) + sizeof(T) * ((size_t)offL + offMP + offM) + 2 * sizeof(int)) + 256;
		NBCO_TRY(c->reserve(c->oct_tree, bytes));
	}
	OctView<T> t;
	{
		char *q = (char *)c->oct_tree.ptr;
		t.csz = (float4 *)q; q += sizeof(float4) * (size_t)ntot;
		t.mpole = (T *)q; q += sizeof(T) * (size_t)ntot * offMP;
		t.local = (T *)q; q += sizeof(T) * (size_t)ntot * offL;
		t.msym = (T *)q; q += sizeof(T) * (size_t)ntot * offM;
		t.mult = (int *)q; q += sizeof(int) * (size_t)ntot;
		t.index = (int *)q;   // ntot + 1 entries
		t.L = L; t.ntot = ntot; t.side = side; t.n = n;
	}
	NBCO_TRY(c->reserve(c->pos4, sizeof(float4) * (size_t)n));
	NBCO_TRY(c->reserve(c->pos4_alt, sizeof(float4) * (size_t)n));
	NBCO_TRY(c->reserve(c->keys, sizeof(uint64_t) * (size_t)n));       // uint32 keys in / out share this buffer's halves
	NBCO_TRY(c->reserve(c->idx, sizeof(uint32_t) * (size_t)n));
	NBCO_TRY(c->reserve(c->idx_alt, sizeof(uint32_t) * (size_t)n));
	NBCO_TRY(c->reserve(c->counters, sizeof(int) * 128));
	// per-cell work counters: [ngroup | ndesc | nchunk | group_off | desc_off | chunk_off], m + 1 entries each; M2L: cnt, start
	NBCO_TRY(c->reserve(c->list_cnt, sizeof(int) * (6 * ((size_t)m + 1) + 2 * ((size_t)ntot + 2))));
	int *ngroup = c->list_cnt.as<int>(), *ndesc = ngroup + (m + 1), *nchunk = ndesc + (m + 1);
	int *group_off = nchunk + (m + 1), *desc_off = group_off + (m + 1), *chunk_off = desc_off + (m + 1);
	int *m2l_cnt = chunk_off + (m + 1), *m2l_start = m2l_cnt + (ntot + 2);

	int *slab = nullptr;            // device: this rank's slab, null = the whole cell order
	long long *pbounds = nullptr;   // device: particle boundaries of all slabs
	if (world > 64 || world < 1 || rank < 0 || rank >= world) return c->fail(NBCO_ERR_ARG, "nbco_fmm_oct_shard: 1 <= world <= 64, 0 <= rank < world");
	float4 *pos_in = c->pos4_alt.as<float4>(), *pos = c->pos4.as<float4>();
	uint32_t *keys_in = c->keys.as<uint32_t>(), *keys = keys_in + n;
	uint32_t *idx_in = c->idx.as<uint32_t>(), *idx = c->idx_alt.as<uint32_t>();

	// ---- build: keys, sort, cell ranges ----------------------------------------------------------------------
	{
		PhaseScope ph(c, NBCO_PH_BUILD);
		NBCO_TRY(launch_pack4(c, pos_in, p, n));
		float *mm = c->small.as<float>() + 64, *scal = c->small.as<float>() + 96;
		NBCO_TRY(launch_minmax4(c, pos_in, n, mm));
		hipLaunchKernelGGL(oct_scalars_kernel, dim3(1), dim3(64), 0, st, (const float *)mm, side, std::sqrt(c->o.eps2), scal);
		hipLaunchKernelGGL(oct_keys_kernel, dim3(grid1d(n)), dim3(kBlock), 0, st, (const float4 *)pos_in, n, (const float *)scal, side, keys_in, idx_in);
		size_t bytes = 0;
		NBCO_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys, idx_in, idx, (size_t)n, 0u, (unsigned)(3 * L), st));
		NBCO_TRY(c->reserve(c->sort_tmp, bytes));
		bytes = c->sort_tmp.bytes;
		NBCO_HIP(rocprim::radix_sort_pairs(c->sort_tmp.ptr, bytes, keys_in, keys, idx_in, idx, (size_t)n, 0u, (unsigned)(3 * L), st));
		hipLaunchKernelGGL(oct_gather4_kernel, dim3(grid1d(n)), dim3(kBlock), 0, st, (const float4 *)pos_in, (const uint32_t *)idx, pos, n);
		hipLaunchKernelGGL(oct_index_kernel<T>, dim3(grid1d(m + 1)), dim3(kBlock), 0, st, t, (const uint32_t *)keys);
		if (world > 1)
		{
			slab = c->counters.as<int>() + 64;
			pbounds = reinterpret_cast<long long *>(c->small.as<float>() + 128);
			hipLaunchKernelGGL(oct_slab_kernel<T>, dim3(1), dim3(128), 0, st, t, world, rank, slab, pbounds);
		}
		NBCO_HIP(hipGetLastError());
	}
	// ---- P2M, M2M ----------------------------------------------------------------------------------------
	{
		PhaseScope ph(c, NBCO_PH_P2M_M2M);
		hipLaunchKernelGGL((oct_leaf_kernel<P, T, SYM>), dim3((m + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, t, (const float4 *)pos);
		for (int l = L - 1; l >= 2; --l) hipLaunchKernelGGL((oct_m2m_kernel<P, T, SYM>), dim3((oct_cnt(l) + kBlock - 1) / kBlock), dim3(kBlock), 0, st, t, l);
		if (!SYM) hipLaunchKernelGGL(oct_expand_kernel<T>, dim3(grid1d(ntot - first)), dim3(kBlock), 0, st, t, P, first);
		NBCO_HIP(hipGetLastError());
	}
	// ---- work lists: M2L stencil entries, P2P groups / descriptors / chunks -----------------------------------------
	int h_tot[4] = {0, 0, 0, 0};   // M2L entries, groups, descriptors, chunks
	const int tpl = [&] {
		const double avg = (double)n / (double)m;
		int v = 8;
		while (v < 64 && v < 2 * avg) v <<= 1;
		return v;
	}();
	int shift = 1;
	while ((1LL << shift) < ntot) ++shift;
	{
		PhaseScope ph(c, NBCO_PH_LISTS);
		hipLaunchKernelGGL((oct_m2l_list_kernel<false, T>), dim3(grid1d(ntot)), dim3(kBlock), 0, st, t, radius, first, m2l_cnt, (const int *)nullptr, shift,
		                   (uint64_t *)nullptr, (const int *)slab);
		NBCO_TRY(scan_ints(c, m2l_cnt, m2l_start, (size_t)ntot + 1));
		if (c->o.coll)
		{
			hipLaunchKernelGGL(oct_p2p_count_kernel<T>, dim3(grid1d(m + 1)), dim3(kBlock), 0, st, t, radius, tpl, ngroup, ndesc, nchunk, (const int *)slab);
			NBCO_TRY(scan_ints(c, ngroup, group_off, (size_t)m + 1));
			NBCO_TRY(scan_ints(c, ndesc, desc_off, (size_t)m + 1));
			NBCO_TRY(scan_ints(c, nchunk, chunk_off, (size_t)m + 1));
		}
		NBCO_HIP(hipMemcpyAsync(&h_tot[0], m2l_start + ntot, sizeof(int), hipMemcpyDeviceToHost, st));
		if (c->o.coll)
		{
			NBCO_HIP(hipMemcpyAsync(&h_tot[1], group_off + m, sizeof(int), hipMemcpyDeviceToHost, st));
			NBCO_HIP(hipMemcpyAsync(&h_tot[2], desc_off + m, sizeof(int), hipMemcpyDeviceToHost, st));
			NBCO_HIP(hipMemcpyAsync(&h_tot[3], chunk_off + m, sizeof(int), hipMemcpyDeviceToHost, st));
		}
		if (pbounds_host && world > 1) NBCO_HIP(hipMemcpyAsync(pbounds_host, pbounds, sizeof(long long) * (size_t)(world + 1), hipMemcpyDeviceToHost, st));
		NBCO_HIP(hipStreamSynchronize(st));   // the one host round trip of the evaluation: sizes of the work lists
		const long long nm2l = h_tot[0], ngr = h_tot[1], nds = h_tot[2], nck = h_tot[3];
		if (nm2l < 0 || nck < 0) return c->fail(NBCO_ERR_CAPACITY, "nbco_fmm_traceless: work list size overflows 32 bits");
		NBCO_TRY(c->reserve(c->m2l_keys_alt, sizeof(uint64_t) * (size_t)(nm2l + 1)));
		if (nm2l > 0)
			hipLaunchKernelGGL((oct_m2l_list_kernel<true, T>), dim3(grid1d(ntot)), dim3(kBlock), 0, st, t, radius, first, (int *)nullptr, (const int *)m2l_start,
			                   shift, c->m2l_keys_alt.as<uint64_t>(), (const int *)slab);
		if (c->o.coll)
		{
			NBCO_TRY(c->reserve(c->oct_groups, sizeof(int) * 3 * (size_t)(ngr + 1)));
			NBCO_TRY(c->reserve(c->p2p_keys, sizeof(int2) * (size_t)(nds + 1)));
			NBCO_TRY(c->reserve(c->p2p_chunks, sizeof(int4) * (size_t)(nck + 1)));
			NBCO_TRY(c->reserve(c->part, sizeof(float4) * (size_t)nck * (size_t)tpl + 256));
			int *grp_index = c->oct_groups.as<int>(), *grp_mult = grp_index + (ngr + 1), *grp_cell = grp_mult + (ngr + 1);
			hipLaunchKernelGGL(oct_p2p_fill_kernel<T>, dim3(grid1d(m)), dim3(kBlock), 0, st, t, radius, tpl, (const int *)group_off, (const int *)desc_off,
			                   grp_index, grp_mult, grp_cell, c->p2p_keys.as<int2>(), (const int *)slab);
			hipLaunchKernelGGL(oct_p2p_chunk_kernel, dim3(grid1d(std::max<long long>(ngr, 1))), dim3(kBlock), 0, st, (const int *)(group_off + m),
			                   (const int *)grp_cell, (const int *)group_off, (const int *)desc_off, (const int *)chunk_off, (const int *)grp_index,
			                   (const int *)grp_mult, c->p2p_chunks.as<int4>());
		}
		NBCO_HIP(hipGetLastError());
	}
	const long long nm2l = h_tot[0], ngr = h_tot[1], nck = h_tot[3];
	// ---- P2P -----------------------------------------------------------------------------------------------
	float4 *near = c->part.as<float4>();
	const bool have_near = c->o.coll && nck > 0;
	if (have_near)
	{
		PhaseScope ph(c, NBCO_PH_P2P);
		const int2 *pd = c->p2p_keys.as<int2>();
		const int4 *pc = c->p2p_chunks.as<int4>();
		const int *pt = chunk_off + m;
		if (tpl == 8) launch_p2p<8>(c, pos, pd, pc, pt, nck, kSrcPiece, tpl, near, n);
		else if (tpl == 16) launch_p2p<16>(c, pos, pd, pc, pt, nck, kSrcPiece, tpl, near, n);
		else if (tpl == 32) launch_p2p<32>(c, pos, pd, pc, pt, nck, kSrcPiece, tpl, near, n);
		else launch_p2p<64>(c, pos, pd, pc, pt, nck, kSrcPiece, tpl, near, n);
		NBCO_HIP(hipGetLastError());
	}
	// ---- M2L, L2L ------------------------------------------------------------------------------------------
	{
		PhaseScope ph(c, NBCO_PH_M2L);
		NBCO_HIP(hipMemsetAsync(t.local, 0, sizeof(T) * (size_t)ntot * offL, st));
		if (nm2l > 0) NBCO_TRY(m2l_lanes(c, P, t.csz, SYM ? t.mpole : t.msym, t.local, c->m2l_keys_alt.as<uint64_t>(), m2l_start, shift, ntot, SYM ? offMP : 0));
	}
	{
		PhaseScope ph(c, NBCO_PH_L2L);
		for (int lc = 3; lc <= L; ++lc)
			hipLaunchKernelGGL((oct_l2l_kernel<P, T>), dim3((oct_cnt(lc) + kBlock - 1) / kBlock), dim3(kBlock), 0, st, t, lc, (const int *)slab);
		NBCO_HIP(hipGetLastError());
	}
	// ---- L2P + near field + rescale -------------------------------------------------------------------------
	{
		PhaseScope ph(c, NBCO_PH_L2P);
		hipLaunchKernelGGL((oct_l2p_kernel<P, T>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, t, (const float4 *)pos, (const uint32_t *)keys,
		                   (const float4 *)near, (const int *)group_off, (const int *)desc_off, (const int *)chunk_off, tpl, have_near ? 1 : 0, param, a,
		                   (const int *)slab);
		NBCO_HIP(hipGetLastError());
	}
	// ---- positions and velocities in cell order (fmm_cart3_traceless.cuh:386-392, :530-535) ------------------------
	{
		PhaseScope ph(c, NBCO_PH_FINISH);
		hipLaunchKernelGGL(oct_unpack4_kernel, dim3(grid1d(n)), dim3(kBlock), 0, st, (const float4 *)pos, p, n);
		NBCO_TRY(c->reserve(c->tmp3, sizeof(float) * 3 * (size_t)n));
		NBCO_TRY(launch_gather3(c, c->tmp3.as<float>(), p + 3 * n, (const int *)idx, n, false));
		NBCO_HIP(hipMemcpyAsync(p + 3 * n, c->tmp3.ptr, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToDevice, st));
		NBCO_HIP(hipGetLastError());
	}
	if (pbounds_host && world == 1) { pbounds_host[0] = 0; pbounds_host[1] = n; }
	OctTreeDev &o = c->oct;
	o.L = L; o.ntot = ntot; o.order = P; o.n = n; o.csz = t.csz; o.mpole = t.mpole; o.local = t.local; o.mult = t.mult; o.index = t.index;
	o.keys = keys; o.perm = idx; o.m2l_entries = nm2l; o.p2p_groups = ngr; o.p2p_desc = h_tot[2]; o.p2p_chunks = nck; o.tpl = tpl;
	o.real_bytes = (int)sizeof(T);
	o.mpole_reals = offMP;
	o.valid = true;
	return NBCO_OK;
}

} // namespace

int fmm_oct_traceless_eval(nbco_ctx *c, float *p, float *a, long long n, const float *param, bool symmetric, int world, int rank, long long *pbounds_host)
{
	if (n <= 0) return c->fail(NBCO_ERR_ARG, "nbco_fmm_traceless: n must be positive");
	if (n > 0x7fffffffLL / 4) return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_fmm_traceless: n too large for 32-bit indices");
	const bool f64 = c->o.far_fp64 != 0;
	if (symmetric)
	{
		// symmetric multipoles of orders 0..p come from the generated operators of order p + 1: p <= 9
#define NBCO_OCT_CASE(PP) case PP: return f64 ? oct_eval<PP, double, true>(c, p, a, n, param, world, rank, pbounds_host) : oct_eval<PP, float, true>(c, p, a, n, param, world, rank, pbounds_host);
		switch (c->o.fmm_order)
		{
		NBCO_OCT_CASE(1) NBCO_OCT_CASE(2) NBCO_OCT_CASE(3) NBCO_OCT_CASE(4) NBCO_OCT_CASE(5)
		NBCO_OCT_CASE(6) NBCO_OCT_CASE(7) NBCO_OCT_CASE(8) NBCO_OCT_CASE(9)
		default: return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_fmm_symmetric: orders 1..9");
		}
#undef NBCO_OCT_CASE
	}
#define NBCO_OCT_CASE(PP) case PP: return f64 ? oct_eval<PP, double, false>(c, p, a, n, param, world, rank, pbounds_host) : oct_eval<PP, float, false>(c, p, a, n, param, world, rank, pbounds_host);
	switch (c->o.fmm_order)
	{
	NBCO_OCT_CASE(1) NBCO_OCT_CASE(2) NBCO_OCT_CASE(3) NBCO_OCT_CASE(4) NBCO_OCT_CASE(5)
	NBCO_OCT_CASE(6) NBCO_OCT_CASE(7) NBCO_OCT_CASE(8) NBCO_OCT_CASE(9) NBCO_OCT_CASE(10)
	default: return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_fmm_traceless: generated operators exist for orders 1..10");
	}
#undef NBCO_OCT_CASE
}

int oct_copy_out(nbco_ctx *c, int which, void *dst, long long bytes)
{
	const OctTreeDev &o = c->oct;
	if (!o.valid) return c->fail(NBCO_ERR_ARG, "nbco_oct_copy: no octree evaluation has run");
	const size_t offL = (size_t)(o.order + 1) * (o.order + 1);
	const void *src = nullptr;
	size_t need = 0;
	switch (which)
	{
	case NBCO_OCT_MULT: src = o.mult; need = 4 * (size_t)o.ntot; break;
	case NBCO_OCT_INDEX: src = o.index; need = 4 * (size_t)o.ntot; break;
	case NBCO_OCT_CENTER4: src = o.csz; need = 16 * (size_t)o.ntot; break;
	case NBCO_OCT_MPOLE: src = o.mpole; need = (size_t)o.real_bytes * (size_t)o.ntot * (size_t)o.mpole_reals; break;
	case NBCO_OCT_LOCAL: src = o.local; need = (size_t)o.real_bytes * (size_t)o.ntot * offL; break;
	case NBCO_OCT_KEYS: src = o.keys; need = 4 * (size_t)o.n; break;
	case NBCO_OCT_PERM: src = o.perm; need = 4 * (size_t)o.n; break;
	default: return c->fail(NBCO_ERR_ARG, "nbco_oct_copy: unknown array");
	}
	if ((long long)need > bytes) return c->fail(NBCO_ERR_ARG, "nbco_oct_copy: destination too small");
	NBCO_HIP(hipStreamSynchronize(c->stream));
	if (need) NBCO_HIP(hipMemcpy(dst, src, need, hipMemcpyDeviceToHost));
	return NBCO_OK;
}

NBCO_CHECKED_COLLECT(nbco_checked_collect_oct)
