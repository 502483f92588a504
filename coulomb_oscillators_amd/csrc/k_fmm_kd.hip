// k_fmm_kd.hip -- 3-D cartesian-tensor FMM on a balanced kd-tree with dual tree traversal.
// Reference behaviour: fmm_cart3_kdtree.cuh (driver :1478-1771 GPU, :1773-1929 CPU), operators of
// fmm_cart_base3.cuh, leaf helpers of appel.cuh.  Default semantics follow the reference's CPU
// driver: the tree is rebuilt every evaluation (opts.tree_steps = 1) and leaf-leaf
// pairs go to P2P before the admissibility test (opts.m2l_first = 0).
//
// gfx950 design (one evaluation; DESIGN.md section 4 has the measured kernel table):
//   build     positions packed to float4.  Levels whose nodes exceed 4096 particles: exact radix selection of the
//             median + unordered partition + tie resolver (k_kdselect.hip), no sort.  The rest of every subtree is
//             built by one workgroup in LDS (kd_subtree_kernel), which first restores the order the reference's chain
//             of stable per-level sorts (:167-202) would have produced, so the result is bit-identical to the oracle's.
//             Fallback after a tie overflow: one stable radix sort of the composite key
//             (node << 32 | order-preserving float bits) per level.  Geometry that feeds admissibility decisions
//             (leaf centroids, parent centres, box diagonals, distances) is evaluated without FMA contraction so
//             that the interaction lists equal the oracle's bit for bit.
//   P2M/M2M   generated straight-line bodies, one thread per leaf / node (k_farfield.hip); orders 9-10: one wave per
//             node, one lane per multipole component, flattened term tables (kernels below).
//   traverse  level-synchronous expansion of the pair frontier (no recursion, no per-block stacks), two tree levels
//             per launch; list slots are reserved with one packed block scan and three atomics per block.
//   lists     counting sort of the directed (target, source) entries by target + per-target rank sort: every target
//             node / leaf gets a contiguous, deterministic source list, so P2P and M2L run without float atomics and
//             are bit-reproducible.
//   P2P       per-target lists cut into chunks of <= 16 source leaves, one wave per chunk (k_p2p.hpp).
//   M2L/L2L   register-resident generated bodies (k_m2l.hip, k_farfield.hip); orders 9-10: table kernels below with
//             dimensionless gradient tensors (no fp32 overflow for p = 10).
//   L2P       one thread per particle; fused with the near-field sum, the final rescale by param[0] and the scatter
//             back to the caller's order.
//   multi-GPU kd-domain sharding: the same two stages (kd_build_upward on the own subtree, kd_interact on the
//             assembled global tree, pruned to the own domain), see the section at the end of this file.
#include "nbco_internal.hpp"
#include "k_p2p.hpp"
#include "kd_common.hpp"
#include <rocprim/rocprim.hpp>
#include <chrono>
#include <functional>
#include <cmath>
#include <algorithm>

namespace {

using kdc::wave_scan_add;
using kdc::wave_min_u32;

// tuple offsets (fmm_cart_base3.cuh:180-188): symmetric orders 0..n-1 hold n(n+1)(n+2)/6 reals, traceless orders 0..n-1 hold n^2
constexpr int sym_off(int n) { return n * (n + 1) * (n + 2) / 6; }
constexpr int tl_off(int n) { return n * n; }

struct TreeView
{
	float *center, *lbound, *rbound;
	float4 *csz;
	float *mpole, *local;
	int *mult, *index, *splitdim;
	int L, ntot;
};

constexpr int kBlock = 256;

__host__ __device__ inline int kd_beg(int l) { return (1 << l) - 1; }
__host__ __device__ inline int kd_cnt(int l) { return 1 << l; }

// kd-domain of a multi-GPU run: this GPU owns the subtree of node 2^d - 1 + g of the global
// tree.  A node "touches" the domain when it lies in that subtree or on the path from its root to the
// global root; d = 0 is the single-GPU case, where every node touches.
struct Dom
{
	int d, g;
};
__host__ __device__ inline bool dom_touch(const Dom dm, int node)
{
#ifdef __HIP_DEVICE_COMPILE__
	const int l = 31 - __clz(node + 1);
#else
	int l = 0;
	while ((2 << l) <= node + 1) ++l;
#endif
	const int pos = node - ((1 << l) - 1);
	return l >= dm.d ? (pos >> (l - dm.d)) == dm.g : (dm.g >> (dm.d - l)) == pos;
}


// ---- tree geometry: every function below must round exactly like the oracle ---------------------
#pragma clang fp contract(off)

__device__ inline int longest_axis(float dx, float dy, float dz)   // fmm_cart3_kdtree.cuh:92,129
{
	return (dx > dy) ? ((dx > dz) ? 0 : 2) : ((dy > dz) ? 1 : 2);
}

// order-preserving 32-bit image of a float (fmm_cart3_kdtree.cuh:175-185)
__device__ inline uint32_t ordered_bits(float f)
{
	uint32_t u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float unordered_bits(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

__global__ void kd_root_kernel(TreeView t, const float *__restrict__ minmax6)   // fmm_cart3_kdtree.cuh:89-97
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	float lx = minmax6[0], ly = minmax6[1], lz = minmax6[2], rx = minmax6[3], ry = minmax6[4], rz = minmax6[5];
	t.lbound[0] = lx; t.lbound[1] = ly; t.lbound[2] = lz;
	t.rbound[0] = rx; t.rbound[1] = ry; t.rbound[2] = rz;
	t.splitdim[0] = longest_axis(rx - lx, ry - ly, rz - lz);
	t.index[0] = 0;
}

// ---- build prologue, one launch ------------------------------------------------------------------------------
// Packs the caller's xyz triplets into float4, writes the identity permutation, clears the selection build's histograms
// and node states, and reduces the bounding box: per-workgroup min / max go into six ordered-bit words with device
// atomics, and the LAST workgroup to finish writes the root node (evalRootBox, fmm_cart3_kdtree.cuh:89-107) and
// re-arms the accumulators for the next build.  Replaces eight small launches.  Atomics on one address retire at ~27 ns
// each: the grid is kept to kPrepGrid large workgroups.
// state: [0..2] min as ordered bits (armed 0xFFFFFFFF), [3..5] max (armed 0), [6] workgroups done (armed 0)
constexpr int kPrepBlock = 1024, kPrepGrid = 128;
__global__ __launch_bounds__(kPrepBlock) void kd_prep_kernel(const float *__restrict__ p3, long long n, float4 *__restrict__ pos,
                                                             int *__restrict__ unsort, uint32_t *__restrict__ zero_a, long long words_a,
                                                             uint32_t *__restrict__ zero_b, long long words_b, int *__restrict__ flag,
                                                             unsigned *__restrict__ state, TreeView t, const float *__restrict__ root6)
{
	float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
	const long long stride = (long long)gridDim.x * kPrepBlock, tid = (long long)blockIdx.x * kPrepBlock + threadIdx.x;
	for (long long i = tid; i < n; i += stride)
	{
		const float x = p3[3 * i], y = p3[3 * i + 1], z = p3[3 * i + 2];
		pos[i] = make_float4(x, y, z, 0.f);
		unsort[i] = (int)i;
		mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
		mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
	}
	for (long long i = tid; i < words_a; i += stride) zero_a[i] = 0u;
	for (long long i = tid; i < words_b; i += stride) zero_b[i] = 0u;
	if (tid == 0) *flag = 0;
	__shared__ float sh[kPrepBlock / 64][6];
	__shared__ unsigned last;
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
	for (int c = 0; c < 3; ++c)
		for (int o = 32; o > 0; o >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o)); }
	if (lane == 0)
#pragma unroll
		for (int c = 0; c < 3; ++c) { sh[w][c] = mn[c]; sh[w][3 + c] = mx[c]; }
	__syncthreads();
	if (threadIdx.x < 6)
	{
		float v = sh[0][threadIdx.x];
		for (int k = 1; k < kPrepBlock / 64; ++k) v = threadIdx.x < 3 ? fminf(v, sh[k][threadIdx.x]) : fmaxf(v, sh[k][threadIdx.x]);
		if (threadIdx.x < 3) atomicMin(&state[threadIdx.x], ordered_bits(v));
		else atomicMax(&state[threadIdx.x], ordered_bits(v));
	}
	// the accumulators are only touched by device atomics and agent-scope loads: completion of this workgroup's atomics is all
	// the ordering the counter needs
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) last = atomicAdd(&state[6], 1u) == gridDim.x - 1 ? 1u : 0u;
	__syncthreads();
	if (!last || threadIdx.x != 0) return;
	float b[6];
	for (int c = 0; c < 6; ++c)
	{
		b[c] = unordered_bits(__hip_atomic_load(&state[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		__hip_atomic_store(&state[c], c < 3 ? 0xFFFFFFFFu : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__hip_atomic_store(&state[6], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (root6)   // a kd-domain keeps the union of its inherited box and the current bounds (see kd_build_upward)
		for (int c = 0; c < 3; ++c) { b[c] = fminf(b[c], root6[c]); b[3 + c] = fmaxf(b[3 + c], root6[3 + c]); }
	t.lbound[0] = b[0]; t.lbound[1] = b[1]; t.lbound[2] = b[2];
	t.rbound[0] = b[3]; t.rbound[1] = b[4]; t.rbound[2] = b[5];
	t.splitdim[0] = longest_axis(b[3] - b[0], b[4] - b[1], b[5] - b[2]);
	t.index[0] = 0;
}

// ---- between two leapfrog steps of nbco_integrate_steps: one pass instead of four ---------------------------------------------
// After the force evaluation of step s the state is: accelerations in tree order (a3), and -- when the tree was rebuilt --
// positions in tree order only as float4 (pos4) with the caller's arrays still in the order before the rebuild.  What follows
// in the step-by-step sequence is  tree order for x, v (reorder_state_kernel)  ->  a -= k o x, v += a ks (finish_kick_kernel)
// -> [step s + 1]  v += a ks, x += v ds (kick_drift_kernel)  ->  pack x, identity permutation, bounding box, root node,
// cleared selection state (kd_prep_kernel).  Here every particle goes through exactly those operations, in that order and
// with the same roundings, in registers.  GATHER: the evaluation rebuilt the tree (velocities come through `unsort`, v_out
// must not be v_in); PREP: 1 = the next evaluation rebuilds (full build prologue), 0 = it reuses the tree (positions packed only).
template <bool GATHER>
__global__ __launch_bounds__(kPrepBlock) void kd_turnaround_kernel(float4 *__restrict__ pos4, int *__restrict__ unsort, float *__restrict__ x3,
                                                                   const float *v_in, float *v_out, const float *__restrict__ a3,
                                                                   const float *__restrict__ param, float ks, float ds, int elastic, long long n, int prep,
                                                                   uint32_t *__restrict__ zero_a, long long words_a, uint32_t *__restrict__ zero_b,
                                                                   long long words_b, int *__restrict__ flag, unsigned *__restrict__ state, TreeView t,
                                                                   const float *__restrict__ root6)
{
	float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
	const long long stride = (long long)gridDim.x * kPrepBlock, tid = (long long)blockIdx.x * kPrepBlock + threadIdx.x;
	const float k3[3] = {param[3], param[4], param[5]};
	// four particles per thread and iteration, loads first: the velocity gather is a chain of two dependent scattered reads, and
	// 128 workgroups (few, because of the bounding-box atomics below) do not hide that latency with occupancy alone
	constexpr int U = 4;
	for (long long i0 = tid; i0 < n; i0 += U * stride)
	{
		float x[U][3], a[U][3], v[U][3];
		long long src[U];
#pragma unroll
		for (int u = 0; u < U; ++u)
		{
			const long long i = i0 + u * stride;
			src[u] = i;
			if (i >= n) continue;
			if (GATHER)
			{
				const float4 q = pos4[i];
				x[u][0] = q.x; x[u][1] = q.y; x[u][2] = q.z;
				src[u] = unsort[i];
			}
			else { x[u][0] = x3[3 * i]; x[u][1] = x3[3 * i + 1]; x[u][2] = x3[3 * i + 2]; }
			a[u][0] = a3[3 * i]; a[u][1] = a3[3 * i + 1]; a[u][2] = a3[3 * i + 2];
		}
#pragma unroll
		for (int u = 0; u < U; ++u)
		{
			if (i0 + u * stride >= n) continue;
			v[u][0] = v_in[3 * src[u]]; v[u][1] = v_in[3 * src[u] + 1]; v[u][2] = v_in[3 * src[u] + 2];
		}
#pragma unroll
		for (int u = 0; u < U; ++u)
		{
			const long long i = i0 + u * stride;
			if (i >= n) continue;
#pragma unroll
			for (int c = 0; c < 3; ++c)
			{
				float ai = a[u][c];                                        // finish_kick_kernel (no rescale: the evaluator did it)
				if (elastic) ai = fmaf(-k3[c], x[u][c], ai);
				float vi = fmaf(ks, ai, v[u][c]);
				vi = fmaf(ks, ai, vi);                                     // kick_drift_kernel
				x[u][c] = fmaf(ds, vi, x[u][c]);
				v[u][c] = vi;
			}
			x3[3 * i] = x[u][0]; x3[3 * i + 1] = x[u][1]; x3[3 * i + 2] = x[u][2];
			v_out[3 * i] = v[u][0]; v_out[3 * i + 1] = v[u][1]; v_out[3 * i + 2] = v[u][2];
			pos4[i] = make_float4(x[u][0], x[u][1], x[u][2], 0.f);          // kd_prep_kernel / pack4
			if (prep) unsort[i] = (int)i;
#pragma unroll
			for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], x[u][c]); mx[c] = fmaxf(mx[c], x[u][c]); }
		}
	}
	if (tid == 0) *flag = 0;
	if (!prep) return;
	for (long long i = tid; i < words_a; i += stride) zero_a[i] = 0u;
	for (long long i = tid; i < words_b; i += stride) zero_b[i] = 0u;
	__shared__ float sh[kPrepBlock / 64][6];
	__shared__ unsigned last;
	const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
	for (int c = 0; c < 3; ++c)
		for (int o = 32; o > 0; o >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o)); }
	if (lane == 0)
#pragma unroll
		for (int c = 0; c < 3; ++c) { sh[w][c] = mn[c]; sh[w][3 + c] = mx[c]; }
	__syncthreads();
	if (threadIdx.x < 6)
	{
		float v = sh[0][threadIdx.x];
		for (int k = 1; k < kPrepBlock / 64; ++k) v = threadIdx.x < 3 ? fminf(v, sh[k][threadIdx.x]) : fmaxf(v, sh[k][threadIdx.x]);
		if (threadIdx.x < 3) atomicMin(&state[threadIdx.x], ordered_bits(v));
		else atomicMax(&state[threadIdx.x], ordered_bits(v));
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) last = atomicAdd(&state[6], 1u) == gridDim.x - 1 ? 1u : 0u;
	__syncthreads();
	if (!last || threadIdx.x != 0) return;
	float b[6];
	for (int c = 0; c < 6; ++c)
	{
		b[c] = unordered_bits(__hip_atomic_load(&state[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
		__hip_atomic_store(&state[c], c < 3 ? 0xFFFFFFFFu : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__hip_atomic_store(&state[6], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (root6)   // a kd-domain keeps the union of its inherited box and the current bounds (as kd_prep_kernel does)
		for (int c = 0; c < 3; ++c) { b[c] = fminf(b[c], root6[c]); b[3 + c] = fmaxf(b[3 + c], root6[3 + c]); }
	t.lbound[0] = b[0]; t.lbound[1] = b[1]; t.lbound[2] = b[2];
	t.rbound[0] = b[3]; t.rbound[1] = b[4]; t.rbound[2] = b[5];
	t.splitdim[0] = longest_axis(b[3] - b[0], b[4] - b[1], b[5] - b[2]);
	t.index[0] = 0;
}

// composite keys of level l (fmm_cart3_kdtree.cuh:167-187): node j = floor(2^l i / n)
__global__ __launch_bounds__(kBlock) void kd_keys_kernel(const float4 *__restrict__ pos, const int *__restrict__ splitdim_l, long long n,
                                                         int l, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
	const long long m = 1LL << l;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		uint64_t j = (uint64_t)(m * i / n);
		float4 p = pos[i];
		int s = splitdim_l[j];
		float v = s == 0 ? p.x : (s == 1 ? p.y : p.z);
		keys[i] = (j << 32) | (uint64_t)ordered_bits(v);
		vals[i] = (uint32_t)i;
	}
}

__global__ __launch_bounds__(kBlock) void kd_permute_kernel(const float4 *__restrict__ pos_in, const int *__restrict__ unsort_in,
                                                            const uint32_t *__restrict__ vals, float4 *__restrict__ pos_out,
                                                            int *__restrict__ unsort_out, long long n)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		uint32_t s = vals[i];
		pos_out[i] = pos_in[s];
		unsort_out[i] = unsort_in[s];
	}
}

__global__ __launch_bounds__(kBlock) void iota_kernel(int *__restrict__ v, long long n)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) v[i] = (int)i;
}

// evalBox (fmm_cart3_kdtree.cuh:109-137): ranges ceil(n i / 2^l); bounds inherited from the parent and
// tightened along the parent's split dimension only
__global__ __launch_bounds__(kBlock) void kd_box_kernel(TreeView t, const float4 *__restrict__ pos, long long n, int l)
{
	const long long m = 1LL << l;
	const int beg = kd_beg(l);
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < m; i += (long long)gridDim.x * kBlock)
	{
		long long start = (i == 0) ? 0 : (n * i - 1) / m + 1;
		long long end = (n * (i + 1) - 1) / m + 1;
		int j = beg + (int)i, parent = (j - 1) >> 1, split = t.splitdim[parent];
		float lb[3] = {t.lbound[3 * parent], t.lbound[3 * parent + 1], t.lbound[3 * parent + 2]};
		float rb[3] = {t.rbound[3 * parent], t.rbound[3 * parent + 1], t.rbound[3 * parent + 2]};
		if (j == 2 * parent + 2)
		{
			float4 q = pos[start];
			float v = split == 0 ? q.x : (split == 1 ? q.y : q.z);
			if (split == 0) lb[0] = v; else if (split == 1) lb[1] = v; else lb[2] = v;
		}
		else
		{
			float4 q = pos[end - 1];
			float v = split == 0 ? q.x : (split == 1 ? q.y : q.z);
			if (split == 0) rb[0] = v; else if (split == 1) rb[1] = v; else rb[2] = v;
		}
		t.lbound[3 * j] = lb[0]; t.lbound[3 * j + 1] = lb[1]; t.lbound[3 * j + 2] = lb[2];
		t.rbound[3 * j] = rb[0]; t.rbound[3 * j + 1] = rb[1]; t.rbound[3 * j + 2] = rb[2];
		t.splitdim[j] = longest_axis(rb[0] - lb[0], rb[1] - lb[1], rb[2] - lb[2]);
		t.index[j] = (int)start;
	}
}

// multLeaves + centerLeaves (appel.cuh:184-197, 226-243): sequential sum in particle order, then one
// division -- the rounding the oracle uses
__global__ __launch_bounds__(kBlock) void kd_leaf_kernel(TreeView t, const float4 *__restrict__ pos, long long n)
{
	const int m = kd_cnt(t.L), beg = kd_beg(t.L);
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock)
	{
		int ind = t.index[beg + i];
		int mlt = (i < m - 1) ? t.index[beg + i + 1] - ind : (int)n - ind;
		float sx = 0.f, sy = 0.f, sz = 0.f;
		for (int j = 0; j < mlt; ++j)
		{
			float4 q = pos[ind + j];
			sx = sx + q.x; sy = sy + q.y; sz = sz + q.z;
		}
		if (mlt > 0) { float d = (float)mlt; sx = sx / d; sy = sy / d; sz = sz / d; }
		t.mult[beg + i] = mlt;
		t.center[3 * (beg + i)] = sx; t.center[3 * (beg + i) + 1] = sy; t.center[3 * (beg + i) + 2] = sz;
	}
}

// centre of charge of a parent (fmm_cart3_kdtree.cuh:339-348)
__device__ inline void parent_centre(const TreeView &t, int k, float c[3], int &mlt)
{
	int c0 = 2 * k + 1, c1 = 2 * k + 2;
	int m0 = t.mult[c0], m1 = t.mult[c1];
	mlt = m0 + m1;
	float f0 = (float)m0, f1 = (float)m1, ft = (float)mlt;
	for (int a = 0; a < 3; ++a)
	{
		float s = f0 * t.center[3 * c0 + a];
		s = s + f1 * t.center[3 * c1 + a];
		c[a] = s / ft;
	}
}

struct AdmTab   // M = (max(mult1,mult2)/N)^(1/(3p+6)) evaluated on the host with libm powf per level
{
	int lo[32];
	float Mlo[32], Mhi[32];
};

// kd_admissible (fmm_cart3_kdtree.cuh:401-414)
__device__ inline bool kd_admissible(const float4 c1, const float4 c2, int n1, int n2, const int *__restrict__ mult, const AdmTab *tabp,
                                     float par)
{
	float dx = c2.x - c1.x, dy = c2.y - c1.y, dz = c2.z - c1.z;
	float dist2 = dx * dx + dy * dy + dz * dz;
	int m1 = mult[n1], m2 = mult[n2];
	int nb = m1 >= m2 ? n1 : n2, mb = m1 >= m2 ? m1 : m2;
	int lev = 31 - __clz(nb + 1);
	float M = (mb == tabp->lo[lev]) ? tabp->Mlo[lev] : tabp->Mhi[lev];
	float parM = par * M;
	float sz = fmaxf(c1.w, c2.w);
	return parM * parM * sz < dist2;
}

// ---- in-LDS subtree build ---------------------------------------------------------------------------
// Once a node holds at most kSubS particles the rest of its subtree is built by ONE workgroup without
// leaving the CU: positions and the cumulative permutation are loaded into LDS once, every remaining
// level is one bitonic sort of 64-bit composites [local node | ordered float key | current position]
// over the whole slice (all nodes of the level at once; the position field makes it the stable sort the
// oracle performs), followed by an in-place permutation through registers and evalBox for the children.
// HBM traffic: one read and one write of the slice instead of ~15 radix passes per level.
constexpr int kSubS = 4096;      // particles per subtree slice (LDS: 32 KB keys + 48 KB xyz + 16 KB permutation + 4 KB split dims)
constexpr int kSubT = 1024;      // threads per workgroup
constexpr int kSubE = kSubS / kSubT;
constexpr int kSelSeg = 32;      // segments up to this size are sorted in a wave's registers; larger ones are split by selection
constexpr int kSelNodes = kSubS / kSelSeg;   // nodes of the first level whose segments are kSelSeg long (128)
constexpr int kSubTieCap = 64;   // pivot ties resolved per node; more -> flag, the caller falls back to the sorting build
struct SubSel
{
	uint32_t prefix, minR, pivot;   // digits chosen so far; smallest key of the right part; the pivot's key
	int rank, neq, cntL, cntR, ntie;
};

__device__ inline float ld_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#ifdef NBCO_SUBTREE_PROF
// profiling build only (make prof): phase timestamps (100 MHz) of one workgroup of kd_subtree_kernel, tools/subtree_prof.py
__device__ long long g_subtree_prof[512];
#define SUBTREE_MARK(k) do { if (blockIdx.x == 37 && threadIdx.x == 0) g_subtree_prof[(k)] = wall_clock64(); } while (0)
extern "C" int nbco_debug_subtree_prof(long long *out512)
{
	return (int)hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_subtree_prof), sizeof(long long) * 512);
}
// ... and of one workgroup of every traversal launch (first pass of its loop)
__device__ long long g_trav_prof[36 * 12];
#define TRAV_MARK(k) do { if (it < 36 && blockIdx.x == 37 && threadIdx.x == 0 && first_pass) g_trav_prof[it * 12 + (k)] = wall_clock64(); } while (0)
extern "C" int nbco_debug_trav_prof(long long *out432)
{
	return (int)hipMemcpyFromSymbol(out432, HIP_SYMBOL(g_trav_prof), sizeof(long long) * 36 * 12);
}
#define TRAV_FIRST_PASS(v) first_pass = (v)
#define TRAV_DEP(v) asm volatile("" :: "v"(v))
#define TRAV_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define SUBTREE_MARK(k)
#define TRAV_MARK(k)
#define TRAV_DEP(v)
#define TRAV_DRAIN()
#define TRAV_FIRST_PASS(v)
#endif

__global__ __launch_bounds__(kSubT) void kd_subtree_kernel(TreeView t, const float4 *__restrict__ pos_in, const int *__restrict__ unsort_in,
                                                           float4 *__restrict__ pos_out, int *__restrict__ unsort_out, long long n, int l0,
                                                           int canon, int two_pass, int *__restrict__ flag)
{
	__shared__ __attribute__((aligned(16))) uint64_t keys[kSubS];
	__shared__ int prio[3];
	__shared__ __attribute__((aligned(16))) float px[kSubS], py[kSubS], pz[kSubS];
	__shared__ int orig[kSubS];
	__shared__ unsigned char sdl[kSubS];   // split dimension of the current level's nodes of this subtree
	// in-LDS selection levels (segments > kSelSeg): per node select state, tie lists and ancestor axes
	__shared__ SubSel sel[kSelNodes / 2];                 // nodes that are split (their children number up to kSelNodes)
	__shared__ int tie_idx[kSelNodes / 2][kSubTieCap];
	__shared__ signed char anc[kSelNodes][3];      // distinct split axes of a node's ancestors, most recent first (-1: none)
	__shared__ signed char anc_next[kSelNodes][3];
	// selection keys are normalised to the node's box along its split axis (subtract the lower face, shift the span up to
	// bit 31): order preserving, and the FIRST radix digit spreads over all bins instead of hammering one LDS counter
	__shared__ float boxs[2][kSelNodes][6];        // lower and upper faces of the current level's nodes, and of their children
	__shared__ uint32_t wmin[kSelNodes], wmin_next[kSelNodes];
	__shared__ int wshl[kSelNodes], wshl_next[kSelNodes];
	const int tid = threadIdx.x;
	SUBTREE_MARK(500);
	const long long j0 = blockIdx.x, m0 = 1LL << l0;
	const long long s0 = (j0 == 0) ? 0 : ((n * j0 - 1) >> l0) + 1;   // (shifts: the divisors are powers of two)
	const long long e0 = ((n * (j0 + 1) - 1) >> l0) + 1;
	const int cnt = (int)(e0 - s0);
	int P2 = 1;
	while (P2 < cnt) P2 <<= 1;
	for (int i = tid; i < cnt; i += kSubT)
	{
		const float4 q = pos_in[s0 + i];
		px[i] = q.x; py[i] = q.y; pz[i] = q.z;
		orig[i] = unsort_in[s0 + i];
	}
	// the selection levels' histograms live in the key buffer; every scan leaves the bins it read at zero, so clearing the
	// buffer once (while the particles are on their way) serves all levels
	for (int q = tid; q < 2 * kSubS; q += kSubT) reinterpret_cast<uint32_t *>(keys)[q] = 0u;
	if (tid < 64)
	{
		// the subtree's root: its split axis, box, and the distinct split axes of its ancestors, most recent first (keys of the
		// stable-sort chain).  One round trip: lane k fetches the axis of the ancestor k + 1 levels up, all six faces are loaded
		// before the axis is known.  (Scalars, not an indexed array: an indexable private array is promoted to LDS by the
		// compiler, and the promoted form reads the workgroup size from the dispatch packet in host memory -- 5 to 25 us at the
		// head of every launch.)
		const int root = kd_beg(l0) + (int)j0;
		const int up = (root + 1) >> (tid + 1);                       // 1-based heap number of that ancestor, 0: above the root
		const int mine = (tid < 31 && up > 0) ? t.splitdim[up - 1] : -1;
		const int a = t.splitdim[root];
		const float l0f = t.lbound[3 * root], l1f = t.lbound[3 * root + 1], l2f = t.lbound[3 * root + 2];
		const float r0f = t.rbound[3 * root], r1f = t.rbound[3 * root + 1], r2f = t.rbound[3 * root + 2];
		int b0 = -1, b1 = -1, b2 = -1;
		for (int k = 0; k < l0 && k < 31; ++k)
		{
			const int ax = __shfl(mine, k);
			if (ax < 0 || ax == b0 || ax == b1 || b2 >= 0) continue;
			if (b0 < 0) b0 = ax; else if (b1 < 0) b1 = ax; else b2 = ax;
		}
		if (tid == 0)
		{
			sdl[0] = (unsigned char)a;
			prio[0] = b0; prio[1] = b1; prio[2] = b2;
			anc[0][0] = (signed char)b0; anc[0][1] = (signed char)b1; anc[0][2] = (signed char)b2;
			const uint32_t lo = ordered_bits(a == 0 ? l0f : (a == 1 ? l1f : l2f)), span = ordered_bits(a == 0 ? r0f : (a == 1 ? r1f : r2f)) - lo;
			wmin[0] = lo; wshl[0] = span ? __clz(span) : 0;
			boxs[0][0][0] = l0f; boxs[0][0][1] = l1f; boxs[0][0][2] = l2f; boxs[0][0][3] = r0f; boxs[0][0][4] = r1f; boxs[0][0][5] = r2f;
			sel[0] = SubSel{0u, 0xFFFFFFFFu, 0u, P2 >> 1, 0, 0, 0, 0};
		}
	}
	__syncthreads();
	SUBTREE_MARK(501);

	// Bitonic sort of keys[0, P2), ascending inside every aligned block of `seg` elements (seg = P2: the whole
	// slice).  Each wave owns a contiguous chunk of 64 R keys in registers (R = 1, 2 or 4 per lane): all
	// compare-exchange stages whose partner distance stays inside the chunk run on shuffles / register
	// swaps with no barrier; only the (at most 10) stages with larger strides go through LDS.
	auto bitonic = [&](int seg) {
		const int nw = kSubT / 64;                                   // 16 waves
		const int R = P2 >= 64 * nw ? P2 / (64 * nw) : 1;           // keys per lane
		const int chunk = 64 * R;
		const int wv = tid >> 6, lane = tid & 63;
		const int wbase = wv * chunk;
		const bool active = wbase < P2;
		uint64_t v[4];
		auto load = [&]() {
#pragma unroll
			for (int r = 0; r < 4; ++r)
				if (r < R && active) v[r] = (wbase + lane + 64 * r) < P2 ? keys[wbase + lane + 64 * r] : ~0ull;
		};
		auto store = [&]() {
#pragma unroll
			for (int r = 0; r < 4; ++r)
				if (r < R && active && (wbase + lane + 64 * r) < P2) keys[wbase + lane + 64 * r] = v[r];
		};
		auto local_stage = [&](int k, int j) {
			if (!active) return;
			if (j < 64)
			{
#pragma unroll
				for (int r = 0; r < 4; ++r)
					if (r < R)
					{
						const int e = wbase + lane + 64 * r;
						const uint64_t o = __shfl_xor(v[r], j);
						const bool up = ((e & ~j & k) == 0) || k == seg;
						const bool keep_min = ((e & j) == 0) == up;
						v[r] = keep_min ? (v[r] < o ? v[r] : o) : (v[r] < o ? o : v[r]);
					}
			}
			else
			{
				const int dr = j >> 6;   // 1 or 2
#pragma unroll
				for (int r = 0; r < 4; ++r)
					if (r < R && (r & dr) == 0)
					{
						const int e = wbase + lane + 64 * r;
						const bool up = ((e & k) == 0) || k == seg;
						const uint64_t a = v[r], b = v[r | dr];
						if ((a > b) == up) { v[r] = b; v[r | dr] = a; }
					}
			}
		};
		bool in_regs = false;
		for (int k = 2; k <= seg; k <<= 1)
			for (int j = k >> 1; j > 0; j >>= 1)
			{
				if (j < chunk)
				{
					if (!in_regs) { load(); in_regs = true; }
					local_stage(k, j);
				}
				else
				{
					if (in_regs) { store(); in_regs = false; __syncthreads(); }
					for (int q = tid; q < (P2 >> 1); q += kSubT)
					{
						const int i = ((q & ~(j - 1)) << 1) | (q & (j - 1));
						const int ixj = i | j;
						const uint64_t a = keys[i], b = keys[ixj];
						const bool up = ((i & k) == 0) || k == seg;
						if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
					}
					__syncthreads();
				}
			}
		if (in_regs) store();
		__syncthreads();
	};
	// n a power of two: every node of every level is an aligned power-of-two block of the slice, so a level's sort
	// only has to merge inside its own nodes
	const bool pow2 = cnt == P2 && (n & (n - 1)) == 0;
	// permute positions and the cumulative permutation in place, through registers
	auto permute = [&]() {
		float rx[kSubE], ry[kSubE], rz[kSubE];
		int ro[kSubE];
#pragma unroll
		for (int e = 0; e < kSubE; ++e)
		{
			const int i = tid + e * kSubT;
			if (i < cnt)
			{
				const int src = (int)(keys[i] & 0xFFF);
				rx[e] = px[src]; ry[e] = py[src]; rz[e] = pz[src]; ro[e] = orig[src];
			}
		}
		__syncthreads();
#pragma unroll
		for (int e = 0; e < kSubE; ++e)
		{
			const int i = tid + e * kSubT;
			if (i < cnt) { px[i] = rx[e]; py[i] = ry[e]; pz[i] = rz[e]; orig[i] = ro[e]; }
		}
		__syncthreads();
	};
	auto coord = [&](int i, int a) { return a == 0 ? px[i] : (a == 1 ? py[i] : pz[i]); };

	// order (u before v) of two slice elements under the keys the stable-sort chain has applied so far: the ancestors'
	// split coordinates, most recent first, then the original index
	auto chain_less = [&](int u, int v, int b1, int b2, int b3) {
		if (b1 >= 0) { const uint32_t a = ordered_bits(coord(u, b1)), b = ordered_bits(coord(v, b1)); if (a != b) return a < b; }
		if (b2 >= 0) { const uint32_t a = ordered_bits(coord(u, b2)), b = ordered_bits(coord(v, b2)); if (a != b) return a < b; }
		if (b3 >= 0) { const uint32_t a = ordered_bits(coord(u, b3)), b = ordered_bits(coord(v, b3)); if (a != b) return a < b; }
		return orig[u] < orig[v];
	};
	// Sort every aligned block of `seg` elements into the order the reference's stable-sort chain would have left it in:
	// by the parent's split coordinate, ties by the next distinct ancestor axes, then by original index.  anc_of(i)
	// gives the ancestor axes of the node that element i belongs to.
	auto canonical_sort = [&](int seg, auto anc_of) {
		for (int i = tid; i < P2; i += kSubT)
		{
			uint64_t k = ~0ull;
			if (i < cnt)
			{
				const int b1 = anc_of(i, 0);
				k = ((uint64_t)(i / seg) << 44) | ((uint64_t)(b1 >= 0 ? ordered_bits(coord(i, b1)) : 0u) << 12) | (uint64_t)i;
			}
			keys[i] = k;
		}
		__syncthreads();
		SUBTREE_MARK(410);
		bitonic(seg);
		SUBTREE_MARK(411);
		permute();
		SUBTREE_MARK(412);
		for (int i = tid; i + 1 < cnt; i += kSubT)
		{
			const int b1 = anc_of(i, 0), b2 = anc_of(i, 1), b3 = anc_of(i, 2);
			const int lo = (i / seg) * seg, hi = min(lo + seg, cnt);   // the run must not leave the node
			const uint32_t k = b1 >= 0 ? ordered_bits(coord(i, b1)) : 0u;
			auto key1 = [&](int q) { return b1 >= 0 ? ordered_bits(coord(q, b1)) : 0u; };
			if (i + 1 >= hi || key1(i + 1) != k || (i > lo && key1(i - 1) == k)) continue;
			int e = i + 1;   // run [i, e] of equal first keys: insertion sort by (b2, b3, original index)
			while (e + 1 < hi && key1(e + 1) == k) ++e;
			for (int u = i + 1; u <= e; ++u)
				for (int v = u; v > i && chain_less(v, v - 1, -1, b2, b3); --v)
				{
					float tx = px[v], ty = py[v], tz = pz[v];
					int to = orig[v];
					px[v] = px[v - 1]; py[v] = py[v - 1]; pz[v] = pz[v - 1]; orig[v] = orig[v - 1];
					px[v - 1] = tx; py[v - 1] = ty; pz[v - 1] = tz; orig[v - 1] = to;
				}
		}
		__syncthreads();
	};

	// The same order for aligned blocks of exactly 32 elements (the leaves of a power-of-two tree), by ranking instead of sorting:
	// an element's place is the number of elements of its block that precede it.  The 32 first keys of a block are read as eight
	// 16-byte broadcasts (the lanes of a half wave share the block), compared in registers; only an element whose first key is
	// not unique in its block walks the chain order.  No dependent shuffle stages, a third of the bitonic network's instructions.
	auto canonical_rank32 = [&](auto anc_of) {
		uint32_t *k32 = reinterpret_cast<uint32_t *>(keys);
#pragma unroll
		for (int e = 0; e < kSubE; ++e)
		{
			const int i = tid + e * kSubT;
			const int b1 = anc_of(i, 0);
			k32[i] = b1 >= 0 ? ordered_bits(coord(i, b1)) : 0u;
		}
		__syncthreads();
		float rx[kSubE], ry[kSubE], rz[kSubE];
		int ro[kSubE], dst[kSubE];
#pragma unroll
		for (int e = 0; e < kSubE; ++e)
		{
			const int i = tid + e * kSubT, base = i & ~31;
			const uint32_t me = k32[i];
			const uint4 *blk = reinterpret_cast<const uint4 *>(k32 + base);
			int less = 0, same = 0;
#pragma unroll
			for (int q = 0; q < 8; ++q)
			{
				const uint4 o = blk[q];
				less += (int)(o.x < me) + (int)(o.y < me) + (int)(o.z < me) + (int)(o.w < me);
				same += (int)(o.x == me) + (int)(o.y == me) + (int)(o.z == me) + (int)(o.w == me);
			}
			if (same > 1)
			{
				const int b2 = anc_of(i, 1), b3 = anc_of(i, 2);
				for (int u = base; u < base + 32; ++u)
					if (u != i && k32[u] == me && chain_less(u, i, -1, b2, b3)) ++less;
			}
			dst[e] = base + less;
			rx[e] = px[i]; ry[e] = py[i]; rz[e] = pz[i]; ro[e] = orig[i];
		}
		__syncthreads();
#pragma unroll
		for (int e = 0; e < kSubE; ++e) { const int d = dst[e]; px[d] = rx[e]; py[d] = ry[e]; pz[d] = rz[e]; orig[d] = ro[e]; }
		__syncthreads();
	};

	int s_begin = 0;   // first sub-level still to be built by sorting
	const bool by_selection = canon && pow2 && P2 > kSelSeg && (t.L - l0) > 0;
	if (by_selection)
	{
		// ---- levels whose node segments exceed kSelSeg: exact median selection + unordered partition in LDS ----------
		// (8-bit radix select over the ordered split coordinate, 4 passes; the k smallest go left.  Elements equal
		// to the pivot are ranked among themselves by the chain order above.)  The canonical order is restored
		// afterwards, once, when the segments fit a wave.
		uint32_t *hist = reinterpret_cast<uint32_t *>(keys);   // [nodes][256]
		const int wv = tid >> 6, lane = tid & 63;
		int s = 0;
		for (; (P2 >> s) > kSelSeg && l0 + s < t.L; ++s)
		{
			const int l = l0 + s, nodes = 1 << s, seg = P2 >> s, half = seg >> 1, lseg = 31 - __clz(seg);
			// radix digits of 8 bits (4 passes) while the histograms of all nodes fit the 32 KB key buffer, 7 bits (5 passes) below
			// two_pass: stop after two digits (16 or 14 bits of the box-normalised key): the pivot's bucket then holds
			// segment / 2^16 elements -- the pivot and its exact ties, practically -- and all of them go to the resolver below,
			// which orders them by (key, ancestor axes, original index)
			const int db = nodes <= 32 ? 8 : 7, bins = 1 << db, npass = two_pass ? 2 : (32 + db - 1) / db;
			const int rest = 32 - min(32, db * npass);   // key bits not looked at by the passes
			SUBTREE_MARK(16 * s);
			// (sel[] of this level's nodes and the zeroed bins were left by the prologue / the previous level)
			uint32_t key[kSubE];
#pragma unroll
			for (int e = 0; e < kSubE; ++e)
			{
				const int i = tid + e * kSubT;
				const int j = i >> lseg;
				key[e] = i < cnt ? (ordered_bits(coord(i, sdl[j])) - wmin[j]) << wshl[j] : 0u;
			}
			for (int pass = 0; pass < npass; ++pass)
			{
				const int hi = 32 - db * pass, lo = max(hi - db, 0), wd = hi - lo;
#pragma unroll
				for (int e = 0; e < kSubE; ++e)
				{
					const int i = tid + e * kSubT;
					if (i >= cnt) continue;
					const int j = i >> lseg;
					if (pass == 0 || (key[e] >> hi) == sel[j].prefix) atomicAdd(&hist[j * bins + ((key[e] >> lo) & ((1u << wd) - 1u))], 1u);
				}
				__syncthreads();
				SUBTREE_MARK(16 * s + 2 + 2 * pass);
				const bool halves = bins == 128;   // 7-bit digits: a node's bins fill half a wave, two nodes per wave
				for (int g = wv; (halves ? 2 * g : g) < nodes; g += kSubT / 64)
				{
					// one wave (or half wave) per node: find the bin holding rank r (1-based among the remaining candidates)
					const int j = halves ? 2 * g + (lane >> 5) : g, ln = halves ? (lane & 31) : lane;
					const int r = sel[j].rank;
					uint32_t cb[4];
					uint32_t sum = 0;
#pragma unroll
					for (int q = 0; q < 4; ++q)
					{
						const int bin = ln * 4 + q;
						cb[q] = hist[j * bins + bin];
						hist[j * bins + bin] = 0;
						sum += cb[q];
					}
					const uint32_t incl = wave_scan_add(sum, halves);
					uint32_t before = incl - sum;
					if ((uint32_t)r > before && (uint32_t)r <= incl)
					{
#pragma unroll
						for (int q = 0; q < 4; ++q)
						{
							if ((uint32_t)r > before && (uint32_t)r <= before + cb[q])
							{
								sel[j].prefix = (sel[j].prefix << wd) | (uint32_t)(ln * 4 + q);
								sel[j].rank = r - (int)before;
								sel[j].neq = (int)cb[q];
							}
							before += cb[q];
						}
					}
				}
				__syncthreads();
				SUBTREE_MARK(16 * s + 3 + 2 * pass);
			}
			if (!two_pass && tid < nodes) sel[tid].pivot = sel[tid].prefix;   // all digits known: the prefix is the pivot
			// classify: 0 left, 1 right, 2 candidate (pivot tie, or pivot bucket after two passes: side decided by its rank
			// among the node's candidates)
			int side[kSubE];
#pragma unroll
			for (int e = 0; e < kSubE; ++e)
			{
				const int i = tid + e * kSubT;
				side[e] = 0;
				if (i >= cnt) continue;
				const int j = i >> lseg;
				const uint32_t pv = sel[j].prefix, kh = two_pass ? key[e] >> rest : key[e];
				if (kh > pv) side[e] = 1;
				else if (kh == pv && (two_pass || sel[j].rank < sel[j].neq))
				{
					side[e] = 2;
					const int slot = atomicAdd(&sel[j].ntie, 1);
					if (slot < kSubTieCap) tie_idx[j][slot] = i;
				}
			}
			__syncthreads();
			SUBTREE_MARK(16 * s + 6);
			float rx[kSubE], ry[kSubE], rz[kSubE];
			int ro[kSubE], dst[kSubE];
#pragma unroll
			for (int e = 0; e < kSubE; ++e)
			{
				const int i = tid + e * kSubT;
				dst[e] = -1;
				if (i >= cnt) continue;
				const int j = i >> lseg;
				if (side[e] == 2)
				{
					const int nt = min(sel[j].ntie, kSubTieCap);
					// candidates differ in the split coordinate itself when the select stopped early: it is the first key, the
					// ancestors' axes other than it follow
					const int a1 = sdl[j];
					int b2 = -1, b3 = -1;
					for (int q = 0; q < 3; ++q)
					{
						const int a = anc[j][q];
						if (a < 0 || a == a1) continue;
						if (b2 < 0) b2 = a; else if (b3 < 0) b3 = a;
					}
					int rk = 0;
					for (int q = 0; q < nt; ++q)
					{
						const int o = tie_idx[j][q];
						if (o != i && chain_less(o, i, a1, b2, b3)) ++rk;
					}
					side[e] = rk < sel[j].rank ? 0 : 1;
					if (rk + 1 == sel[j].rank) sel[j].pivot = key[e];   // the last element of the left child
				}
				rx[e] = px[i]; ry[e] = py[i]; rz[e] = pz[i]; ro[e] = orig[i];
			}
			SUBTREE_MARK(16 * s + 11);
			// slots: the 64 lanes of a wave hold consecutive elements of ONE node (segments are >= 64 long), so one LDS atomic
			// per wave and side reserves the slots and a ballot prefix hands them out
			// (the live lanes of a wave are a prefix of it, so lane 0 is live whenever one is; it issues the atomics of all four
			// elements back to back -- one wait instead of four)
			uint64_t mLs[kSubE], mRs[kSubE];
			uint32_t kmins[kSubE];
			int bLs[kSubE], bRs[kSubE];
#pragma unroll
			for (int e = 0; e < kSubE; ++e)
			{
				const bool on = tid + e * kSubT < cnt;
				mLs[e] = __ballot(on && side[e] == 0);
				mRs[e] = __ballot(on && side[e] != 0);
				kmins[e] = wave_min_u32((on && side[e] != 0) ? key[e] : 0xFFFFFFFFu);
				bLs[e] = 0; bRs[e] = 0;
			}
			if (lane == 0)
			{
#pragma unroll
				for (int e = 0; e < kSubE; ++e)
				{
					const int j = (tid + e * kSubT) >> lseg;
					if (mLs[e]) bLs[e] = atomicAdd(&sel[j].cntL, __popcll(mLs[e]));
					if (mRs[e]) { bRs[e] = atomicAdd(&sel[j].cntR, __popcll(mRs[e])); atomicMin(&sel[j].minR, kmins[e]); }
				}
			}
#pragma unroll
			for (int e = 0; e < kSubE; ++e)
			{
				const int i = tid + e * kSubT;
				const int j = i >> lseg;
				const uint64_t below = (1ull << lane) - 1ull;
				const int baseL = __builtin_amdgcn_readfirstlane(bLs[e]), baseR = __builtin_amdgcn_readfirstlane(bRs[e]);
				if (i < cnt) dst[e] = side[e] == 0 ? j * seg + baseL + __popcll(mLs[e] & below) : j * seg + half + baseR + __popcll(mRs[e] & below);
			}
			__syncthreads();
			SUBTREE_MARK(16 * s + 7);
			if (tid < nodes && sel[tid].ntie > kSubTieCap) *flag = 1;   // unresolved ties: the host redoes the build by sorting
#pragma unroll
			for (int e = 0; e < kSubE; ++e)
				if (dst[e] >= 0)
				{
					// a tie overflow can leave a side over-full; keep the stores inside the slice (the result is discarded)
					const int d = min(max(dst[e], 0), cnt - 1);
					px[d] = rx[e]; py[d] = ry[e]; pz[d] = rz[e]; orig[d] = ro[e];
				}
			__syncthreads();
			SUBTREE_MARK(16 * s + 8);
			// evalBox for the children (fmm_cart3_kdtree.cuh:109-137): the left child's upper face is the pivot (its last
			// particle in sorted order), the right child's lower face its smallest coordinate
			const long long m = 1LL << l, mc = m << 1, jbase = j0 << s;
			const int nchild = 2 << s;
			int sdc = 0;
			if (tid < nchild)
			{
				const int cidx = tid, j = cidx >> 1;
				const long long jc = (jbase << 1) + cidx;
				const long long start = (jc == 0) ? 0 : ((n * jc - 1) >> (l + 1)) + 1;
				const int node = kd_beg(l + 1) + (int)jc, parent = (node - 1) >> 1, split = sdl[j];
				float lb[3], rb[3];
				for (int a = 0; a < 3; ++a) { lb[a] = boxs[s & 1][j][a]; rb[a] = boxs[s & 1][j][3 + a]; }   // the parent's box, kept in LDS
				if (cidx & 1) lb[split] = unordered_bits((sel[j].minR >> wshl[j]) + wmin[j]);
				else rb[split] = unordered_bits((sel[j].pivot >> wshl[j]) + wmin[j]);
				for (int a = 0; a < 3; ++a) { st_agent(&t.lbound[3 * node + a], lb[a]); st_agent(&t.rbound[3 * node + a], rb[a]); }
				if (cidx < kSelNodes)
					for (int a = 0; a < 3; ++a) { boxs[(s & 1) ^ 1][cidx][a] = lb[a]; boxs[(s & 1) ^ 1][cidx][3 + a] = rb[a]; }
				sdc = longest_axis(rb[0] - lb[0], rb[1] - lb[1], rb[2] - lb[2]);
				t.splitdim[node] = sdc;
				t.index[node] = (int)start;
				// ancestor axes of the child: the parent's split axis first, then the parent's own list without it
				signed char o1 = -1, o2 = -1;
#pragma unroll
				for (int q = 0; q < 3; ++q)
				{
					const signed char a = anc[j][q];
					if (a < 0 || a == split) continue;
					if (o1 < 0) o1 = a; else if (o2 < 0) o2 = a;
				}
				if (cidx < kSelNodes)
				{
					anc_next[cidx][0] = (signed char)split; anc_next[cidx][1] = o1; anc_next[cidx][2] = o2;
					const uint32_t lo = ordered_bits(lb[sdc]), span = ordered_bits(rb[sdc]) - lo;
					wmin_next[cidx] = lo; wshl_next[cidx] = span ? __clz(span) : 0;
				}
			}
			__syncthreads();
			SUBTREE_MARK(16 * s + 9);
			if (tid < nchild)
			{
				sdl[tid] = (unsigned char)sdc;
				if (tid < kSelNodes)
				{
					anc[tid][0] = anc_next[tid][0]; anc[tid][1] = anc_next[tid][1]; anc[tid][2] = anc_next[tid][2];
					wmin[tid] = wmin_next[tid]; wshl[tid] = wshl_next[tid];
				}
				if (tid < kSelNodes / 2) sel[tid] = SubSel{0u, 0xFFFFFFFFu, 0u, seg >> 2, 0, 0, 0, 0};   // the next level's nodes
			}
			__syncthreads();
			SUBTREE_MARK(16 * s + 10);
		}
		s_begin = s;
		SUBTREE_MARK(400);
		// every remaining node (or leaf, if the selection levels reached the bottom) is an aligned block of `seg` elements
		// with its own ancestors: restore the canonical order inside each
		const int seg = P2 >> s_begin;
		if (seg == 32 && P2 == kSubS) canonical_rank32([&](int i, int q) { return (int)anc[i >> 5][q]; });
		else canonical_sort(seg, [&](int i, int q) { return (int)anc[i / seg][q]; });
		SUBTREE_MARK(401);
	}
	else if (canon && l0 > 0)
	{
		// The selection passes above this level deliver the right particle SET in arbitrary order: restore the order of
		// the reference's stable-sort chain for the whole slice.
		canonical_sort(P2, [&](int, int q) { return prio[q]; });
	}

	for (int l = l0 + s_begin; l < t.L; ++l)
	{
		const int s = l - l0;                     // sub-level
		SUBTREE_MARK(200 + 8 * s);
		const long long m = 1LL << l;
		const long long jbase = j0 << s;          // first node of this subtree at level l
		// (a) composite keys (fmm_cart3_kdtree.cuh:167-187): node = floor(2^l i / n)
		for (int i = tid; i < P2; i += kSubT)
		{
			uint64_t k = ~0ull;
			if (i < cnt)
			{
				const long long jl = (m * (s0 + i)) / n - jbase;
				const int sd = sdl[jl];
				const float v = sd == 0 ? px[i] : (sd == 1 ? py[i] : pz[i]);
				k = ((uint64_t)jl << 44) | ((uint64_t)ordered_bits(v) << 12) | (uint64_t)i;
			}
			keys[i] = k;
		}
		__syncthreads();
		SUBTREE_MARK(200 + 8 * s + 1);
		// (b) bitonic sort, ascending; (c) apply the permutation
		bitonic(pow2 ? (P2 >> s) : P2);
		SUBTREE_MARK(200 + 8 * s + 2);
		permute();
		SUBTREE_MARK(200 + 8 * s + 3);
		// (d) evalBox for the children (fmm_cart3_kdtree.cuh:109-137); parents' boxes were written by this
		// workgroup (or by the global pass for l = l0): read them past the L1
		const long long mc = m << 1;
		const int nchild = 2 << s;
		for (int cidx = tid; cidx < nchild; cidx += kSubT)
		{
			const long long jc = (jbase << 1) + cidx;
			const long long start = (jc == 0) ? 0 : (n * jc - 1) / mc + 1;
			const long long end = (n * (jc + 1) - 1) / mc + 1;
			const int node = kd_beg(l + 1) + (int)jc, parent = (node - 1) >> 1, split = sdl[cidx >> 1];
			float lb[3], rb[3];
			for (int a = 0; a < 3; ++a) { lb[a] = ld_agent(&t.lbound[3 * parent + a]); rb[a] = ld_agent(&t.rbound[3 * parent + a]); }
			if (cidx & 1)
			{
				const int i = (int)(start - s0);
				lb[split] = split == 0 ? px[i] : (split == 1 ? py[i] : pz[i]);
			}
			else
			{
				const int i = (int)(end - 1 - s0);
				rb[split] = split == 0 ? px[i] : (split == 1 ? py[i] : pz[i]);
			}
			for (int a = 0; a < 3; ++a) { st_agent(&t.lbound[3 * node + a], lb[a]); st_agent(&t.rbound[3 * node + a], rb[a]); }
			const int sdc = longest_axis(rb[0] - lb[0], rb[1] - lb[1], rb[2] - lb[2]);
			t.splitdim[node] = sdc;
			t.index[node] = (int)start;
			// the split dims of the next level are consumed by this workgroup only; stage them after the barrier
			keys[cidx] = (uint64_t)sdc;
		}
		__syncthreads();
		for (int cidx = tid; cidx < nchild; cidx += kSubT) sdl[cidx] = (unsigned char)keys[cidx];
		__syncthreads();
		SUBTREE_MARK(200 + 8 * s + 4);
	}
	SUBTREE_MARK(402);
	for (int i = tid; i < cnt; i += kSubT)
	{
		pos_out[s0 + i] = make_float4(px[i], py[i], pz[i], 0.f);
		unsort_out[s0 + i] = orig[i];
	}
	SUBTREE_MARK(403);
	// multiplicity and centre of charge of this slice's leaves, while their particles are still in LDS (what kd_leaf_kernel
	// does from HBM: sequential sum in particle order, one division)
	{
		const int sl = t.L - l0;
		const long long mL = 1LL << t.L, jb = j0 << sl;
		if (pow2 && (cnt >> sl) == 32)
		{
			// leaves of exactly 32 particles at multiples of 32: one thread per leaf and axis, the particles as eight 16-byte reads
			// (the leaves of a wave's lanes all start in the same LDS bank: a quarter of the conflicts of 32 single reads)
			const int nl = 1 << sl;
			for (int w = tid; w < 3 * nl; w += kSubT)
			{
				const int axis = w / nl, i = w - axis * nl;
				const float4 *src = reinterpret_cast<const float4 *>((axis == 0 ? px : (axis == 1 ? py : pz)) + 32 * i);
				float sum = 0.f;
#pragma unroll
				for (int q = 0; q < 8; ++q)
				{
					const float4 v = src[q];
					sum = sum + v.x; sum = sum + v.y; sum = sum + v.z; sum = sum + v.w;
				}
				const int node = kd_beg(t.L) + (int)(jb + i);
				if (axis == 0) t.mult[node] = 32;
				t.center[3 * node + axis] = sum / 32.f;
			}
		}
		else
		for (int i = tid; i < (1 << sl); i += kSubT)
		{
			const long long jc = jb + i;
			const long long st = (jc == 0) ? 0 : (n * jc - 1) / mL + 1, en = (n * (jc + 1) - 1) / mL + 1;
			const int mlt = (int)(en - st);
			float sx = 0.f, sy = 0.f, sz = 0.f;
			for (int k = (int)(st - s0); k < (int)(en - s0); ++k) { sx = sx + px[k]; sy = sy + py[k]; sz = sz + pz[k]; }
			if (mlt > 0) { const float d = (float)mlt; sx = sx / d; sy = sy / d; sz = sz / d; }
			const int node = kd_beg(t.L) + (int)jc;
			t.mult[node] = mlt;
			t.center[3 * node] = sx; t.center[3 * node + 1] = sy; t.center[3 * node + 2] = sz;
		}
	}
	SUBTREE_MARK(404);
}

#pragma clang fp contract(fast)

// ---- dual tree traversal -----------------------------------------------------------------------------
// counters: [0] p2p count, [1] m2l count, [2] overflow flag, [4 + it] frontier size of iteration it

// exclusive scan of a packed 3-field counter over the 256 threads of a block (fields: bits 0-19,
// 20-39, 40-59; every block total stays far below 2^20)
__device__ inline uint64_t block_exclusive_scan3(uint64_t v, uint64_t *sh_wave, uint64_t &total)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	// (three 20-bit fields, scanned one by one on DPP: a 64-bit __shfl_up is two LDS round trips per step)
	const uint64_t incl = (uint64_t)wave_scan_add((uint32_t)(v & 0xFFFFFu)) | ((uint64_t)wave_scan_add((uint32_t)((v >> 20) & 0xFFFFFu)) << 20) |
	                      ((uint64_t)wave_scan_add((uint32_t)(v >> 40)) << 40);
	if (lane == 63) sh_wave[w] = incl;
	__syncthreads();
	uint64_t base = 0, tot = 0;
	for (int k = 0; k < 4; ++k)
	{
		uint64_t s = sh_wave[k];
		if (k < w) base += s;
		tot += s;
	}
	total = tot;
	return base + incl - v;
}

// classification of one node pair (fmm_cart3_kdtree.cuh:586-609 CPU order, :504-542 GPU order):
// 0 nothing, 1 P2P, 2 M2L, 3 self pair -> 3 children, 4 split the second node, 5 split the first node
__device__ inline int classify_pair(const TreeView &t, const AdmTab *tab, int2 np, float par, int m2l_first, const Dom dm)
{
	const int ntot = t.ntot;
	if (dm.d > 0 && !dom_touch(dm, np.x) && !dom_touch(dm, np.y)) return 0;   // nothing below this pair reaches the domain
	const bool leaf1 = 2 * np.x + 1 >= ntot, leaf2 = 2 * np.y + 1 >= ntot;
	if (!m2l_first && leaf1 && leaf2) return (np.x != np.y) ? 1 : 0;
	if (np.x == np.y) return leaf1 ? 0 : 3;
	const float4 c1 = t.csz[np.x], c2 = t.csz[np.y];
	if (kd_admissible(c1, c2, np.x, np.y, t.mult, tab, par)) return 2;
	if (leaf1 && leaf2) return 1;
	return (leaf1 || (!leaf2 && c1.w <= c2.w)) ? 4 : 5;
}

// the same on preloaded traversal records (centre + size, multiplicity): the traversal kernel fetches the records of a pair's
// nodes AND of their children in one round trip, before it knows how the pair splits
struct NodeRec { float4 c; int m; };
#pragma clang fp contract(off)   // the admissibility test must round exactly like the oracle's (kd_admissible above): no fused multiply-adds
__device__ inline bool kd_admissible_rec(const NodeRec a, const NodeRec b, int n1, int n2, const AdmTab *tabp, float par)
{
	float dx = b.c.x - a.c.x, dy = b.c.y - a.c.y, dz = b.c.z - a.c.z;
	float dist2 = dx * dx + dy * dy + dz * dz;
	int nb = a.m >= b.m ? n1 : n2, mb = a.m >= b.m ? a.m : b.m;
	int lev = 31 - __clz(nb + 1);
	float M = (mb == tabp->lo[lev]) ? tabp->Mlo[lev] : tabp->Mhi[lev];
	float parM = par * M;
	float sz = fmaxf(a.c.w, b.c.w);
	return parM * parM * sz < dist2;
}
#pragma clang fp contract(fast)
__device__ inline int classify_rec(int ntot, const AdmTab *tab, int2 np, const NodeRec a, const NodeRec b, float par, int m2l_first, const Dom dm)
{
	if (dm.d > 0 && !dom_touch(dm, np.x) && !dom_touch(dm, np.y)) return 0;   // nothing below this pair reaches the domain
	const bool leaf1 = 2 * np.x + 1 >= ntot, leaf2 = 2 * np.y + 1 >= ntot;
	if (!m2l_first && leaf1 && leaf2) return (np.x != np.y) ? 1 : 0;
	if (np.x == np.y) return leaf1 ? 0 : 3;
	if (kd_admissible_rec(a, b, np.x, np.y, tab, par)) return 2;
	if (leaf1 && leaf2) return 1;
	return (leaf1 || (!leaf2 && a.c.w <= b.c.w)) ? 4 : 5;
}

// children of a split pair, by value (an int2[] written through a pointer ends up in scratch memory, i.e. in extra
// round trips on the traversal's dependency chain)
struct PairKids
{
	int2 a, b, c;
	int n;
};
__device__ inline PairKids pair_children(int kd, int2 np)
{
	// branch-free selects on scalars (kd: 3 self pair -> 3 children, 4 split the second node, 5 split the first)
	const int x1 = 2 * np.x + 1, x2 = 2 * np.x + 2, y1 = 2 * np.y + 1, y2 = 2 * np.y + 2;
	PairKids k;
	k.a.x = kd == 4 ? np.x : x1;
	k.a.y = kd == 5 ? np.y : (kd == 4 ? y1 : x1);
	k.b.x = kd == 4 ? np.x : (kd == 5 ? x2 : x1);
	k.b.y = kd == 5 ? np.y : (kd == 4 ? y2 : x2);
	k.c.x = x2;
	k.c.y = x2;
	k.n = kd == 3 ? 3 : (kd >= 4 ? 2 : 0);
	return k;
}

// One launch advances the pair frontier by TWO traversal steps: every thread classifies its pair and,
// if it splits, classifies the (up to 3) children as well; only grandchildren go back to the frontier.
// That halves the number of dependent launches of this latency-bound phase.  Output slots are reserved
// with one packed block scan and three atomics per block.
//
// A returning atomic on ONE address costs ~27 ns on this part (measured), and every workgroup needs one per output
// list before it can write: with a single counter per list the 1024 workgroups of a launch queue up for ~28 us.
// The frontier and the two pair lists are therefore kept as kTravK independent regions (capacity cap / kTravK each,
// workgroup b appends to region b mod kTravK): kTravK short queues instead of one long one.  Readers map a dense
// index to (region, offset) with the regions' prefix sums.
constexpr int kTravK = 16;
constexpr int kTcFrontier = 0;                    // [it][kTravK] sizes of the frontier regions before iteration it (it < 36)
constexpr int kTcP2P = 36 * kTravK;               // [kTravK] region sizes of the P2P pair list
constexpr int kTcM2L = kTcP2P + kTravK;           // [kTravK]                    M2L pair list
constexpr int kTcP2PPref = kTcM2L + kTravK;       // [kTravK + 1] exclusive prefix sums (traverse_finish_kernel)
constexpr int kTcM2LPref = kTcP2PPref + kTravK + 1;
constexpr int kTcInts = kTcM2LPref + kTravK + 1;
// traversal launches beyond the tree depth: every launch performs two traversal steps, and L launches empty the frontier in
// every case tried (edge sizes, deep trees, sharded trees); one spare, and traverse_finish_kernel checks the outcome
#ifndef NBCO_TRAV_EXTRA
#define NBCO_TRAV_EXTRA 1
#endif

// dense index -> slot of a region-structured list
__device__ inline long long region_slot(const int *__restrict__ pref, long long capR, long long i)
{
	int r = 0;
#pragma unroll
	for (int q = 1; q < kTravK; ++q) r += (i >= pref[q]) ? 1 : 0;
	return (long long)r * capR + (i - pref[r]);
}

__global__ __launch_bounds__(kBlock) void traverse_kernel(TreeView t, AdmTab tab_arg, const int2 *__restrict__ fin, int2 *__restrict__ fout,
                                                          int2 *__restrict__ p2p, int2 *__restrict__ m2l, int2 *__restrict__ p2p_rank,
                                                          int2 *__restrict__ m2l_rank, int *__restrict__ counters,
                                                          int *__restrict__ tctr, int it, long long capR, float par, int m2l_first,
                                                          unsigned *__restrict__ cnt_p2p, unsigned *__restrict__ cnt_m2l, const Dom dm)
{
	__shared__ uint64_t sh_wave[4];
	__shared__ int sh_base[3];
	__shared__ int in_pref[kTravK + 1];
	__shared__ AdmTab tab;   // LDS copy: a lane-indexed read of the kernel argument would be one more global round trip per test
#ifdef NBCO_SUBTREE_PROF
	bool first_pass;
#endif
	TRAV_FIRST_PASS(true);
	TRAV_MARK(0);
	for (int q = threadIdx.x; q < (int)(sizeof(AdmTab) / sizeof(int)); q += kBlock) reinterpret_cast<int *>(&tab)[q] = reinterpret_cast<const int *>(&tab_arg)[q];
	if (threadIdx.x < 64)
	{
		// prefix sums of the input frontier's region sizes
		const int lane = threadIdx.x;
		// (a region that ran over its capacity holds capR valid pairs; the overflow flag is already up)
		const int v = lane < kTravK ? (int)min((long long)tctr[kTcFrontier + it * kTravK + lane], capR) : 0, incl = (int)wave_scan_add((uint32_t)v);
		if (lane < kTravK) in_pref[lane] = incl - v;
		if (lane == kTravK - 1) in_pref[kTravK] = incl;
	}
	__syncthreads();
	TRAV_MARK(1);
	const int nin = in_pref[kTravK];
	const int lbeg = kd_beg(t.L);
	const int rout = (blockIdx.x + 5 * it) & (kTravK - 1);   // rotate, so that a busy part of the frontier does not keep feeding one region
	const long long obase = (long long)rout * capR;
	for (long long base = (long long)blockIdx.x * kBlock; base < nin; base += (long long)gridDim.x * kBlock)
	{
		const long long i = base + threadIdx.x;
		// up to 4 classified pairs per thread: the input pair and its children (named scalars: nothing goes to scratch)
		int2 p0 = make_int2(0, 0);
		int k0 = 0, k1 = 0, k2 = 0, k3 = 0;
		PairKids ch;
		ch.a = ch.b = ch.c = make_int2(0, 0);
		ch.n = 0;
		if (i < nin)
		{
			p0 = fin[region_slot(in_pref, capR, i)];
			if (!NBCO_CHECKED_OK((unsigned)p0.x < (unsigned)t.ntot && (unsigned)p0.y < (unsigned)t.ntot, NBCO_CHK_FRONTIER)) p0 = make_int2(0, 0);
			TRAV_DEP(p0.x);
			TRAV_MARK(2);
			// records of x, y and of their children, all in flight together (a leaf's "children" are clamped and never used)
			const int last = t.ntot - 1;
			const int ix1 = min(2 * p0.x + 1, last), ix2 = min(2 * p0.x + 2, last), iy1 = min(2 * p0.y + 1, last), iy2 = min(2 * p0.y + 2, last);
			const NodeRec X{t.csz[p0.x], t.mult[p0.x]}, Y{t.csz[p0.y], t.mult[p0.y]};
			const NodeRec X1{t.csz[ix1], t.mult[ix1]}, X2{t.csz[ix2], t.mult[ix2]}, Y1{t.csz[iy1], t.mult[iy1]}, Y2{t.csz[iy2], t.mult[iy2]};
			TRAV_DEP(Y2.m); TRAV_DEP(X.c.x); TRAV_DEP(Y.c.x); TRAV_DEP(X1.c.x); TRAV_DEP(X2.c.x); TRAV_DEP(Y1.c.x); TRAV_DEP(Y2.c.x);
			TRAV_MARK(3);
			k0 = classify_rec(t.ntot, &tab, p0, X, Y, par, m2l_first, dm);
			ch = pair_children(k0, p0);
			// children (pair_children): 3 -> (x1,x1) (x1,x2) (x2,x2); 4 -> (x,y1) (x,y2); 5 -> (x1,y) (x2,y)
			const NodeRec A1 = k0 == 4 ? X : X1, A2 = k0 == 5 ? Y : (k0 == 4 ? Y1 : X1);
			const NodeRec B1 = k0 == 4 ? X : (k0 == 5 ? X2 : X1), B2 = k0 == 5 ? Y : (k0 == 4 ? Y2 : X2);
			if (ch.n > 0) k1 = classify_rec(t.ntot, &tab, ch.a, A1, A2, par, m2l_first, dm);
			if (ch.n > 1) k2 = classify_rec(t.ntot, &tab, ch.b, B1, B2, par, m2l_first, dm);
			if (ch.n > 2) k3 = classify_rec(t.ntot, &tab, ch.c, X2, X2, par, m2l_first, dm);
		}
		const int nch = ch.n;
		TRAV_DEP(k0 + k1 + k2 + k3);
		TRAV_MARK(4);
		auto weight = [](int q) { return (uint64_t)(q == 3 ? 3 : (q >= 4 ? 2 : 0)) | ((uint64_t)(q == 1) << 20) | ((uint64_t)(q == 2) << 40); };
		// a split input pair itself emits nothing
		const uint64_t cnt = nch > 0 ? weight(k1) + weight(k2) + weight(k3) : weight(k0);
		uint64_t tot;
		const uint64_t off = block_exclusive_scan3(cnt, sh_wave, tot);
		const int tf = (int)(tot & 0xFFFFF), tp = (int)((tot >> 20) & 0xFFFFF), tm = (int)(tot >> 40);
		TRAV_MARK(5);
		if (threadIdx.x == 0) sh_base[0] = tf ? atomicAdd(&tctr[kTcFrontier + (it + 1) * kTravK + rout], tf) : 0;
		if (threadIdx.x == 64) sh_base[1] = tp ? atomicAdd(&tctr[kTcP2P + rout], tp) : 0;
		if (threadIdx.x == 128) sh_base[2] = tm ? atomicAdd(&tctr[kTcM2L + rout], tm) : 0;
		// the per-target entry counts of the directed lists are accumulated here, under the traversal's latency; the value
		// an atomic returns is the entry's slot inside its target's range, kept beside the pair so that filling the
		// directed lists needs no second round of atomics (device-scope atomics retire at ~17 G/s on this part: two
		// per entry were 60 us of every evaluation).  -1: the node belongs to another domain.  All atomics of a thread
		// are issued before any of their results is used, and before the barrier that publishes the block's reservations, so
		// they share that round trip.  (They count even when a region turns out to be full: traverse_finish_kernel clears
		// the per-target counts of an overflowed traversal.)
		auto slots = [&](int q, int2 np, bool ok) {
			int2 r = make_int2(-1, -1);
			if (q == 1 && ok)
			{
				if (dm.d == 0 || dom_touch(dm, np.x)) r.x = (int)atomicAdd(&cnt_p2p[np.x - lbeg], 1u);
				if (dm.d == 0 || dom_touch(dm, np.y)) r.y = (int)atomicAdd(&cnt_p2p[np.y - lbeg], 1u);
			}
			if (q == 2 && ok)
			{
				if (dm.d == 0 || dom_touch(dm, np.x)) r.x = (int)atomicAdd(&cnt_m2l[np.x], 1u);
				if (dm.d == 0 || dom_touch(dm, np.y)) r.y = (int)atomicAdd(&cnt_m2l[np.y], 1u);
			}
			return r;
		};
		const int2 r0 = slots(k0, p0, nch == 0 && i < nin), r1 = slots(k1, ch.a, nch > 0), r2 = slots(k2, ch.b, nch > 1), r3 = slots(k3, ch.c, nch > 2);
		// The barrier publishes the block's three reservations (sh_base, LDS).  It must NOT wait for the slot atomics above, which
		// take 2-8 us to come back in a wide launch: an LDS-only barrier (no workgroup fence, which would drain the vector-memory
		// counter), the pairs and the next frontier are stored under that wait, the slots last.
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		TRAV_MARK(7);
		long long bf = sh_base[0], bp = sh_base[1], bm = sh_base[2];
		const bool okf = bf + tf <= capR, okp = bp + tp <= capR, okm = bm + tm <= capR;
		if (threadIdx.x == 0 && !(okf && okp && okm)) counters[2] = 1;
		// The workgroup whose reservation crosses the end of a frontier region writes nothing, and the next launch reads the
		// region up to its capacity: give the unwritten tail a well-formed pair (the root against itself), or that launch
		// would classify whatever the buffer held before.  (The evaluation is lost anyway -- NBCO_ERR_CAPACITY -- but every
		// kernel queued behind the traversal must still run on valid indices.)  The P2P / M2L regions are not read back on
		// overflow (traverse_finish_kernel declares the lists empty).
		if (!okf && bf < capR)
			for (long long k = bf + threadIdx.x; k < capR; k += kBlock) fout[obase + k] = make_int2(0, 0);
		bf += (long long)(off & 0xFFFFF); bp += (long long)((off >> 20) & 0xFFFFF); bm += (long long)(off >> 40);
		// returns the entry's place in its pair list (-1: none)
		auto emit = [&](int q, int2 np) {
			long long at = -1;
			if (q == 1 && okp) { p2p[obase + bp] = np; at = obase + bp; }
			if (q == 2 && okm) { m2l[obase + bm] = np; at = obase + bm; }
			const PairKids g = pair_children(q, np);
			if (q >= 3 && okf)
			{
				fout[obase + bf] = g.a;
				fout[obase + bf + 1] = g.b;
				if (g.n > 2) fout[obase + bf + 2] = g.c;
			}
			// cursors advance by selects (an if / else chain over them is turned into a scratch array by the compiler)
			bp += q == 1 ? 1 : 0;
			bm += q == 2 ? 1 : 0;
			bf += g.n;
			return at;
		};
		long long at0 = -1, at1 = -1, at2 = -1, at3 = -1;
		if (nch == 0) at0 = emit(k0, p0);
		else
		{
			at1 = emit(k1, ch.a);
			at2 = emit(k2, ch.b);
			if (nch > 2) at3 = emit(k3, ch.c);
		}
		TRAV_DEP(r0.x + r1.x + r2.x + r3.x + r0.y + r1.y + r2.y + r3.y);
		TRAV_MARK(6);
		auto put_slots = [&](int q, long long at, int2 r) {
			if (at < 0) return;
			if (q == 1) p2p_rank[at] = r; else m2l_rank[at] = r;
		};
		put_slots(k0, at0, r0); put_slots(k1, at1, r1); put_slots(k2, at2, r2); put_slots(k3, at3, r3);
		TRAV_MARK(8);
		TRAV_DRAIN();
		TRAV_MARK(9);
		__syncthreads();
		TRAV_MARK(10);
		TRAV_FIRST_PASS(false);
	}
	TRAV_MARK(11);
}

// start state of a traversal: the root pair in the frontier, counters cleared, and (all workgroups) the per-target entry
// counts of both lists cleared
// One (leaf, leaf) self entry per own leaf of the P2P list (fmm_cart3_kdtree.cuh:1059-1071) is counted here: it owns slot 0
// of its target's range, the slots handed out by the traversal's atomics start at 1.
__global__ __launch_bounds__(kBlock) void traverse_init_kernel(int2 *frontier, int *counters, int nctr, int *tctr, unsigned *__restrict__ list_cnt,
                                                               long long words, long long self0, long long nself)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < words; i += (long long)gridDim.x * kBlock)
		list_cnt[i] = (i >= self0 && i < self0 + nself) ? 1u : 0u;
	if (blockIdx.x != 0) return;
	for (int i = threadIdx.x; i < nctr; i += kBlock) counters[i] = 0;
	for (int i = threadIdx.x; i < kTcInts; i += kBlock) tctr[i] = 0;
	__syncthreads();
	if (threadIdx.x == 0) { frontier[0] = make_int2(0, 0); tctr[kTcFrontier] = 1; }
}

// region prefix sums and totals of the two pair lists (counters[0] = P2P pairs, counters[1] = M2L pairs).  If a region ran
// over its capacity (counters[2], reported to the caller as NBCO_ERR_CAPACITY once the host looks at the flags) the
// lists are declared empty and the per-target counts cleared, so that everything already queued behind the traversal
// runs on a consistent -- if useless -- state.
// The counts and flags the host looks at after the evaluation go straight to pinned host memory (`host_flags`: P2P pairs, M2L
// pairs, list overflow, tie flag of the build) -- two device-to-host copies less on the critical path.
__global__ __launch_bounds__(1024) void traverse_finish_kernel(int *counters, int *tctr, long long capR, unsigned *cnt_all, long long ncnt,
                                                               unsigned *cnt_self, int nself, int *__restrict__ host_flags, int iters, int seq)
{
	const int lane = threadIdx.x;
	const bool overflow = counters[2] != 0;
	if (overflow)
	{
		// empty lists; the own leaves keep their self entries
		for (long long i = threadIdx.x; i < ncnt; i += blockDim.x) cnt_all[i] = 0u;
		__syncthreads();
		for (int i = threadIdx.x; i < nself; i += blockDim.x) cnt_self[i] = 1u;
	}
	if (lane >= 64) return;
	for (int which = 0; which < 2; ++which)
	{
		const int src = which ? kTcM2L : kTcP2P, dst = which ? kTcM2LPref : kTcP2PPref;
		int v = (lane < kTravK && !overflow) ? tctr[src + lane] : 0, incl = v;
		for (int o = 1; o < kTravK; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
		if (lane < kTravK) tctr[dst + lane] = incl - v;
		if (lane == kTravK - 1) { tctr[dst + kTravK] = incl; counters[which] = incl; host_flags[which] = incl; }
	}
	// the frontier the last launch wrote must be empty (it is after L + 1 launches of two levels each; checked, not assumed)
	int left = lane < kTravK ? tctr[kTcFrontier + iters * kTravK + lane] : 0;
	for (int o = 32; o > 0; o >>= 1) left += __shfl_xor(left, o);
	if (lane == 0) { host_flags[2] = overflow ? 1 : (left != 0 ? 2 : 0); host_flags[3] = counters[110]; }
	// the host spins on this word (nbco_ctx::wait_flags): it must land after the four values above
	__threadfence_system();
	if (lane == 0) __hip_atomic_store(&host_flags[4], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// dense copy of a region-structured pair list (nbco_kd_copy)
__global__ __launch_bounds__(kBlock) void list_compact_kernel(const int2 *__restrict__ src, const int *__restrict__ pref, long long capR,
                                                              int2 *__restrict__ dst)
{
	const long long n = pref[kTravK];
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) dst[i] = src[region_slot(pref, capR, i)];
}

// ---- directed lists by counting sort ---------------------------------------------------------------
// key = target << shift | source.  P2P works on leaf numbers (node - kd_beg(L)) and gets one (leaf, leaf) self entry per
// leaf (fmm_cart3_kdtree.cuh:1059-1071); M2L works on node numbers.
// count (during the traversal, which also hands every entry its slot) -> exclusive scan -> scatter ->
// per-target sort of the (short) source ranges.  The last step makes the lists, and with them every
// floating-point sum downstream, identical from run to run.

__global__ __launch_bounds__(kBlock) void list_fill_kernel(const int2 *__restrict__ pairs, const int2 *__restrict__ ranks, const int *__restrict__ pref,
                                                           long long capR, int sub, int self0, int nself, int shift, const int *__restrict__ start,
                                                           uint64_t *__restrict__ keys, const int *__restrict__ chunk_off, int ntargets,
                                                           const int *__restrict__ leaf_index, const int *__restrict__ leaf_mult,
                                                           int4 *__restrict__ chunk, int pidmode)
{
	// pidmode (mutual near field): an entry is (source << 32 | index of its unordered pair), so that after the per-target sort the
	// two directions of a pair still know each other; self entries carry 0xFFFFFFFF
	// P2P list: the work-unit table of the pair kernel only needs the two prefix sums, like the fill: same launch.
	// chunk record: {first particle of the target leaf, first entry, end entry, particles of the target leaf}
	if (chunk)
		for (int i = blockIdx.x * kBlock + threadIdx.x; i < ntargets; i += gridDim.x * kBlock)
		{
			const int b = start[i], e = start[i + 1], o = chunk_off[i], n = chunk_off[i + 1] - o;
			const int ind = leaf_index[i], mlt = leaf_mult[i];
			// equal shares (a list of 17 becomes 9 + 8, not 16 + 1: a wave that only gets one source leaf spends its life
			// in the chain of dependent loads at the head of a chunk)
			const int per = n > 0 ? (e - b + n - 1) / n : 0;
			for (int k = 0; k < n; ++k) chunk[o + k] = make_int4(ind, min(b + k * per, e), min(b + (k + 1) * per, e), mlt);
		}
	const long long npairs = pref[kTravK];   // the pair count never leaves the device (regions are clamped to their capacity)
	const long long total = npairs + nself;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock)
	{
		if (i < npairs)
		{
			const long long slot = region_slot(pref, capR, i);
			if (!NBCO_CHECKED_OK(slot >= 0 && slot < capR * kTravK, NBCO_CHK_FILL)) continue;
			const int2 p = pairs[slot], r = ranks[slot];
			const uint64_t a = (uint64_t)(p.x - sub), b = (uint64_t)(p.y - sub);
			if (!NBCO_CHECKED_OK(a < (uint64_t)ntargets && b < (uint64_t)ntargets && (r.x < 0 || start[a] + r.x < start[a + 1])
			                         && (r.y < 0 || start[b] + r.y < start[b + 1]), NBCO_CHK_FILL)) continue;
			if (pidmode)
			{
				if (r.x >= 0) keys[start[a] + r.x] = (b << 32) | (uint64_t)(uint32_t)i;
				if (r.y >= 0) keys[start[b] + r.y] = (a << 32) | (uint64_t)(uint32_t)i;
			}
			else
			{
				if (r.x >= 0) keys[start[a] + r.x] = (a << shift) | b;
				if (r.y >= 0) keys[start[b] + r.y] = (b << shift) | a;
			}
		}
		else
		{
			// self entries of the domain's own leaves: counted first (traverse_init_kernel), so they own slot 0
			const uint64_t t = (uint64_t)(self0 + (i - npairs));
			keys[start[t]] = pidmode ? ((t << 32) | 0xFFFFFFFFull) : ((t << shift) | t);
		}
	}
}

// one wave per target: rank sort of its source range (distinct keys) in registers up to 512 entries (the BASELINE ball
// has 17 on average, 254..300 at most; a bitonic network in LDS for the 257..512 class cost 30 us of a 45 us kernel), and
// beyond that (wide opening radii, outliers late in a long run) a radix sort of the range through global memory
#ifndef NBCO_P2P_CHUNK
#define NBCO_P2P_CHUNK 16
#endif
constexpr int kP2PChunk = NBCO_P2P_CHUNK;   // source leaves per near-field work unit (see the P2P section)

// diagnostics only: the near-field launch's arrays as one binary file {header 8 x int64: n, entries, chunks, mlt_max, stride, 0, 0, 0;
// float4 pos[n]; int2 desc[entries]; int4 chunk[chunks]; float4 partial[chunks * stride]} for tools/p2p_lab.hip.  Synchronises.
static int p2p_dump(nbco_ctx *c, const char *path, const float4 *pos, const int2 *desc, const int4 *chunk, const int *nchunks, const int *nentries, int mlt_max,
                    const float4 *partial, long long n)
{
	NBCO_HIP(hipStreamSynchronize(c->stream));
	int hc = 0, he = 0;
	NBCO_HIP(hipMemcpy(&hc, nchunks, sizeof(int), hipMemcpyDeviceToHost));
	NBCO_HIP(hipMemcpy(&he, nentries, sizeof(int), hipMemcpyDeviceToHost));
	FILE *f = std::fopen(path, "wb");
	if (!f) return c->fail(NBCO_ERR_ARG, "NBCO_P2P_DUMP: cannot open the file");
	const long long head[8] = {n, he, hc, mlt_max, mlt_max, 0, 0, 0};
	std::fwrite(head, sizeof head, 1, f);
	auto put = [&](const void *dev, size_t bytes) {
		std::vector<char> h(bytes);
		if (hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
		return std::fwrite(h.data(), 1, bytes, f) == bytes;
	};
	const bool ok = put(pos, sizeof(float4) * (size_t)n) && put(desc, sizeof(int2) * (size_t)he) && put(chunk, sizeof(int4) * (size_t)hc) &&
	                put(partial, sizeof(float4) * (size_t)hc * (size_t)mlt_max);
	std::fclose(f);
	return ok ? NBCO_OK : c->fail(NBCO_ERR_HIP, "NBCO_P2P_DUMP: copy failed");
}
// DESC: the list is the P2P list -- also emit the source descriptor (first particle, multiplicity) of every sorted entry,
// so the pair kernel does no dependent index -> mult -> position loads
// what the per-target sort of the P2P list also produces for the mutual near-field kernel (k_p2p.hpp); desc4 == nullptr: off
struct MutualLists
{
	int4 *desc4 = nullptr;            // per sorted entry: {first particle of the source leaf, its multiplicity, pair index, code}
	int4 *chunk = nullptr;            // work units {first particle of the target leaf, first entry, end entry, multiplicity}
	const int *chunk_off = nullptr;   // chunk slots of target t: [chunk_off[t], chunk_off[t + 1])
	int2 *sec_range = nullptr;        // per target: its entries whose sums other waves deliver (code 2)
	int self0 = 0, nself = 0;         // leaves of the own kd-domain
};
template <bool DESC>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void list_segsort_kernel(const int *__restrict__ start, int ntargets, uint64_t *in, uint64_t *out, int shift,
                                                              const int *__restrict__ leaf_index, const int *__restrict__ leaf_mult,
                                                              int2 *__restrict__ desc, MutualLists mu)
{
	__shared__ unsigned digit_off[kBlock / 64][256];   // long ranges only: per-wave digit offsets of the radix passes
	const uint64_t smask = (1ull << shift) - 1;
	// mutual near field (mu.desc4 set): entries are (source << 32 | pair index); the sorted list is written in the usual
	// (target << shift | source) form, and every entry gets a descriptor {first particle, multiplicity, pair index, code}
	// with code 1 = this target's wave evaluates the pair for both leaves (source after target), 2 = the source's wave does
	// (source before target), 0 = one direction only (the leaf itself, or a source outside the kd-domain [self0, self0 + nself))
	const int lowbit = (DESC && mu.desc4) ? 32 : 0;
	// (li, lm: first particle and multiplicity of the source leaf, fetched by the caller as soon as it knows the source -- beside the
	// ranking, not behind it)
	auto emit = [&](int t, int slot, uint64_t key, int li, int lm) {
		const int src = (int)((key >> lowbit) & smask);
		if (!NBCO_CHECKED_OK(src >= 0 && src < ntargets && slot >= start[t] && slot < start[t + 1], NBCO_CHK_SORT)) return;
		if (DESC && mu.desc4)
		{
			out[slot] = ((uint64_t)t << shift) | (uint64_t)src;
			const int code = (src == t || src < mu.self0 || src >= mu.self0 + mu.nself) ? 0 : (src > t ? 1 : 2);
			mu.desc4[slot] = make_int4(li, lm, (int)(uint32_t)key, code);
			return;
		}
		out[slot] = key;
		if (DESC) desc[slot] = make_int2(li, lm);
	};
	// mutual near field: work units of target t once its list is sorted.  Sorted order = [sources of lower kd-domains: nf
	// entries][own sources before t: delivered by their waves][t itself][sources after t]; the work entries (all but the
	// second group) are dealt to the target's chunk slots in equal shares; a unit that spans the gap skips it by code.
	auto finish_target = [&](int t, int s, int cnt, int nf, int nq, int lane) {
		const int o = mu.chunk_off[t], n = mu.chunk_off[t + 1] - o;
		const int w = nf + (cnt - nq), per = n > 0 ? (w + n - 1) / n : 0;
		const int ind = leaf_index[t], mlt = leaf_mult[t];
		for (int i = lane; i < n; i += 64)
		{
			const int jb = min(i * per, w), je = min((i + 1) * per, w);
			mu.chunk[o + i] = make_int4(ind, s + (jb < nf ? jb : jb - nf + nq), s + (je <= nf ? je : je - nf + nq), mlt);
		}
		if (lane == 0) mu.sec_range[t] = make_int2(s + nf, s + nq);
	};
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	// ranges of up to 64 * J entries: every lane keeps J entries in registers and ranks them against all entries of the
	// range, which are broadcast one by one with v_readlane (an SGPR lane index: the loop is scalar, no LDS or memory
	// latency inside).  All entries of a range share the target, so comparing the 32-bit source indices orders the keys.
	auto rank_in_registers = [&](auto jtag, int t, int s, int cnt) {
		constexpr int J = decltype(jtag)::value;
		uint64_t key[J];
		unsigned src[J];
		// (the descriptors are fetched beside the ranking only while they fit the register budget of 8 waves per SIMD: with more
		// than four entries per lane they would spill to scratch, and a kernel with a scratch segment pays for it on every launch)
		constexpr bool PRE = DESC && J <= 4;
		int rank[J], li[PRE ? J : 1], lm[PRE ? J : 1];
#pragma unroll
		for (int j = 0; j < J; ++j)
		{
			key[j] = lane + 64 * j < cnt ? in[s + lane + 64 * j] : ~0ull;
			src[j] = (unsigned)((key[j] >> lowbit) & smask);
			rank[j] = 0;
		}
#pragma unroll
		for (int j = 0; j < J; ++j)
			if (PRE)
			{
				const int sidx = (int)min(src[j], (unsigned)(ntargets - 1));   // (idle lanes hold the all-ones key)
				li[j] = leaf_index[sidx]; lm[j] = leaf_mult[sidx];
			}
#pragma unroll
		for (int jb = 0; jb < J; ++jb)
		{
			const int lim = min(64, cnt - 64 * jb);   // wave-uniform
			for (int q = 0; q < lim; ++q)
			{
				const unsigned other = (unsigned)__builtin_amdgcn_readlane((int)src[jb], q);
#pragma unroll
				for (int j = 0; j < J; ++j) rank[j] += other < src[j] ? 1 : 0;
			}
		}
		int nf = 0, nq = 0;
		if (DESC && mu.desc4)
		{
#pragma unroll
			for (int j = 0; j < J; ++j)
			{
				const bool valid = lane + 64 * j < cnt;
				nf += __popcll(__ballot(valid && (int)src[j] < mu.self0));
				nq += __popcll(__ballot(valid && (int)src[j] < t));
			}
		}
#pragma unroll
		for (int j = 0; j < J; ++j)
			if (lane + 64 * j < cnt)
			{
				if (PRE) emit(t, s + rank[j], key[j], li[j], lm[j]);
				else
				{
					const int sidx = DESC ? (int)min(src[j], (unsigned)(ntargets - 1)) : 0;
					emit(t, s + rank[j], key[j], DESC ? leaf_index[sidx] : 0, DESC ? leaf_mult[sidx] : 0);
				}
			}
		if (DESC && mu.desc4) finish_target(t, s, cnt, nf, nq, lane);
	};
	for (int t = blockIdx.x * (kBlock / 64) + wv; t < ntargets; t += gridDim.x * (kBlock / 64))
	{
		// wave-uniform by construction; telling the compiler so keeps the loops scalar
		const int s = __builtin_amdgcn_readfirstlane(start[t]), cnt = __builtin_amdgcn_readfirstlane(start[t + 1]) - s;
		if (cnt <= 64) rank_in_registers(std::integral_constant<int, 1>{}, t, s, cnt);
		else if (cnt <= 128) rank_in_registers(std::integral_constant<int, 2>{}, t, s, cnt);
		else if (cnt <= 192) rank_in_registers(std::integral_constant<int, 3>{}, t, s, cnt);
		else if (cnt <= 256) rank_in_registers(std::integral_constant<int, 4>{}, t, s, cnt);
		else if (cnt <= 320) rank_in_registers(std::integral_constant<int, 5>{}, t, s, cnt);
		else if (cnt <= 384) rank_in_registers(std::integral_constant<int, 6>{}, t, s, cnt);
		else if (cnt <= 512) rank_in_registers(std::integral_constant<int, 8>{}, t, s, cnt);
		else
		{
			// Long range (an outlier's leaf can be paired with most of the tree: tens of thousands of entries): the wave sorts
			// it by source index with a stable LSD radix sort, 8 bits per pass, ping-ponging between its slices of the
			// unsorted and the sorted key arrays.  O(cnt) per pass, where ranking would be O(cnt^2).
			unsigned *off = digit_off[wv];
			uint64_t *src = in + s, *dst = out + s;
			const uint64_t below = (1ull << lane) - 1ull;
			for (int bit = lowbit; bit < lowbit + shift; bit += 8)
			{
				for (int b = lane; b < 256; b += 64) off[b] = 0u;
				wave_lds_sync();
				for (int i = lane; i < cnt; i += 64) atomicAdd(&off[(unsigned)(src[i] >> bit) & 255u], 1u);
				wave_lds_sync();
				// exclusive scan of the 256 digit counts (four per lane)
				unsigned v0 = off[4 * lane], v1 = off[4 * lane + 1], v2 = off[4 * lane + 2], v3 = off[4 * lane + 3];
				const unsigned sum = v0 + v1 + v2 + v3;
				const unsigned incl = wave_scan_add(sum);
				const unsigned base = incl - sum;
				wave_lds_sync();
				off[4 * lane] = base; off[4 * lane + 1] = base + v0; off[4 * lane + 2] = base + v0 + v1; off[4 * lane + 3] = base + v0 + v1 + v2;
				wave_lds_sync();
				// stable scatter, 64 entries at a time in range order
				for (int i0 = 0; i0 < cnt; i0 += 64)
				{
					const int i = i0 + lane;
					const bool valid = i < cnt;
					const uint64_t key = valid ? src[i] : 0ull;
					const unsigned dg = (unsigned)(key >> bit) & 255u;
					uint64_t same = __ballot(valid);   // lanes holding the same digit
#pragma unroll
					for (int b = 0; b < 8; ++b)
					{
						const uint64_t bal = __ballot((dg >> b) & 1u);
						same &= ((dg >> b) & 1u) ? bal : ~bal;
					}
					const unsigned before = (unsigned)__popcll(same & below);
					const unsigned o = valid ? off[dg] : 0u;
					if (valid) dst[o + before] = key;
					wave_lds_sync();
					if (valid && before == 0u) off[dg] = o + (unsigned)__popcll(same);
					wave_lds_sync();
				}
				// the next pass reads what this one wrote (same wave, through memory)
				__threadfence_block();
				uint64_t *t2 = src; src = dst; dst = t2;
			}
			// `src` holds the sorted range now
			int nf = 0, nq = 0;
			for (int i0 = 0; i0 < cnt; i0 += 64)
			{
				const int i = i0 + lane;
				const uint64_t key = i < cnt ? src[i] : 0ull;
				const int sv = (int)((key >> lowbit) & smask);
				if (i < cnt) emit(t, s + i, key, DESC ? leaf_index[min(sv, ntargets - 1)] : 0, DESC ? leaf_mult[min(sv, ntargets - 1)] : 0);
				nf += __popcll(__ballot(i < cnt && sv < mu.self0));
				nq += __popcll(__ballot(i < cnt && sv < t));
			}
			if (DESC && mu.desc4) finish_target(t, s, cnt, nf, nq, lane);
		}
	}
}

// directed pair interactions = sum over the directed P2P entries of mult[target] * mult[source]
// (the self entries contribute mult^2), SURVEY 8(d)
__global__ __launch_bounds__(kBlock) void pair_count_kernel(const int *__restrict__ leaf_mult, const uint64_t *__restrict__ keys,
                                                            const int *__restrict__ count_ptr, int shift, unsigned long long *__restrict__ out)
{
	const long long count = *count_ptr;
	const uint64_t mask = (1ull << shift) - 1;
	unsigned long long s = 0;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock)
		s += (unsigned long long)leaf_mult[(int)(keys[i] >> shift)] * (unsigned long long)leaf_mult[(int)(keys[i] & mask)];
	for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
	if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// ---- P2P ---------------------------------------------------------------------------------------------
// Work units.  The number of source leaves per target leaf is very uneven (Gaussian ball, N = 1M: mean 17,
// max > 250), so one wave per target leaf leaves a tail as long as the rest of the kernel.  Every
// target leaf's sorted source range is therefore cut into chunks of at most kP2PChunk entries; a wave
// evaluates one chunk and stores the partial sums of its 32 targets, and the L2P kernel adds a leaf's
// chunks in list order (fixed order: still bit-reproducible, still no atomics).  Source descriptors and chunk
// counts come out of the per-target sort (list_segsort_kernel<true>).
// tree order for the caller's state in one pass: positions unpacked to xyz triplets, velocities gathered into a scratch
// copy (the gather cannot run in place)
__global__ __launch_bounds__(kBlock) void reorder_state_kernel(const float4 *__restrict__ pos, const int *__restrict__ unsort,
                                                               const float *__restrict__ v_in, float *__restrict__ p_out, float *__restrict__ v_tmp,
                                                               long long n, const int *__restrict__ order_in, int *__restrict__ order_out)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		const float4 q = pos[i];
		p_out[3 * i] = q.x; p_out[3 * i + 1] = q.y; p_out[3 * i + 2] = q.z;
		const long long s = unsort[i];
		v_tmp[3 * i] = v_in[3 * s]; v_tmp[3 * i + 1] = v_in[3 * s + 1]; v_tmp[3 * i + 2] = v_in[3 * s + 2];
		// opts.track_order: the particle now at position i was at position s before this rebuild, i.e. it is particle
		// order_in[s] of the state the tracking started from (order_in == nullptr: this is the first permutation)
		if (order_out) order_out[i] = order_in ? order_in[s] : (int)s;
	}
}

// sorted positions back to xyz triplets
__global__ __launch_bounds__(kBlock) void unpack4_kernel(const float4 *__restrict__ src, float *__restrict__ dst, long long n)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		float4 p = src[i];
		dst[3 * i] = p.x; dst[3 * i + 1] = p.y; dst[3 * i + 2] = p.z;
	}
}

static int grid1d(long long n, int cap = 2048)
{
	long long b = (n + kBlock - 1) / kBlock;
	if (b < 1) b = 1;
	if (b > cap) b = cap;
	return (int)b;
}


// ---- host side ----------------------------------------------------------------------------------------

// kd levels, fmm_cart3_kdtree.cuh:1502-1515 (GPU driver honours tree_L; the CPU driver's formula is the same otherwise)
static int kd_levels(long long n, int p, float dens_inhom, int tree_L)
{
	int L;
	if (tree_L == 0)
	{
		float s = (float)(p * p);
		L = (int)std::round(std::log2(dens_inhom * (float)n / s));
	}
	else L = tree_L;
	L = std::max(2, std::min(30, L));
	while ((1LL << L) > n) --L;
	return L;
}

static TreeView view_of(const KdTreeDev &k)
{
	TreeView t;
	t.center = k.center; t.lbound = k.lbound; t.rbound = k.rbound; t.csz = k.csz; t.mpole = k.mpole; t.local = k.local;
	t.mult = k.mult; t.index = k.index; t.splitdim = k.splitdim; t.L = k.L; t.ntot = k.ntot;
	return t;
}

static int sort_pairs_u64(nbco_ctx *c, uint64_t *kin, uint64_t *kout, uint32_t *vin, uint32_t *vout, long long n, int end_bit)
{
	size_t bytes = 0;
	NBCO_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, c->stream));
	NBCO_TRY(c->reserve(c->sort_tmp, bytes));
	bytes = c->sort_tmp.bytes;
	NBCO_HIP(rocprim::radix_sort_pairs(c->sort_tmp.ptr, bytes, kin, kout, vin, vout, (size_t)n, 0u, (unsigned)end_bit, c->stream));
	return NBCO_OK;
}

// directed, per-target sorted list of `pairs` (+ one self entry for each of the targets [self0, self0 + nself)) into
// keys_out; start[0..T].  cnt[0..T) holds the per-target pair-entry counts accumulated by the traversal, fill[0..T) is zero.
static int exclusive_scan_ints(nbco_ctx *c, int *in, int *out, size_t count, DevBuf &tmp)
{
	// (a one-workgroup scan in a single launch was tried for these 32K..64K-element arrays: 3x slower than rocPRIM's two launches)
	hipStream_t st = c->stream;
	size_t bytes = 0;
	NBCO_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, count, rocprim::plus<int>(), st));
	NBCO_TRY(c->reserve(tmp, bytes));
	bytes = tmp.bytes;
	NBCO_HIP(rocprim::exclusive_scan(tmp.ptr, bytes, in, out, 0, count, rocprim::plus<int>(), st));
	return NBCO_OK;
}

// One scan for two prefix sums of the P2P list: entries per target (low word -> start[]) and chunks per target (high
// word -> chunk_off[]); the chunk count follows from the entry count, so the work-unit table no longer waits for the sort.
struct PackCounts
{
	__host__ __device__ uint64_t operator()(unsigned cnt) const { return (uint64_t)cnt | ((uint64_t)((cnt + kP2PChunk - 1) / kP2PChunk) << 32); }
};
struct SplitRef
{
	int *s, *o;
	__host__ __device__ SplitRef &operator=(uint64_t v) { *s = (int)(uint32_t)v; *o = (int)(v >> 32); return *this; }
};
struct SplitIter
{
	using iterator_category = std::random_access_iterator_tag;
	using value_type = uint64_t;
	using difference_type = std::ptrdiff_t;
	using pointer = uint64_t *;
	using reference = SplitRef;
	int *s, *o;
	__host__ __device__ SplitRef operator*() const { return SplitRef{s, o}; }
	__host__ __device__ SplitRef operator[](difference_type i) const { return SplitRef{s + i, o + i}; }
	__host__ __device__ SplitIter operator+(difference_type i) const { return SplitIter{s + i, o + i}; }
	__host__ __device__ SplitIter operator-(difference_type i) const { return SplitIter{s - i, o - i}; }
	__host__ __device__ SplitIter &operator+=(difference_type i) { s += i; o += i; return *this; }
	__host__ __device__ SplitIter &operator++() { ++s; ++o; return *this; }
	__host__ __device__ difference_type operator-(const SplitIter &b) const { return s - b.s; }
};
static int exclusive_scan_counts_and_chunks(nbco_ctx *c, const unsigned *cnt, int *start, int *chunk_off, size_t count, DevBuf &tmp)
{
	hipStream_t st = c->stream;
	auto in = rocprim::make_transform_iterator(cnt, PackCounts{});
	size_t bytes = 0;
	NBCO_HIP(rocprim::exclusive_scan(nullptr, bytes, in, SplitIter{start, chunk_off}, (uint64_t)0, count, rocprim::plus<uint64_t>(), st));
	NBCO_TRY(c->reserve(tmp, bytes));
	bytes = tmp.bytes;
	NBCO_HIP(rocprim::exclusive_scan(tmp.ptr, bytes, in, SplitIter{start, chunk_off}, (uint64_t)0, count, rocprim::plus<uint64_t>(), st));
	return NBCO_OK;
}

static int build_directed_list(nbco_ctx *c, const int2 *pairs, const int2 *ranks, const int *pref_dev, long long capR, long long npairs_hint, int sub,
                               int self0, int nself, int ntargets, int shift, unsigned *cnt, int *start, uint64_t *keys_tmp, uint64_t *keys_out,
                               DevBuf &scan_tmp, const int *leaf_index = nullptr, const int *leaf_mult = nullptr, int2 *desc = nullptr,
                               int *chunk_off = nullptr, int4 *chunks = nullptr, const MutualLists *mutual = nullptr)
{
	hipStream_t st = c->stream;   // (the self entries were added to cnt by traverse_finish_kernel)
	if (desc)
	{
		// P2P list: entry offsets and chunk offsets from one scan (the work-unit table is written by the fill kernel)
		NBCO_TRY(exclusive_scan_counts_and_chunks(c, cnt, start, chunk_off, (size_t)(ntargets + 1), scan_tmp));
	}
	else
		NBCO_TRY(exclusive_scan_ints(c, (int *)cnt, start, (size_t)(ntargets + 1), scan_tmp));
	hipLaunchKernelGGL(list_fill_kernel, dim3(grid1d(npairs_hint + nself)), dim3(kBlock), 0, st, pairs, ranks, pref_dev, capR, sub, self0, nself, shift,
	                   (const int *)start, keys_tmp, (const int *)chunk_off, ntargets, leaf_index, leaf_mult,
	                   (desc && !mutual) ? chunks : (int4 *)nullptr, mutual ? 1 : 0);
	if (desc)
	{
		MutualLists mu;
		if (mutual) { mu = *mutual; mu.chunk = chunks; mu.chunk_off = chunk_off; mu.self0 = self0; mu.nself = nself; }
		hipLaunchKernelGGL(list_segsort_kernel<true>, dim3((ntargets + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, (const int *)start, ntargets,
		                   keys_tmp, keys_out, shift, leaf_index, leaf_mult, desc, mu);
	}
	else
		hipLaunchKernelGGL(list_segsort_kernel<false>, dim3((ntargets + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, (const int *)start,
		                   ntargets, keys_tmp, keys_out, shift, (const int *)nullptr, (const int *)nullptr, (int2 *)nullptr, MutualLists{});
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

// node arrays of a tree with ntot nodes carved out of `buf`
static int kd_carve(nbco_ctx *c, DevBuf &buf, KdTreeDev &k, int ntot, int offM, int offL)
{
	// (opts.far_fp64: the multipole / local tuples are doubles; everything else of the tree stays fp32)
	const size_t rb = c->o.far_fp64 ? 8 : 4;
	k.real_bytes = (int)rb;
	size_t bytes = (size_t)ntot * (3 * 3 * sizeof(float) + sizeof(float4) + (size_t)(offM + offL) * rb + 3 * sizeof(int)) + 256 + 64;
	NBCO_TRY(c->reserve(buf, bytes));
	char *q = (char *)buf.ptr;
	k.csz = (float4 *)q; q += sizeof(float4) * (size_t)ntot;
	k.center = (float *)q; q += 12 * (size_t)ntot;
	k.lbound = (float *)q; q += 12 * (size_t)ntot;
	k.rbound = (float *)q; q += 12 * (size_t)ntot;
	q = (char *)(((uintptr_t)q + 15) & ~(uintptr_t)15);   // (the double tuples want 8-byte alignment: 36 ntot bytes of fp32 geometry precede them)
	k.mpole = (float *)q; q += rb * (size_t)ntot * offM;
	q = (char *)(((uintptr_t)q + 15) & ~(uintptr_t)15);
	k.local = (float *)q; q += rb * (size_t)ntot * offL;
	q = (char *)(((uintptr_t)q + 15) & ~(uintptr_t)15);
	k.mult = (int *)q; q += 4 * (size_t)ntot;
	k.index = (int *)q; q += 4 * (size_t)ntot;
	k.splitdim = (int *)q;
	return NBCO_OK;
}

// Levels [0, l0) of a tree over pos[0..n): exact median selection + partition (k_kdselect.hip) or, after a tie
// overflow, one global stable radix sort per level.  The root's box / split axis must be in place; on return
// pos / unsort point at the buffers holding the result and the boxes of level l0 are written.
static int kd_build_top(nbco_ctx *c, const TreeView &tv, float4 *&pos, float4 *&pos_alt, int *&unsort, int *&unsort_alt, long long n, int l0,
                        bool use_select, bool select_ready = false, bool warm = false)
{
	hipStream_t st = c->stream;
	if (use_select && l0 > 0 && !select_ready) NBCO_TRY(kd_select_begin(c, l0));
	for (int l = 0; l < l0; ++l)
	{
		if (use_select)
			NBCO_TRY(kd_select_level(c, l, n, pos, unsort, pos_alt, unsort_alt, tv.lbound, tv.rbound, tv.splitdim, tv.index,
			                         c->counters.as<int>() + 110, warm));
		else
		{
			if (l > 0) hipLaunchKernelGGL(kd_box_kernel, dim3(grid1d(kd_cnt(l))), dim3(kBlock), 0, st, tv, pos, n, l);
			hipLaunchKernelGGL(kd_keys_kernel, dim3(grid1d(n)), dim3(kBlock), 0, st, pos, tv.splitdim + kd_beg(l), n, l,
			                   c->keys.as<uint64_t>(), c->idx.as<uint32_t>());
			NBCO_TRY(sort_pairs_u64(c, c->keys.as<uint64_t>(), c->keys_alt.as<uint64_t>(), c->idx.as<uint32_t>(), c->idx_alt.as<uint32_t>(), n, 32 + l));
			hipLaunchKernelGGL(kd_permute_kernel, dim3(grid1d(n)), dim3(kBlock), 0, st, pos, unsort, c->idx_alt.as<uint32_t>(), pos_alt, unsort_alt, n);
		}
		std::swap(pos, pos_alt);
		std::swap(unsort, unsort_alt);
	}
	if (l0 > 0 && !use_select) hipLaunchKernelGGL(kd_box_kernel, dim3(grid1d(kd_cnt(l0))), dim3(kBlock), 0, st, tv, pos, n, l0);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

static int kd_reserve_particles(nbco_ctx *c, long long n)
{
	NBCO_TRY(c->reserve(c->pos4, sizeof(float4) * (size_t)n));
	NBCO_TRY(c->reserve(c->pos4_alt, sizeof(float4) * (size_t)n));
	NBCO_TRY(c->reserve(c->unsort, sizeof(int) * (size_t)n));
	NBCO_TRY(c->reserve(c->unsort_alt, sizeof(int) * (size_t)n));
	NBCO_TRY(c->reserve(c->keys, sizeof(uint64_t) * (size_t)n));
	NBCO_TRY(c->reserve(c->keys_alt, sizeof(uint64_t) * (size_t)n));
	NBCO_TRY(c->reserve(c->idx, sizeof(uint32_t) * (size_t)n));
	NBCO_TRY(c->reserve(c->idx_alt, sizeof(uint32_t) * (size_t)n));
	NBCO_TRY(c->reserve(c->counters, sizeof(int) * 128));
	return NBCO_OK;
}

// ---- stage 1: tree over p[0..n) with L levels + upward pass ---------------------------------------------
// root6 (device, {lbound, rbound}) overrides the root box: the box of a kd-domain is inherited from the
// global tree's top splits.  On return c->pos4 / c->unsort hold the tree-ordered positions and the map
// back to the caller's order.
// stage: 0 = build + upward, 1 = build only, 2 = upward only (of the tree built by the previous stage-1 call)
static int kd_build_upward(nbco_ctx *c, const float *p, long long n, int L, const float *root6, bool &rebuild, int stage = 0)
{
	const int P = c->o.fmm_order;
	const int ntot = (1 << (L + 1)) - 1, nleaf = 1 << L;
	const int mlt_max = (int)((n - 1) / nleaf + 1);
	hipStream_t st = c->stream;
	if (stage != 2)
	{
		NBCO_TRY(c->join_aux());   // e.g. the multipole chain of an evaluation that is being redone
		KdTreeDev &k = c->kd;
		const int old_real = k.real_bytes;
		NBCO_TRY(kd_carve(c, c->treebuf, k, ntot, sym_off(P), tl_off(P + 1)));
		const bool topo_change = k.L != L || k.ntot != ntot || k.order != P || k.n != n || k.real_bytes != old_real;
		if (topo_change) c->tree_valid = false;
		k.L = L; k.ntot = ntot; k.order = P; k.mlt_max = mlt_max; k.n = n;
		NBCO_TRY(kd_reserve_particles(c, n));
		rebuild = c->o.unsort || !c->tree_valid || (c->eval_counter % c->o.tree_steps) == 0;
	}
	TreeView tv = view_of(c->kd);
	float4 *pos = c->pos4.as<float4>(), *pos_alt = c->pos4_alt.as<float4>();
	int *unsort = c->unsort.as<int>(), *unsort_alt = c->unsort_alt.as<int>();
	if (stage != 2)
	{
		PhaseScope ph(c, NBCO_PH_BUILD);
		if (rebuild)
		{
			// levels whose nodes exceed the LDS slice
			int l0 = 0;
			while (l0 < L && (n + (1LL << l0) - 1) / (1LL << l0) > kSubS) ++l0;
			const bool use_select = !c->force_sort_build;
			long long words_a = 0, words_b = 0;
			if (use_select && l0 > 0) NBCO_TRY(kd_select_begin(c, l0, false, &words_a, &words_b));
			if (!c->prep_state.ptr)
			{
				NBCO_TRY(c->reserve(c->prep_state, 64));
				const unsigned arm[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
				NBCO_HIP(hipMemcpyAsync(c->prep_state.ptr, arm, sizeof arm, hipMemcpyHostToDevice, st));
				NBCO_HIP(hipStreamSynchronize(st));   // `arm` lives on this stack frame; happens once per context
			}
			// A pivot with more ties than the resolver takes (flag -> the sorting build redoes the evaluation) leaves slots
			// of a level's output unwritten, and the rest of this evaluation still runs on them: every slot of both
			// permutation buffers must hold an index below n at all times.  From the second level on the buffers hold older
			// permutations; the alternate buffer is primed once per particle count.
			if (c->perm_primed_n != n)
			{
				hipLaunchKernelGGL(iota_kernel, dim3(grid1d(n)), dim3(kBlock), 0, st, unsort_alt, n);
				hipLaunchKernelGGL(iota_kernel, dim3(grid1d(n)), dim3(kBlock), 0, st, unsort, n);
				c->perm_primed_n = n;
			}
			// pack + identity permutation + bounding box + root node + cleared selection state: one launch
			// (nbco_integrate_steps: the pass between two steps has done all of it, kd_turnaround)
			if (c->skip_prep != 1)
			hipLaunchKernelGGL(kd_prep_kernel, dim3(kPrepGrid), dim3(kPrepBlock), 0, st, p, n, pos, unsort, c->sel_hist.as<uint32_t>(), words_a,
			                   c->sel_nodes.as<uint32_t>(), words_b, c->counters.as<int>() + 110, c->prep_state.as<unsigned>(), tv, root6);
			// the previous build's boxes are still in the tree arrays: select around its pivots (one pass per level instead of two)
			bool warm = use_select && l0 > 0 && c->sel_warm_enabled && c->tree_valid && !c->sel_three_pass;
			if (warm && c->sel_warm_cooldown > 0) { --c->sel_warm_cooldown; warm = false; }
			c->sel_warm_used = warm;
			if (warm) ++c->sel_warm_builds;
			NBCO_TRY(kd_build_top(c, tv, pos, pos_alt, unsort, unsort_alt, n, l0, use_select, true, warm));
			// the rest of every level-l0 subtree inside one workgroup's LDS
			hipLaunchKernelGGL(kd_subtree_kernel, dim3(kd_cnt(l0)), dim3(kSubT), 0, st, tv, (const float4 *)pos, (const int *)unsort, pos_alt, unsort_alt, n, l0,
			                   use_select ? 1 : 0, c->sel_three_pass ? 0 : 1, c->counters.as<int>() + 110);
#ifdef NBCO_SUBTREE_PROF
			// the kernel only reads its inputs: a second launch right behind the first one repeats it with warm instruction caches
			if (std::getenv("NBCO_SUBTREE_TWICE"))
				hipLaunchKernelGGL(kd_subtree_kernel, dim3(kd_cnt(l0)), dim3(kSubT), 0, st, tv, (const float4 *)pos, (const int *)unsort, pos_alt, unsort_alt, n, l0,
				                   use_select ? 1 : 0, c->sel_three_pass ? 0 : 1, c->counters.as<int>() + 110);
#endif
			std::swap(pos, pos_alt);
			std::swap(unsort, unsort_alt);
			NBCO_HIP(hipGetLastError());
			// keep the "current" buffers in the primary slots
			if (pos != c->pos4.as<float4>()) { std::swap(c->pos4, c->pos4_alt); std::swap(c->unsort, c->unsort_alt); }
			pos = c->pos4.as<float4>();
		}
		else
		{
			// tree reused: the caller's positions are already in tree order
			c->sel_warm_used = false;
			if (c->skip_prep == 0)
			{
				NBCO_TRY(launch_pack4(c, pos, p, n));
				NBCO_HIP(hipMemsetAsync(c->counters.as<int>() + 110, 0, sizeof(int), st));
			}
		}
		c->skip_prep = 0;
		if (!rebuild) hipLaunchKernelGGL(kd_leaf_kernel, dim3(grid1d(nleaf)), dim3(kBlock), 0, st, tv, pos, n);   // (a rebuild's subtree kernel did it)
		// centres and traversal records of all nodes first (2 launches): that is all the traversal needs, so the multipole
		// chain (P2M + M2M, generated register-resident bodies of k_farfield.hip) runs beside it on the second stream
		NBCO_TRY(launch_kd_centres(c, tv.center, tv.mult, L, tv.lbound, tv.rbound, tv.csz));
		NBCO_HIP(hipGetLastError());
	}
	if (stage == 1) return NBCO_OK;
	{
		NBCO_TRY(c->fork_aux());
		StreamScope on_aux(c, c->aux);
		PhaseScope ph(c, NBCO_PH_P2M_M2M);
		NBCO_TRY(launch_upward_gen(c, P, pos, tv.center, tv.mpole, tv.mult, tv.index, L, 0, c->kd.real_bytes == 8));
	}
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

// LET exchange of the sharded evaluation (see "Locally-essential-tree exchange" below): what has arrived, per global node / leaf
struct LetHave { const unsigned char *node, *leaf; };
constexpr int kLetWord = 16;   // h_flags[16], [17]: the guard's two words
// every source of a sorted directed list (key = target << shift | source) must have arrived; host_word = 1 + a missing id
__global__ __launch_bounds__(kBlock) void let_guard_kernel(const uint64_t *__restrict__ keys, const int *__restrict__ total, int shift,
                                                           const unsigned char *__restrict__ have, int *__restrict__ host_word)
{
	const uint64_t mask = (1ull << shift) - 1;
	const int n = *total;
	int missing = 0;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
	{
		const int id = (int)(keys[i] & mask);
		if (!have[id]) missing = id + 1;
	}
	if (missing) { *host_word = missing; __threadfence_system(); }
}
static int launch_let_guard(nbco_ctx *c, const uint64_t *keys, const int *total, int shift, const unsigned char *have, int word, long long hint)
{
	hipLaunchKernelGGL(let_guard_kernel, dim3(grid1d(hint, 2048)), dim3(kBlock), 0, c->stream, keys, total, shift, have, c->h_flags + kLetWord + word);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}


// ---- stage 2: traversal, lists, P2P, M2L, L2L, L2P on the tree `tv` over pos[0..n) --------------------------
// Only targets touching the domain `dm` are served; accelerations come out for the particles
// [own0, own0 + own_n) (tree order, or scattered through `unsort` when opts.unsort is set).
struct KdCounts
{
	long long np2p = 0, nm2l = 0;
	int sel_overflow = 0, react_overflow = 0;
};

// phase 0: everything; 1: up to and including the traversal (its flags on their way to the host); 2: the rest, for a tree on
// which phase 1 has run.  `pre_far` (phase 0 / 2) is called on the second stream ahead of the far-field chain: the sharded
// evaluation unpacks the multipoles there, which arrive after the traversal has started.
static int kd_interact(nbco_ctx *c, const TreeView &tv, const float4 *pos, long long n, int mlt_max, const Dom dm, long long own0, long long own_n,
                       const int *unsort, float *a, const float *param, KdCounts &out, int phase = 0, const std::function<int()> *pre_far = nullptr,
                       const LetHave *let = nullptr)
{
	const int P = c->o.fmm_order;
	const int L = tv.L, ntot = tv.ntot, nleaf = 1 << L, beg = kd_beg(L);
	const int offL = tl_off(P + 1);
	out = KdCounts{};
	const int self0 = dm.g << (L - dm.d), nself = 1 << (L - dm.d);   // the domain's own leaves
	hipStream_t st = c->stream;

	// capacity of the frontier and of the two pair lists, split into kTravK regions
	// (regions fill unevenly: each gets twice its share of list_factor * ntot pairs)
	const long long capR = 2 * (((long long)c->o.list_factor * c->list_growth * ntot + 4096 + kTravK - 1) / kTravK), cap = capR * kTravK;
	NBCO_TRY(c->reserve(c->frontier_a, sizeof(int2) * (size_t)cap));
	NBCO_TRY(c->reserve(c->frontier_b, sizeof(int2) * (size_t)cap));
	// pairs [0, cap) and, behind them, the slot of either direction inside its target's range [cap, 2 cap)
	NBCO_TRY(c->reserve(c->p2p_list, sizeof(int2) * 2 * (size_t)cap));
	NBCO_TRY(c->reserve(c->m2l_list, sizeof(int2) * 2 * (size_t)cap));
	NBCO_TRY(c->reserve(c->counters, sizeof(int) * 128));
	c->list_cap = cap;

	// ---- dual tree traversal ----------------------------------------------------------------------------
	if (phase != 2)
	{
		PhaseScope ph(c, NBCO_PH_TRAVERSE);
		AdmTab tab;
		for (int l = 0; l < 32; ++l)
		{
			long long lo = l <= L ? (n >> l) : 0;
			long long hi = l <= L ? ((n + (1LL << l) - 1) >> l) : 0;
			tab.lo[l] = (int)lo;
			const float e = 1.f / (float)(3 * P + 6);
			tab.Mlo[l] = lo > 0 ? std::pow((float)lo / (float)n, e) : 0.f;   // fmm_cart3_kdtree.cuh:410
			tab.Mhi[l] = hi > 0 ? std::pow((float)hi / (float)n, e) : 0.f;
		}
		int *ctr = c->counters.as<int>();
		int2 *fa = c->frontier_a.as<int2>(), *fb = c->frontier_b.as<int2>();
		// per-target entry counters / fill cursors of the two directed lists: [cnt_p2p | fill_p2p | cnt_m2l | fill_m2l]
		const size_t np_ = (size_t)nleaf + 2, nm_ = (size_t)ntot + 2;
		NBCO_TRY(c->reserve(c->list_cnt, sizeof(unsigned) * 2 * (np_ + nm_)));

		unsigned *cnt_p2p = c->list_cnt.as<unsigned>(), *cnt_m2l = cnt_p2p + 2 * np_;
		NBCO_TRY(c->reserve(c->trav_ctr, sizeof(int) * 1024));
		static_assert(kTcInts <= 1024, "traversal counter block");
		int *tctr = c->trav_ctr.as<int>();
		NBCO_TRY(c->flags_begin());
		hipLaunchKernelGGL(traverse_init_kernel, dim3(grid1d((long long)(2 * (np_ + nm_)) / 8 + 1, 256)), dim3(kBlock), 0, st, fa, ctr, 104, tctr,
		                   c->list_cnt.as<unsigned>(), (long long)(2 * (np_ + nm_)), (long long)self0, (long long)(c->o.coll ? nself : 0));
		// (counters[110] is the selection-build flag)
		const int iters = L + NBCO_TRAV_EXTRA;   // every launch performs two traversal steps; traverse_finish_kernel checks that none is left
		for (int it = 0; it < iters; ++it)
		{
			hipLaunchKernelGGL(traverse_kernel, dim3(1024), dim3(kBlock), 0, st, tv, tab, (const int2 *)fa, fb, c->p2p_list.as<int2>(),
			                   c->m2l_list.as<int2>(), c->p2p_list.as<int2>() + cap, c->m2l_list.as<int2>() + cap, ctr, tctr, it, capR,
			                   c->o.tree_radius, c->o.m2l_first, cnt_p2p, cnt_m2l, dm);
			std::swap(fa, fb);
		}
		hipLaunchKernelGGL(traverse_finish_kernel, dim3(1), dim3(1024), 0, st, ctr, tctr, capR, c->list_cnt.as<unsigned>(), (long long)(2 * (np_ + nm_)),
		                   cnt_p2p + self0, c->o.coll ? nself : 0, c->h_flags, iters, ++c->flags_seq);
		NBCO_HIP(hipGetLastError());
		// counts and flags are in pinned host memory once this event has passed; the host looks at them only after it has
		// enqueued the rest of the evaluation (every later kernel takes its counts from the device), so the GPU never
		// waits for a host round trip
		NBCO_HIP(hipEventRecord(c->ev_flags, st));
	}
	if (phase == 1) return NBCO_OK;
	const int *p2p_pref = c->trav_ctr.as<int>() + kTcP2PPref, *m2l_pref = c->trav_ctr.as<int>() + kTcM2LPref;
	const int shift = L + 1;
	// capacities (the traversal never writes more than `cap` pairs) and launch-size hints from the previous evaluation
	const long long dp2p_cap = c->o.coll ? 2 * cap + nself : 0, dm2l_cap = 2 * cap;
	const long long np2p_hint = c->hint_np2p > 0 ? std::min(cap, c->hint_np2p + c->hint_np2p / 4 + 1024) : cap;
	const long long nm2l_hint = c->hint_nm2l > 0 ? std::min(cap, c->hint_nm2l + c->hint_nm2l / 4 + 1024) : cap;
	const long long dp2p_hint = 2 * np2p_hint + nself;
	const long long max_chunks = c->o.coll ? dp2p_cap / kP2PChunk + nself : 0, chunks_hint = dp2p_hint / kP2PChunk + nself;
	// Near field in the mutual (Newton III) form when the leaves fill two 16-lane rows (k_p2p.hpp).  Its reaction records are
	// indexed by the unordered pair: one per pair of the P2P list, sized from the previous evaluation's count -- the first
	// evaluation of a context waits for the traversal's count instead (once), and an evaluation whose list outgrew the buffer
	// is repeated by the caller (the kernels never write beyond it).
	const int mutual_th = (c->o.coll && c->o.p2p_mutual) ? p2p_mutual_halves(mlt_max) : 0;   // 32-particle halves per leaf, 0 = one-directional kernel
	const bool mutual = mutual_th > 0;
	c->info.p2p_halves = mutual_th;
	long long react_cap = 0;
	if (mutual)
	{
		if (c->hint_np2p <= 0)
		{
			NBCO_TRY(c->wait_flags());
			c->hint_np2p = c->h_flags[0];
			c->hint_nm2l = c->h_flags[1];
		}
		// one record slot per DIRECTED entry (only the slots of entries delivered by other waves are ever written)
		const long long pairs_room = std::min(cap, c->hint_np2p + c->hint_np2p / 4 + 1024);
		react_cap = 2 * pairs_room + nself;
		NBCO_TRY(c->reserve(c->p2p_react, sizeof(float4) * 32 * (size_t)mutual_th * (size_t)react_cap));
	}
	float4 *near = nullptr;
	auto enqueue_p2p = [&]() -> int {
		NBCO_TRY(c->reserve(c->part, sizeof(float4) * (size_t)max_chunks * (size_t)mlt_max));
		near = c->part.as<float4>();
		PhaseScope ph(c, NBCO_PH_P2P);
		const int4 *pc = c->p2p_chunks.as<int4>();
		const int *pt = c->p2p_chunk_off.as<int>() + nleaf;   // total number of chunks
		const int2 *pd = c->p2p_desc.as<int2>();
		if (mutual)
		{
			launch_p2p_mutual(c, mutual_th, pos, c->p2p_desc.as<int4>(), pc, pt, chunks_hint, mlt_max, near, c->p2p_react.as<float4>(), react_cap, n);
		}
		else if (mlt_max <= 8) launch_p2p<8>(c, pos, pd, pc, pt, chunks_hint, mlt_max, mlt_max, near, n);
		else if (mlt_max <= 16) launch_p2p<16>(c, pos, pd, pc, pt, chunks_hint, mlt_max, mlt_max, near, n);
		else if (mlt_max <= 32) launch_p2p<32>(c, pos, pd, pc, pt, chunks_hint, mlt_max, mlt_max, near, n);
		else launch_p2p<64>(c, pos, pd, pc, pt, chunks_hint, mlt_max, mlt_max, near, n);
		NBCO_HIP(hipGetLastError());
		// diagnostics (tools/p2p_lab.hip): NBCO_P2P_DUMP=<file> writes the inputs and the partial sums of the near-field launch
		static const char *dump_path = std::getenv("NBCO_P2P_DUMP");
		if (dump_path && !mutual) NBCO_TRY(p2p_dump(c, dump_path, pos, pd, pc, pt, c->p2p_start.as<int>() + nleaf, mlt_max, near, n));
		return NBCO_OK;
	};
	// ---- directed sorted lists --------------------------------------------------------------------------
	{
		NBCO_TRY(c->reserve(c->p2p_keys, sizeof(uint64_t) * (size_t)(dp2p_cap + 1)));
		NBCO_TRY(c->reserve(c->p2p_keys_alt, sizeof(uint64_t) * (size_t)(dp2p_cap + 1)));
		NBCO_TRY(c->reserve(c->m2l_keys, sizeof(uint64_t) * (size_t)(dm2l_cap + 1)));
		NBCO_TRY(c->reserve(c->m2l_keys_alt, sizeof(uint64_t) * (size_t)(dm2l_cap + 1)));
		NBCO_TRY(c->reserve(c->p2p_start, sizeof(int) * (size_t)(nleaf + 2)));
		NBCO_TRY(c->reserve(c->m2l_start, sizeof(int) * (size_t)(ntot + 2)));
		// critical path first (the host needs several microseconds per launch): the P2P list chain on the main stream;
		// the far-field chain on the second stream only has to wait for the traversal (marked here)
		NBCO_TRY(c->fork_mark());
		if (c->o.coll)
		{
			PhaseScope ph(c, NBCO_PH_LISTS);
			unsigned *cp = c->list_cnt.as<unsigned>();
			NBCO_TRY(c->reserve(c->p2p_desc, (mutual ? sizeof(int4) : sizeof(int2)) * (size_t)(dp2p_cap + 1)));
			MutualLists mu;
			if (mutual)
			{
				NBCO_TRY(c->reserve(c->p2p_sec, sizeof(int2) * (size_t)(nleaf + 2)));
				mu.desc4 = c->p2p_desc.as<int4>();
				mu.sec_range = c->p2p_sec.as<int2>();
			}
			NBCO_TRY(c->reserve(c->p2p_chunk_off, sizeof(int) * (size_t)(nleaf + 2)));
			NBCO_TRY(c->reserve(c->p2p_chunks, sizeof(int4) * (size_t)max_chunks));
			NBCO_TRY(build_directed_list(c, c->p2p_list.as<int2>(), c->p2p_list.as<int2>() + cap, p2p_pref, capR, np2p_hint, beg, self0, nself, nleaf, shift, cp,
			                             c->p2p_start.as<int>(), c->p2p_keys.as<uint64_t>(), c->p2p_keys_alt.as<uint64_t>(), c->sort_tmp,
			                             tv.index + beg, tv.mult + beg, c->p2p_desc.as<int2>(), c->p2p_chunk_off.as<int>(), c->p2p_chunks.as<int4>(),
			                             mutual ? &mu : nullptr));
			if (mutual) launch_p2p_link(c, c->p2p_desc.as<int4>(), c->p2p_keys_alt.as<uint64_t>(), c->p2p_start.as<int>(), c->p2p_start.as<int>() + nleaf, shift, dp2p_hint);
			if (let) NBCO_TRY(launch_let_guard(c, c->p2p_keys_alt.as<uint64_t>(), c->p2p_start.as<int>() + nleaf, shift, let->leaf, 1, dp2p_hint));
			// remembered for nbco_kd_get_info (the directed pair count is evaluated on demand)
			c->pc_mult = tv.mult + beg; c->pc_shift = shift; c->pc_total = c->p2p_start.as<int>() + nleaf;
		}
		if (c->o.coll) NBCO_TRY(enqueue_p2p());   // before the host spends its time on the far-field chain below: the pair kernel is next on this stream
		// the far field does not depend on the P2P list: M2L list, M2L and L2L run on the second stream, behind the
		// multipole chain, and overlap the P2P list chain and the start of P2P
		// (the locals are cleared on the second stream before it starts waiting for the traversal)
		const int f64 = c->o.far_fp64 ? 1 : 0;   // (the tree this view belongs to was carved under the same option)
		NBCO_HIP(hipMemsetAsync(tv.local, 0, (f64 ? sizeof(double) : sizeof(float)) * (size_t)ntot * offL, c->aux));
		NBCO_TRY(c->fork_wait());
		{
			StreamScope on_aux(c, c->aux);
			if (pre_far) NBCO_TRY((*pre_far)());
			unsigned *cm = c->list_cnt.as<unsigned>() + 2 * ((size_t)nleaf + 2);
			NBCO_TRY(build_directed_list(c, c->m2l_list.as<int2>(), c->m2l_list.as<int2>() + cap, m2l_pref, capR, nm2l_hint, 0, 0, 0, ntot, shift, cm,
			                             c->m2l_start.as<int>(), c->m2l_keys.as<uint64_t>(), c->m2l_keys_alt.as<uint64_t>(), c->scan_tmp_aux));
			if (let) NBCO_TRY(launch_let_guard(c, c->m2l_keys_alt.as<uint64_t>(), c->m2l_start.as<int>() + ntot, shift, let->node, 0, 2 * nm2l_hint));
			{
				PhaseScope ph(c, NBCO_PH_M2L);
				// register-resident generated bodies, one interaction per lane (k_m2l.hip)
				if (f64) NBCO_TRY(launch_m2l_lanes_f64(c, P, tv.csz, (const double *)tv.mpole, (double *)tv.local, c->m2l_keys_alt.as<uint64_t>(), c->m2l_start.as<int>(), shift, ntot));
				else NBCO_TRY(launch_m2l_lanes(c, P, tv.csz, tv.mpole, tv.local, c->m2l_keys_alt.as<uint64_t>(), c->m2l_start.as<int>(), shift, ntot));
			}
			{
				PhaseScope ph(c, NBCO_PH_L2L);
				NBCO_TRY(launch_downward_gen(c, P, tv.center, tv.local, L, dm.d, dm.g, f64));
			}
		}
		NBCO_HIP(hipGetLastError());
	}
	// ---- L2P + rescale + (un)sort -------------------------------------------------------------------------
	NBCO_TRY(c->join_aux());   // far field (multipoles, M2L list, M2L, L2L) complete
	{
		PhaseScope ph(c, NBCO_PH_L2P);
		NBCO_TRY(launch_l2p_gen(c, P, pos, tv.center, tv.local, near, c->p2p_chunk_off.as<int>(), tv.index, mlt_max, unsort, c->o.unsort ? 1 : 0,
		                        param, a, c->o.coll ? 1 : 0, n, L, own0, own_n, mutual ? c->p2p_sec.as<int2>() : nullptr,
		                        mutual ? c->p2p_react.as<float4>() : nullptr, react_cap, 32 * mutual_th, c->o.far_fp64 ? 1 : 0));
	}
	// ---- now look at what the traversal reported (long finished: the GPU is busy with the work queued above) ----------
	{
		const auto t0 = std::chrono::steady_clock::now();
		NBCO_TRY(c->wait_flags());
		c->host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	}
	const int *h = c->h_flags;
	if (h[3] != 0)
	{
		// a node had more pivot ties than the selection build resolves (degenerate coordinates): everything queued so
		// far ran on a tree that is not the reference's, but only the acceleration array has been written -- the caller
		// redoes the evaluation with the sorting build
		out.sel_overflow = 1;
		return NBCO_OK;
	}
	if (h[2] == 2) return c->fail(NBCO_ERR_HIP, "internal error: the dual tree traversal did not finish in its launches");
	if (h[2] != 0) return c->fail(NBCO_ERR_CAPACITY, "dual tree traversal exceeded the list capacity (raise opts.list_factor or set opts.list_grow)");
	out.np2p = h[0]; out.nm2l = h[1];
	c->hint_np2p = h[0]; c->hint_nm2l = h[1];
	// what nbco_energy_fmm needs of this evaluation: tree, tree-ordered positions, the two sorted lists
	{
		nbco_ctx::LastEval &le = c->last_eval;
		le.valid = true;
		le.center = tv.center; le.csz = tv.csz; le.mpole = tv.mpole; le.mult = tv.mult; le.index = tv.index;
		le.L = L; le.ntot = ntot; le.order = P; le.shift = shift; le.real_bytes = c->o.far_fp64 ? 8 : 4;
		le.pos = pos; le.n = n; le.own0 = own0; le.own_n = own_n;
		le.have_p2p = c->o.coll != 0;
	}
	if (mutual && 2LL * h[0] + nself > react_cap) out.react_overflow = 1;   // the pair list outgrew the reaction records: same evaluation again, sized from h[0]
	return NBCO_OK;
}

// positions in tree order; velocities follow (fmm_cart3_kdtree.cuh:1755-1760)
static int kd_finish_order(nbco_ctx *c, float *p, long long n)
{
	PhaseScope ph(c, NBCO_PH_FINISH);
	hipStream_t st = c->stream;
	NBCO_TRY(c->reserve(c->tmp3, sizeof(float) * 3 * (size_t)n));
	const int *order_in = nullptr;
	int *order_out = nullptr;
	if (c->o.track_order)
	{
		// cumulative permutation across rebuilds (the reference keeps none: its snapshots are in tree order, main3.cu:855-858)
		NBCO_TRY(c->reserve(c->order, sizeof(int) * (size_t)n));
		NBCO_TRY(c->reserve(c->order_alt, sizeof(int) * (size_t)n));
		if (c->order_n == n) order_in = c->order.as<int>();
		order_out = c->order_alt.as<int>();
	}
	else c->order_n = -1;
	hipLaunchKernelGGL(reorder_state_kernel, dim3(grid1d(n)), dim3(kBlock), 0, st, (const float4 *)c->pos4.as<float4>(), (const int *)c->unsort.as<int>(),
	                   (const float *)(p + 3 * n), p, c->tmp3.as<float>(), n, order_in, order_out);
	if (order_out) { std::swap(c->order, c->order_alt); c->order_n = n; }
	if (c->defer_v_copy) c->v_deferred = c->tmp3.as<float>();   // the caller's next pass over the velocities reads them from here
	else NBCO_HIP(hipMemcpyAsync(p + 3 * n, c->tmp3.ptr, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToDevice, st));
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

// ---- FMM potential energy (no reference driver computes an energy) -------------------------------
// phi_i = sum_{j != i} (|x_i - x_j|^2 + eps2)^(-1/2) with the interaction lists of the last evaluation: the leaf's P2P list
// pair by pair, and for the M2L list of the leaf and of each of its ancestors the source node's multipole expansion evaluated
// AT THE PARTICLE (the reference's m2p_pot3, fmm_cart_base3.cuh:1474-1490, instead of an order-0 local: no second truncation).
// With b_K = d^K f / K!, f = 1/|d|:  phi = sum_K M[K] |K|! b_K(d), d = x_i - c_s, and the Taylor coefficients follow
//   k R^2 b_K + (2k - 1) sum_a d_a b_{K - e_a} + (k - 1) sum_a b_{K - 2 e_a} = 0,   k = |K|, R^2 = |d|^2
// (tests/test_oracle_closed_form.py derives the same numbers from the polynomial form of the derivatives).  fp64 throughout:
// this is a diagnostic that runs once per snapshot, and energy drifts are read at the 1e-6 level.
__host__ __device__ constexpr int sym_index(int x, int z, int n) { return (n * (n + 1) - (n - z) * (n - z + 1)) / 2 + n - x; }
__host__ __device__ constexpr int sym_offset(int n) { return n * (n + 1) * (n + 2) / 6; }

template <int P, typename T>
__device__ inline double m2p_potential(const T *__restrict__ M, double dx, double dy, double dz, double eps2)
{
	constexpr int offM = sym_offset(P);
	double B[offM > 0 ? offM : 1];
	const double R2 = dx * dx + dy * dy + dz * dz + eps2, iR2 = 1.0 / R2;
	B[0] = sqrt(iR2);
	double phi = (double)M[0] * B[0], fact = 1.0;
#pragma unroll
	for (int k = 1; k < P; ++k)
	{
		fact *= (double)k;
		const double c1 = -(double)(2 * k - 1) * iR2 / (double)k, c2 = -(double)(k - 1) * iR2 / (double)k;
		double s = 0.0;
#pragma unroll
		for (int z = 0; z <= k; ++z)
#pragma unroll
			for (int x = k - z; x >= 0; --x)
			{
				const int y = k - x - z;
				double t1 = 0.0, t2 = 0.0;
				if (x >= 1) t1 += dx * B[sym_offset(k - 1) + sym_index(x - 1, z, k - 1)];
				if (y >= 1) t1 += dy * B[sym_offset(k - 1) + sym_index(x, z, k - 1)];
				if (z >= 1) t1 += dz * B[sym_offset(k - 1) + sym_index(x, z - 1, k - 1)];
				if (k >= 2)
				{
					if (x >= 2) t2 += B[sym_offset(k - 2) + sym_index(x - 2, z, k - 2)];
					if (y >= 2) t2 += B[sym_offset(k - 2) + sym_index(x, z, k - 2)];
					if (z >= 2) t2 += B[sym_offset(k - 2) + sym_index(x, z - 2, k - 2)];
				}
				const double b = c1 * t1 + c2 * t2;
				B[sym_offset(k) + sym_index(x, z, k)] = b;
				s += (double)M[sym_offset(k) + sym_index(x, z, k)] * b;
			}
		phi += fact * s;
	}
	return phi;
}

template <int P, typename T>
__global__ __launch_bounds__(kBlock) void kd_potential_kernel(nbco_ctx::LastEval le, const uint64_t *__restrict__ m2l_keys, const int *__restrict__ m2l_start,
                                                              const uint64_t *__restrict__ p2p_keys, const int *__restrict__ p2p_start, float eps2f,
                                                              double *__restrict__ part)
{
	constexpr int offM = sym_offset(P);
	const long long io = (long long)blockIdx.x * kBlock + threadIdx.x;
	double phi = 0.0;
	if (io < le.own_n)
	{
		const long long i = le.own0 + io;
		const int beg = kd_beg(le.L), lf = (int)(((1LL << le.L) * i) / le.n);
		const uint64_t mask = (1ull << le.shift) - 1;
		const float4 p = le.pos[i];
		const double eps2 = (double)eps2f;
		if (le.have_p2p)
			for (int e = p2p_start[lf]; e < p2p_start[lf + 1]; ++e)
			{
				const int src = beg + (int)(p2p_keys[e] & mask);
				const int is = le.index[src], ms = le.mult[src];
				for (int j = 0; j < ms; ++j)
				{
					if (is + j == i) continue;
					const float4 q = le.pos[is + j];
					const double dx = (double)p.x - (double)q.x, dy = (double)p.y - (double)q.y, dz = (double)p.z - (double)q.z;
					phi += 1.0 / sqrt(dx * dx + dy * dy + dz * dz + eps2);
				}
			}
		for (int node = beg + lf;; node = (node - 1) >> 1)
		{
			for (int e = m2l_start[node]; e < m2l_start[node + 1]; ++e)
			{
				const int sn = (int)(m2l_keys[e] & mask);
				const float4 cs = le.csz[sn];
				phi += m2p_potential<P, T>(reinterpret_cast<const T *>(le.mpole) + (size_t)sn * offM, (double)p.x - (double)cs.x, (double)p.y - (double)cs.y, (double)p.z - (double)cs.z, eps2);
			}
			if (node == 0) break;
		}
	}
	// block sum -> one partial per block (summed in a fixed order on the host)
	__shared__ double sh[kBlock / 64];
	for (int o = 32; o > 0; o >>= 1) phi += __shfl_xor(phi, o);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = phi;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		double t = 0.0;
		for (int k = 0; k < kBlock / 64; ++k) t += sh[k];
		part[blockIdx.x] = t;
	}
}

template <int P> static void launch_potential(nbco_ctx *c, int grid, double *part)
{
	// (the multipoles of the last evaluation are doubles when it ran with opts.far_fp64: LastEval::real_bytes)
	if (c->last_eval.real_bytes == 8)
		hipLaunchKernelGGL((kd_potential_kernel<P, double>), dim3(grid), dim3(kBlock), 0, c->stream, c->last_eval, (const uint64_t *)c->m2l_keys_alt.as<uint64_t>(),
		                   (const int *)c->m2l_start.as<int>(), (const uint64_t *)c->p2p_keys_alt.as<uint64_t>(), (const int *)c->p2p_start.as<int>(), c->o.eps2,
		                   part);
	else
		hipLaunchKernelGGL((kd_potential_kernel<P, float>), dim3(grid), dim3(kBlock), 0, c->stream, c->last_eval, (const uint64_t *)c->m2l_keys_alt.as<uint64_t>(),
		                   (const int *)c->m2l_start.as<int>(), (const uint64_t *)c->p2p_keys_alt.as<uint64_t>(), (const int *)c->p2p_start.as<int>(), c->o.eps2,
		                   part);
}

} // namespace

// sum over the own particles of phi_i / 2 (the caller multiplies by param[0] = xi / N)
int kd_energy_fmm(nbco_ctx *c, long long n_own, double *half_phi_sum)
{
	const nbco_ctx::LastEval &le = c->last_eval;
	if (!le.valid || !c->tree_valid) return c->fail(NBCO_ERR_ARG, "nbco_energy_fmm: no kd-tree evaluation to take the lists from");
	if (le.own_n != n_own) return c->fail(NBCO_ERR_ARG, "nbco_energy_fmm: particle count differs from the last evaluation's");
	const int grid = (int)((le.own_n + kBlock - 1) / kBlock);
	NBCO_TRY(c->reserve(c->part, sizeof(double) * (size_t)grid));
	double *part = c->part.as<double>();
	switch (le.order)
	{
	case 1: launch_potential<1>(c, grid, part); break;
	case 2: launch_potential<2>(c, grid, part); break;
	case 3: launch_potential<3>(c, grid, part); break;
	case 4: launch_potential<4>(c, grid, part); break;
	case 5: launch_potential<5>(c, grid, part); break;
	case 6: launch_potential<6>(c, grid, part); break;
	case 7: launch_potential<7>(c, grid, part); break;
	case 8: launch_potential<8>(c, grid, part); break;
	case 9: launch_potential<9>(c, grid, part); break;
	case 10: launch_potential<10>(c, grid, part); break;
	default: return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_energy_fmm: order");
	}
	NBCO_HIP(hipGetLastError());
	std::vector<double> h((size_t)grid);
	NBCO_HIP(hipMemcpyAsync(h.data(), part, sizeof(double) * (size_t)grid, hipMemcpyDeviceToHost, c->stream));
	NBCO_HIP(hipStreamSynchronize(c->stream));
	double s = 0.0;
	for (double v : h) s += v;
	*half_phi_sum = 0.5 * s;
	return NBCO_OK;
}

namespace {
} // namespace

// the re-ordering an evaluation with defer_order set has left undone (nbco_integrate_steps, last step)
int kd_finish_pending_order(nbco_ctx *c, float *p, long long n)
{
	if (!c->order_pending) return NBCO_OK;
	c->order_pending = false;
	return kd_finish_order(c, p, n);
}

// nbco_integrate_steps, between the force evaluation of one leapfrog step and that of the next (kd_turnaround_kernel).  v_in: where
// the current velocities are (the caller's array or the scratch copy of the turnaround before); returns where they are now.
int kd_turnaround(nbco_ctx *c, float *p, const float *v_in, const float **v_now, const float *param, float ks, float ds, bool elastic, long long n,
                  const float *root6)
{
	const bool gather = c->order_pending;
	c->order_pending = false;
	float *x = p, *v = p + 3 * n, *a = p + 6 * n;
	const int L = c->kd.L;
	// will the next evaluation rebuild?  (kd_build_upward's rule; the options cannot change inside nbco_integrate_steps)
	const bool next_rebuild = (c->eval_counter % c->o.tree_steps) == 0;
	long long words_a = 0, words_b = 0;
	if (next_rebuild)
	{
		int l0 = 0;
		while (l0 < L && (n + (1LL << l0) - 1) / (1LL << l0) > kSubS) ++l0;
		if (!c->force_sort_build && l0 > 0) NBCO_TRY(kd_select_begin(c, l0, false, &words_a, &words_b));
	}
	float *v_out = v;
	if (gather)
	{
		NBCO_TRY(c->reserve(c->tmp3, sizeof(float) * 3 * (size_t)n));
		NBCO_TRY(c->reserve(c->tmp3b, sizeof(float) * 3 * (size_t)n));
		v_out = v_in == c->tmp3.as<float>() ? c->tmp3b.as<float>() : c->tmp3.as<float>();
	}
	else if (v_in != v) v_out = const_cast<float *>(v_in);   // in place, wherever they are
	PhaseScope ph(c, NBCO_PH_AXPY);
	TreeView tv = view_of(c->kd);
	hipStream_t st = c->stream;
#define NBCO_TURN_ARGS c->pos4.as<float4>(), c->unsort.as<int>(), x, v_in, v_out, (const float *)a, param, ks, ds, elastic ? 1 : 0, n, next_rebuild ? 1 : 0, \
	c->sel_hist.as<uint32_t>(), words_a, c->sel_nodes.as<uint32_t>(), words_b, c->counters.as<int>() + 110, c->prep_state.as<unsigned>(), tv, root6
	if (gather) hipLaunchKernelGGL(kd_turnaround_kernel<true>, dim3(kPrepGrid), dim3(kPrepBlock), 0, st, NBCO_TURN_ARGS);
	else hipLaunchKernelGGL(kd_turnaround_kernel<false>, dim3(kPrepGrid), dim3(kPrepBlock), 0, st, NBCO_TURN_ARGS);
#undef NBCO_TURN_ARGS
	NBCO_HIP(hipGetLastError());
	c->skip_prep = next_rebuild ? 1 : 2;
	c->order_n = -1;
	c->last_eval.valid = false;   // the particles have moved (and pos4 may hold the next build's input): nbco_energy_fmm must follow an evaluation
	*v_now = v_out;
	return NBCO_OK;
}

int fmm_kdtree_eval(nbco_ctx *c, float *p, float *a, long long n, const float *param)
{
	if (n <= 0) return c->fail(NBCO_ERR_ARG, "nbco_fmm_kdtree: n must be positive");
	if (n > 0x7fffffffLL / 4) return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_fmm_kdtree: n too large for 32-bit tree indices");
	const int P = c->o.fmm_order;
	const int L = kd_levels(n, P, c->o.dens_inhom, c->o.tree_L);
	bool rebuild = false;
	NBCO_TRY(kd_build_upward(c, p, n, L, nullptr, rebuild));
	KdCounts cnt;
	const Dom whole{0, 0};
	{
		const int rc = kd_interact(c, view_of(c->kd), c->pos4.as<float4>(), n, c->kd.mlt_max, whole, 0, n, c->unsort.as<int>(), a, param, cnt);
		if (rc == NBCO_ERR_CAPACITY && c->grow_lists(c->kd.ntot))
		{
			// the lists have outgrown their buffers (the caller's arrays are untouched): twice the room, same evaluation -- on the
			// same rebuild schedule (a forced rebuild here would make the trees of the following tree_steps - 1 evaluations depend
			// on when a buffer happened to fill up)
			return fmm_kdtree_eval(c, p, a, n, param);
		}
		if (rc != NBCO_OK) return rc;
	}
	if (cnt.react_overflow && !cnt.sel_overflow) return fmm_kdtree_eval(c, p, a, n, param);   // (same schedule, see above)
	if (cnt.sel_overflow && c->sel_warm_used)
	{
		// the one-pass select missed a median (or what it left behind tripped the tie flag): the same evaluation with the cold
		// two-pass select, nothing escalated
		c->note_warm_miss();
		c->tree_valid = false;
		return fmm_kdtree_eval(c, p, a, n, param);
	}
	if (rebuild && c->sel_warm_used) c->note_warm_ok();
	if (cnt.sel_overflow)
	{
		// next more conservative build: three radix passes, then the sorting build
		if (!c->escalate_build()) return c->fail(NBCO_ERR_UNSUPPORTED, "kd-tree build: tie flag raised by the sorting build");
		c->tree_valid = false;
		return fmm_kdtree_eval(c, p, a, n, param);
	}
	c->order_pending = false;
	if (!c->o.unsort && rebuild)
	{
		if (c->defer_order) c->order_pending = true;   // nbco_integrate_steps folds the re-ordering into its pass between two steps
		else NBCO_TRY(kd_finish_order(c, p, n));
	}

	c->tree_valid = true;
	c->tree_n = n;
	c->tree_order = P;
	c->eval_counter += 1;
	nbco_kd_info &info = c->info;
	info.L = L; info.ntot = c->kd.ntot; info.order = P; info.mlt_max = c->kd.mlt_max; info.n = n;
	info.p2p_pairs = cnt.np2p; info.m2l_pairs = cnt.nm2l; info.rebuilt = rebuild ? 1 : 0;
	info.directed_p2p = -1;   // read back from the device counter on demand (nbco_kd_get_info)
	return NBCO_OK;
}

// =====================================================================================================
// Multi-GPU: kd-domain sharding.  GPU g of G = 2^d owns the subtree of node 2^d - 1 + g of
// the GLOBAL balanced kd-tree: N / G particles, the global levels d .. L.  One force evaluation is
//   local    build levels d .. L of the own subtree + P2M/M2M up to its root          (kd_dist_local)
//   exchange all-gather of {centre+size, multipoles} of every domain's nodes and of the
//            tree-ordered positions -- done by the caller (RCCL), this library never communicates
//   finish   assemble the global node arrays, M2M for levels d-1 .. 0, dual traversal pruned to pairs
//            that touch the own domain, P2P / M2L / L2L / L2P for the own targets only    (kd_dist_finish)
// Cross-domain pairs are evaluated one-directionally on the owner of the target, so no force reduction
// is needed and every target's sums run in the single-GPU order: the result equals the 1-GPU result bit
// for bit (particles with exactly tied coordinates excepted: their order inside a leaf may differ).
// The top d median splits (kd_dist_partition) run redundantly on every GPU over the gathered state.
namespace {

int log2_exact(int v)
{
	int d = 0;
	while ((1 << d) < v) ++d;
	return (1 << d) == v ? d : -1;
}

// global node id of local node k of domain r
__host__ __device__ inline int dist_global_id(int k, int r, int d)
{
#ifdef __HIP_DEVICE_COMPILE__
	const int l = 31 - __clz(k + 1);
#else
	int l = 0;
	while ((2 << l) <= k + 1) ++l;
#endif
	return (1 << (l + d)) - 1 + (r << l) + (k - ((1 << l) - 1));
}

// gathered per-rank blocks -> global node arrays (levels >= d).  `blocks` points at rank 0's data, consecutive ranks are
// block_bytes apart: the traversal records (float4 csz[ntot_loc]) and the multipoles (float mpole[ntot_loc][offM]) either
// travel in one block per rank (nbco_dist_finish) or in two all-gathers (nbco_dist_finish_traverse / _rest)
__global__ __launch_bounds__(kBlock) void dist_unpack_nodes_kernel(TreeView t, const char *__restrict__ blocks, size_t block_bytes, int ntot_loc, int G,
                                                                   int d)
{
	const long long total = (long long)G * ntot_loc;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock)
	{
		const int r = (int)(i / ntot_loc), k = (int)(i % ntot_loc);
		const float4 cs = reinterpret_cast<const float4 *>(blocks + (size_t)r * block_bytes)[k];
		const int gid = dist_global_id(k, r, d);
		t.csz[gid] = cs;
		t.center[3 * gid] = cs.x; t.center[3 * gid + 1] = cs.y; t.center[3 * gid + 2] = cs.z;
	}
}
template <typename T>   // float, or double tuples with opts.far_fp64
__global__ __launch_bounds__(kBlock) void dist_unpack_mpole_kernel(TreeView t, const char *__restrict__ blocks, size_t block_bytes, int ntot_loc, int G,
                                                                   int r0, int d, int offM)
{
	const long long per = (long long)ntot_loc * offM, total = (long long)G * per;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock)
	{
		const int r = (int)(i / per);
		const long long e = i % per;
		const int k = (int)(e / offM), comp = (int)(e % offM);
		const T *src = reinterpret_cast<const T *>(blocks + (size_t)r * block_bytes);
		reinterpret_cast<T *>(t.mpole)[(size_t)dist_global_id(k, r0 + r, d) * offM + comp] = src[e];
	}
}
// ranges of evalBox's rule for every node of the global tree (fmm_cart3_kdtree.cuh:109-137)
__global__ __launch_bounds__(kBlock) void dist_ranges_kernel(TreeView t, long long n)
{
	for (int j = blockIdx.x * kBlock + threadIdx.x; j < t.ntot; j += gridDim.x * kBlock)
	{
		const int l = 31 - __clz(j + 1);
		const long long m = 1LL << l, i = j - (m - 1);
		const long long start = (i == 0) ? 0 : (n * i - 1) / m + 1, end = (n * (i + 1) - 1) / m + 1;
		t.index[j] = (int)start;
		t.mult[j] = (int)(end - start);
	}
}
__global__ void dist_root6_kernel(const float *__restrict__ lb, const float *__restrict__ rb, int node, float *__restrict__ out6)
{
	if (threadIdx.x < 3) { out6[threadIdx.x] = lb[3 * node + threadIdx.x]; out6[3 + threadIdx.x] = rb[3 * node + threadIdx.x]; }
}

// top-tree arrays (levels 0 .. d) inside c->dist_top
struct TopView
{
	float *lbound, *rbound;
	int *splitdim, *index;
};
TopView top_view(nbco_ctx *c, int ntop)
{
	TopView v;
	char *q = (char *)c->dist_top.ptr;
	v.lbound = (float *)q; q += 12 * (size_t)ntop;
	v.rbound = (float *)q; q += 12 * (size_t)ntop;
	v.splitdim = (int *)q; q += 4 * (size_t)ntop;
	v.index = (int *)q;
	return v;
}

} // namespace

int kd_dist_layout(nbco_ctx *c, long long n_global, int world, int rank, nbco_dist_layout *out)
{
	const int d = log2_exact(world);
	if (d < 0 || world > 64) return c->fail(NBCO_ERR_ARG, "nbco_dist: the number of domains must be a power of two <= 64");
	if (rank < 0 || rank >= world) return c->fail(NBCO_ERR_ARG, "nbco_dist: rank out of range");
	if (n_global <= 0 || n_global % world != 0) return c->fail(NBCO_ERR_ARG, "nbco_dist: n must be a positive multiple of the number of domains");
	if (n_global > 0x7fffffffLL / 4) return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_dist: n too large for 32-bit tree indices");
	const int P = c->o.fmm_order;
	const int L = kd_levels(n_global, P, c->o.dens_inhom, c->o.tree_L);
	if (L - d < 2) return c->fail(NBCO_ERR_ARG, "nbco_dist: too few particles per domain (the local tree needs >= 2 levels)");
	if (d > 0 && n_global / world < 4096) return c->fail(NBCO_ERR_ARG, "nbco_dist: at least 4096 particles per domain are required");
	out->world = world; out->rank = rank; out->d = d; out->L = L; out->L_local = L - d; out->order = P;
	out->ntot_local = (1 << (L - d + 1)) - 1;
	out->n_global = n_global; out->n_local = n_global / world;
	out->csz_bytes = (long long)out->ntot_local * (long long)sizeof(float4);
	const long long rb = c->o.far_fp64 ? 8 : 4;   // bytes per real of a multipole tuple
	out->mpole_bytes = (long long)out->ntot_local * rb * (long long)sym_off(P);
	out->nodes_bytes = out->csz_bytes + out->mpole_bytes;
	out->pos_bytes = (long long)out->n_local * (long long)sizeof(float4);
	out->let_node_bytes = rb * (((sym_off(P) + 1 + 3) / 4) * 4);
	out->let_counts = 2 * world + 2;
	return NBCO_OK;
}

// the top-tree arrays for k_dpart.hip (levels 0 .. d)
int kd_dist_top_arrays(nbco_ctx *c, int ntop, float **lb, float **rb, int **sd, int **index)
{
	NBCO_TRY(c->reserve(c->dist_top, (size_t)ntop * 32 + 64));
	const TopView v = top_view(c, ntop);
	*lb = v.lbound; *rb = v.rbound; *sd = v.splitdim; *index = v.index;
	return NBCO_OK;
}
// what nbco_dist_partition leaves behind besides the domain's state and the top boxes
// between the force evaluation of one leapfrog step of a sharded run and that of the next: elastic term, both half kicks, drift and
// the next local build's prologue in one pass over the domain's state (kd_turnaround_kernel; the state is in tree order already)
int kd_dist_turnaround(nbco_ctx *c, float *buf_local, long long n_local, const float *param, float ks, float ds, bool elastic)
{
	if (!c->dist.partitioned || n_local != c->dist.n_local || !c->tree_valid)
		return c->fail(NBCO_ERR_ARG, "nbco_dist_turnaround: call it right after a sharded force evaluation");
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	const int d = lay.d, ntop = (1 << (d + 1)) - 1;
	TopView top = top_view(c, ntop);
	float *root6 = c->small.as<float>() + 80;
	hipLaunchKernelGGL(dist_root6_kernel, dim3(1), dim3(64), 0, c->stream, (const float *)top.lbound, (const float *)top.rbound, (1 << d) - 1 + lay.rank, root6);
	const float *v_now = nullptr;
	c->order_pending = false;
	NBCO_TRY(kd_turnaround(c, buf_local, buf_local + 3 * n_local, &v_now, param, ks, ds, elastic, n_local, root6));
	return NBCO_OK;
}

int kd_dist_set_partitioned(nbco_ctx *c, long long n_global, int world, int rank)
{
	c->skip_prep = 0;   // (a prologue done by nbco_dist_turnaround belonged to the state before the cut)
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, n_global, world, rank, &lay));
	c->dist.world = world; c->dist.rank = rank; c->dist.d = lay.d; c->dist.n_global = n_global; c->dist.n_local = lay.n_local; c->dist.L = lay.L;
	c->dist.partitioned = true;
	c->dist.build_done = c->dist.local_done = c->dist.traversed = c->dist.let_selected = c->dist.let_packed = false;
	c->tree_valid = false;
	return NBCO_OK;
}

// state_all = [pos N x 3 | vel N x 3] (every rank passes the same gathered state), state_local = [pos | vel] of
// the rank's domain in partition order.
int kd_dist_partition(nbco_ctx *c, const float *state_all, long long n_global, int world, int rank, float *state_local)
{
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, n_global, world, rank, &lay));
	const int d = lay.d, ntop = (1 << (d + 1)) - 1;
	const long long n = n_global, nl = lay.n_local;
	hipStream_t st = c->stream;
	NBCO_TRY(c->reserve(c->dist_top, (size_t)ntop * 32 + 64));
	NBCO_TRY(kd_reserve_particles(c, n));
	TopView top = top_view(c, ntop);
	TreeView tv{};
	tv.lbound = top.lbound; tv.rbound = top.rbound; tv.splitdim = top.splitdim; tv.index = top.index; tv.L = d; tv.ntot = ntop;
	for (int attempt = 0; attempt < 3; ++attempt)
	{
		float4 *pos = c->pos4.as<float4>(), *pos_alt = c->pos4_alt.as<float4>();
		int *unsort = c->unsort.as<int>(), *unsort_alt = c->unsort_alt.as<int>();
		PhaseScope ph(c, NBCO_PH_BUILD);
		NBCO_TRY(launch_pack4(c, pos, state_all, n));
		NBCO_HIP(hipMemsetAsync(c->counters.as<int>() + 110, 0, sizeof(int), st));
		float *mm = c->small.as<float>() + 64;
		NBCO_TRY(launch_minmax4(c, pos, n, mm));
		hipLaunchKernelGGL(kd_root_kernel, dim3(1), dim3(64), 0, st, tv, (const float *)mm);
		hipLaunchKernelGGL(iota_kernel, dim3(grid1d(n)), dim3(kBlock), 0, st, unsort, n);
		c->perm_primed_n = -1;   // the permutation buffers now hold n_global-range indices: the next local build primes them again
		const bool use_select = !c->force_sort_build;
		NBCO_TRY(kd_build_top(c, tv, pos, pos_alt, unsort, unsort_alt, n, d, use_select));
		int flag = 0;
		NBCO_HIP(hipMemcpyAsync(&flag, c->counters.as<int>() + 110, sizeof(int), hipMemcpyDeviceToHost, st));
		NBCO_HIP(hipStreamSynchronize(st));
		if (flag && use_select) { c->escalate_build(); continue; }
		// the domain's slice of the partitioned state
		hipLaunchKernelGGL(unpack4_kernel, dim3(grid1d(nl)), dim3(kBlock), 0, st, (const float4 *)(pos + (size_t)rank * nl), state_local, nl);
		NBCO_TRY(launch_gather3(c, state_local + 3 * nl, state_all + 3 * n, unsort + (size_t)rank * nl, nl, false));
		NBCO_HIP(hipGetLastError());
		break;
	}
	c->dist.world = world; c->dist.rank = rank; c->dist.d = d; c->dist.n_global = n_global; c->dist.n_local = nl; c->dist.L = lay.L;
	c->dist.partitioned = true;
	c->tree_valid = false;
	c->skip_prep = 0;
	return NBCO_OK;
}

// stage 1 (pos_send != null): subtree build, tree-ordered positions into pos_send; stage 2 (nodes_send != null): upward
// pass, node block into nodes_send.  Both pointers: the whole local stage.  The split lets the caller start the all-gather
// of the positions while the multipoles are still being computed.
// csz_send / mpole_send: the two halves of the node block on their own (the traversal records are known after the build,
// ahead of the multipoles)
// let_stage (LET exchange, nothing but the traversal records is copied out): 1 = build, 2 = upward pass
int kd_dist_local(nbco_ctx *c, float *buf_local, long long n_local, void *nodes_send, void *pos_send, void *csz_send, void *mpole_send, int let_stage)
{
	if (!c->dist.partitioned || n_local != c->dist.n_local)
		return c->fail(NBCO_ERR_ARG, "nbco_dist_local: call nbco_dist_partition first (and pass its local particle count)");
	if (c->o.unsort) return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_dist: opts.unsort is not available with kd-domain sharding");
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	if (lay.L != c->dist.L) return c->fail(NBCO_ERR_ARG, "nbco_dist_local: options changed since nbco_dist_partition");
	const int d = lay.d, ntop = (1 << (d + 1)) - 1;
	hipStream_t st = c->stream;
	bool rebuild = c->dist.rebuilt;
	if (pos_send || let_stage == 1)
	{
		TopView top = top_view(c, ntop);
		float *root6 = c->small.as<float>() + 80;
		hipLaunchKernelGGL(dist_root6_kernel, dim3(1), dim3(64), 0, st, (const float *)top.lbound, (const float *)top.rbound, (1 << d) - 1 + lay.rank, root6);
		if (let_stage == 1 && c->dist.let_selected)
		{
			// called again after a round in which some rank's build was flagged (the flags travel with the LET counts, so the LET
			// path needs no host round trip behind the build): repeat this rank's build more conservatively if it was the one
			NBCO_TRY(c->wait_flags());
			if (c->h_flags[3] != 0)
			{
				if (c->sel_warm_used) c->note_warm_miss();
				else if (!c->escalate_build()) return c->fail(NBCO_ERR_UNSUPPORTED, "kd-tree build: tie flag raised by the sorting build");
				c->tree_valid = false;
			}
			c->dist.let_selected = c->dist.let_packed = c->dist.traversed = c->dist.local_done = false;
		}
		NBCO_TRY(kd_build_upward(c, buf_local, n_local, lay.L_local, root6, rebuild, 1));
		while (rebuild && !c->force_sort_build && let_stage != 1)
		{
			// A tie overflow of the selection build has to be caught BEFORE the exchange (the other domains are
			// about to consume these positions and nodes); the retry with a more conservative build is purely local.
			int flag = 0;
			NBCO_HIP(hipMemcpyAsync(&flag, c->counters.as<int>() + 110, sizeof(int), hipMemcpyDeviceToHost, st));
			NBCO_HIP(hipStreamSynchronize(st));
			if (!flag) { if (c->sel_warm_used) c->note_warm_ok(); break; }
			// (a flagged build that ran the warm select is repeated cold first, nothing escalated)
			if (c->sel_warm_used) c->note_warm_miss();
			else c->escalate_build();
			c->tree_valid = false;
			NBCO_TRY(kd_build_upward(c, buf_local, n_local, lay.L_local, root6, rebuild, 1));
		}
		c->dist.rebuilt = rebuild;
		if (pos_send) NBCO_HIP(hipMemcpyAsync(pos_send, c->pos4.ptr, sizeof(float4) * (size_t)n_local, hipMemcpyDeviceToDevice, st));
		if (csz_send) NBCO_HIP(hipMemcpyAsync(csz_send, c->kd.csz, sizeof(float4) * (size_t)lay.ntot_local, hipMemcpyDeviceToDevice, st));
		c->dist.build_done = true;
	}
	if (nodes_send || mpole_send || let_stage == 2)
	{
		if (!c->dist.build_done) return c->fail(NBCO_ERR_ARG, "nbco_dist_local_upward: the build stage has not run");
		c->dist.build_done = false;
		NBCO_TRY(kd_build_upward(c, buf_local, n_local, lay.L_local, nullptr, rebuild, 2));
		c->dist.local_done = true;
		if (let_stage == 2) return NBCO_OK;   // (nbco_dist_let_pack waits for the second stream)
		NBCO_TRY(c->join_aux());   // the multipoles are about to leave the GPU
		const int offM = sym_off(lay.order);
		if (nodes_send)
		{
			NBCO_HIP(hipMemcpyAsync(nodes_send, c->kd.csz, sizeof(float4) * (size_t)lay.ntot_local, hipMemcpyDeviceToDevice, st));
			mpole_send = (char *)nodes_send + sizeof(float4) * (size_t)lay.ntot_local;
		}
		NBCO_HIP(hipMemcpyAsync(mpole_send, c->kd.mpole, (size_t)c->kd.real_bytes * (size_t)lay.ntot_local * offM, hipMemcpyDeviceToDevice, st));
		c->dist.local_done = true;
	}
	return NBCO_OK;
}

// The global tree of a sharded evaluation: node arrays carved from dist_tree; ranges from evalBox's rule.
static int dist_global_tree(nbco_ctx *c, const nbco_dist_layout &lay, KdTreeDev &g)
{
	const int L = lay.L, P = lay.order, ntot = (1 << (L + 1)) - 1;
	NBCO_TRY(kd_carve(c, c->dist_tree, g, ntot, sym_off(P), tl_off(P + 1)));
	g.L = L; g.ntot = ntot; g.order = P; g.n = lay.n_global; g.mlt_max = (int)((lay.n_global - 1) / (1LL << L) + 1);
	return NBCO_OK;
}

// =====================================================================================================
// Locally-essential-tree exchange (north_star; the reference is single-GPU).  Instead of every domain's whole node block and
// all of its positions, a rank sends every other rank exactly what that rank's evaluation reads.
//   * The traversal records (centre + squared box diagonal, 16 B per node) still travel as one small all-gather, so every
//     rank traverses the same global geometry (no rank ever meets a node it has no geometry for).
//   * The dual traversal is symmetric and every rank runs it on identical inputs with identical code: rank g emits the pair
//     (x, y) whenever x or y touches its domain, and so does every other rank the pair touches.  The pairs rank g holds
//     therefore ARE the list of what others need from it: an M2L pair (x, y) with x in g's subtree means every domain that
//     touches y (the owner of y, or all domains below y when y lies above the domain roots) reads x's multipole; a P2P
//     pair means the owner of leaf y reads the positions of leaf x.  The domain roots' multipoles go to everyone (M2M of
//     the levels above the domains).  Nothing is estimated, nothing conservative is sent.
//   * Per receiver the selected nodes / leaves are compacted into a contiguous segment of self-describing records (global
//     node id + multipole; position + global particle index), exchanged with one all-to-all of variable splits each, and
//     scattered into the receiver's global arrays.
//   * A guard on the receiver checks every source of its sorted M2L and P2P lists against what has arrived
//     (let_guard_kernel): a miss -- which would mean the ranks' traversals disagreed -- fails loudly (nbco_dist_let_check,
//     and the next nbco_dist_let_pack) instead of reading stale memory.
namespace {

__device__ inline uint64_t dom_mask(int node, int d)
{
	const int l = 31 - __clz(node + 1), pos = node - ((1 << l) - 1);
	if (l >= d) return 1ull << (pos >> (l - d));
	const int w = 1 << (d - l);
	return (w >= 64 ? ~0ull : ((1ull << w) - 1ull)) << (pos * w);
}
// local index of global node `node` (level >= d) inside the subtree of domain g
__device__ inline int dom_local_id(int node, int d, int g)
{
	const int l = 31 - __clz(node + 1), pos = node - ((1 << l) - 1), ll = l - d;
	return (1 << ll) - 1 + (pos - (g << ll));
}

// need masks from the raw pair lists of the traversal (regions + prefix sums, see traverse_kernel)
__global__ __launch_bounds__(kBlock) void let_mark_kernel(const int2 *__restrict__ m2l, const int *__restrict__ m2l_pref, const int2 *__restrict__ p2p,
                                                          const int *__restrict__ p2p_pref, long long capR, int d, int g, int L_loc,
                                                          unsigned long long *__restrict__ need_node, unsigned long long *__restrict__ need_leaf)
{
	const uint64_t me = 1ull << g;
	const long long nm = m2l_pref[kTravK], np = p2p ? p2p_pref[kTravK] : 0;
	const int leaf0 = (1 << L_loc) - 1;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < nm + np; i += (long long)gridDim.x * kBlock)
	{
		const bool far = i < nm;
		const int2 pr = far ? m2l[region_slot(m2l_pref, capR, i)] : p2p[region_slot(p2p_pref, capR, i - nm)];
		const uint64_t mx = dom_mask(pr.x, d), my = dom_mask(pr.y, d);
		if (mx == me && (my & ~me))
		{
			const int k = dom_local_id(pr.x, d, g);
			if (far) atomicOr(&need_node[k], (unsigned long long)(my & ~me)); else atomicOr(&need_leaf[k - leaf0], (unsigned long long)(my & ~me));
		}
		if (my == me && (mx & ~me))
		{
			const int k = dom_local_id(pr.y, d, g);
			if (far) atomicOr(&need_node[k], (unsigned long long)(mx & ~me)); else atomicOr(&need_leaf[k - leaf0], (unsigned long long)(mx & ~me));
		}
	}
}

// blockIdx.y = receiver r: compact the nodes / leaves r needs.  sel_node[r][slot] = local node, sel_leaf[r][slot] = {local leaf
// node, first record of its particles in r's position segment}; cursors[r] = {nodes, leaves, particles, -}.  The order inside
// a segment depends on the order in which blocks arrive; the receiver scatters by id, so results do not.
constexpr int kLetBlock = 1024;
__global__ __launch_bounds__(kLetBlock) void let_slots_kernel(const unsigned long long *__restrict__ need_node, const unsigned long long *__restrict__ need_leaf,
                                                              const int *__restrict__ mult, int ntot_loc, int L_loc, int g, int G,
                                                              int *__restrict__ sel_node, int2 *__restrict__ sel_leaf, int *__restrict__ cursors)
{
	__shared__ int wsum[3][kLetBlock / 64];
	__shared__ int base[3];
	const int r = blockIdx.y, k = blockIdx.x * kLetBlock + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const int leaf0 = (1 << L_loc) - 1, nleaf = 1 << L_loc;
	if (r == g) return;
	bool bn = false, bl = false;
	int m = 0;
	if (k < ntot_loc)
	{
		bn = k == 0 || ((need_node[k] >> r) & 1ull);   // the domain root's multipole goes to everyone
		if (k >= leaf0) { bl = (need_leaf[k - leaf0] >> r) & 1ull; m = bl ? mult[k] : 0; }
	}
	const unsigned long long mn = __ballot(bn), ml = __ballot(bl);
	const unsigned long long below = (1ull << lane) - 1ull;
	int incl = m;
	for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o); if (lane >= o) incl += y; }
	int on = __popcll(mn & below), ol = __popcll(ml & below), op = incl - m;
	if (lane == 63) { wsum[0][wv] = __popcll(mn); wsum[1][wv] = __popcll(ml); wsum[2][wv] = incl; }
	__syncthreads();
	if (threadIdx.x < 3)
	{
		int tot = 0;
		for (int w = 0; w < kLetBlock / 64; ++w) { const int x = wsum[threadIdx.x][w]; wsum[threadIdx.x][w] = tot; tot += x; }
		base[threadIdx.x] = tot ? atomicAdd(&cursors[4 * r + threadIdx.x], tot) : 0;
	}
	__syncthreads();
	if (bn) sel_node[(size_t)r * ntot_loc + base[0] + wsum[0][wv] + on] = k;
	if (bl) sel_leaf[(size_t)r * nleaf + base[1] + wsum[1][wv] + ol] = make_int2(k, base[2] + wsum[2][wv] + op);
}
// counts for the other ranks: [2 r] node records, [2 r + 1] particles for rank r; [2 G] = the traversal ran out of list room
__global__ void let_counts_kernel(const int *__restrict__ cursors, const int *__restrict__ counters, int G, long long *__restrict__ counts)
{
	const int r = threadIdx.x;
	if (r < G) { counts[2 * r] = cursors[4 * r]; counts[2 * r + 1] = cursors[4 * r + 2]; }
	if (r == 0) { counts[2 * G] = counters[2] != 0; counts[2 * G + 1] = counters[110] != 0; }   // list overflow; the build's tie / warm-miss flag
}

struct LetBases { long long v[65]; };   // first record of every receiver's (sender's) segment

// the node id in a record of reals: its bit pattern in a float, its value in a double (exact up to 2^53)
__device__ inline float let_id_enc(int id, float) { return __int_as_float(id); }
__device__ inline double let_id_enc(int id, double) { return (double)id; }
__device__ inline int let_id_dec(float v) { return __float_as_int(v); }
__device__ inline int let_id_dec(double v) { return (v >= 0.0 && v < 2147483648.0) ? (int)v : -1; }
// node records: {global node id, multipole[offM]} padded to rec reals (floats, or doubles with opts.far_fp64); blockIdx.y = receiver
template <typename T>
__global__ __launch_bounds__(kBlock) void let_pack_mpole_kernel(const T *__restrict__ mpole, int offM, int rec, int ntot_loc, int d, int g,
                                                                const int *__restrict__ sel_node, const int *__restrict__ cursors, LetBases nb,
                                                                T *__restrict__ out)
{
	const int r = blockIdx.y;
	const long long total = (long long)cursors[4 * r] * rec;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock)
	{
		const int comp = (int)(i % rec), slot = (int)(i / rec);
		const int k = sel_node[(size_t)r * ntot_loc + slot];
		T v = T(0);
		if (comp == 0) v = let_id_enc(dist_global_id(k, g, d), T());
		else if (comp <= offM) v = mpole[(size_t)k * offM + comp - 1];
		out[(nb.v[r] + slot) * rec + comp] = v;
	}
}
// position records: float4 {x, y, z, bits(global particle index)}; W (power of two >= the largest leaf) lanes per leaf
__global__ __launch_bounds__(kBlock) void let_pack_pos_kernel(const float4 *__restrict__ pos, const int *__restrict__ index, const int *__restrict__ mult, int L_loc,
                                                              int wlog, long long first_global, const int2 *__restrict__ sel_leaf,
                                                              const int *__restrict__ cursors, LetBases pb, float4 *__restrict__ out)
{
	const int r = blockIdx.y;
	const long long total = (long long)cursors[4 * r + 1] << wlog;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock)
	{
		const int j = (int)(i & ((1 << wlog) - 1));
		const int2 sl = sel_leaf[((size_t)r << L_loc) + (i >> wlog)];
		if (j >= mult[sl.x]) continue;
		const int q = index[sl.x] + j;
		float4 p = pos[q];
		p.w = __int_as_float((int)(first_global + q));
		out[pb.v[r] + sl.y + j] = p;
	}
}
__global__ __launch_bounds__(kBlock) void let_unpack_pos_kernel(const float4 *__restrict__ rec, long long count, float4 *__restrict__ pos_all, long long n_global,
                                                                int L, unsigned char *__restrict__ have_leaf)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock)
	{
		float4 p = rec[i];
		const long long idx = __float_as_int(p.w);
		if (idx < 0 || idx >= n_global) continue;   // (cannot happen; the guard reports the leaf as missing)
		p.w = 0.f;
		pos_all[idx] = p;
		have_leaf[(int)(((1LL << L) * idx) / n_global)] = 1;
	}
}
template <typename T>
__global__ __launch_bounds__(kBlock) void let_unpack_mpole_kernel(const T *__restrict__ recs, long long count, int rec, int offM, int ntot,
                                                                  T *__restrict__ mpole, unsigned char *__restrict__ have_node)
{
	const long long total = count * rec;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock)
	{
		const int comp = (int)(i % rec);
		const long long q = i / rec;
		const int gid = let_id_dec(recs[q * rec]);
		if (gid < 0 || gid >= ntot) continue;
		if (comp == 0) have_node[gid] = 1;
		else if (comp <= offM) mpole[(size_t)gid * offM + comp - 1] = recs[i];
	}
}
// own subtree and the levels above the domains are always there
__global__ __launch_bounds__(kBlock) void let_have_own_kernel(unsigned char *__restrict__ have_node, unsigned char *__restrict__ have_leaf, int ntot, int L, Dom dm)
{
	for (int j = blockIdx.x * kBlock + threadIdx.x; j < ntot; j += gridDim.x * kBlock)
	{
		const int l = 31 - __clz(j + 1);
		const bool mine = l < dm.d || dom_touch(dm, j);
		have_node[j] = mine ? 1 : 0;
		if (l == L) have_leaf[j - ((1 << L) - 1)] = mine ? 1 : 0;
	}
}
int let_rec_floats(int offM) { return ((offM + 1 + 3) / 4) * 4; }

// what the guard of the evaluation before reported (the caller has synchronised since: it holds counts that the kernels queued
// behind that evaluation produced)
int let_report(nbco_ctx *c)
{
	volatile int *w = c->h_flags + kLetWord;
	const int node = w[0], leaf = w[1];
	w[0] = 0; w[1] = 0;
	if (node || leaf)
	{
		char msg[200];
		snprintf(msg, sizeof msg, "LET exchange incomplete on rank %d: %s %d is in an interaction list but was not received", c->dist.rank,
		         node ? "the multipole of node" : "leaf", node ? node - 1 : leaf - 1);
		return c->fail(NBCO_ERR_HIP, msg);
	}
	return NBCO_OK;
}

struct LetView
{
	unsigned long long *need_node, *need_leaf;
	int *sel_node, *cursors;
	int2 *sel_leaf;
};
int let_view(nbco_ctx *c, const nbco_dist_layout &lay, LetView &v)
{
	const size_t nt = (size_t)lay.ntot_local, nl = (size_t)1 << lay.L_local, G = (size_t)lay.world;
	NBCO_TRY(c->reserve(c->let_sel, 8 * (nt + nl) + 4 * G * nt + 8 * G * nl + 16 * G + 64));
	char *q = (char *)c->let_sel.ptr;
	v.need_node = (unsigned long long *)q; q += 8 * nt;
	v.need_leaf = (unsigned long long *)q; q += 8 * nl;
	v.sel_leaf = (int2 *)q; q += 8 * G * nl;
	v.sel_node = (int *)q; q += 4 * G * nt;
	v.cursors = (int *)q;
	return NBCO_OK;
}

} // namespace

static int dist_geometry(nbco_ctx *c, const nbco_dist_layout &lay, const TreeView &tv, const char *csz_blocks, size_t csz_stride);

// csz_all: the gathered traversal records (world x csz_bytes, rank order).  Global geometry, dual traversal, and from its pair
// lists the segments this rank owes every other one; counts (device, 2 world + 2 values, see let_counts_kernel) are what the
// caller all-gathers next.  Called again after a round in which some rank reported list overflow, it repeats the traversal
// with more room where that happened and does nothing elsewhere.
int kd_dist_let_select(nbco_ctx *c, const void *csz_all, long long *counts)
{
	if (!c->dist.local_done && !c->dist.build_done) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_select: call nbco_dist_let_local_geom first");
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	const int G = lay.world, d = lay.d;
	hipStream_t st = c->stream;
	KdTreeDev g;
	NBCO_TRY(dist_global_tree(c, lay, g));
	TreeView tv = view_of(g);
	if (c->dist.let_selected)
	{
		NBCO_TRY(c->wait_flags());
		if (c->h_flags[2] != 1) return NBCO_OK;   // this rank's lists had room
		if (!c->grow_lists(g.ntot)) return c->fail(NBCO_ERR_CAPACITY, "dual tree traversal exceeded the list capacity (raise opts.list_factor or set opts.list_grow)");
	}
	else NBCO_TRY(dist_geometry(c, lay, tv, (const char *)csz_all, (size_t)lay.csz_bytes));
	NBCO_TRY(c->reserve(c->dist_pos, sizeof(float4) * (size_t)lay.n_global));
	KdCounts cnt;
	const Dom dm{d, lay.rank};
	NBCO_TRY(kd_interact(c, tv, c->dist_pos.as<float4>(), lay.n_global, g.mlt_max, dm, (long long)lay.rank * lay.n_local, lay.n_local, c->unsort.as<int>(),
	                     nullptr, nullptr, cnt, 1));
	c->dist.pos_all = c->dist_pos.ptr;
	c->dist.traversed = true;
	LetView v;
	NBCO_TRY(let_view(c, lay, v));
	{
		PhaseScope ph(c, NBCO_PH_TRAVERSE);
		const size_t nt = (size_t)lay.ntot_local, nl = (size_t)1 << lay.L_local;
		NBCO_HIP(hipMemsetAsync(v.need_node, 0, 8 * (nt + nl), st));
		NBCO_HIP(hipMemsetAsync(v.cursors, 0, 16 * (size_t)G, st));
		const long long capR = c->list_cap / kTravK;
		const int *tctr = c->trav_ctr.as<int>();
		const long long hint = (c->hint_nm2l > 0 ? c->hint_nm2l + c->hint_np2p : c->list_cap) + 1024;
		hipLaunchKernelGGL(let_mark_kernel, dim3(grid1d(hint, 4096)), dim3(kBlock), 0, st, (const int2 *)c->m2l_list.as<int2>(), tctr + kTcM2LPref,
		                   c->o.coll ? (const int2 *)c->p2p_list.as<int2>() : nullptr, tctr + kTcP2PPref, capR, d, lay.rank, lay.L_local, v.need_node, v.need_leaf);
		hipLaunchKernelGGL(let_slots_kernel, dim3((lay.ntot_local + kLetBlock - 1) / kLetBlock, G), dim3(kLetBlock), 0, st, (const unsigned long long *)v.need_node,
		                   (const unsigned long long *)v.need_leaf, (const int *)c->kd.mult, lay.ntot_local, lay.L_local, lay.rank, G, v.sel_node, v.sel_leaf, v.cursors);
		hipLaunchKernelGGL(let_counts_kernel, dim3(1), dim3(64), 0, st, (const int *)v.cursors, (const int *)c->counters.as<int>(), G, counts);
		NBCO_HIP(hipGetLastError());
	}
	c->dist.let_selected = true;
	return NBCO_OK;
}

// counts_all (host): the all-gathered counts, [sender][2 world + 2].  Fills the two send buffers: for receiver r (rank order)
// counts_all[me][2 r + 1] position records of 16 bytes, counts_all[me][2 r] node records of let_node_bytes.
int kd_dist_let_pack(nbco_ctx *c, const long long *counts_all, void *pos_send, void *mpole_send)
{
	if (!c->dist.let_selected || !c->dist.local_done) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_pack: selection or multipoles missing");
	NBCO_TRY(let_report(c));
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	const int G = lay.world, S = 2 * G + 2, offM = sym_off(lay.order), rec = let_rec_floats(offM);
	for (int s = 0; s < G; ++s)
	{
		if (counts_all[(size_t)s * S + 2 * G]) return c->fail(NBCO_ERR_CAPACITY, "nbco_dist_let_pack: a rank reported list overflow; repeat nbco_dist_let_select on every rank");
		if (counts_all[(size_t)s * S + 2 * G + 1])
			return c->fail(NBCO_ERR_CAPACITY, "nbco_dist_let_pack: a rank's tree build was flagged; repeat the evaluation from nbco_dist_let_local_geom on every rank");
	}
	if (c->sel_warm_used && c->dist.rebuilt) c->note_warm_ok();
	const long long *mine = counts_all + (size_t)lay.rank * S;
	LetBases nb{}, pb{};
	long long nmax = 0, lmax = 0;
	for (int r = 0; r < G; ++r)
	{
		nb.v[r + 1] = nb.v[r] + mine[2 * r];
		pb.v[r + 1] = pb.v[r] + mine[2 * r + 1];
		nmax = std::max(nmax, mine[2 * r]);
		lmax = std::max(lmax, mine[2 * r + 1]);
	}
	LetView v;
	NBCO_TRY(let_view(c, lay, v));
	int wlog = 0;
	while ((1 << wlog) < c->kd.mlt_max) ++wlog;
	PhaseScope ph(c, NBCO_PH_P2M_M2M);
	if (lmax > 0)   // (an upper bound of the lane count: every selected leaf holds at least one particle)
		hipLaunchKernelGGL(let_pack_pos_kernel, dim3(grid1d(lmax << wlog, 2048), G), dim3(kBlock), 0, c->stream, (const float4 *)c->pos4.as<float4>(),
		                   (const int *)c->kd.index, (const int *)c->kd.mult, lay.L_local, wlog, (long long)lay.rank * lay.n_local, (const int2 *)v.sel_leaf,
		                   (const int *)v.cursors, pb, (float4 *)pos_send);
	NBCO_TRY(c->join_aux());   // the upward pass
	if (nmax > 0)
	{
		if (c->kd.real_bytes == 8)
			hipLaunchKernelGGL(let_pack_mpole_kernel<double>, dim3(grid1d(nmax * rec, 2048), G), dim3(kBlock), 0, c->stream, (const double *)c->kd.mpole, offM, rec,
			                   lay.ntot_local, lay.d, lay.rank, (const int *)v.sel_node, (const int *)v.cursors, nb, (double *)mpole_send);
		else
			hipLaunchKernelGGL(let_pack_mpole_kernel<float>, dim3(grid1d(nmax * rec, 2048), G), dim3(kBlock), 0, c->stream, (const float *)c->kd.mpole, offM, rec,
			                   lay.ntot_local, lay.d, lay.rank, (const int *)v.sel_node, (const int *)v.cursors, nb, (float *)mpole_send);
	}
	NBCO_HIP(hipGetLastError());
	c->dist.let_packed = true;
	return NBCO_OK;
}

static int dist_finish_rest(nbco_ctx *c, const char *mp_blocks, size_t mp_stride, float *buf_local, float *a_local, const float *param, const long long *let_counts,
                            const void *let_pos);

// pos_recv / mpole_recv: the records received from ranks 0, 1, .. (counts_all[s][2 me + 1] / counts_all[s][2 me] from rank s)
int kd_dist_let_finish(nbco_ctx *c, const long long *counts_all, const void *pos_recv, const void *mpole_recv, float *buf_local, float *a_local, const float *param)
{
	if (!c->dist.let_packed) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_finish: call nbco_dist_let_pack first");
	c->dist.let_packed = false; c->dist.let_selected = false;
	return dist_finish_rest(c, (const char *)mpole_recv, 0, buf_local, a_local, param, counts_all, pos_recv);
}

int kd_dist_let_check(nbco_ctx *c)
{
	NBCO_HIP(hipStreamSynchronize(c->stream));
	return let_report(c);
}

// global tree geometry from the gathered traversal records: csz_blocks points at rank 0's records, consecutive ranks are
// csz_stride bytes apart
static int dist_geometry(nbco_ctx *c, const nbco_dist_layout &lay, const TreeView &tv, const char *csz_blocks, size_t csz_stride)
{
	const int d = lay.d, G = lay.world, ntop = (1 << (d + 1)) - 1;
	hipStream_t st = c->stream;
	PhaseScope ph(c, NBCO_PH_P2M_M2M);
	hipLaunchKernelGGL(dist_ranges_kernel, dim3(grid1d(tv.ntot)), dim3(kBlock), 0, st, tv, lay.n_global);
	hipLaunchKernelGGL(dist_unpack_nodes_kernel, dim3(grid1d((long long)G * lay.ntot_local)), dim3(kBlock), 0, st, tv, csz_blocks, csz_stride, lay.ntot_local, G, d);
	if (d > 0)
	{
		// centres, multiplicities and traversal records of the d levels above the domains (boxes from the partition step)
		TopView top = top_view(c, ntop);
		NBCO_TRY(launch_kd_centres_top(c, tv.center, tv.mult, d - 1, top.lbound, top.rbound, tv.csz));
	}
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}

// First half of the finish stage: needs the traversal records and the positions of all domains, not the multipoles.
static int dist_finish_traverse(nbco_ctx *c, const char *csz_blocks, size_t csz_stride, const void *pos_all)
{
	if (!c->dist.local_done && !c->dist.build_done) return c->fail(NBCO_ERR_ARG, "nbco_dist_finish: call nbco_dist_local first");
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	KdTreeDev g;
	NBCO_TRY(dist_global_tree(c, lay, g));
	TreeView tv = view_of(g);
	NBCO_TRY(dist_geometry(c, lay, tv, csz_blocks, csz_stride));
	KdCounts cnt;
	const Dom dm{lay.d, lay.rank};
	NBCO_TRY(kd_interact(c, tv, (const float4 *)pos_all, lay.n_global, g.mlt_max, dm, (long long)lay.rank * lay.n_local, lay.n_local,
	                     c->unsort.as<int>(), nullptr, nullptr, cnt, 1));
	c->dist.pos_all = pos_all;
	c->dist.traversed = true;
	return NBCO_OK;
}

// Second half: multipoles of all domains (rank 0's at mp_blocks, consecutive ranks mp_stride bytes apart), lists, near and far
// field, L2P for the own particles.
// LET exchange (let_counts != null): mp_blocks / let_pos are the received records, let_counts the gathered count matrix
static int dist_finish_rest(nbco_ctx *c, const char *mp_blocks, size_t mp_stride, float *buf_local, float *a_local, const float *param, const long long *let_counts,
                            const void *let_pos)
{
	if (!c->dist.traversed || !c->dist.local_done) return c->fail(NBCO_ERR_ARG, "nbco_dist_finish_rest: the traversal half or the multipoles are missing");
	c->dist.traversed = false;
	c->dist.local_done = false;
	c->dist.build_done = false;
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	const int d = lay.d, G = lay.world, L = lay.L, P = lay.order;
	const int offM = sym_off(P);
	const long long nl = lay.n_local;
	KdTreeDev g;
	NBCO_TRY(dist_global_tree(c, lay, g));
	TreeView tv = view_of(g);
	const Dom dm{d, lay.rank};
	// LET exchange: what arrived, per global node / leaf; received positions into the (sparse) global position array
	long long nodes_in = 0, parts_in = 0;
	const int rec = let_rec_floats(offM);
	LetHave have{nullptr, nullptr};
	if (let_counts)
	{
		const int S = 2 * G + 2, nleaf = 1 << L;
		for (int sdr = 0; sdr < G; ++sdr) { nodes_in += let_counts[(size_t)sdr * S + 2 * lay.rank]; parts_in += let_counts[(size_t)sdr * S + 2 * lay.rank + 1]; }
		NBCO_TRY(c->reserve(c->let_have, (size_t)g.ntot + nleaf + 64));
		unsigned char *hn = c->let_have.as<unsigned char>(), *hl = hn + g.ntot;
		have = LetHave{hn, hl};
		float4 *pos_all = c->dist_pos.as<float4>();
		PhaseScope ph(c, NBCO_PH_P2M_M2M);
		hipLaunchKernelGGL(let_have_own_kernel, dim3(grid1d(g.ntot)), dim3(kBlock), 0, c->stream, hn, hl, g.ntot, L, dm);
		NBCO_HIP(hipMemcpyAsync(pos_all + (size_t)lay.rank * nl, c->pos4.ptr, sizeof(float4) * (size_t)nl, hipMemcpyDeviceToDevice, c->stream));
		if (parts_in > 0)
			hipLaunchKernelGGL(let_unpack_pos_kernel, dim3(grid1d(parts_in, 4096)), dim3(kBlock), 0, c->stream, (const float4 *)let_pos, parts_in, pos_all, lay.n_global, L, hl);
		NBCO_HIP(hipGetLastError());
	}
	// on the second stream, ahead of the M2L list: multipoles into the global arrays, M2M for the levels above the domains
	const std::function<int()> pre_far = [&]() -> int {
		PhaseScope ph(c, NBCO_PH_P2M_M2M);
		const bool f64 = g.real_bytes == 8;   // (dist_global_tree carved the global arrays under the same opts.far_fp64 as the local tree)
		auto unpack_blocks = [&](const char *blocks, size_t stride, int nblocks, int r0) {
			const int grid = grid1d((long long)nblocks * lay.ntot_local * offM);
			if (f64) hipLaunchKernelGGL(dist_unpack_mpole_kernel<double>, dim3(grid), dim3(kBlock), 0, c->stream, tv, blocks, stride, lay.ntot_local, nblocks, r0, d, offM);
			else hipLaunchKernelGGL(dist_unpack_mpole_kernel<float>, dim3(grid), dim3(kBlock), 0, c->stream, tv, blocks, stride, lay.ntot_local, nblocks, r0, d, offM);
		};
		if (offM > 0 && !let_counts) unpack_blocks(mp_blocks, mp_stride, G, 0);
		if (offM > 0 && let_counts)
		{
			// the own subtree straight from the local tree, the rest from the records
			unpack_blocks((const char *)c->kd.mpole, (size_t)0, 1, lay.rank);
			if (nodes_in > 0)
			{
				if (f64)
					hipLaunchKernelGGL(let_unpack_mpole_kernel<double>, dim3(grid1d(nodes_in * rec, 4096)), dim3(kBlock), 0, c->stream, (const double *)mp_blocks, nodes_in, rec,
					                   offM, g.ntot, (double *)tv.mpole, const_cast<unsigned char *>(have.node));
				else
					hipLaunchKernelGGL(let_unpack_mpole_kernel<float>, dim3(grid1d(nodes_in * rec, 4096)), dim3(kBlock), 0, c->stream, (const float *)mp_blocks, nodes_in, rec,
					                   offM, g.ntot, tv.mpole, const_cast<unsigned char *>(have.node));
			}
		}
		if (d > 0) NBCO_TRY(launch_m2m_top_gen(c, P, tv.center, tv.mpole, tv.mult, d - 1, 0, f64 ? 1 : 0));
		NBCO_HIP(hipGetLastError());
		return NBCO_OK;
	};
	const LetHave *let = let_counts ? &have : nullptr;
	KdCounts cnt;
	int rc = kd_interact(c, tv, (const float4 *)c->dist.pos_all, lay.n_global, g.mlt_max, dm, (long long)lay.rank * nl, nl, c->unsort.as<int>(), a_local, param,
	                     cnt, 2, &pre_far, let);
	while ((rc == NBCO_ERR_CAPACITY && c->grow_lists(g.ntot)) || (rc == NBCO_OK && cnt.react_overflow && !cnt.sel_overflow))
		// twice the room (or reaction records sized from the count just seen), traversal and the rest again (the global arrays,
		// multipoles included, are in place; purely local)
		rc = kd_interact(c, tv, (const float4 *)c->dist.pos_all, lay.n_global, g.mlt_max, dm, (long long)lay.rank * nl, nl, c->unsort.as<int>(), a_local, param,
		                 cnt, 0, nullptr, let);
	if (rc != NBCO_OK) return rc;
	if (cnt.sel_overflow) return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_dist_finish: unresolved tie overflow of the selection build");
	if (c->dist.rebuilt) NBCO_TRY(kd_finish_order(c, buf_local, nl));
	c->tree_valid = true;
	c->tree_n = nl;
	c->tree_order = P;
	c->eval_counter += 1;
	nbco_kd_info &info = c->info;
	info.L = L; info.ntot = g.ntot; info.order = P; info.mlt_max = g.mlt_max; info.n = lay.n_global;
	info.p2p_pairs = cnt.np2p; info.m2l_pairs = cnt.nm2l; info.rebuilt = c->dist.rebuilt ? 1 : 0;
	info.directed_p2p = -1;
	return NBCO_OK;
}

int kd_dist_finish_traverse(nbco_ctx *c, const void *csz_all, const void *pos_all)
{
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	return dist_finish_traverse(c, (const char *)csz_all, (size_t)lay.csz_bytes, pos_all);
}
int kd_dist_finish_rest(nbco_ctx *c, const void *mpole_all, float *buf_local, float *a_local, const float *param)
{
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	return dist_finish_rest(c, (const char *)mpole_all, (size_t)lay.mpole_bytes, buf_local, a_local, param, nullptr, nullptr);
}
// both halves on node blocks gathered as a whole (per rank: records, then multipoles)
int kd_dist_finish(nbco_ctx *c, const void *nodes_all, const void *pos_all, float *buf_local, float *a_local, const float *param)
{
	if (!c->dist.local_done) return c->fail(NBCO_ERR_ARG, "nbco_dist_finish: call nbco_dist_local first");
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	NBCO_TRY(dist_finish_traverse(c, (const char *)nodes_all, (size_t)lay.nodes_bytes, pos_all));
	return dist_finish_rest(c, (const char *)nodes_all + lay.csz_bytes, (size_t)lay.nodes_bytes, buf_local, a_local, param, nullptr, nullptr);
}

// directed pair interactions of the last evaluation (nbco_kd_get_info): sum over the sorted P2P entries
int kd_count_pairs(nbco_ctx *c, long long *out)
{
	if (!c->pc_mult || !c->p2p_keys_alt.ptr) { *out = 0; return NBCO_OK; }
	unsigned long long *ctr = (unsigned long long *)(c->counters.as<int>() + 100);
	NBCO_HIP(hipMemsetAsync(ctr, 0, sizeof(unsigned long long), c->stream));
	hipLaunchKernelGGL(pair_count_kernel, dim3(1024), dim3(kBlock), 0, c->stream, c->pc_mult, (const uint64_t *)c->p2p_keys_alt.as<uint64_t>(), c->pc_total,
	                   c->pc_shift, ctr);
	NBCO_HIP(hipGetLastError());
	unsigned long long v = 0;
	NBCO_HIP(hipMemcpyAsync(&v, ctr, sizeof v, hipMemcpyDeviceToHost, c->stream));
	NBCO_HIP(hipStreamSynchronize(c->stream));
	*out = (long long)v;
	return NBCO_OK;
}

int kd_copy_out(nbco_ctx *c, int which, void *dst, long long bytes)
{
	const KdTreeDev &k = c->kd;
	if (!c->tree_valid) return c->fail(NBCO_ERR_ARG, "nbco_kd_copy: no kd-tree evaluation has run");
	const void *src = nullptr;
	size_t need = 0;
	const int offM = sym_off(k.order), offL = tl_off(k.order + 1);
	switch (which)
	{
	case NBCO_KD_MULT: src = k.mult; need = 4 * (size_t)k.ntot; break;
	case NBCO_KD_INDEX: src = k.index; need = 4 * (size_t)k.ntot; break;
	case NBCO_KD_SPLITDIM: src = k.splitdim; need = 4 * (size_t)k.ntot; break;
	case NBCO_KD_CENTER: src = k.center; need = 12 * (size_t)k.ntot; break;
	case NBCO_KD_LBOUND: src = k.lbound; need = 12 * (size_t)k.ntot; break;
	case NBCO_KD_RBOUND: src = k.rbound; need = 12 * (size_t)k.ntot; break;
	case NBCO_KD_MPOLE: src = k.mpole; need = (size_t)k.real_bytes * (size_t)k.ntot * offM; break;   // doubles after an evaluation with opts.far_fp64
	case NBCO_KD_LOCAL: src = k.local; need = (size_t)k.real_bytes * (size_t)k.ntot * offL; break;
	case NBCO_KD_P2P_LIST:
	case NBCO_KD_M2L_LIST:
	{
		// the pair lists live in regions: make a dense copy in the (idle) frontier buffer
		const bool p2p = which == NBCO_KD_P2P_LIST;
		need = 8 * (size_t)(p2p ? c->info.p2p_pairs : c->info.m2l_pairs);
		const long long capR = c->list_cap / kTravK;
		hipLaunchKernelGGL(list_compact_kernel, dim3(1024), dim3(kBlock), 0, c->stream, (const int2 *)(p2p ? c->p2p_list.ptr : c->m2l_list.ptr),
		                   (const int *)(c->trav_ctr.as<int>() + (p2p ? kTcP2PPref : kTcM2LPref)), capR, c->frontier_a.as<int2>());
		NBCO_HIP(hipGetLastError());
		src = c->frontier_a.ptr;
		break;
	}
	case NBCO_KD_UNSORT: src = c->unsort.ptr; need = 4 * (size_t)k.n; break;
	case NBCO_KD_ORDER:
		if (!c->o.track_order || c->order_n != k.n) return c->fail(NBCO_ERR_ARG, "nbco_kd_copy: NBCO_KD_ORDER needs opts.track_order and unsort = 0 evaluations");
		src = c->order.ptr; need = 4 * (size_t)k.n; break;
	default: return c->fail(NBCO_ERR_ARG, "nbco_kd_copy: unknown array");
	}
	if ((long long)need > bytes) return c->fail(NBCO_ERR_ARG, "nbco_kd_copy: destination too small");
	NBCO_HIP(hipStreamSynchronize(c->stream));
	if (need) NBCO_HIP(hipMemcpy(dst, src, need, hipMemcpyDeviceToHost));
	return NBCO_OK;
}

NBCO_CHECKED_COLLECT(nbco_checked_collect_kd)
