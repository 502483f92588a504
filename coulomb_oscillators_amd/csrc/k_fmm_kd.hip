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
#include <vector>

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


#include "kd_build_kernels.hpp"   // tree geometry (rounded like the oracle), build prologue, the fused turnaround pass, per-level kernels of the sorting build, the in-LDS subtree build
#include "kd_traverse_kernels.hpp"   // dual tree traversal: admissibility, the level-synchronous frontier kernel, its init / finish kernels
#include "kd_list_kernels.hpp"   // directed lists by counting sort: fill, per-target sort with source descriptors and work units, pair count; state reorder kernels
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
		// Long ranges (leaves stretched by ejected particles, late in a run) go to a kernel of their own when the previous
		// evaluation's list says they are to be expected: more than 48 entries per target on average (17 in the benchmark's first
		// steps, 109 a thousand steps in).  Either way every range is sorted; the choice only moves the long ones.
		const bool long_kernel = shift >= 11 && shift <= 16 && c->hint_np2p > 0 && 2 * c->hint_np2p > 48LL * (nself > 0 ? nself : ntargets);   // (per OWN target)
		const int long_from = long_kernel ? 512 : 0;
		c->info.long_lists = long_kernel ? 1 : 0;
		hipLaunchKernelGGL(list_segsort_kernel<true>, dim3((ntargets + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, (const int *)start, ntargets,
		                   keys_tmp, keys_out, shift, leaf_index, leaf_mult, desc, mu, long_from);
		if (long_kernel)
			hipLaunchKernelGGL(list_longsort_kernel, dim3(256), dim3(kLongBlock), 0, st, (const int *)start,
			                   ntargets, (const uint64_t *)keys_tmp, keys_out, shift, leaf_index, leaf_mult, desc, mu, long_from);
	}
	else
		hipLaunchKernelGGL(list_segsort_kernel<false>, dim3((ntargets + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, (const int *)start,
		                   ntargets, keys_tmp, keys_out, shift, (const int *)nullptr, (const int *)nullptr, (int2 *)nullptr, MutualLists{}, 0);
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
			                   use_select ? 1 : 0, c->sel_three_pass ? 0 : 1, c->counters.as<int>() + 110, c->top_sd, c->top_root1);
#ifdef NBCO_SUBTREE_PROF
			// the kernel only reads its inputs: a second launch right behind the first one repeats it with warm instruction caches
			if (std::getenv("NBCO_SUBTREE_TWICE"))
				hipLaunchKernelGGL(kd_subtree_kernel, dim3(kd_cnt(l0)), dim3(kSubT), 0, st, tv, (const float4 *)pos, (const int *)unsort, pos_alt, unsort_alt, n, l0,
				                   use_select ? 1 : 0, c->sel_three_pass ? 0 : 1, c->counters.as<int>() + 110, c->top_sd, c->top_root1);
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
constexpr int kLetWord = 16;   // h_flags[16..19]: the guard's two words, one pair per parity of the attempt (DistState::let_epoch)
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
	hipLaunchKernelGGL(let_guard_kernel, dim3(grid1d(hint, 2048)), dim3(kBlock), 0, c->stream, keys, total, shift, have,
	                   c->h_flags + kLetWord + 2 * (int)(c->dist.let_epoch & 1) + word);
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

#include "kd_energy_kernels.hpp"   // FMM potential energy: multipole-to-particle potential and the per-particle pass
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

#include "kd_dist.hpp"   // multi-GPU: kd-domain sharding (layout, partition, local stage, global tree)
#include "kd_let.hpp"   // multi-GPU: locally-essential-tree exchange and the second half of the sharded evaluation
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
