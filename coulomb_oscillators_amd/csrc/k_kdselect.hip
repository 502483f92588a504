// k_kdselect.hip -- top levels of the kd-tree build by exact median SELECTION instead of sorting.
//
// The reference sorts every node's particles along its split axis at every level (stable sort chain,
// fmm_cart3_kdtree.cuh:1858-1871) and cuts at the median.  The order that chain produces inside node v
// is the lexicographic order on (c[a1], c[a2], c[a3], original index) with a1 = splitdim(v) and a2, a3
// the next distinct split axes of v's ancestors (most recent first); the children are the first k
// and the remaining elements of that order.  Only the children's particle SETS and the two boundary
// coordinates enter the tree, so while a node is too large for one workgroup's LDS the level is
// built by
//   1. a radix select of the k-th smallest ordered key c[a1] per node: 11 + 11 bits, which leaves a handful of candidates
//      in the pivot's bucket (three passes, 11 + 11 + 10 bits of the ordered float, for nodes above 2^22 particles or after a
//      bucket overflow),
//   2. an unordered partition into < pivot | > pivot with per-node atomics (block-aggregated),
//   3. an exact resolution of the candidates -- the elements that share the pivot's bucket or tie with the pivot -- by
//      (c[a1], c[a2], c[a3], index),
//   4. evalBox for the children from the pivot value and the smallest key of the right part.
// The canonical order is re-established in LDS when the subtree kernel takes over (k_fmm_kd.hip).
// If more than kTieCap elements tie with a pivot (degenerate inputs) a flag is raised and the caller
// rebuilds with the sorting path.
#include "nbco_internal.hpp"
#include "kd_common.hpp"

namespace {

using namespace kdc;

// Workgroups of BLOCK threads take chunks of 8 * BLOCK consecutive elements (4 * BLOCK in the warm kernels); a chunk must touch at most two nodes.
// Atomics on one address retire at ~27 ns each on this part, and every workgroup of a node adds to the node's histogram
// bins, completion counters and partition cursors: levels whose nodes hold >= 8192 particles use 1024-thread workgroups
// (8192-element chunks), a quarter of the workgroups per node; the rest (nodes of 4097..8191) use 256 / 2048.
constexpr int kBlockBig = 1024, kBlockSmall = 256;
constexpr int kBins = 2048;
constexpr int kTieCap = 64;
constexpr int kTieWords = 8;   // a candidate's record: {k1, k2, k3, original index, x, y, z, -}

// per-node state of one level (zeroed for all levels by one memset per build)
struct SelNode
{
	// select state after pass 0 ([0]) and after pass 1 ([1]): digits chosen so far and the 0-based rank of the pivot inside
	// the remaining candidates.  Written during the launch that resolves the pass, read by the launch after it (two slots: a
	// launch never overwrites what its own workgroups are reading)
	uint32_t prefix[2], r[2];
	uint32_t done_part; // blocks that have finished their part of the partition
	uint32_t cntL, cntR, tiecnt;
	uint32_t minR;     // smallest ordered key of the right child, stored inverted (~key) so that zero = "none yet"
	uint32_t below;    // warm select: elements whose bucket lies below the histogram window
};

struct SelPivot { uint32_t prefix, r, neq, need; };

__device__ inline uint32_t ld_agent_u32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Locate the bin holding rank r in one 2048-bin histogram and descend into it.  Called by all BLOCK threads.
template <int BLOCK>
__device__ inline void descend(const uint32_t *__restrict__ h, int bits, SelPivot &pv, uint32_t *sh /* [BLOCK/64 + 4] */)
{
	constexpr int R = BLOCK / 64;   // result slots follow the wave sums
	constexpr int PER = kBins / BLOCK;
	uint32_t v[PER], s = 0;
#pragma unroll
	for (int q = 0; q < PER; ++q) { v[q] = ld_agent_u32(&h[threadIdx.x * PER + q]); s += v[q]; }
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t incl = s;
	for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(incl, o); if (lane >= o) incl += y; }
	__syncthreads();
	if (lane == 63) sh[w] = incl;
	__syncthreads();
	uint32_t base = 0;
	for (int q = 0; q < w; ++q) base += sh[q];
	uint32_t cum = base + incl - s;
#pragma unroll
	for (int q = 0; q < PER; ++q)
	{
		if (pv.r >= cum && pv.r < cum + v[q]) { sh[R] = threadIdx.x * PER + q; sh[R + 1] = cum; sh[R + 2] = v[q]; }
		cum += v[q];
	}
	__syncthreads();
	pv.prefix = (pv.prefix << bits) | sh[R];
	pv.r -= sh[R + 1];
	pv.neq = sh[R + 2];
	pv.need = pv.r + 1;
}

// All keys of a node lie between the ordered images of its box faces along the split axis.  Subtracting the lower
// face and shifting the span up to bit 31 is order preserving and spreads the FIRST radix digit over all bins
// (the raw top 11 bits of neighbouring coordinates are nearly identical: every LDS atomic would hit one bin).
__device__ inline void key_window(const float *__restrict__ lbound, const float *__restrict__ rbound, const int *__restrict__ sd_l, int l,
                                  long long j, uint32_t &kmin, int &shl)
{
	const int node = (1 << l) - 1 + (int)j, a = sd_l[j];
	kmin = ordered_bits(lbound[3 * node + a]);
	const uint32_t span = ordered_bits(rbound[3 * node + a]) - kmin;
	shl = span ? __clz(span) : 0;
}

// Bucket key of the early-stop select (two passes): any monotone non-decreasing map of the coordinate will do, because the
// elements of the pivot's bucket are ordered exactly afterwards.  The ordered float bits are piecewise linear (2^23 keys per
// binade: a box that reaches down to ~0 keeps half of its particles in 1/256 of the key range); this map is linear over the
// node's box, so buckets hold node size / 2^22 elements times the density contrast inside the box.  Subtraction,
// multiplication and the truncating conversion are each monotone, and every kernel evaluates the same expression.
__device__ inline void lin_window(const float *__restrict__ lbound, const float *__restrict__ rbound, const int *__restrict__ sd_l, int l,
                                  long long j, float &lo, float &scale)
{
	const int node = (1 << l) - 1 + (int)j, a = sd_l[j];
	lo = lbound[3 * node + a];
	const float span = __fsub_rn(rbound[3 * node + a], lo);
	scale = span > 0.f ? __fdiv_rn(4294967040.f, span) : 0.f;
}
__device__ inline uint32_t lin_key(float x, float lo, float scale)
{
	const float v = __fmul_rn(__fsub_rn(x, lo), scale);
	return (uint32_t)fminf(fmaxf(v, 0.f), 4294967040.f);
}

// number of CHUNK-sized blocks whose element range overlaps node j
template <int CHUNK>
__device__ inline uint32_t chunks_of_node(long long n, long long j, long long m)
{
	const long long s = range_start(n, j, m), e = range_start(n, j + 1, m);
	return (uint32_t)((e - 1) / CHUNK - s / CHUNK + 1);
}

// The block that completes node j's histogram of pass PASS descends into the bin holding the pivot rank and
// publishes the new select state (the classic last-block-done pattern: nobody waits).
// Select state of node j after the passes before PASS, computed by EVERY workgroup that touches the node at the start
// of the launch of pass PASS (PASS = 3: the partition): the result of the passes before the previous one is read from
// nodes[j] (written during the previous launch), the previous pass's histogram is descended here.  Compared with letting
// the last workgroup of each pass do it, this takes the atomics drain, the completion counter and the serial descent off
// the tail of every launch (three dependent round trips) and puts one descent beside the element loads of the next one.
// The workgroup that holds the node's first element records the state for the launches after it.
template <int PASS, int BLOCK>
__device__ inline SelPivot resolve_before(const uint32_t *__restrict__ hist, SelNode *__restrict__ nodes, long long n, int l, long long j,
                                          long long i0, uint32_t *sh)
{
	constexpr int CHUNK = 8 * BLOCK;
	const long long m = 1LL << l;
	SelPivot pv;
	pv.prefix = 0; pv.neq = 0; pv.need = 0;
	pv.r = (uint32_t)(range_start(n, 2 * j + 1, 2 * m) - range_start(n, j, m) - 1);
	if (PASS == 0) return pv;
	if (PASS >= 2) { pv.prefix = nodes[j].prefix[PASS - 2]; pv.r = nodes[j].r[PASS - 2]; }
	descend<BLOCK>(hist + ((size_t)(PASS - 1) * m + j) * kBins, PASS == 3 ? 10 : 11, pv, sh);
	const long long first = range_start(n, j, m);
	if (PASS <= 2 && threadIdx.x == 0 && first >= i0 && first < i0 + CHUNK) { nodes[j].prefix[PASS - 1] = pv.prefix; nodes[j].r[PASS - 1] = pv.r; }
	return pv;
}

template <int PASS, int BLOCK, bool LIN>
__global__ __launch_bounds__(BLOCK) void sel_hist_kernel(const float4 *__restrict__ pos, const int *__restrict__ sd_l, uint32_t *__restrict__ hist,
                                                          SelNode *__restrict__ nodes, const float *__restrict__ lbound,
                                                          const float *__restrict__ rbound, long long n, int l)
{
	constexpr int CHUNK = 8 * BLOCK;
	__shared__ uint32_t h[2][kBins];
	__shared__ uint32_t sh[BLOCK / 64 + 4];
	const long long m = 1LL << l;
	const long long i0 = (long long)blockIdx.x * CHUNK;
	// element loads first: they are independent of the node state resolved below
	constexpr int PER = CHUNK / BLOCK;
	float4 p[PER];
#pragma unroll
	for (int e = 0; e < PER; ++e)
	{
		const long long i = i0 + e * BLOCK + threadIdx.x;
		if (i < n) p[e] = pos[i];
	}
	for (int t = threadIdx.x; t < 2 * kBins; t += BLOCK) (&h[0][0])[t] = 0;
	const long long ilast = (i0 + CHUNK < n ? i0 + CHUNK : n) - 1;
	const long long j0 = div_floor_small(m * i0, n), j1 = div_floor_small(m * ilast, n);
	uint32_t pfx[2] = {0, 0}, kmin[2] = {0, 0};
	int shl[2] = {0, 0};
	float lo[2] = {0.f, 0.f}, scale[2] = {0.f, 0.f};
	for (int jj = 0; jj < 2; ++jj)
		if (j0 + jj <= j1)
		{
			pfx[jj] = resolve_before<PASS, BLOCK>(hist, nodes, n, l, j0 + jj, i0, sh).prefix;   // uniform over the block
			if (LIN) lin_window(lbound, rbound, sd_l, l, j0 + jj, lo[jj], scale[jj]);
			else key_window(lbound, rbound, sd_l, l, j0 + jj, kmin[jj], shl[jj]);
		}
	__syncthreads();
	const long long split = j1 > j0 ? range_start(n, j1, m) : n;
	const int sd[2] = {sd_l[j0], sd_l[j1]};
#pragma unroll
	for (int e = 0; e < PER; ++e)
	{
		const long long i = i0 + e * BLOCK + threadIdx.x;
		if (i < n)
		{
			const int jj = i >= split ? 1 : 0;
			const float x = axis_of(p[e], sd[jj]);
			const uint32_t key = LIN ? lin_key(x, lo[jj], scale[jj]) : (ordered_bits(x) - kmin[jj]) << shl[jj];
			bool ok = true;
			uint32_t d = key >> 21;
			if (PASS == 1) { ok = (key >> 21) == pfx[jj]; d = (key >> 10) & 0x7FFu; }
			if (PASS == 2) { ok = (key >> 10) == pfx[jj]; d = key & 0x3FFu; }
			if (ok) atomicAdd(&h[jj][d], 1u);
		}
	}
	__syncthreads();
	for (int t = threadIdx.x; t < 2 * kBins; t += BLOCK)
	{
		const uint32_t v = (&h[0][0])[t];
		const long long j = j0 + (t / kBins);
		if (v && j <= j1) atomicAdd(&hist[((size_t)PASS * m + j) * kBins + (t % kBins)], v);
	}
}

// ---- warm select: ONE histogram pass per level instead of two ---------------------------------------------------------------------
// Between two builds of a simulation the particles move by ~1e-5 of a box, and so do the medians.  The previous build's pivot of
// node j is still in the tree arrays when level l is selected -- it is the upper face, along the split axis, of the left child's
// OLD box, which this level's partition overwrites only at its very end.  The pass histograms only a WINDOW of kWarmBins buckets
// of the box-linear key (bucket = key >> drop, drop chosen per level so that a bucket holds well under one element on average)
// centred on that prediction, and counts what lies below it; the partition finds the pivot's bucket in it exactly as it does in
// the second pass's histogram.  A window that misses the median (first build after a jump, a changed split axis, ..) raises
// flag 2: the partition leaves the node untouched -- exactly what it does with a tie overflow -- and the host repeats the
// evaluation with the two-pass select.
constexpr int kWarmBins = 3 * kBins;
__device__ inline uint32_t warm_window_start(const float *__restrict__ lbound, const float *__restrict__ rbound, const int *__restrict__ sd_l, int l,
                                             long long j, float lo, float scale, int drop)
{
	const int left = (2 << l) - 1 + 2 * (int)j, a = sd_l[j];
	const uint32_t c = lin_key(rbound[3 * left + a], lo, scale) >> drop;   // bucket of the previous pivot under the current box
	return c > (uint32_t)(kWarmBins / 2) ? c - (uint32_t)(kWarmBins / 2) : 0u;
}

template <int BLOCK, int EPT = 8>
__global__ __launch_bounds__(BLOCK) void sel_hist_warm_kernel(const float4 *__restrict__ pos, const int *__restrict__ sd_l, uint32_t *__restrict__ hist,
                                                               SelNode *__restrict__ nodes, const float *__restrict__ lbound,
                                                               const float *__restrict__ rbound, long long n, int l, int drop)
{
	constexpr int CHUNK = EPT * BLOCK;
	__shared__ uint32_t h[2][kWarmBins];
	__shared__ uint32_t below[2];
	const long long m = 1LL << l;
	const long long i0 = (long long)blockIdx.x * CHUNK;
	constexpr int PER = CHUNK / BLOCK;
	float4 p[PER];
#pragma unroll
	for (int e = 0; e < PER; ++e)
	{
		const long long i = i0 + e * BLOCK + threadIdx.x;
		if (i < n) p[e] = pos[i];
	}
	for (int t = threadIdx.x; t < 2 * kWarmBins; t += BLOCK) (&h[0][0])[t] = 0;
	if (threadIdx.x < 2) below[threadIdx.x] = 0;
	const long long ilast = (i0 + CHUNK < n ? i0 + CHUNK : n) - 1;
	const long long j0 = div_floor_small(m * i0, n), j1 = div_floor_small(m * ilast, n);
	// split axis, box and the left child's old box of the chunk's one or two nodes in one round trip (all three axes are
	// fetched before the axis is known; the partition does the same)
	__shared__ uint32_t meta[2][16];
	if (threadIdx.x < 32)
	{
		const int jj = threadIdx.x >> 4, t = threadIdx.x & 15;
		const long long j = j0 + jj;
		uint32_t v = 0;
		if (j <= j1)
		{
			const int node = (int)(m - 1 + j), left = 2 * node + 1;
			if (t == 0) v = (uint32_t)sd_l[j];
			else if (t < 4) v = __float_as_uint(lbound[3 * node + (t - 1)]);
			else if (t < 7) v = __float_as_uint(rbound[3 * node + (t - 4)]);
			else if (t < 10) v = __float_as_uint(rbound[3 * left + (t - 7)]);
		}
		meta[jj][t] = v;
	}
	__syncthreads();
	float lo[2] = {0.f, 0.f}, scale[2] = {0.f, 0.f};
	uint32_t w0[2] = {0, 0};
	for (int jj = 0; jj < 2; ++jj)
		if (j0 + jj <= j1)
		{
			const int a = (int)meta[jj][0];
			lo[jj] = __uint_as_float(meta[jj][1 + a]);                                         // lin_window
			const float span = __fsub_rn(__uint_as_float(meta[jj][4 + a]), lo[jj]);
			scale[jj] = span > 0.f ? __fdiv_rn(4294967040.f, span) : 0.f;
			const uint32_t c = lin_key(__uint_as_float(meta[jj][7 + a]), lo[jj], scale[jj]) >> drop;   // warm_window_start
			w0[jj] = c > (uint32_t)(kWarmBins / 2) ? c - (uint32_t)(kWarmBins / 2) : 0u;
		}
	const long long split = j1 > j0 ? range_start(n, j1, m) : n;
	const int sd[2] = {(int)meta[0][0], j1 > j0 ? (int)meta[1][0] : (int)meta[0][0]};
	uint32_t nb[2] = {0, 0};
#pragma unroll
	for (int e = 0; e < PER; ++e)
	{
		const long long i = i0 + e * BLOCK + threadIdx.x;
		if (i < n)
		{
			const int jj = i >= split ? 1 : 0;
			const uint32_t b = lin_key(axis_of(p[e], sd[jj]), lo[jj], scale[jj]) >> drop;
			if (b < w0[jj]) nb[jj] += 1;
			else if (b - w0[jj] < (uint32_t)kWarmBins) atomicAdd(&h[jj][b - w0[jj]], 1u);
		}
	}
#pragma unroll
	for (int q = 0; q < 2; ++q)
	{
		uint32_t v = nb[q];
		for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
		if ((threadIdx.x & 63) == 0 && v) atomicAdd(&below[q], v);
	}
	__syncthreads();
	for (int t = threadIdx.x; t < 2 * kWarmBins; t += BLOCK)
	{
		const uint32_t v = (&h[0][0])[t];
		const long long j = j0 + (t / kWarmBins);
		const int bin = t % kWarmBins;
		// window part w = bin / kBins of node j lives where pass w's histogram would (same storage, zeroed by the build prologue)
		if (v && j <= j1) atomicAdd(&hist[((size_t)(bin / kBins) * m + j) * kBins + (bin % kBins)], v);
	}
	if (threadIdx.x < 2 && j0 + threadIdx.x <= j1 && below[threadIdx.x]) atomicAdd(&nodes[j0 + threadIdx.x].below, below[threadIdx.x]);
}

// the window histogram of node j -> the pivot's bucket; pv.need == 0 reports a window that does not hold the median
// (the thread's bins v[] and the count below the window were fetched by the caller, next to everything else it needs to know
// about the node: warm_bins)
template <int BLOCK>
__device__ inline void warm_bins(const uint32_t *__restrict__ hist, long long m, long long j, uint32_t (&v)[kWarmBins / BLOCK])
{
#pragma unroll
	for (int q = 0; q < kWarmBins / BLOCK; ++q)
	{
		const int bin = threadIdx.x * (kWarmBins / BLOCK) + q;
		v[q] = ld_agent_u32(&hist[((size_t)(bin / kBins) * m + j) * kBins + (bin % kBins)]);
	}
}
template <int BLOCK>
__device__ inline SelPivot resolve_warm(const uint32_t (&v)[kWarmBins / BLOCK], uint32_t below, long long n, int l, long long j, uint32_t w0,
                                        uint32_t *sh /* [BLOCK/64 + 4] */)
{
	constexpr int R = BLOCK / 64, PER = kWarmBins / BLOCK;
	const long long m = 1LL << l;
	SelPivot pv;
	pv.prefix = 0; pv.neq = 0; pv.need = 0;
	const long long rank = range_start(n, 2 * j + 1, 2 * m) - range_start(n, j, m) - 1 - (long long)below;
	uint32_t s = 0;
#pragma unroll
	for (int q = 0; q < PER; ++q) s += v[q];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t incl = wave_scan_add(s);
	__syncthreads();
	if (lane == 63) sh[w] = incl;
	if (threadIdx.x == 0) sh[R + 3] = 0;
	__syncthreads();
	// the waves' totals, scanned by every wave for itself
	const uint32_t wt = lane < R ? sh[lane] : 0u;
	const uint32_t base = __shfl(wave_scan_add(wt) - wt, w);
	uint32_t cum = base + incl - s;
	if (rank >= 0)
	{
		const uint32_t r = (uint32_t)rank;
#pragma unroll
		for (int q = 0; q < PER; ++q)
		{
			if (r >= cum && r < cum + v[q]) { sh[R] = threadIdx.x * PER + q; sh[R + 1] = cum; sh[R + 2] = v[q]; sh[R + 3] = 1; }
			cum += v[q];
		}
	}
	__syncthreads();
	if (sh[R + 3])
	{
		pv.prefix = w0 + sh[R];
		pv.r = (uint32_t)rank - sh[R + 1];
		pv.neq = sh[R + 2];
		pv.need = pv.r + 1;
	}
	__syncthreads();
	return pv;
}

// exclusive scan over the block of four 16-bit counters packed in a uint64 (each block total <= CHUNK < 2^16)
template <int BLOCK>
__device__ inline uint64_t block_scan4(uint64_t v, uint64_t *sh_wave, uint64_t &total)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint64_t incl = v;
	for (int o = 1; o < 64; o <<= 1)
	{
		const uint64_t y = __shfl_up(incl, o);
		if (lane >= o) incl += y;
	}
	if (lane == 63) sh_wave[w] = incl;
	__syncthreads();
	uint64_t base = 0, tot = 0;
	for (int k = 0; k < BLOCK / 64; ++k)
	{
		const uint64_t t = sh_wave[k];
		if (k < w) base += t;
		tot += t;
	}
	total = tot;
	return base + incl - v;
}

#ifdef NBCO_SUBTREE_PROF
// profiling build only (make prof): phase timestamps (100 MHz) of one workgroup of the partition of level 7, tools/subtree_prof.py
__device__ long long g_part_prof[64];
#define PART_MARK(k) do { if (l == 7 && blockIdx.x == 37 && threadIdx.x == 0) g_part_prof[(k)] = wall_clock64(); } while (0)
extern "C" int nbco_debug_partition_prof(long long *out64)
{
	return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_part_prof), sizeof(long long) * 64);
}
#else
#define PART_MARK(k)
#endif

#pragma clang fp contract(off)
// What a workgroup has to know about one of its nodes, fetched in ONE round trip at the head of the partition by 64 of its threads
// (every address follows from the node's number): words 0 .. 31 the split axes of the ancestors (nearest first), then the
// node's own split axis, its box, the OLD box of its left child (warm select: the previous build's pivot) and the count below
// the warm window.
constexpr int kMetaSd = 32, kMetaLb = 33, kMetaRb = 36, kMetaOld = 39, kMetaBelow = 42, kMetaWords = 64;

// Run by ONE wave once every workgroup of node j has finished its part of the partition (last-block-done).  It first
// orders the elements that tie with the pivot by the remaining keys of the stable-sort chain -- the next distinct
// ancestor split axes, then the original index -- and hands the first `need` of them to the left child; then evalBox
// for the two children (fmm_cart3_kdtree.cuh:109-137): the sorted order's boundary elements are the pivot (largest
// key of the left child) and the smallest key of the right child.
__device__ inline void ties_and_boxes(float4 *__restrict__ pos_out, int *__restrict__ unsort_out, float *__restrict__ lbound, float *__restrict__ rbound,
                                      int *__restrict__ splitdim, int *__restrict__ index, SelNode *__restrict__ nodes,
                                      const uint32_t *__restrict__ tielist, int *__restrict__ flag, long long n, int l, int j, int lane,
                                      const SelPivot pv /* three passes: prefix = the pivot as an ordered key */, const uint32_t *meta)
{
	const int m = 1 << l;
	const int a1 = (int)meta[kMetaSd];
	// one round trip: the node's counters and (whether or not there are any) the lane's candidate record.  The candidates are
	// the elements that tie with the pivot (three passes) or share its bucket (two passes, warm select); the workgroups that
	// met them left complete records {k1, k2, k3, original index, x, y, z}
	const uint32_t *rec = tielist + ((size_t)j * kTieCap + lane) * kTieWords;
	const uint32_t nt = ld_agent_u32(&nodes[j].tiecnt), cl = ld_agent_u32(&nodes[j].cntL), cr = ld_agent_u32(&nodes[j].cntR);
	const uint32_t inv = ld_agent_u32(&nodes[j].minR);   // inverted, 0 = no element above the candidates
	const uint32_t k1 = ld_agent_u32(rec), k2 = ld_agent_u32(rec + 1), k3 = ld_agent_u32(rec + 2), org = ld_agent_u32(rec + 3);
	const uint32_t bx = ld_agent_u32(rec + 4), by = ld_agent_u32(rec + 5), bz = ld_agent_u32(rec + 6);
	// the pivot (largest key of the left child) and the smallest candidate key that went right
	uint32_t pivot = pv.prefix, cand_right = 0xFFFFFFFFu;
	if (nt > kTieCap) { if (lane == 0) *flag = 1; }
	else if (nt > 0)
	{
		// Their order in the stable-sort chain is (c[a1], c[a2], c[a3], original index); the first `need` belong to the left
		// child.  Every other element of the node has its slot by now, so the candidates' slots follow the final cursors.
		const bool mine = (uint32_t)lane < nt;
		uint32_t rank = 0;
		for (uint32_t q = 0; q < nt; ++q)
		{
			const uint32_t q1 = __shfl(k1, q), q2 = __shfl(k2, q), q3 = __shfl(k3, q), qo = __shfl(org, q);
			const bool before = q1 < k1 || (q1 == k1 && (q2 < k2 || (q2 == k2 && (q3 < k3 || (q3 == k3 && qo < org)))));
			rank += before ? 1u : 0u;
		}
		if (mine)
		{
			const long long dst = rank < pv.need ? range_start(n, j, m) + cl + rank : range_start(n, 2 * j + 1, 2LL * m) + cr + (rank - pv.need);
			pos_out[dst] = make_float4(__uint_as_float(bx), __uint_as_float(by), __uint_as_float(bz), 0.f);
			unsort_out[dst] = (int)org;
		}
		const unsigned long long at_pivot = __ballot(mine && rank + 1 == pv.need), after = __ballot(mine && rank == pv.need);
		if (at_pivot) pivot = __shfl(k1, __ffsll((long long)at_pivot) - 1);
		if (after) cand_right = __shfl(k1, __ffsll((long long)after) - 1);
	}
	if (lane < 2)
	{
		const int c = 2 * j + lane, child = 2 * m - 1 + c;
		float lb[3] = {__uint_as_float(meta[kMetaLb]), __uint_as_float(meta[kMetaLb + 1]), __uint_as_float(meta[kMetaLb + 2])};
		float rb[3] = {__uint_as_float(meta[kMetaRb]), __uint_as_float(meta[kMetaRb + 1]), __uint_as_float(meta[kMetaRb + 2])};
		if (pv.need == 0) { /* warm select, window missed: nothing was moved; the children inherit the box (flag 2 is up) */ }
		else if (c & 1)
		{
			const uint32_t above = inv ? ~inv : 0xFFFFFFFFu;
			const float v = unordered_bits(cand_right < above ? cand_right : above);
			if (a1 == 0) lb[0] = v; else if (a1 == 1) lb[1] = v; else lb[2] = v;
		}
		else
		{
			const float v = unordered_bits(pivot);
			if (a1 == 0) rb[0] = v; else if (a1 == 1) rb[1] = v; else rb[2] = v;
		}
		lbound[3 * child] = lb[0]; lbound[3 * child + 1] = lb[1]; lbound[3 * child + 2] = lb[2];
		rbound[3 * child] = rb[0]; rbound[3 * child + 1] = rb[1]; rbound[3 * child + 2] = rb[2];
		splitdim[child] = longest_axis(rb[0] - lb[0], rb[1] - lb[1], rb[2] - lb[2]);
		index[child] = (int)range_start(n, c, 2LL * m);
	}
}
#pragma clang fp contract(on)

// Unordered partition of every node into [keys below the pivot | keys above the pivot]; elements equal to the
// pivot go left when all of them belong there and to the tie list otherwise.  All loads of a thread's 8 elements
// are issued up front, slots are reserved with one packed block scan and four concurrent global atomics.
// WARM: the select was one windowed pass (sel_hist_warm_kernel), buckets are key >> drop; otherwise drop = 10 (two passes of 11 bits)
//
// A workgroup is a chain of dependent round trips through L2 (~1.5 us each; the 40 MB it moves per level hide beside them), so
// the kernel is laid out to have few: (1) the elements, the node's description (kMeta*) and its window histogram leave
// together; (2) the cursors of the four output streams AND the slots of the workgroup's candidates are reserved by one batch of
// atomics; (3) the stores drain; (4) the completion counter; (5) the completing wave fetches counters and candidate records in
// one batch.  (Round 2 started with eleven: axis -> box -> old pivot -> histogram -> ancestors' axes, one atomic per
// candidate, and three rounds of loads in the tail.)
template <int BLOCK, int NP, bool WARM = false, int EPT = 8>
__global__ __launch_bounds__(BLOCK) void sel_partition_kernel(const float4 *__restrict__ pos_in, const int *__restrict__ unsort_in,
                                                               float4 *__restrict__ pos_out, int *__restrict__ unsort_out,
                                                               const int *__restrict__ sd_l, SelNode *__restrict__ nodes,
                                                               uint32_t *__restrict__ tielist, long long n, int l, float *__restrict__ lbound,
                                                               float *__restrict__ rbound, int *__restrict__ splitdim, int *__restrict__ index,
                                                               int *__restrict__ flag, const uint32_t *__restrict__ hist, int drop,
                                                               const int *__restrict__ top_sd, int top_root1)
{
	constexpr int CHUNK = EPT * BLOCK;
	static_assert(CHUNK < 65536, "packed 16-bit block counters");
	static_assert(EPT == 8 || WARM, "the cold select's passes agree on 8 elements per thread");
	static_assert(BLOCK >= 128, "two nodes' descriptions are fetched by 128 threads");
	__shared__ uint32_t base_s[6], mR[2], ntie_s[2];   // base_s[4 + jj]: first slot of the workgroup's candidates in node jj's list
	__shared__ uint32_t meta[2][kMetaWords];
	const long long m = 1LL << l;
	const long long i0 = (long long)blockIdx.x * CHUNK;
	// element loads first: they are independent of the node state resolved below
	constexpr int PER = CHUNK / BLOCK;
	float4 p[PER];
	int org[PER];
	PART_MARK(0);
#pragma unroll
	for (int e = 0; e < PER; ++e)
	{
		const long long i = i0 + e * BLOCK + threadIdx.x;
		if (i < n) { p[e] = pos_in[i]; org[e] = unsort_in[i]; }
	}
	const long long ilast = (i0 + CHUNK < n ? i0 + CHUNK : n) - 1;
	const long long j0 = div_floor_small(m * i0, n), j1 = div_floor_small(m * ilast, n);
	PART_MARK(1);
	// the description of the chunk's one or two nodes (see kMeta*), and the first node's window histogram, in the same round trip
	if (threadIdx.x < 128)
	{
		const int jj = threadIdx.x >> 6, t = threadIdx.x & 63;
		const long long j = j0 + jj;
		uint32_t v = 0;
		if (j <= j1)
		{
			const int node = (int)(m - 1 + j), left = 2 * node + 1;
			if (t < 32)
			{
				// ancestors inside this tree, then -- the local build of a kd-domain -- the domain root's ancestors in the global
				// tree: the stable-sort chain's keys reach back to the global root (3 = none)
				v = 3u;
				if (t < l) v = (uint32_t)splitdim[(((int)(m + j)) >> (t + 1)) - 1];
				else if (top_sd) { const int up = top_root1 >> (t - l + 1); if (up > 0) v = (uint32_t)top_sd[up - 1]; }
			}
			else if (t == kMetaSd) v = (uint32_t)sd_l[j];
			else if (t < kMetaRb) v = __float_as_uint(lbound[3 * node + (t - kMetaLb)]);
			else if (t < kMetaOld) v = __float_as_uint(rbound[3 * node + (t - kMetaRb)]);
			// (the children's old boxes are still in place: they are rewritten by the workgroup that completes the node, and no
			// workgroup of the node completes before this one has)
			else if (t < kMetaBelow) v = WARM ? __float_as_uint(rbound[3 * left + (t - kMetaOld)]) : 0u;
			else if (t == kMetaBelow) v = WARM ? ld_agent_u32(&nodes[j].below) : 0u;
		}
		meta[jj][t] = v;
	}
	uint32_t wbins[kWarmBins / BLOCK];
	if (WARM) warm_bins<BLOCK>(hist, m, j0, wbins);
	if (threadIdx.x < 2) { mR[threadIdx.x] = 0xFFFFFFFFu; ntie_s[threadIdx.x] = 0; }
	// first element of node j1 (only meaningful when the chunk straddles two nodes)
	const long long split = j1 > j0 ? range_start(n, j1, m) : n;
	__syncthreads();
	PART_MARK(2);
	// the next two distinct split axes of a node's ancestors (keys two and three of the stable-sort chain, for the candidates' records)
	__shared__ signed char anc23[2][2];
	if (threadIdx.x < 2 && j0 + threadIdx.x <= j1)
	{
		const int a1 = (int)meta[threadIdx.x][kMetaSd];
		int a2 = -1, a3 = -1;
		for (int t = 0; t < 32; ++t)
		{
			const int a = (int)meta[threadIdx.x][t];
			if (a > 2) break;   // (above the global root)
			if (a == a1 || a == a2) continue;
			if (a2 < 0) a2 = a;
			else { a3 = a; break; }
		}
		anc23[threadIdx.x][0] = (signed char)a2; anc23[threadIdx.x][1] = (signed char)a3;
	}
	// the third pass's histogram is descended here (see resolve_before): pivot back to the un-normalised ordered key
	__shared__ uint32_t sh[BLOCK / 64 + 4];
	// NP = 2: the select stopped after two passes -- piv is the 22-bit bucket of the pivot under the box-linear bucket key,
	// every element of that bucket is a candidate for the resolver; NP = 3: piv is the pivot itself (an ordered key)
	constexpr bool EARLY = NP < 3;
	const int kDrop = drop;   // key bits below the bucket
	uint32_t piv[2] = {0, 0}, all_left[2] = {0, 0};
	float lo[2] = {0.f, 0.f}, scale[2] = {0.f, 0.f};
	bool miss[2] = {false, false};   // warm select: the window did not hold the median -- the node is left alone, flag 2
	SelPivot pvs[2];
	for (int jj = 0; jj < 2; ++jj)
	{
		pvs[jj] = SelPivot{0, 0, 0, 0};
		if (j0 + jj <= j1)
		{
			const int a = (int)meta[jj][kMetaSd];
			const float blo = __uint_as_float(meta[jj][kMetaLb + a]), bhi = __uint_as_float(meta[jj][kMetaRb + a]);
			if (EARLY)
			{
				// lin_window
				lo[jj] = blo;
				const float span = __fsub_rn(bhi, blo);
				scale[jj] = span > 0.f ? __fdiv_rn(4294967040.f, span) : 0.f;
			}
			if (WARM)
			{
				// warm_window_start: the bucket of the previous pivot under the current box
				const uint32_t c = lin_key(__uint_as_float(meta[jj][kMetaOld + a]), lo[jj], scale[jj]) >> drop;
				const uint32_t w0 = c > (uint32_t)(kWarmBins / 2) ? c - (uint32_t)(kWarmBins / 2) : 0u;
				if (jj == 1) warm_bins<BLOCK>(hist, m, j0 + 1, wbins);   // a straddling chunk: one more round trip
				pvs[jj] = resolve_warm<BLOCK>(wbins, meta[jj][kMetaBelow], n, l, j0 + jj, w0, sh);
				miss[jj] = pvs[jj].need == 0;
				if (miss[jj] && threadIdx.x == 0) *flag = 2;
				piv[jj] = pvs[jj].prefix;
				continue;
			}
			pvs[jj] = resolve_before<NP, BLOCK>(hist, nodes, n, l, j0 + jj, i0, sh);
			if (!EARLY)
			{
				// all passes done: back from the node's key window (key_window) to the pivot as an ordered key
				const uint32_t kmin = ordered_bits(blo), span = ordered_bits(bhi) - kmin;
				pvs[jj].prefix = (pvs[jj].prefix >> (span ? __clz(span) : 0)) + kmin;
			}
			piv[jj] = pvs[jj].prefix;
			all_left[jj] = !EARLY && pvs[jj].need == pvs[jj].neq;
		}
	}
	if (j1 == j0) { piv[1] = piv[0]; all_left[1] = all_left[0]; lo[1] = lo[0]; scale[1] = scale[0]; miss[1] = miss[0]; }
	const int sd[2] = {(int)meta[0][kMetaSd], j1 > j0 ? (int)meta[1][kMetaSd] : (int)meta[0][kMetaSd]};
	__syncthreads();
	PART_MARK(3);
	// Destination slots.  The four output streams of the chunk (left / right of its one or two nodes) are filled in
	// element order: consecutive lanes with the same destination stream write consecutive slots, so the stores coalesce
	// (per-thread cursors made every lane of a store instruction hit a different sector: twice the HBM write traffic).
	// Per round e (one element per thread) and wave: four ballots give the lane's rank inside its stream and the wave's
	// counts; an exclusive scan over the (round, wave) grid turns the counts into offsets.
	constexpr int W = BLOCK / 64;
	__shared__ uint64_t pw[PER * W + 1];   // four 16-bit counters per (round, wave); [PER * W] = block totals
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint64_t below = (1ull << lane) - 1ull;
	int cat[PER];          // -1 tie / nothing, else 2 * (second node ?) + (right ?)
	int tslot[PER];        // candidates: place among the workgroup's candidates of the node, else -1
	uint32_t lp[PER];      // rank among the wave's elements of the same stream in this round
	uint32_t tmin[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
	for (int e = 0; e < PER; ++e)
	{
		const long long i = i0 + e * BLOCK + threadIdx.x;
		cat[e] = -1;
		tslot[e] = -1;
		if (i < n)
		{
			const int jj = i >= split ? 1 : 0;
			const uint32_t key = ordered_bits(axis_of(p[e], sd[jj]));
			const uint32_t cmp = EARLY ? lin_key(axis_of(p[e], sd[jj]), lo[jj], scale[jj]) >> kDrop : key;
			if (WARM && miss[jj]) { /* nothing moves */ }
			else if (cmp < piv[jj] || (cmp == piv[jj] && all_left[jj])) cat[e] = 2 * jj;
			else if (cmp > piv[jj]) { cat[e] = 2 * jj + 1; tmin[jj] = key < tmin[jj] ? key : tmin[jj]; }
			else tslot[e] = (int)atomicAdd(&ntie_s[jj], 1u);   // a candidate: its record is written once the node's list has room for it
		}
		uint64_t packed = 0;
		lp[e] = 0;
#pragma unroll
		for (int q = 0; q < 4; ++q)
		{
			if (q >= 2 && j1 == j0) break;   // one node: two streams
			const uint64_t bq = __ballot(cat[e] == q);
			packed |= (uint64_t)__popcll(bq) << (16 * q);
			if (cat[e] == q) lp[e] = (uint32_t)__popcll(bq & below);
		}
		if (lane == 0) pw[e * W + wv] = packed;
	}
#pragma unroll
	for (int q = 0; q < 2; ++q)
	{
		const uint32_t v = wave_min_u32(tmin[q]);
		if (lane == 0 && v != 0xFFFFFFFFu) atomicMin(&mR[q], v);
	}
	PART_MARK(10);
	__syncthreads();
	PART_MARK(4);
	if (threadIdx.x < 64)
	{
		// exclusive scan of the PER * W packed counters (two consecutive entries per lane; the 16-bit fields cannot carry: a
		// workgroup holds fewer than 2^16 elements)
		constexpr int NE = PER * W;
		const int k0 = 2 * lane, k1 = 2 * lane + 1;
		const uint64_t v0 = k0 < NE ? pw[k0] : 0, v1 = k1 < NE ? pw[k1] : 0;
		const uint64_t both = v0 + v1;
		const uint64_t incl = (uint64_t)wave_scan_add((uint32_t)both) | ((uint64_t)wave_scan_add((uint32_t)(both >> 32)) << 32);
		const uint64_t excl = incl - both;
		if (k0 < NE) pw[k0] = excl;
		if (k1 < NE) pw[k1] = excl + v0;
		// the one batch of atomics: the cursors of the four streams, the places of the candidates, the smallest key that went right
		const uint64_t tot = __shfl(incl, 63);
		if (lane < 4)
		{
			const int jj = lane >> 1, right = lane & 1;
			const uint32_t c = (uint32_t)((tot >> (16 * lane)) & 0xFFFF);
			base_s[lane] = c ? atomicAdd(right ? &nodes[j0 + jj].cntR : &nodes[j0 + jj].cntL, c) : 0;
		}
		else if (lane < 6)
		{
			const uint32_t c = ntie_s[lane - 4];
			base_s[lane] = c ? atomicAdd(&nodes[j0 + (lane - 4)].tiecnt, c) : 0;
		}
		else if (lane < 8)
		{
			if (mR[lane - 6] != 0xFFFFFFFFu) atomicMax(&nodes[j0 + (lane - 6)].minR, ~mR[lane - 6]);
		}
	}
	__syncthreads();
	PART_MARK(5);
	// first slot of each stream: [left0, right0, left1, right1]
	long long first[4];
#pragma unroll
	for (int q = 0; q < 4; ++q)
	{
		const long long j = j0 + (q >> 1);
		first[q] = ((q & 1) ? range_start(n, 2 * j + 1, 2 * m) : range_start(n, j, m)) + base_s[q];
	}
#pragma unroll
	for (int e = 0; e < PER; ++e)
	{
		if (tslot[e] >= 0)
		{
			const int jj = (i0 + e * BLOCK + threadIdx.x) >= split ? 1 : 0;
			const uint32_t t = base_s[4 + jj] + (uint32_t)tslot[e];
			// read back inside this launch by the workgroup that completes the node: stores past the (non-coherent) L2
			if (t < kTieCap)
			{
				uint32_t *rec = tielist + ((size_t)(j0 + jj) * kTieCap + t) * kTieWords;
				const int a2 = anc23[jj][0], a3 = anc23[jj][1];
				const uint32_t w[7] = {ordered_bits(axis_of(p[e], sd[jj])), a2 >= 0 ? ordered_bits(axis_of(p[e], a2)) : 0u,
				                       a3 >= 0 ? ordered_bits(axis_of(p[e], a3)) : 0u, (uint32_t)org[e],
				                       __float_as_uint(p[e].x), __float_as_uint(p[e].y), __float_as_uint(p[e].z)};
#pragma unroll
				for (int q = 0; q < 7; ++q) __hip_atomic_store(&rec[q], w[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		if (cat[e] < 0) continue;
		const int q = cat[e];
		const long long dst = (q == 0 ? first[0] : (q == 1 ? first[1] : (q == 2 ? first[2] : first[3])))
		                      + (long long)((pw[e * W + wv] >> (16 * q)) & 0xFFFF) + lp[e];
		pos_out[dst] = p[e];
		unsort_out[dst] = org[e];
	}
	// The workgroup that completes a node resolves its pivot ties and writes the children's boxes (last-block-done: the
	// cursors, tie list and minimum are only touched by device atomics and read back with agent-scope loads).
	__shared__ uint32_t fin[2];
	PART_MARK(6);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	PART_MARK(7);
	if (threadIdx.x < 2) fin[threadIdx.x] = (j0 + threadIdx.x <= j1) ? atomicAdd(&nodes[j0 + threadIdx.x].done_part, 1u) : 0xFFFFFFFFu;
	__syncthreads();
	PART_MARK(8);
	if (threadIdx.x < 64)
		for (int jj = 0; jj < 2; ++jj)
			if (j0 + jj <= j1 && fin[jj] == chunks_of_node<CHUNK>(n, j0 + jj, m) - 1)
				ties_and_boxes(pos_out, unsort_out, lbound, rbound, splitdim, index, nodes, tielist, flag, n, l, (int)(j0 + jj),
				               (int)threadIdx.x, pvs[jj], meta[jj]);
	PART_MARK(9);
}


} // namespace

// Storage of the pass histograms and node counters of levels 0 .. l0-1.  They have to be zero before the first level:
// zero = true does it with two memsets, zero = false leaves it to the caller (the build prologue kernel clears the
// two ranges returned in words_a / words_b while it packs the positions).
int kd_select_begin(nbco_ctx *c, int l0, bool zero, long long *words_a, long long *words_b)
{
	const size_t nodes = ((size_t)1 << l0) - 1;
	NBCO_TRY(c->reserve(c->sel_hist, sizeof(uint32_t) * 3 * nodes * kBins));
	NBCO_TRY(c->reserve(c->sel_nodes, sizeof(SelNode) * nodes));
	NBCO_TRY(c->reserve(c->sel_ties, sizeof(uint32_t) * nodes * kTieCap * kTieWords));
	if (words_a) *words_a = (long long)(3 * nodes * kBins);
	if (words_b) *words_b = (long long)(sizeof(SelNode) / sizeof(uint32_t) * nodes);
	if (zero)
	{
		NBCO_HIP(hipMemsetAsync(c->sel_hist.ptr, 0, sizeof(uint32_t) * 3 * nodes * kBins, c->stream));
		NBCO_HIP(hipMemsetAsync(c->sel_nodes.ptr, 0, sizeof(SelNode) * nodes, c->stream));
	}
	return NBCO_OK;
}

// Split every node of level l (all of which hold more than 4096 particles) and write the boxes of level
// l + 1.  `flag` (device int) is set when a node had more ties than the resolver handles.
template <int BLOCK, int NP>
static void select_level_launch(nbco_ctx *c, int l, long long n, const float4 *pos_in, const int *unsort_in, float4 *pos_out, int *unsort_out,
                                float *lbound, float *rbound, int *splitdim, int *index, int *flag, int warm_drop)
{
	const int m = 1 << l;
	// level l uses the slices [m - 1, 2m - 1) of the per-build arrays
	SelNode *nodes = c->sel_nodes.as<SelNode>() + (m - 1);
	uint32_t *hist = c->sel_hist.as<uint32_t>() + (size_t)3 * (m - 1) * kBins;
	uint32_t *ties = c->sel_ties.as<uint32_t>() + (size_t)(m - 1) * kTieCap * kTieWords;
	const int *sd_l = splitdim + (m - 1);
	hipStream_t st = c->stream;
	constexpr int CHUNK = 8 * BLOCK;
	const int gchunks = (int)((n + CHUNK - 1) / CHUNK);
	if (NP == 2 && warm_drop > 0)
	{
		// The warm kernels take FOUR elements per thread (twice the workgroups of the cold passes: 256 of them at N = 1M).  A
		// workgroup's time is its threads' serial work -- classify, ballots, slot arithmetic per element -- not the node's
		// atomics: 8 elements per thread took 18.8 us per partition, 4 take 16; 2 are worse again (0.86 ms per step against
		// 0.80), and 512-thread workgroups with 8 elements each gain nothing.
		constexpr int EPT = BLOCK == kBlockBig ? 4 : 8;
		const int wchunks = (int)((n + EPT * BLOCK - 1) / (EPT * BLOCK));
		hipLaunchKernelGGL((sel_hist_warm_kernel<BLOCK, EPT>), dim3(wchunks), dim3(BLOCK), 0, st, pos_in, sd_l, hist, nodes, (const float *)lbound, (const float *)rbound,
		                   n, l, warm_drop);
		hipLaunchKernelGGL((sel_partition_kernel<BLOCK, 2, true, EPT>), dim3(wchunks), dim3(BLOCK), 0, st, pos_in, unsort_in, pos_out, unsort_out, sd_l, nodes, ties, n, l,
		                   lbound, rbound, splitdim, index, flag, (const uint32_t *)hist, warm_drop, c->top_sd, c->top_root1);
		return;
	}
	hipLaunchKernelGGL((sel_hist_kernel<0, BLOCK, NP == 2>), dim3(gchunks), dim3(BLOCK), 0, st, pos_in, sd_l, hist, nodes, (const float *)lbound, (const float *)rbound, n, l);
	hipLaunchKernelGGL((sel_hist_kernel<1, BLOCK, NP == 2>), dim3(gchunks), dim3(BLOCK), 0, st, pos_in, sd_l, hist, nodes, (const float *)lbound, (const float *)rbound, n, l);
	if (NP >= 3)
		hipLaunchKernelGGL((sel_hist_kernel<2, BLOCK, false>), dim3(gchunks), dim3(BLOCK), 0, st, pos_in, sd_l, hist, nodes, (const float *)lbound, (const float *)rbound, n, l);
	hipLaunchKernelGGL((sel_partition_kernel<BLOCK, NP>), dim3(gchunks), dim3(BLOCK), 0, st, pos_in, unsort_in, pos_out, unsort_out, sd_l, nodes, ties, n, l,
	                   lbound, rbound, splitdim, index, flag, (const uint32_t *)hist, 10, c->top_sd, c->top_root1);
}

int kd_select_level(nbco_ctx *c, int l, long long n, const float4 *pos_in, const int *unsort_in, float4 *pos_out, int *unsort_out,
                    float *lbound, float *rbound, int *splitdim, int *index, int *flag, bool warm)
{
	// Two radix passes over a bucket key that is linear across the node's box (lin_key) leave the pivot's bucket with about
	// node size / 2^22 elements times the density contrast inside the box; the tie resolver orders them exactly as long as
	// they are at most kTieCap.  Nodes of up to 2^22 particles stop after two passes (a handful of candidates), larger ones
	// run all three passes over the ordered float bits.  A fuller bucket (tightly clustered input) raises the tie flag and the
	// caller falls back to three passes everywhere (sel_three_pass), then to the sorting build.
	const long long node = n >> l;
	const int np = (c->sel_three_pass || node > (1LL << 22)) ? 3 : 2;
	const bool big = node >= 8 * kBlockBig;
	// warm select (one pass): buckets of key >> drop with node / 2^(32 - drop) ~ 1/64 element per bucket on average, at most 22 bits
	int warm_drop = 0;
	if (warm && np == 2)
	{
		int lg = 0;
		while ((1LL << lg) < node) ++lg;
		// (after misses: 2 bits coarser per miss, i.e. a window 4x as wide, down to 12 bits -- the resolver takes 64 candidates)
		warm_drop = 32 - std::max(12, std::min(22, std::max(16, lg + 6)) - 2 * c->sel_warm_coarsen);
	}
#define NBCO_SEL_ARGS c, l, n, pos_in, unsort_in, pos_out, unsort_out, lbound, rbound, splitdim, index, flag, warm_drop
	if (big)
	{
		if (np == 2) select_level_launch<kBlockBig, 2>(NBCO_SEL_ARGS);
		else select_level_launch<kBlockBig, 3>(NBCO_SEL_ARGS);
	}
	else
	{
		if (np == 2) select_level_launch<kBlockSmall, 2>(NBCO_SEL_ARGS);
		else select_level_launch<kBlockSmall, 3>(NBCO_SEL_ARGS);
	}
#undef NBCO_SEL_ARGS
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}
