// k_kdselect.hip -- top levels of the kd-tree build by exact median SELECTION instead of sorting.
//
// The reference sorts every node's particles along its split axis at every level (stable sort chain,
// fmm_cart3_kdtree.cuh:1858-1871) and cuts at the median.  The order that chain produces inside node v
// is the lexicographic order on (c[a1], c[a2], c[a3], original index) with a1 = splitdim(v) and a2, a3
// the next distinct split axes of v's ancestors (most recent first); the children are the first k
// and the remaining elements of that order.  Only the children's particle SETS and the two boundary
// coordinates enter the tree, so while a node is too large for one workgroup's LDS the level is
// built by
//   1. a three-pass (11 + 11 + 10 bit) radix select of the k-th smallest ordered key c[a1] per node,
//   2. an unordered partition into < pivot | > pivot with per-node atomics (block-aggregated),
//   3. an exact resolution of the elements that tie with the pivot (a handful) by (c[a2], c[a3], index),
//   4. evalBox for the children from the pivot value and the smallest key of the right part.
// The canonical order is re-established in LDS when the subtree kernel takes over (k_fmm_kd.hip).
// If more than kTieCap elements tie with a pivot (degenerate inputs) a flag is raised and the caller
// rebuilds with the sorting path.
#include "nbco_internal.hpp"
#include "kd_common.hpp"

namespace {

using namespace kdc;

constexpr int kBlock = 256;
constexpr int kChunk = 2048;            // elements per block; nodes handled here hold > 4096 particles,
                                        // so a chunk touches at most two nodes
constexpr int kBins = 2048;
constexpr int kTieCap = 64;

struct SelNode
{
	uint32_t prefix;   // selected digits so far (after the last pass: the pivot key)
	uint32_t r;        // 0-based rank of the pivot inside the current candidate set
	uint32_t nless;    // elements with key < candidate prefix
	uint32_t neq;      // elements equal to the pivot key
	uint32_t need;     // how many of them belong to the left child
	uint32_t cntL, cntR, tiecnt;
	uint32_t minR;     // smallest ordered key of the right child
	uint32_t pad[7];
};

__global__ __launch_bounds__(kBlock) void sel_init_kernel(SelNode *__restrict__ nodes, uint32_t *__restrict__ hist, long long n, int l)
{
	const int m = 1 << l;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < 3LL * m * kBins; i += (long long)gridDim.x * kBlock) hist[i] = 0;
	const int j = blockIdx.x * kBlock + threadIdx.x;
	if (j < m)
	{
		const long long start = range_start(n, j, m), mid = range_start(n, 2 * j + 1, 2 * m);
		SelNode s{};
		s.r = (uint32_t)(mid - start - 1);
		s.minR = 0xFFFFFFFFu;
		nodes[j] = s;
	}
}

template <int PASS>
__global__ __launch_bounds__(kBlock) void sel_hist_kernel(const float4 *__restrict__ pos, const int *__restrict__ sd_l, const SelNode *__restrict__ nodes,
                                                          uint32_t *__restrict__ hist, long long n, int l)
{
	__shared__ uint32_t h[2][kBins];
	const long long m = 1LL << l;
	for (int t = threadIdx.x; t < 2 * kBins; t += kBlock) (&h[0][0])[t] = 0;
	__syncthreads();
	const long long i0 = (long long)blockIdx.x * kChunk;
	const long long j0 = (m * i0) / n;
	for (int e = 0; e < kChunk / kBlock; ++e)
	{
		const long long i = i0 + e * kBlock + threadIdx.x;
		if (i < n)
		{
			const long long j = (m * i) / n;
			const uint32_t key = ordered_bits(axis_of(pos[i], sd_l[j]));
			bool ok = true;
			uint32_t d = key >> 21;
			if (PASS == 1) { ok = (key >> 21) == nodes[j].prefix; d = (key >> 10) & 0x7FFu; }
			if (PASS == 2) { ok = (key >> 10) == nodes[j].prefix; d = key & 0x3FFu; }
			if (ok) atomicAdd(&h[j - j0][d], 1u);
		}
	}
	__syncthreads();
	for (int t = threadIdx.x; t < 2 * kBins; t += kBlock)
	{
		const uint32_t v = (&h[0][0])[t];
		const long long j = j0 + (t / kBins);
		if (v && j < m) atomicAdd(&hist[((size_t)PASS * m + j) * kBins + (t % kBins)], v);
	}
}

// one block per node: locate the bin that holds rank r, descend into it
template <int PASS>
__global__ __launch_bounds__(kBlock) void sel_scan_kernel(SelNode *__restrict__ nodes, const uint32_t *__restrict__ hist, int l)
{
	__shared__ uint32_t wsum[kBlock / 64];
	__shared__ uint32_t found[3];
	const int m = 1 << l, j = blockIdx.x;
	const uint32_t *h = hist + ((size_t)PASS * m + j) * kBins;
	constexpr int PER = kBins / kBlock;
	uint32_t v[PER], s = 0;
#pragma unroll
	for (int q = 0; q < PER; ++q) { v[q] = h[threadIdx.x * PER + q]; s += v[q]; }
	// exclusive scan of the per-thread sums
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	uint32_t incl = s;
	for (int o = 1; o < 64; o <<= 1) { uint32_t y = __shfl_up(incl, o); if (lane >= o) incl += y; }
	if (lane == 63) wsum[w] = incl;
	__syncthreads();
	uint32_t base = 0;
	for (int q = 0; q < w; ++q) base += wsum[q];
	uint32_t cum = base + incl - s;
	const uint32_t r = nodes[j].r;
#pragma unroll
	for (int q = 0; q < PER; ++q)
	{
		if (r >= cum && r < cum + v[q]) { found[0] = threadIdx.x * PER + q; found[1] = cum; found[2] = v[q]; }
		cum += v[q];
	}
	__syncthreads();
	if (threadIdx.x == 0)
	{
		SelNode nd = nodes[j];
		const int bits = PASS == 2 ? 10 : 11;
		nd.prefix = (nd.prefix << bits) | found[0];
		nd.nless += found[1];
		nd.r -= found[1];
		if (PASS == 2) { nd.neq = found[2]; nd.need = nd.r + 1; }
		nodes[j] = nd;
	}
}

__global__ __launch_bounds__(kBlock) void sel_partition_kernel(const float4 *__restrict__ pos_in, const int *__restrict__ unsort_in,
                                                               float4 *__restrict__ pos_out, int *__restrict__ unsort_out,
                                                               const int *__restrict__ sd_l, SelNode *__restrict__ nodes,
                                                               uint32_t *__restrict__ tielist, long long n, int l)
{
	__shared__ uint32_t cL[2], cR[2], bL[2], bR[2], mR[2];
	const long long m = 1LL << l;
	if (threadIdx.x < 2) { cL[threadIdx.x] = 0; cR[threadIdx.x] = 0; mR[threadIdx.x] = 0xFFFFFFFFu; }
	__syncthreads();
	const long long i0 = (long long)blockIdx.x * kChunk;
	const long long j0 = (m * i0) / n;
	constexpr int PER = kChunk / kBlock;
	int cls[PER];          // 0 none, 1 left, 2 right
	uint32_t rank[PER];
	for (int e = 0; e < PER; ++e)
	{
		const long long i = i0 + e * kBlock + threadIdx.x;
		cls[e] = 0;
		rank[e] = 0;
		if (i < n)
		{
			const long long j = (m * i) / n;
			const int jj = (int)(j - j0);
			const uint32_t key = ordered_bits(axis_of(pos_in[i], sd_l[j]));
			const uint32_t piv = nodes[j].prefix;
			if (key < piv || (key == piv && nodes[j].need == nodes[j].neq)) { cls[e] = 1; rank[e] = atomicAdd(&cL[jj], 1u); }
			else if (key > piv) { cls[e] = 2; rank[e] = atomicAdd(&cR[jj], 1u); atomicMin(&mR[jj], key); }
			else
			{
				const uint32_t t = atomicAdd(&nodes[j].tiecnt, 1u);
				if (t < kTieCap) tielist[(size_t)j * kTieCap + t] = (uint32_t)i;
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < 2)
	{
		const long long j = j0 + threadIdx.x;
		if (j < m)
		{
			bL[threadIdx.x] = cL[threadIdx.x] ? atomicAdd(&nodes[j].cntL, cL[threadIdx.x]) : 0;
			bR[threadIdx.x] = cR[threadIdx.x] ? atomicAdd(&nodes[j].cntR, cR[threadIdx.x]) : 0;
			if (mR[threadIdx.x] != 0xFFFFFFFFu) atomicMin(&nodes[j].minR, mR[threadIdx.x]);
		}
	}
	__syncthreads();
	for (int e = 0; e < PER; ++e)
	{
		if (!cls[e]) continue;
		const long long i = i0 + e * kBlock + threadIdx.x;
		const long long j = (m * i) / n;
		const int jj = (int)(j - j0);
		const long long dst = cls[e] == 1 ? range_start(n, j, m) + bL[jj] + rank[e] : range_start(n, 2 * j + 1, 2 * m) + bR[jj] + rank[e];
		pos_out[dst] = pos_in[i];
		unsort_out[dst] = unsort_in[i];
	}
}

// one wave per node: order the elements that tie with the pivot by the remaining keys of the stable-sort
// chain -- the next distinct ancestor split axes, then the original index -- and hand the first `need`
// of them to the left child
__global__ __launch_bounds__(64) void sel_ties_kernel(const float4 *__restrict__ pos_in, const int *__restrict__ unsort_in,
                                                      float4 *__restrict__ pos_out, int *__restrict__ unsort_out, const int *__restrict__ splitdim,
                                                      SelNode *__restrict__ nodes, const uint32_t *__restrict__ tielist, int *__restrict__ flag,
                                                      long long n, int l)
{
	const int m = 1 << l, j = blockIdx.x, lane = threadIdx.x;
	const uint32_t nt = nodes[j].tiecnt;
	if (nt == 0) return;
	if (nt > kTieCap) { if (lane == 0) *flag = 1; return; }
	// next two distinct axes above this node
	const int node = m - 1 + j, a1 = splitdim[node];
	int a2 = -1, a3 = -1;
	for (int anc = node; anc > 0;)
	{
		anc = (anc - 1) >> 1;
		const int a = splitdim[anc];
		if (a == a1 || a == a2) continue;
		if (a2 < 0) a2 = a;
		else { a3 = a; break; }
	}
	uint32_t idx = 0, k2 = 0, k3 = 0, org = 0;
	if ((uint32_t)lane < nt)
	{
		idx = tielist[(size_t)j * kTieCap + lane];
		const float4 p = pos_in[idx];
		k2 = a2 >= 0 ? ordered_bits(axis_of(p, a2)) : 0;
		k3 = a3 >= 0 ? ordered_bits(axis_of(p, a3)) : 0;
		org = (uint32_t)unsort_in[idx];
	}
	uint32_t rank = 0;
	for (uint32_t q = 0; q < nt; ++q)
	{
		const uint32_t q2 = __shfl(k2, q), q3 = __shfl(k3, q), qo = __shfl(org, q);
		const bool before = q2 < k2 || (q2 == k2 && (q3 < k3 || (q3 == k3 && qo < org)));
		rank += before ? 1u : 0u;
	}
	if ((uint32_t)lane < nt)
	{
		const uint32_t need = nodes[j].need;
		long long dst;
		if (rank < need) dst = range_start(n, j, m) + atomicAdd(&nodes[j].cntL, 1u);
		else
		{
			dst = range_start(n, 2 * j + 1, 2LL * m) + atomicAdd(&nodes[j].cntR, 1u);
			atomicMin(&nodes[j].minR, nodes[j].prefix);
		}
		pos_out[dst] = pos_in[idx];
		unsort_out[dst] = unsort_in[idx];
	}
}

#pragma clang fp contract(off)
// evalBox for the children of level l (fmm_cart3_kdtree.cuh:109-137): the sorted order's boundary
// elements are the pivot (largest key of the left child) and the smallest key of the right child
__global__ __launch_bounds__(kBlock) void sel_box_kernel(float *__restrict__ lbound, float *__restrict__ rbound, int *__restrict__ splitdim,
                                                         int *__restrict__ index, const SelNode *__restrict__ nodes, long long n, int l)
{
	const int m = 1 << l;
	const int c = blockIdx.x * kBlock + threadIdx.x;
	if (c >= 2 * m) return;
	const int j = c >> 1, parent = m - 1 + j, node = 2 * m - 1 + c, split = splitdim[parent];
	float lb[3] = {lbound[3 * parent], lbound[3 * parent + 1], lbound[3 * parent + 2]};
	float rb[3] = {rbound[3 * parent], rbound[3 * parent + 1], rbound[3 * parent + 2]};
	if (c & 1)
	{
		const float v = unordered_bits(nodes[j].minR);
		if (split == 0) lb[0] = v; else if (split == 1) lb[1] = v; else lb[2] = v;
	}
	else
	{
		const float v = unordered_bits(nodes[j].prefix);
		if (split == 0) rb[0] = v; else if (split == 1) rb[1] = v; else rb[2] = v;
	}
	lbound[3 * node] = lb[0]; lbound[3 * node + 1] = lb[1]; lbound[3 * node + 2] = lb[2];
	rbound[3 * node] = rb[0]; rbound[3 * node + 1] = rb[1]; rbound[3 * node + 2] = rb[2];
	splitdim[node] = longest_axis(rb[0] - lb[0], rb[1] - lb[1], rb[2] - lb[2]);
	index[node] = (int)range_start(n, c, 2LL * m);
}
#pragma clang fp contract(on)

} // namespace

// Split every node of level l (all of which hold more than 4096 particles) and write the boxes of level
// l + 1.  `flag` (device int) is set when a node had more ties than the resolver handles.
int kd_select_level(nbco_ctx *c, int l, long long n, const float4 *pos_in, const int *unsort_in, float4 *pos_out, int *unsort_out,
                    float *lbound, float *rbound, int *splitdim, int *index, int *flag)
{
	const int m = 1 << l;
	const size_t hist_bytes = sizeof(uint32_t) * 3 * (size_t)m * kBins;
	NBCO_TRY(c->reserve(c->sel_hist, hist_bytes));
	NBCO_TRY(c->reserve(c->sel_nodes, sizeof(SelNode) * (size_t)m));
	NBCO_TRY(c->reserve(c->sel_ties, sizeof(uint32_t) * (size_t)m * kTieCap));
	SelNode *nodes = c->sel_nodes.as<SelNode>();
	uint32_t *hist = c->sel_hist.as<uint32_t>();
	uint32_t *ties = c->sel_ties.as<uint32_t>();
	const int *sd_l = splitdim + (m - 1);
	hipStream_t st = c->stream;
	const int gchunks = (int)((n + kChunk - 1) / kChunk);
	int ginit = (int)((3LL * m * kBins + kBlock - 1) / kBlock);
	if (ginit > 2048) ginit = 2048;
	hipLaunchKernelGGL(sel_init_kernel, dim3(ginit), dim3(kBlock), 0, st, nodes, hist, n, l);
	hipLaunchKernelGGL(sel_hist_kernel<0>, dim3(gchunks), dim3(kBlock), 0, st, pos_in, sd_l, (const SelNode *)nodes, hist, n, l);
	hipLaunchKernelGGL(sel_scan_kernel<0>, dim3(m), dim3(kBlock), 0, st, nodes, (const uint32_t *)hist, l);
	hipLaunchKernelGGL(sel_hist_kernel<1>, dim3(gchunks), dim3(kBlock), 0, st, pos_in, sd_l, (const SelNode *)nodes, hist, n, l);
	hipLaunchKernelGGL(sel_scan_kernel<1>, dim3(m), dim3(kBlock), 0, st, nodes, (const uint32_t *)hist, l);
	hipLaunchKernelGGL(sel_hist_kernel<2>, dim3(gchunks), dim3(kBlock), 0, st, pos_in, sd_l, (const SelNode *)nodes, hist, n, l);
	hipLaunchKernelGGL(sel_scan_kernel<2>, dim3(m), dim3(kBlock), 0, st, nodes, (const uint32_t *)hist, l);
	hipLaunchKernelGGL(sel_partition_kernel, dim3(gchunks), dim3(kBlock), 0, st, pos_in, unsort_in, pos_out, unsort_out, sd_l, nodes, ties, n, l);
	hipLaunchKernelGGL(sel_ties_kernel, dim3(m), dim3(64), 0, st, pos_in, unsort_in, pos_out, unsort_out, (const int *)splitdim, nodes,
	                   (const uint32_t *)ties, flag, n, l);
	hipLaunchKernelGGL(sel_box_kernel, dim3((2 * m + kBlock - 1) / kBlock), dim3(kBlock), 0, st, lbound, rbound, splitdim, index,
	                   (const SelNode *)nodes, n, l);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}
