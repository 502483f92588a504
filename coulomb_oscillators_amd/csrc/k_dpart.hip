// k_dpart.hip -- distributed re-partition of the kd-domains: the top log2(G) median splits of
// the balanced kd-tree (fmm_cart3_kdtree.cuh:109-137, :1858-1871) WITHOUT gathering the state.
//
// nbco_dist_partition all-gathers [pos | vel] of all N particles to every GPU and selects redundantly: O(N_global) memory and
// traffic per GPU.  Here every rank keeps its n_local particles; per level the exact median of every node is found by a
// radix select over the ordered float keys (11 + 11 + 10 bits) whose histograms are summed across ranks, elements that tie
// with the pivot are ordered by the stable-sort chain's remaining keys (next distinct ancestor axes, then the original index
// = rank * n_local + i, the index nbco_dist_partition's gathered order gives them), the children's boxes follow evalBox's
// rule, and the local array is partitioned in place.  After log2(G) levels the particles are grouped by destination rank and
// one all-to-all moves [pos | vel] of exactly the particles that change owner (plus the local copy of those that stay).
// The result -- top boxes, split axes, the particles of every domain AND THEIR ORDER (source rank, then index in the source's
// state: the order of the gathered state) -- equals nbco_dist_partition's.
//
// The library never communicates: nbco_dist_repartition_begin / _next run the local stages and tell the caller which
// collective to run on which part of the caller's workspace before the next call (a small state machine, nbco_dist_step).
#include "nbco_internal.hpp"
#include "kd_common.hpp"
#include <rocprim/rocprim.hpp>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <algorithm>

namespace {

using namespace kdc;

#pragma clang fp contract(off)   // (the only float arithmetic here: differences of box faces for longest_axis)

constexpr int kB = 1024, kBins = 2048, kTie = 64, kGX = 64;

// order-preserving image of an unsigned key in a SIGNED int (the collectives reduce int32)
__device__ inline int skey(uint32_t u) { return (int)(u ^ 0x80000000u); }
__device__ inline uint32_t ukey(int s) { return (uint32_t)s ^ 0x80000000u; }

// select state of one node of the current level (device, identical on every rank)
struct DpNode
{
	uint32_t prefix, r, neq, need;   // digits chosen so far, rank of the pivot among the remaining candidates, size of its bin, how many of them go left
	uint32_t kmin, pivot;            // ordered key of the box's lower face (key window); the pivot as an ordered key (after the third pass)
	int shl, axis, a2, a3;           // key window shift; split axis and the next two distinct ancestor axes (-1: none)
};

struct TopArrays { float *lb, *rb; int *sd, *index; };

__global__ void dp_fill_kernel(int *v, int value, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = value; }

// positions -> float4 {x, y, z, bits(local index)}; local bounds as signed keys: mm[c] = min key, mm[3 + c] = min of the INVERTED
// key (= the maximum), so that one MIN all-reduce serves both
__global__ __launch_bounds__(kB) void dp_pack_kernel(const float *__restrict__ p3, long long n, float4 *__restrict__ P, int *__restrict__ mm)
{
	uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
	for (long long i = (long long)blockIdx.x * kB + threadIdx.x; i < n; i += (long long)gridDim.x * kB)
	{
		const float x = p3[3 * i], y = p3[3 * i + 1], z = p3[3 * i + 2];
		P[i] = make_float4(x, y, z, __int_as_float((int)i));
		const uint32_t k[3] = {ordered_bits(x), ordered_bits(y), ordered_bits(z)};
#pragma unroll
		for (int c = 0; c < 3; ++c) { lo[c] = min(lo[c], k[c]); hi[c] = min(hi[c], ~k[c]); }
	}
#pragma unroll
	for (int c = 0; c < 3; ++c)
	{
		for (int o = 32; o > 0; o >>= 1) { lo[c] = min(lo[c], (uint32_t)__shfl_xor((int)lo[c], o)); hi[c] = min(hi[c], (uint32_t)__shfl_xor((int)hi[c], o)); }
		if ((threadIdx.x & 63) == 0) { atomicMin(&mm[c], skey(lo[c])); atomicMin(&mm[3 + c], skey(hi[c])); }
	}
}

// select state of the nodes of level l from their boxes and split axes (top arrays of levels 0 .. l)
__global__ void dp_nodes_kernel(TopArrays t, int l, long long n_global, DpNode *__restrict__ nd)
{
	const int m = 1 << l, j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const int node = m - 1 + j, a1 = t.sd[node];
	DpNode s;
	s.prefix = 0; s.neq = 0; s.need = 0; s.pivot = 0;
	s.r = (uint32_t)(range_start(n_global, 2 * j + 1, 2LL * m) - range_start(n_global, j, m) - 1);
	s.axis = a1;
	s.kmin = ordered_bits(t.lb[3 * node + a1]);
	const uint32_t span = ordered_bits(t.rb[3 * node + a1]) - s.kmin;
	s.shl = span ? __clz(span) : 0;
	s.a2 = -1; s.a3 = -1;
	for (int anc = node; anc > 0;)
	{
		anc = (anc - 1) >> 1;
		const int a = t.sd[anc];
		if (a == a1 || a == s.a2) continue;
		if (s.a2 < 0) s.a2 = a;
		else { s.a3 = a; break; }
	}
	nd[j] = s;
}

// root box from the reduced bounds (evalRootBox, fmm_cart3_kdtree.cuh:89-107)
__global__ void dp_root_kernel(const int *__restrict__ mm, TopArrays t)
{
	if (threadIdx.x != 0) return;
	float b[6];
	for (int c = 0; c < 3; ++c) { b[c] = unordered_bits(ukey(mm[c])); b[3 + c] = unordered_bits(~ukey(mm[3 + c])); }
	t.lb[0] = b[0]; t.lb[1] = b[1]; t.lb[2] = b[2];
	t.rb[0] = b[3]; t.rb[1] = b[4]; t.rb[2] = b[5];
	t.sd[0] = longest_axis(b[3] - b[0], b[4] - b[1], b[5] - b[2]);
	t.index[0] = 0;
}

// A particle as a whole.  With `axis_of(P[i], axis)` the compiler narrows the 16-byte load to one dword at a SELECTED ADDRESS, and
// the code it emitted for that select (hipcc 7.2, gfx950) leaves the address register unset when axis == 2: a wild load, i.e. a
// GPU memory fault as soon as a node splits along z (found on the first uniform test input).  The empty asm makes all four
// components live, so the load stays a dwordx4 and the select happens on values.
__device__ inline float4 load_particle(const float4 *__restrict__ P, long long i)
{
	float4 v = P[i];
	asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
	return v;
}

__device__ inline uint32_t window_key(const float4 p, const DpNode &n) { return (ordered_bits(axis_of(p, n.axis)) - n.kmin) << n.shl; }

// local histogram of one radix pass; blockIdx.y = node, the node's local particles are P[seg[j] .. seg[j + 1])
__global__ __launch_bounds__(kB) void dp_hist_kernel(const float4 *__restrict__ P, const int *__restrict__ seg, const DpNode *__restrict__ nd, int pass,
                                                     int *__restrict__ hist)
{
	__shared__ uint32_t h[kBins];
	const int j = blockIdx.y, s = seg[j], e = seg[j + 1];
	if (s + (long long)blockIdx.x * kB >= e) return;
	for (int t = threadIdx.x; t < kBins; t += kB) h[t] = 0;
	__syncthreads();
	const DpNode n = nd[j];
	for (long long i = s + (long long)blockIdx.x * kB + threadIdx.x; i < e; i += (long long)gridDim.x * kB)
	{
		const uint32_t key = window_key(load_particle(P, i), n);
		bool ok = true;
		uint32_t d = key >> 21;
		if (pass == 1) { ok = (key >> 21) == n.prefix; d = (key >> 10) & 0x7FFu; }
		if (pass == 2) { ok = (key >> 10) == n.prefix; d = key & 0x3FFu; }
		if (ok) atomicAdd(&h[d], 1u);
	}
	__syncthreads();
	for (int t = threadIdx.x; t < kBins; t += kB)
		if (h[t]) atomicAdd(&hist[(size_t)j * kBins + t], (int)h[t]);
}

// the bin of the (summed) histogram that holds rank r; one block per node
__global__ __launch_bounds__(kB) void dp_descend_kernel(const int *__restrict__ hist, DpNode *__restrict__ nd, int pass)
{
	__shared__ uint32_t wsum[kB / 64];
	__shared__ uint32_t res[3];
	const int j = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t v0 = (uint32_t)hist[(size_t)j * kBins + 2 * threadIdx.x], v1 = (uint32_t)hist[(size_t)j * kBins + 2 * threadIdx.x + 1];
	uint32_t incl = v0 + v1;
	for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)incl, o); if (lane >= o) incl += y; }
	if (lane == 63) wsum[w] = incl;
	__syncthreads();
	uint32_t cum = incl - (v0 + v1);
	for (int q = 0; q < w; ++q) cum += wsum[q];
	const uint32_t r = nd[j].r;
	if (r >= cum && r < cum + v0) { res[0] = 2 * threadIdx.x; res[1] = cum; res[2] = v0; }
	else if (r >= cum + v0 && r < cum + v0 + v1) { res[0] = 2 * threadIdx.x + 1; res[1] = cum + v0; res[2] = v1; }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		DpNode n = nd[j];
		n.prefix = (n.prefix << (pass == 2 ? 10 : 11)) | res[0];
		n.r -= res[1];
		n.neq = res[2];
		n.need = n.r + 1;
		if (pass == 2) n.pivot = (n.prefix >> n.shl) + n.kmin;   // back from the key window to the ordered key
		nd[j] = n;
	}
}

// elements that tie with the pivot of a node that has to split its ties: records {k2, k3, original index, -} for the exchange
__global__ __launch_bounds__(kB) void dp_ties_kernel(const float4 *__restrict__ P, const int *__restrict__ seg, const DpNode *__restrict__ nd, long long first_global,
                                                     unsigned char *__restrict__ cand, uint4 *__restrict__ rec, int *__restrict__ cnt, int *__restrict__ flag)
{
	const int j = blockIdx.y, s = seg[j], e = seg[j + 1];
	const DpNode n = nd[j];
	for (long long i = s + (long long)blockIdx.x * kB + threadIdx.x; i < e; i += (long long)gridDim.x * kB)
	{
		unsigned char c = 0;
		const float4 p = load_particle(P, i);
		if (n.need != n.neq && ordered_bits(axis_of(p, n.axis)) == n.pivot)
		{
			const int slot = atomicAdd(&cnt[j], 1);
			if (slot < kTie)
			{
				rec[(size_t)j * kTie + slot] = make_uint4(n.a2 >= 0 ? ordered_bits(axis_of(p, n.a2)) : 0u, n.a3 >= 0 ? ordered_bits(axis_of(p, n.a3)) : 0u,
				                                         (uint32_t)(first_global + __float_as_int(p.w)), 0u);
				c = (unsigned char)(slot + 1);
			}
			else *flag = 1;
		}
		cand[i] = c;
	}
}

// all ranks' tie records of node j (block b of `all`: [m][kTie] records, then [m] counts): the first `need` in the order
// (k2, k3, original index) go left.  dec[j][slot] = 1 for this rank's records that do.
__global__ __launch_bounds__(kB) void dp_decide_kernel(const char *__restrict__ all, size_t block_bytes, int m, int world, int rank, const DpNode *__restrict__ nd,
                                                       unsigned char *__restrict__ dec, int *__restrict__ flag)
{
	__shared__ uint4 recs[kTie * 32];   // at most 32 ranks (dpart_begin)
	__shared__ int base[33];
	const int j = blockIdx.x;
	if (nd[j].need == nd[j].neq) return;
	if (threadIdx.x == 0)
	{
		int tot = 0;
		for (int b = 0; b < world; ++b)
		{
			base[b] = tot;
			const int *cnt = reinterpret_cast<const int *>(all + (size_t)b * block_bytes + sizeof(uint4) * (size_t)m * kTie);
			tot += min(cnt[j], kTie);
			if (cnt[j] > kTie) *flag = 1;
		}
		base[world] = tot;
	}
	__syncthreads();
	const int tot = base[world];
	for (int b = 0; b < world; ++b)
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(all + (size_t)b * block_bytes) + (size_t)j * kTie;
		for (int q = threadIdx.x; q < base[b + 1] - base[b]; q += kB) recs[base[b] + q] = src[q];
	}
	__syncthreads();
	const uint32_t need = nd[j].need;
	for (int q = base[rank] + threadIdx.x; q < base[rank + 1]; q += kB)
	{
		const uint4 me = recs[q];
		uint32_t before = 0;
		for (int o = 0; o < tot; ++o)
		{
			const uint4 x = recs[o];
			before += (x.x < me.x || (x.x == me.x && (x.y < me.y || (x.y == me.y && x.z < me.z)))) ? 1u : 0u;
		}
		dec[(size_t)j * kTie + (q - base[rank])] = before < need ? 1 : 0;
	}
}

// 0 = left child, 1 = right child
__device__ inline int dp_side(const float4 p, const DpNode &n, int j, unsigned char cand, const unsigned char *__restrict__ dec, uint32_t &key)
{
	key = ordered_bits(axis_of(p, n.axis));
	if (key < n.pivot) return 0;
	if (key > n.pivot) return 1;
	if (n.need == n.neq || cand == 0) return 0;   // (cand == 0 with split ties: the tie list overflowed, flag already up)
	return dec[(size_t)j * kTie + cand - 1] ? 0 : 1;
}

// per node: local size of the left child, smallest key of the right child (signed image, INT_MAX = none)
__global__ __launch_bounds__(kB) void dp_count_kernel(const float4 *__restrict__ P, const int *__restrict__ seg, const DpNode *__restrict__ nd,
                                                      const unsigned char *__restrict__ cand, const unsigned char *__restrict__ dec, int *__restrict__ cntL,
                                                      int *__restrict__ minR)
{
	__shared__ int wl[kB / 64], wm[kB / 64];
	const int j = blockIdx.y, s = seg[j], e = seg[j + 1];
	if (s + (long long)blockIdx.x * kB >= e) return;
	const DpNode n = nd[j];
	int left = 0, mr = INT_MAX;
	for (long long i = s + (long long)blockIdx.x * kB + threadIdx.x; i < e; i += (long long)gridDim.x * kB)
	{
		uint32_t key;
		const int side = dp_side(load_particle(P, i), n, j, cand[i], dec, key);
		left += side == 0;
		if (side == 1) mr = min(mr, skey(key));
	}
	for (int o = 32; o > 0; o >>= 1) { left += __shfl_xor(left, o); mr = min(mr, __shfl_xor(mr, o)); }
	if ((threadIdx.x & 63) == 0) { wl[threadIdx.x >> 6] = left; wm[threadIdx.x >> 6] = mr; }
	__syncthreads();
	if (threadIdx.x == 0)
	{
		for (int q = 1; q < kB / 64; ++q) { left += wl[q]; mr = min(mr, wm[q]); }
		if (left) atomicAdd(&cntL[j], left);
		if (mr != INT_MAX) atomicMin(&minR[j], mr);
	}
}

// evalBox for the children of level l's nodes (fmm_cart3_kdtree.cuh:109-137): the pivot is the largest key of the left child,
// minR the smallest of the right one; local segments of the children; cursors cleared
__global__ void dp_boxes_kernel(TopArrays t, int l, long long n_global, const DpNode *__restrict__ nd, const int *__restrict__ minR, const int *__restrict__ seg,
                                const int *__restrict__ cntL, int *__restrict__ seg_next, int *__restrict__ cursor)
{
	const int m = 1 << l, j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const int node = m - 1 + j, a1 = nd[j].axis;
	for (int side = 0; side < 2; ++side)
	{
		const int c = 2 * j + side, child = 2 * m - 1 + c;
		float lb[3] = {t.lb[3 * node], t.lb[3 * node + 1], t.lb[3 * node + 2]};
		float rb[3] = {t.rb[3 * node], t.rb[3 * node + 1], t.rb[3 * node + 2]};
		if (side) { const float v = unordered_bits(ukey(minR[j])); if (a1 == 0) lb[0] = v; else if (a1 == 1) lb[1] = v; else lb[2] = v; }
		else { const float v = unordered_bits(nd[j].pivot); if (a1 == 0) rb[0] = v; else if (a1 == 1) rb[1] = v; else rb[2] = v; }
		t.lb[3 * child] = lb[0]; t.lb[3 * child + 1] = lb[1]; t.lb[3 * child + 2] = lb[2];
		t.rb[3 * child] = rb[0]; t.rb[3 * child + 1] = rb[1]; t.rb[3 * child + 2] = rb[2];
		t.sd[child] = longest_axis(rb[0] - lb[0], rb[1] - lb[1], rb[2] - lb[2]);
		t.index[child] = (int)range_start(n_global, c, 2LL * m);
		cursor[c] = 0;
	}
	seg_next[2 * j] = seg[j];
	seg_next[2 * j + 1] = seg[j] + cntL[j];
	if (j == m - 1) seg_next[2 * m] = seg[m];
}

// in-place (ping-pong) partition of every node's local particles into [left child | right child]
__global__ __launch_bounds__(kB) void dp_scatter_kernel(const float4 *__restrict__ P, float4 *__restrict__ Pout, const int *__restrict__ seg,
                                                        const int *__restrict__ seg_next, const DpNode *__restrict__ nd, const unsigned char *__restrict__ cand,
                                                        const unsigned char *__restrict__ dec, int *__restrict__ cursor)
{
	__shared__ int wc[kB / 64][2], bs[2];
	const int j = blockIdx.y, s = seg[j], e = seg[j + 1];
	const DpNode n = nd[j];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (long long base = s + (long long)blockIdx.x * kB; base < e; base += (long long)gridDim.x * kB)
	{
		const long long i = base + threadIdx.x;
		int side = -1;
		float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
		if (i < e) { uint32_t key; p = load_particle(P, i); side = dp_side(p, n, j, cand[i], dec, key); }
		const unsigned long long bl = __ballot(side == 0), br = __ballot(side == 1), below = (1ull << lane) - 1ull;
		const int lr = side == 0 ? __popcll(bl & below) : __popcll(br & below);
		if (lane == 0) { wc[w][0] = __popcll(bl); wc[w][1] = __popcll(br); }
		__syncthreads();
		if (threadIdx.x < 2)
		{
			int tot = 0;
			for (int q = 0; q < kB / 64; ++q) { const int x = wc[q][threadIdx.x]; wc[q][threadIdx.x] = tot; tot += x; }
			bs[threadIdx.x] = tot ? atomicAdd(&cursor[2 * j + threadIdx.x], tot) : 0;
		}
		__syncthreads();
		if (side >= 0) Pout[seg_next[2 * j + side] + bs[side] + wc[w][side] + lr] = p;
		__syncthreads();
	}
}

__global__ void dp_counts_kernel(const int *__restrict__ seg, int world, int *__restrict__ counts)
{
	const int g = threadIdx.x;
	if (g < world) counts[g] = seg[g + 1] - seg[g];
}

// destination rank of every particle by its LOCAL INDEX (P is grouped by destination: seg[g] .. seg[g + 1]), the key of the
// stable one-pass sort that puts every destination's particles back into the order of the local state
__global__ __launch_bounds__(kB) void dp_dest_kernel(const float4 *__restrict__ P, const int *__restrict__ seg, int world, long long n, uint32_t *__restrict__ dest)
{
	for (long long i = (long long)blockIdx.x * kB + threadIdx.x; i < n; i += (long long)gridDim.x * kB)
	{
		int g = 0;
		while (g + 1 < world && seg[g + 1] <= i) ++g;
		dest[__float_as_int(load_particle(P, i).w)] = (uint32_t)g;
	}
}
// records {x, y, z, vx, vy, vz}: slot i holds the particle order[i] of the local state (grouped by destination rank, by local
// index inside a group)
__global__ __launch_bounds__(kB) void dp_send_kernel(const uint32_t *__restrict__ order, const float *__restrict__ pos, const float *__restrict__ vel, long long n,
                                                     float *__restrict__ out)
{
	for (long long i = (long long)blockIdx.x * kB + threadIdx.x; i < n; i += (long long)gridDim.x * kB)
	{
		const long long o = order[i];
		out[6 * i] = pos[3 * o]; out[6 * i + 1] = pos[3 * o + 1]; out[6 * i + 2] = pos[3 * o + 2];
		out[6 * i + 3] = vel[3 * o]; out[6 * i + 4] = vel[3 * o + 1]; out[6 * i + 5] = vel[3 * o + 2];
	}
}
__global__ __launch_bounds__(kB) void dp_recv_kernel(const float *__restrict__ in, long long n, float *__restrict__ pos, float *__restrict__ vel)
{
	for (long long i = (long long)blockIdx.x * kB + threadIdx.x; i < n; i += (long long)gridDim.x * kB)
	{
		pos[3 * i] = in[6 * i]; pos[3 * i + 1] = in[6 * i + 1]; pos[3 * i + 2] = in[6 * i + 2];
		vel[3 * i] = in[6 * i + 3]; vel[3 * i + 1] = in[6 * i + 4]; vel[3 * i + 2] = in[6 * i + 5];
	}
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout
{
	size_t tie_block, coll_send, coll_recv, send, recv, total;
};
Layout layout_of(long long n_local, int world)
{
	int d = 0;
	while ((1 << d) < world) ++d;
	const size_t mmax = d > 0 ? (size_t)1 << (d - 1) : 1;
	Layout L;
	L.tie_block = align256(mmax * (sizeof(uint4) * kTie + sizeof(int)));
	L.coll_send = align256(std::max<size_t>(std::max<size_t>(mmax * kBins * sizeof(int), L.tie_block), 1024));
	L.coll_recv = align256(std::max<size_t>((size_t)world * L.tie_block, (size_t)world * world * sizeof(int) + 1024));
	L.send = align256((size_t)n_local * 24);
	L.recv = align256((size_t)n_local * 24);
	L.total = L.coll_send + L.coll_recv + L.send + L.recv;
	return L;
}

enum Stage { ST_IDLE = 0, ST_BOUNDS, ST_HIST, ST_TIES, ST_MINR, ST_COUNTS, ST_MOVE };

} // namespace

int kd_dist_top_arrays(nbco_ctx *c, int ntop, float **lb, float **rb, int **sd, int **index);
int kd_dist_set_partitioned(nbco_ctx *c, long long n_global, int world, int rank);

int dpart_workspace(nbco_ctx *c, long long n_global, int world, long long *bytes)
{
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, n_global, world, 0, &lay));
	*bytes = (long long)layout_of(lay.n_local, world).total;
	return NBCO_OK;
}

// runs the local stages that follow the collective just completed (or the first ones) and describes the next collective
static int dpart_advance(nbco_ctx *c, nbco_dist_step *out)
{
	static const bool trace = getenv("NBCO_DPART_TRACE") != nullptr;
	nbco_ctx::DPart &s = c->dpart;
	hipStream_t st = c->stream;
	const int G = s.world, d = s.d;
	const long long nl = s.n_local;
	const Layout L = layout_of(nl, G);
	char *W = (char *)s.work;
	int *coll = reinterpret_cast<int *>(W);
	char *coll_recv = W + L.coll_send;
	float *sendbuf = reinterpret_cast<float *>(W + L.coll_send + L.coll_recv), *recvbuf = reinterpret_cast<float *>(W + L.coll_send + L.coll_recv + L.send);
	TopArrays t;
	NBCO_TRY(kd_dist_top_arrays(c, (1 << (d + 1)) - 1, &t.lb, &t.rb, &t.sd, &t.index));
	// scratch of the state machine: [nodes | seg a | seg b | cntL | cursor | flag] ints + cand bytes + dec bytes
	const size_t mmax = (size_t)1 << std::max(d, 1);
	NBCO_TRY(c->reserve(c->dpart_buf, sizeof(DpNode) * mmax + sizeof(int) * (6 * mmax + 16) + (size_t)nl + mmax * kTie + 256));
	DpNode *nd = c->dpart_buf.as<DpNode>();
	int *seg_a = reinterpret_cast<int *>(nd + mmax), *seg_b = seg_a + (mmax + 1), *cntL = seg_b + (mmax + 1), *cursor = cntL + mmax, *flag = cursor + 2 * mmax;
	unsigned char *cand = reinterpret_cast<unsigned char *>(flag + 4), *dec = cand + nl;
	int *seg = s.seg_flip ? seg_b : seg_a, *seg_next = s.seg_flip ? seg_a : seg_b;
	float4 *P = (s.pos_flip ? c->pos4_alt : c->pos4).as<float4>(), *Palt = (s.pos_flip ? c->pos4 : c->pos4_alt).as<float4>();
	auto step = [&](int op, size_t send_off, size_t recv_off, long long count) {
		*out = nbco_dist_step{};
		out->op = op; out->send_off = (long long)send_off; out->recv_off = (long long)recv_off; out->count = count;
		return (int)NBCO_OK;
	};
	auto start_hist = [&]() {
		const int m = 1 << s.level;
		NBCO_HIP(hipMemsetAsync(coll, 0, sizeof(int) * (size_t)m * kBins, st));
		hipLaunchKernelGGL(dp_hist_kernel, dim3(kGX, m), dim3(kB), 0, st, (const float4 *)P, (const int *)seg, (const DpNode *)nd, s.pass, coll);
		NBCO_HIP(hipGetLastError());
		s.stage = ST_HIST;
		return step(NBCO_COLL_ALLREDUCE_SUM_I32, 0, 0, (long long)m * kBins);
	};
	auto start_counts = [&]() {
		hipLaunchKernelGGL(dp_counts_kernel, dim3(1), dim3(64), 0, st, (const int *)seg, G, coll);
		NBCO_HIP(hipGetLastError());
		s.stage = ST_COUNTS;
		return step(NBCO_COLL_ALLGATHER, 0, L.coll_send, (long long)(sizeof(int) * G));
	};
	if (trace)
	{
		const hipError_t e = hipStreamSynchronize(st);
		fprintf(stderr, "[dpart rank %d] stage %d level %d pass %d (%s)\n", s.rank, s.stage, s.level, s.pass, hipGetErrorString(e));
	}
	switch (s.stage)
	{
	case ST_BOUNDS:
	{
		hipLaunchKernelGGL(dp_root_kernel, dim3(1), dim3(64), 0, st, (const int *)coll, t);
		NBCO_HIP(hipGetLastError());
		s.level = 0; s.pass = 0;
		if (d == 0) return start_counts();
		hipLaunchKernelGGL(dp_nodes_kernel, dim3(1), dim3(64), 0, st, t, 0, s.n_global, nd);
		return start_hist();
	}
	case ST_HIST:
	{
		const int m = 1 << s.level;
		hipLaunchKernelGGL(dp_descend_kernel, dim3(m), dim3(kB), 0, st, (const int *)coll, nd, s.pass);
		NBCO_HIP(hipGetLastError());
		if (s.pass < 2) { ++s.pass; return start_hist(); }
		// ties: this rank's block = [m][kTie] records + [m] counts
		uint4 *rec = reinterpret_cast<uint4 *>(coll);
		int *cnt = reinterpret_cast<int *>(rec + (size_t)m * kTie);
		NBCO_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)m, st));
		hipLaunchKernelGGL(dp_ties_kernel, dim3(kGX, m), dim3(kB), 0, st, (const float4 *)P, (const int *)seg, (const DpNode *)nd, (long long)s.rank * nl, cand, rec, cnt,
		                   flag);
		NBCO_HIP(hipGetLastError());
		s.stage = ST_TIES;
		s.tie_bytes = sizeof(uint4) * (size_t)m * kTie + sizeof(int) * (size_t)m;
		return step(NBCO_COLL_ALLGATHER, 0, L.coll_send, (long long)s.tie_bytes);
	}
	case ST_TIES:
	{
		const int m = 1 << s.level;
		hipLaunchKernelGGL(dp_decide_kernel, dim3(m), dim3(kB), 0, st, (const char *)coll_recv, s.tie_bytes, m, G, s.rank, (const DpNode *)nd, dec, flag);
		NBCO_HIP(hipMemsetAsync(cntL, 0, sizeof(int) * (size_t)m, st));
		hipLaunchKernelGGL(dp_fill_kernel, dim3(1), dim3(64), 0, st, coll, INT_MAX, m);
		hipLaunchKernelGGL(dp_count_kernel, dim3(kGX, m), dim3(kB), 0, st, (const float4 *)P, (const int *)seg, (const DpNode *)nd, (const unsigned char *)cand,
		                   (const unsigned char *)dec, cntL, coll);
		NBCO_HIP(hipGetLastError());
		s.stage = ST_MINR;
		return step(NBCO_COLL_ALLREDUCE_MIN_I32, 0, 0, m);
	}
	case ST_MINR:
	{
		const int m = 1 << s.level;
		hipLaunchKernelGGL(dp_boxes_kernel, dim3(1), dim3(64), 0, st, t, s.level, s.n_global, (const DpNode *)nd, (const int *)coll, (const int *)seg, (const int *)cntL,
		                   seg_next, cursor);
		hipLaunchKernelGGL(dp_scatter_kernel, dim3(kGX, m), dim3(kB), 0, st, (const float4 *)P, Palt, (const int *)seg, (const int *)seg_next, (const DpNode *)nd,
		                   (const unsigned char *)cand, (const unsigned char *)dec, cursor);
		NBCO_HIP(hipGetLastError());
		s.seg_flip = !s.seg_flip; s.pos_flip = !s.pos_flip;
		std::swap(seg, seg_next); std::swap(P, Palt);
		++s.level; s.pass = 0;
		if (s.level == d) return start_counts();
		hipLaunchKernelGGL(dp_nodes_kernel, dim3(1), dim3(64), 0, st, t, s.level, s.n_global, nd);
		return start_hist();
	}
	case ST_COUNTS:
	{
		// the count matrix and the tie flag on the host: the one synchronisation of the re-partition
		std::vector<int> M((size_t)G * G);
		int hflag = 0;
		NBCO_HIP(hipMemcpyAsync(M.data(), coll_recv, sizeof(int) * (size_t)G * G, hipMemcpyDeviceToHost, st));
		NBCO_HIP(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st));
		NBCO_HIP(hipStreamSynchronize(st));
		if (hflag) { s.stage = ST_IDLE; return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_dist_repartition: more than 64 particles of one rank tie with a pivot (use nbco_dist_partition)"); }
		long long in = 0;
		for (int g = 0; g < G; ++g) { s.rows_send[g] = M[(size_t)s.rank * G + g]; s.rows_recv[g] = M[(size_t)g * G + s.rank]; in += s.rows_recv[g]; }
		if (in != nl) { s.stage = ST_IDLE; return c->fail(NBCO_ERR_HIP, "internal error: nbco_dist_repartition produced unbalanced domains"); }
		// The scatter passes leave a destination's particles in the order their workgroups happened to finish.  The receiver's
		// local build takes the local index as the last key of its stable-sort chain, so the order has to be the gathered
		// partition's (source rank, then index in the source's state): one stable radix pass over the destination ranks, taken in
		// the order of the local state.
		{
			NBCO_TRY(c->reserve(c->idx, sizeof(uint32_t) * (size_t)nl));
			NBCO_TRY(c->reserve(c->idx_alt, sizeof(uint32_t) * (size_t)nl));
			NBCO_TRY(c->reserve(c->keys, sizeof(uint64_t) * (size_t)nl));
			uint32_t *dest = c->idx.as<uint32_t>(), *dest_sorted = c->idx_alt.as<uint32_t>(), *order = c->keys.as<uint32_t>();
			hipLaunchKernelGGL(dp_dest_kernel, dim3(256), dim3(kB), 0, st, (const float4 *)P, (const int *)seg, G, nl, dest);
			NBCO_HIP(hipGetLastError());
			const unsigned bits = (unsigned)std::max(d, 1);
			rocprim::counting_iterator<uint32_t> iota(0u);
			size_t bytes = 0;
			NBCO_HIP(rocprim::radix_sort_pairs(nullptr, bytes, dest, dest_sorted, iota, order, (size_t)nl, 0u, bits, st));
			NBCO_TRY(c->reserve(c->sort_tmp, bytes));
			bytes = c->sort_tmp.bytes;
			NBCO_HIP(rocprim::radix_sort_pairs(c->sort_tmp.ptr, bytes, dest, dest_sorted, iota, order, (size_t)nl, 0u, bits, st));
			hipLaunchKernelGGL(dp_send_kernel, dim3(256), dim3(kB), 0, st, (const uint32_t *)order, (const float *)s.state, (const float *)(s.state + 3 * nl), nl, sendbuf);
		}
		NBCO_HIP(hipGetLastError());
		s.stage = ST_MOVE;
		NBCO_TRY(step(NBCO_COLL_ALLTOALL, L.coll_send + L.coll_recv, L.coll_send + L.coll_recv + L.send, nl));
		out->row_bytes = 24;
		for (int g = 0; g < G; ++g) { out->rows_send[g] = s.rows_send[g]; out->rows_recv[g] = s.rows_recv[g]; }
		return NBCO_OK;
	}
	case ST_MOVE:
	{
		hipLaunchKernelGGL(dp_recv_kernel, dim3(256), dim3(kB), 0, st, (const float *)recvbuf, nl, s.state, s.state + 3 * nl);
		NBCO_HIP(hipGetLastError());
		s.stage = ST_IDLE;
		NBCO_TRY(kd_dist_set_partitioned(c, s.n_global, G, s.rank));
		return step(NBCO_COLL_DONE, 0, 0, 0);
	}
	default:
		return c->fail(NBCO_ERR_ARG, "nbco_dist_repartition_next: no re-partition in progress");
	}
}

int dpart_begin(nbco_ctx *c, float *state_local, long long n_global, int world, int rank, void *work, long long work_bytes, nbco_dist_step *out)
{
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, n_global, world, rank, &lay));
	const Layout L = layout_of(lay.n_local, world);
	if (work_bytes < (long long)L.total) return c->fail(NBCO_ERR_ARG, "nbco_dist_repartition_begin: workspace too small (nbco_dist_repartition_workspace)");
	if (world > 32) return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_dist_repartition: at most 32 domains (use nbco_dist_partition)");
	nbco_ctx::DPart &s = c->dpart;
	s = nbco_ctx::DPart{};
	c->last_eval.valid = false;   // pos4 is this stage's scratch from here on: the last evaluation's lists no longer have their positions
	s.world = world; s.rank = rank; s.d = lay.d; s.n_global = n_global; s.n_local = lay.n_local; s.state = state_local; s.work = work;
	const long long nl = lay.n_local;
	hipStream_t st = c->stream;
	NBCO_TRY(c->reserve(c->pos4, sizeof(float4) * (size_t)nl));
	NBCO_TRY(c->reserve(c->pos4_alt, sizeof(float4) * (size_t)nl));
	const size_t mmax = (size_t)1 << std::max(lay.d, 1);
	NBCO_TRY(c->reserve(c->dpart_buf, sizeof(DpNode) * mmax + sizeof(int) * (6 * mmax + 16) + (size_t)nl + mmax * kTie + 256));
	DpNode *nd = c->dpart_buf.as<DpNode>();
	int *seg_a = reinterpret_cast<int *>(nd + mmax), *flag = seg_a + 2 * (mmax + 1) + mmax + 2 * mmax;
	int *coll = reinterpret_cast<int *>(work);
	const int seg0[2] = {0, (int)nl};
	NBCO_HIP(hipMemcpyAsync(seg_a, seg0, sizeof seg0, hipMemcpyHostToDevice, st));
	NBCO_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
	hipLaunchKernelGGL(dp_fill_kernel, dim3(1), dim3(64), 0, st, coll, INT_MAX, 6);
	hipLaunchKernelGGL(dp_pack_kernel, dim3(128), dim3(kB), 0, st, (const float *)state_local, nl, c->pos4.as<float4>(), coll);
	NBCO_HIP(hipGetLastError());
	NBCO_HIP(hipStreamSynchronize(st));   // (seg0 lives on this stack frame)
	s.stage = ST_BOUNDS;
	*out = nbco_dist_step{};
	out->op = NBCO_COLL_ALLREDUCE_MIN_I32; out->send_off = 0; out->recv_off = 0; out->count = 6;
	return NBCO_OK;
}

int dpart_next(nbco_ctx *c, nbco_dist_step *out) { return dpart_advance(c, out); }
