// kd_build_kernels.hpp -- part of k_fmm_kd.hip (included there, in this place: one translation unit, one anonymous namespace)
// tree geometry (rounded like the oracle), build prologue, the fused turnaround pass, per-level kernels of the sorting build, the in-LDS subtree build
// (no include guard on purpose: this is a section of that file, not a header)
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
                                                           int canon, int two_pass, int *__restrict__ flag, const int *__restrict__ top_sd, int top_root1)
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
		// (the local build of a kd-domain: behind the ancestors inside this tree come the domain root's ancestors in the global tree)
		int mine = -1;
		if (tid < 31)
		{
			if (up > 0) mine = t.splitdim[up - 1];
			else if (top_sd) { const int upg = top_root1 >> (tid - l0 + 1); if (upg > 0) mine = top_sd[upg - 1]; }
		}
		const int a = t.splitdim[root];
		const float l0f = t.lbound[3 * root], l1f = t.lbound[3 * root + 1], l2f = t.lbound[3 * root + 2];
		const float r0f = t.rbound[3 * root], r1f = t.rbound[3 * root + 1], r2f = t.rbound[3 * root + 2];
		int b0 = -1, b1 = -1, b2 = -1;
		for (int k = 0; k < 31; ++k)
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

