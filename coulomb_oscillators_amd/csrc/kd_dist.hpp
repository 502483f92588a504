// kd_dist.hpp -- part of k_fmm_kd.hip (included there, in this place: one translation unit, one anonymous namespace)
// multi-GPU: kd-domain sharding (layout, partition, local stage, global tree)
// (no include guard on purpose: this is a section of that file, not a header)
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
		// the domain's slice of the partitioned state, IN THE ORDER OF THE GATHERED STATE: the selection levels leave a node's
		// particles in the order their workgroups happened to finish, and the local build takes the local index as the last key of
		// its stable-sort chain (pivot ties on every axis of the chain: a node near the domain root whose ancestors all split
		// along its own axis) -- with the gathered order restored that key is the single-GPU tree's, whatever the launch order
		// was (`tools/soak_dist.py` found domains that differed between two runs of the same input)
		{
			uint32_t *own = reinterpret_cast<uint32_t *>(unsort + (size_t)rank * nl), *sorted = c->idx.as<uint32_t>();
			int bits = 1;
			while ((1LL << bits) < n) ++bits;
			size_t bytes = 0;
			NBCO_HIP(rocprim::radix_sort_keys(nullptr, bytes, own, sorted, (size_t)nl, 0u, (unsigned)bits, st));
			NBCO_TRY(c->reserve(c->sort_tmp, bytes));
			bytes = c->sort_tmp.bytes;
			NBCO_HIP(rocprim::radix_sort_keys(c->sort_tmp.ptr, bytes, own, sorted, (size_t)nl, 0u, (unsigned)bits, st));
			NBCO_TRY(launch_gather3(c, state_local, state_all, reinterpret_cast<const int *>(sorted), nl, false));
			NBCO_TRY(launch_gather3(c, state_local + 3 * nl, state_all + 3 * n, reinterpret_cast<const int *>(sorted), nl, false));
		}
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
		// the stable-sort chain's tie-breaking keys reach back to the GLOBAL root: the build takes the split axes above the domain
		// root from the top tree (found at N = 2^24, where pivots with equal split coordinates are common: without them 531 leaves of
		// one domain held other particles than the single-GPU tree's, `tests/test_gpu_dist.py::test_config_four_...`)
		struct TopAxes
		{
			nbco_ctx *c;
			TopAxes(nbco_ctx *c_, const int *sd, int root1) : c(c_) { c->top_sd = sd; c->top_root1 = root1; }
			~TopAxes() { c->top_sd = nullptr; c->top_root1 = 1; }
		} top_axes(c, d > 0 ? top.splitdim : nullptr, (1 << d) + lay.rank);
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

