// kd_let.hpp -- part of k_fmm_kd.hip (included there, in this place: one translation unit, one anonymous namespace)
// multi-GPU: locally-essential-tree exchange and the second half of the sharded evaluation
// (no include guard on purpose: this is a section of that file, not a header)
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
	// (the segment holds exactly cursors[4 r] records, or -- capped form -- what the caller sized it for: records beyond are dropped)
	const long long total = std::min((long long)cursors[4 * r], nb.v[r + 1] - nb.v[r]) * rec;
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
	const long long total = (long long)cursors[4 * r + 1] << wlog, room = pb.v[r + 1] - pb.v[r];
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock)
	{
		const int j = (int)(i & ((1 << wlog) - 1));
		const int2 sl = sel_leaf[((size_t)r << L_loc) + (i >> wlog)];
		if (j >= mult[sl.x] || sl.y + j >= room) continue;
		const int q = index[sl.x] + j;
		float4 p = pos[q];
		p.w = __int_as_float((int)(first_global + q));
		out[pb.v[r] + sl.y + j] = p;
	}
}
// capped form: the records of a segment nobody filled say so (node id / particle index -1; the unpack kernels skip them)
template <typename T>
__global__ __launch_bounds__(kBlock) void let_pad_kernel(const int *__restrict__ cursors, LetBases nb, LetBases pb, int rec, T *__restrict__ mp_out,
                                                         float4 *__restrict__ pos_out)
{
	const int r = blockIdx.y;
	const long long ncap = nb.v[r + 1] - nb.v[r], pcap = pb.v[r + 1] - pb.v[r];
	const long long n0 = std::min((long long)cursors[4 * r], ncap), p0 = std::min((long long)cursors[4 * r + 2], pcap);
	const long long nfree = ncap - n0, total = nfree + (pcap - p0);
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock)
	{
		if (i < nfree) mp_out[(nb.v[r] + n0 + i) * rec] = let_id_enc(-1, T());
		else pos_out[pb.v[r] + p0 + (i - nfree)] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
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
// settled_only: the last attempt's kernels may still be running (capped form: the caller has waited for its counts, not for
// its end) -- look at the pair of the attempt before only
int let_report(nbco_ctx *c, bool settled_only = false)
{
	int node = 0, leaf = 0;
	for (int par = 0; par < 2; ++par)
	{
		if (settled_only && par != (int)(c->dist.let_epoch & 1)) continue;   // (the last attempt was let_epoch - 1)
		volatile int *w = c->h_flags + kLetWord + 2 * par;
		const int wn = w[0], wl = w[1];
		w[0] = 0; w[1] = 0;
		if (c->dist.let_ignore[par]) { c->dist.let_ignore[par] = false; continue; }   // an attempt the caller has repeated since
		if (!node) node = wn;
		if (!leaf) leaf = wl;
	}
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

// the two send buffers: for receiver r (rank order) room[2 r + 1] position records of 16 bytes, room[2 r] node records of
// let_node_bytes.  capped: the segments are what the caller sized them for, not what was selected -- records that do not fit
// are dropped, free records are marked invalid.
static int let_pack_segments(nbco_ctx *c, const nbco_dist_layout &lay, const long long *room, void *pos_send, void *mpole_send, bool capped)
{
	const int G = lay.world, offM = sym_off(lay.order), rec = let_rec_floats(offM);
	LetBases nb{}, pb{};
	long long nmax = 0, lmax = 0;
	for (int r = 0; r < G; ++r)
	{
		if (room[2 * r] < 0 || room[2 * r + 1] < 0 || (r == lay.rank && (room[2 * r] || room[2 * r + 1])))
			return c->fail(NBCO_ERR_ARG, "nbco_dist_let_pack: negative segment size, or a segment for the own rank");
		nb.v[r + 1] = nb.v[r] + room[2 * r];
		pb.v[r + 1] = pb.v[r] + room[2 * r + 1];
		nmax = std::max(nmax, room[2 * r]);
		lmax = std::max(lmax, room[2 * r + 1]);
	}
	if ((nb.v[G] > 0 && !mpole_send) || (pb.v[G] > 0 && !pos_send)) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_pack: null send buffer");
	LetView v;
	NBCO_TRY(let_view(c, lay, v));
	int wlog = 0;
	while ((1 << wlog) < c->kd.mlt_max) ++wlog;
	PhaseScope ph(c, NBCO_PH_P2M_M2M);
	const bool f64 = c->kd.real_bytes == 8;
	if (capped && nmax + lmax > 0)
	{
		if (f64) hipLaunchKernelGGL(let_pad_kernel<double>, dim3(grid1d(nmax + lmax, 256), G), dim3(kBlock), 0, c->stream, (const int *)v.cursors, nb, pb, rec,
		                            (double *)mpole_send, (float4 *)pos_send);
		else hipLaunchKernelGGL(let_pad_kernel<float>, dim3(grid1d(nmax + lmax, 256), G), dim3(kBlock), 0, c->stream, (const int *)v.cursors, nb, pb, rec,
		                        (float *)mpole_send, (float4 *)pos_send);
	}
	if (lmax > 0)   // (an upper bound of the lane count: every selected leaf holds at least one particle)
		hipLaunchKernelGGL(let_pack_pos_kernel, dim3(grid1d(lmax << wlog, 2048), G), dim3(kBlock), 0, c->stream, (const float4 *)c->pos4.as<float4>(),
		                   (const int *)c->kd.index, (const int *)c->kd.mult, lay.L_local, wlog, (long long)lay.rank * lay.n_local, (const int2 *)v.sel_leaf,
		                   (const int *)v.cursors, pb, (float4 *)pos_send);
	NBCO_TRY(c->join_aux());   // the upward pass
	if (nmax > 0)
	{
		if (f64)
			hipLaunchKernelGGL(let_pack_mpole_kernel<double>, dim3(grid1d(nmax * rec, 2048), G), dim3(kBlock), 0, c->stream, (const double *)c->kd.mpole, offM, rec,
			                   lay.ntot_local, lay.d, lay.rank, (const int *)v.sel_node, (const int *)v.cursors, nb, (double *)mpole_send);
		else
			hipLaunchKernelGGL(let_pack_mpole_kernel<float>, dim3(grid1d(nmax * rec, 2048), G), dim3(kBlock), 0, c->stream, (const float *)c->kd.mpole, offM, rec,
			                   lay.ntot_local, lay.d, lay.rank, (const int *)v.sel_node, (const int *)v.cursors, nb, (float *)mpole_send);
	}
	NBCO_HIP(hipGetLastError());
	c->dist.let_packed = true;
	c->dist.let_capped = capped;
	c->dist.let_flagged = false;
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
	const int G = lay.world, S = 2 * G + 2;
	for (int s = 0; s < G; ++s)
	{
		if (counts_all[(size_t)s * S + 2 * G]) return c->fail(NBCO_ERR_CAPACITY, "nbco_dist_let_pack: a rank reported list overflow; repeat nbco_dist_let_select on every rank");
		if (counts_all[(size_t)s * S + 2 * G + 1])
			return c->fail(NBCO_ERR_CAPACITY, "nbco_dist_let_pack: a rank's tree build was flagged; repeat the evaluation from nbco_dist_let_local_geom on every rank");
	}
	if (c->sel_warm_used && c->dist.rebuilt) c->note_warm_ok();
	return let_pack_segments(c, lay, counts_all + (size_t)lay.rank * S, pos_send, mpole_send, false);
}

// The exchange without a host round trip in the middle of the evaluation.  The caller sizes the segments BEFORE it knows this
// evaluation's counts (from the evaluation before, with head room): caps_out[2 r] node records and caps_out[2 r + 1] position
// records for receiver r.  It gathers the counts as before, but looks at them only once the whole evaluation is queued, and
// then says what it found (kd_dist_let_settle): every count within its segment and no flag -- the evaluation stands; else it
// is void and repeated with exact sizes (nbco_dist_let_pack).  Nothing here reads the counts on the host.
int kd_dist_let_pack_capped(nbco_ctx *c, const long long *caps_out, void *pos_send, void *mpole_send)
{
	if (!c->dist.let_selected || !c->dist.local_done) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_pack_capped: selection or multipoles missing");
	nbco_dist_layout lay;
	NBCO_TRY(kd_dist_layout(c, c->dist.n_global, c->dist.world, c->dist.rank, &lay));
	return let_pack_segments(c, lay, caps_out, pos_send, mpole_send, true);
}

static int dist_finish_rest(nbco_ctx *c, const char *mp_blocks, size_t mp_stride, float *buf_local, float *a_local, const float *param, const long long *let_counts,
                            const void *let_pos);

// pos_recv / mpole_recv: the records received from ranks 0, 1, .. (counts_all[s][2 me + 1] / counts_all[s][2 me] from rank s)
int kd_dist_let_finish(nbco_ctx *c, const long long *counts_all, const void *pos_recv, const void *mpole_recv, float *buf_local, float *a_local, const float *param)
{
	if (!c->dist.let_packed || c->dist.let_capped) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_finish: call nbco_dist_let_pack first");
	c->dist.let_packed = false; c->dist.let_selected = false;
	const int rc = dist_finish_rest(c, (const char *)mpole_recv, 0, buf_local, a_local, param, counts_all, pos_recv);
	c->dist.let_epoch += 1;
	return rc;
}

// caps_in[2 s], caps_in[2 s + 1]: the sizes of the segments rank s packed for this rank (what it passed as caps_out[2 me ..]);
// pos_recv / mpole_recv hold those segments in rank order, free records marked
int kd_dist_let_finish_capped(nbco_ctx *c, const long long *caps_in, const void *pos_recv, const void *mpole_recv, float *buf_local, float *a_local, const float *param)
{
	if (!c->dist.let_packed || !c->dist.let_capped) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_finish_capped: call nbco_dist_let_pack_capped first");
	c->dist.let_packed = false; c->dist.let_selected = false;
	const int G = c->dist.world, S = 2 * G + 2, me = c->dist.rank;
	std::vector<long long> as_counts((size_t)G * S, 0);   // (dist_finish_rest sums column `me` of a count matrix)
	for (int s = 0; s < G; ++s)
	{
		if (caps_in[2 * s] < 0 || caps_in[2 * s + 1] < 0) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_finish_capped: negative segment size");
		as_counts[(size_t)s * S + 2 * me] = caps_in[2 * s];
		as_counts[(size_t)s * S + 2 * me + 1] = caps_in[2 * s + 1];
	}
	const int rc = dist_finish_rest(c, (const char *)mpole_recv, 0, buf_local, a_local, param, as_counts.data(), pos_recv);
	c->dist.let_epoch += 1;
	return rc;
}

// After nbco_dist_let_finish_capped, once the caller has looked at the gathered counts.  ok != 0: every count fitted its
// segment and no rank raised a flag -- the evaluation stands (and the guard of the evaluation before is reported here).
// ok == 0: it is void; this rank's build is repeated more conservatively if it was a flagged one, the guard words of the
// void attempt are dropped when they arrive, and the caller repeats the evaluation from nbco_dist_let_local_geom.
int kd_dist_let_settle(nbco_ctx *c, int ok)
{
	if (!c->dist.let_capped) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_settle: no capped exchange to settle");
	c->dist.let_capped = false;
	NBCO_TRY(let_report(c, true));
	if (ok)
	{
		if (c->dist.let_flagged) return c->fail(NBCO_ERR_ARG, "nbco_dist_let_settle: this rank's build was flagged (its counts said so); the evaluation cannot stand");
		if (c->sel_warm_used && c->dist.rebuilt) c->note_warm_ok();
		return NBCO_OK;
	}
	c->dist.let_ignore[(c->dist.let_epoch - 1) & 1] = true;
	if (c->dist.let_flagged)
	{
		if (c->sel_warm_used) c->note_warm_miss();
		else if (!c->escalate_build()) return c->fail(NBCO_ERR_UNSUPPORTED, "kd-tree build: tie flag raised by the sorting build");
	}
	else c->eval_counter -= 1;   // (the repeat takes the void attempt's place in the rebuild schedule of opts.tree_steps)
	c->dist.let_flagged = false;
	return NBCO_OK;
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
		if (d > 0) NBCO_TRY(launch_m2m_top_gen(c, P, tv.center, tv.mpole, tv.mult, d - 1, L, 0, f64 ? 1 : 0));
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
	if (cnt.sel_overflow && c->dist.let_capped)
	{
		// capped exchange: the flag travels with the counts the caller is about to look at; it declares the evaluation void
		// (nbco_dist_let_settle) and everybody repeats it.  buf_local still holds the state in the order before this build.
		c->dist.let_flagged = true;
		c->tree_valid = false;
		return NBCO_OK;
	}
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

