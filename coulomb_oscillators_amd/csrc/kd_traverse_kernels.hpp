// kd_traverse_kernels.hpp -- part of k_fmm_kd.hip (included there, in this place: one translation unit, one anonymous namespace)
// dual tree traversal: admissibility, the level-synchronous frontier kernel, its init / finish kernels
// (no include guard on purpose: this is a section of that file, not a header)
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
		// Lanes of a wave that count the SAME target share one atomic: the group of the first such lane is served by that lane
		// (count = the group's size, every member takes the returned base plus its place in the group), everybody else goes
		// alone.  In the benchmark's first steps neighbouring pairs rarely share a target; a thousand steps in, a leaf stretched
		// by an ejected particle is a partner of ten thousand pairs in a row -- all lanes of wave after wave on one address, each
		// atomic retiring ~27 ns after the one before (the four widest launches took 77-280 us each, `profiles/r03o_late_kernels_before.txt`).
		// Still issued before any result is used: a request keeps {serving lane, place, pending result of the serving lane}.
		const int lane = threadIdx.x & 63;
		const int m2l_at = (int)(cnt_m2l - cnt_p2p);   // (one allocation: [cnt_p2p | fill_p2p | cnt_m2l | fill_m2l])
		struct Req { int from, place; unsigned pend; };
		auto request = [&](int key, bool on) {
			Req r{lane, 0, 0u};
			const uint64_t todo = __ballot(on);
			int size = 1;
			if (todo)   // wave-uniform
			{
				const int first = __builtin_amdgcn_readfirstlane(__ffsll((unsigned long long)todo) - 1);
				const int v = __builtin_amdgcn_readlane(key, first);
				const uint64_t same = __ballot(on && key == v);
				if (on && key == v) { r.from = first; r.place = __popcll(same & ((1ull << lane) - 1ull)); size = __popcll(same); }
			}
			if (on && r.from == lane) r.pend = atomicAdd(&cnt_p2p[key], (unsigned)size);
			return r;
		};
		auto slots = [&](int q, int2 np, bool ok, Req &rx, Req &ry) {
			const bool lst = ok && (q == 1 || q == 2);
			const int kx = q == 1 ? np.x - lbeg : m2l_at + np.x, ky = q == 1 ? np.y - lbeg : m2l_at + np.y;
			rx = request(kx, lst && (dm.d == 0 || dom_touch(dm, np.x)));
			ry = request(ky, lst && (dm.d == 0 || dom_touch(dm, np.y)));
			// (-1: the node belongs to another domain)
			return make_int2(lst && (dm.d == 0 || dom_touch(dm, np.x)) ? 0 : -1, lst && (dm.d == 0 || dom_touch(dm, np.y)) ? 0 : -1);
		};
		Req q0x, q0y, q1x, q1y, q2x, q2y, q3x, q3y;
		int2 r0 = slots(k0, p0, nch == 0 && i < nin, q0x, q0y), r1 = slots(k1, ch.a, nch > 0, q1x, q1y), r2 = slots(k2, ch.b, nch > 1, q2x, q2y),
		     r3 = slots(k3, ch.c, nch > 2, q3x, q3y);
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
		// the slots: the serving lane's result plus the place in its group (all lanes take part in the exchange)
		auto settle = [&](const Req &rq, int &slot) {
			const int base = __shfl((int)rq.pend, rq.from);
			if (slot >= 0) slot = base + rq.place;
		};
		settle(q0x, r0.x); settle(q0y, r0.y); settle(q1x, r1.x); settle(q1y, r1.y);
		settle(q2x, r2.x); settle(q2y, r2.y); settle(q3x, r3.x); settle(q3y, r3.y);
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

