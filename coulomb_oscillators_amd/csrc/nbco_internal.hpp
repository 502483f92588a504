// nbco_internal.hpp -- context, error plumbing and launch declarations shared by the HIP
// translation units of libnbco_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/nbco.h"

#define NBCO_HIP(call)                                                                           \
	do {                                                                                         \
		hipError_t e_ = (call);                                                                  \
		if (e_ != hipSuccess) return c->fail_hip(e_, #call, __FILE__, __LINE__);                 \
	} while (0)

#define NBCO_HIP_M(ctx_, call)                                                                   \
	do {                                                                                         \
		hipError_t e_ = (call);                                                                  \
		if (e_ != hipSuccess) return (ctx_)->fail_hip(e_, #call, __FILE__, __LINE__);            \
	} while (0)

#define NBCO_TRY(call)                   \
	do {                                 \
		int rc_ = (call);                \
		if (rc_ != NBCO_OK) return rc_;  \
	} while (0)

// ---- checked build (-DNBCO_CHECKED, libnbco_hip_checked.so) ------------------------------------------------------------
// Every index that a kernel READS FROM A LIST (frontier pairs, unordered pair lists and their slots, sorted entries, source
// descriptors, work units) is range-checked before it is used as an address: a violation is counted per site and the index
// replaced by a harmless one, so that a corrupted or stale list shows up as a number (nbco_debug_violations) instead of a GPU
// memory fault.  The production build compiles the checks away.
enum
{
	NBCO_CHK_FRONTIER = 0,   // traverse_kernel: node ids of a frontier pair
	NBCO_CHK_FILL = 1,       // list_fill_kernel: region slot, targets, slot inside the target's range
	NBCO_CHK_SORT = 2,       // list_segsort_kernel: source index of an entry, output slot
	NBCO_CHK_CHUNK = 3,      // pair kernels: work-unit record
	NBCO_CHK_DESC = 4,       // pair kernels: source descriptor
	NBCO_CHK_L2P = 5,        // l2p_gen_kernel: chunk range, reaction record
	NBCO_CHK_SITES = 8
};
#ifdef NBCO_CHECKED
namespace { __device__ unsigned nbco_violation_count[NBCO_CHK_SITES]; }
#define NBCO_CHECKED_OK(cond, site) ((cond) ? true : (atomicAdd(&nbco_violation_count[site], 1u), false))
#define NBCO_CHECKED_COLLECT(fn)                                                                                      \
	int fn(unsigned *out)                                                                                             \
	{                                                                                                                 \
		unsigned v[NBCO_CHK_SITES];                                                                                   \
		if (hipMemcpyFromSymbol(v, HIP_SYMBOL(nbco_violation_count), sizeof v) != hipSuccess) return NBCO_ERR_HIP;    \
		for (int i = 0; i < NBCO_CHK_SITES; ++i) out[i] += v[i];                                                      \
		return NBCO_OK;                                                                                               \
	}
#else
#define NBCO_CHECKED_OK(cond, site) (true)
#define NBCO_CHECKED_COLLECT(fn) int fn(unsigned *) { return NBCO_OK; }
#endif
int nbco_checked_collect_kd(unsigned *out);
int nbco_checked_collect_oct(unsigned *out);
int nbco_checked_collect_far(unsigned *out);

// device buffer that only grows (the reference's evaluators keep their scratch in function-local
// statics that grow monotonically, fmm_cart3_kdtree.cuh:1480-1498)
struct DevBuf
{
	void *ptr = nullptr;
	size_t bytes = 0;
	template <class T> T *as() const { return (T *)ptr; }
};

struct PhaseTimer
{
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
	double total_ms = 0;
	long long launches = 0;
};

constexpr int kMaxOrder = 10;

// kd-tree state in HBM (fmmTree_kd of fmm_cart3_kdtree.cuh:25-31, plus the packed arrays the
// gfx950 kernels read)
struct KdTreeDev
{
	int L = 0, ntot = 0, order = 0, mlt_max = 0;
	long long n = 0;
	float *center = nullptr, *lbound = nullptr, *rbound = nullptr;   // [ntot][3]
	float4 *csz = nullptr;                                           // [ntot] centre xyz + box diagonal^2
	float *mpole = nullptr, *local = nullptr;                        // [ntot][offM], [ntot][offL]; DOUBLE tuples behind these pointers when real_bytes == 8
	int *mult = nullptr, *index = nullptr, *splitdim = nullptr;      // [ntot]
	int real_bytes = 4;                                              // element size of mpole / local: 8 with opts.far_fp64
};

// uniform-octree state of the last nbco_fmm_traceless call (fmmTree of fmm_cart3_symmetric.cuh:24-30)
struct OctTreeDev
{
	bool valid = false;
	int L = 0, ntot = 0, order = 0, tpl = 0;
	long long n = 0, m2l_entries = 0, p2p_groups = 0, p2p_desc = 0, p2p_chunks = 0;
	float4 *csz = nullptr;
	void *mpole = nullptr, *local = nullptr;   // float or double (real_bytes)
	int real_bytes = 4;
	int mpole_reals = 0;   // reals per multipole tuple: (p+1)^2 traceless, (p+1)(p+2)(p+3)/6 symmetric
	int *mult = nullptr, *index = nullptr;
	uint32_t *keys = nullptr, *perm = nullptr;
};

struct nbco_ctx
{
	nbco_opts o;
	hipStream_t stream = nullptr;   // the stream work is enqueued on (the caller's, or `aux` inside a StreamScope)
	// second stream for the chains that do not depend on each other (multipoles || traversal, M2L lists || P2P lists);
	// created lazily, ordered against `stream` with events only
	hipStream_t aux = nullptr;
	hipEvent_t ev_fork = nullptr, ev_join = nullptr;
	bool aux_pending = false;
	// the kd evaluator may leave the tree-ordered velocities in tmp3 for the caller's next pass over v (leapfrog's closing
	// kick) instead of copying them back itself: requested with defer_v_copy, reported in v_deferred
	int list_growth = 1;   // the lists hold list_growth * opts.list_factor * nodes pairs (doubled after an overflow, opts.list_grow)
	bool grow_lists(int ntot)   // false: growth is off or exhausted
	{
		if (!o.list_grow || list_growth >= 64) return false;
		// 32-bit offsets: the DIRECTED P2P list holds up to 2 * cap + leaves entries, cap = 2 * (growth * factor * nodes + 4096),
		// and its prefix sums (start[], chunk_off[], the low word of the packed scan) are ints: the doubled capacity must keep
		// 2 * cap + nodes below 2^31
		if (8LL * list_growth * o.list_factor * (long long)ntot + 16384 + ntot >= (1LL << 31)) return false;
		list_growth *= 2;
		return true;
	}
	bool defer_v_copy = false;
	// nbco_integrate_steps: leave the re-ordering of the caller's state after a rebuild to the pass between two steps
	// (order_pending: such a re-ordering is due); skip_prep: that pass has already done the next build's prologue (1) or
	// packed the positions for a reused tree (2)
	bool defer_order = false, order_pending = false;
	int skip_prep = 0;
	const float *v_deferred = nullptr;
	long long perm_primed_n = -1;   // particle count for which both permutation buffers were last filled with valid indices
	double host_wait_s = 0, host_call_s = 0;   // diagnostics (NBCO_HOST_TIMING): time blocked on the flags event / inside nbco_integrate
	long long host_calls = 0;
	bool aux_is_main = false;   // NBCO_AUX_SERIAL=1 (diagnostics): the second stream is the main stream
	bool poison = false;        // NBCO_POISON=1 (diagnostics): every new scratch allocation is filled with 0x7f bytes, the way a recycled
	                            // allocation holds stale data: nothing may depend on fresh memory being zero
	// traversal counts / flags land in pinned host memory; looked at after the rest of the evaluation is enqueued
	int *h_flags = nullptr;
	hipEvent_t ev_flags = nullptr;
	long long hint_np2p = 0, hint_nm2l = 0;   // list sizes of the previous evaluation (launch-size hints only)
	int flags_begin();
	// local build of a kd-domain: the split axes of the domain root's ancestors in the GLOBAL tree (the top tree of the partition
	// step, device array) and the root's 1-based heap number there; nullptr / 1: the tree's root is the global root
	const int *top_sd = nullptr;
	int top_root1 = 1;
	int flags_seq = 0;   // sequence number of the traversal whose results h_flags[0..3] hold (h_flags[4], written last)
	int wait_flags();    // until the traversal enqueued last has reported: spins on the pinned word, no interrupt-driven wake-up
	DevBuf scan_tmp_aux;
	int fork_aux();   // aux waits for everything enqueued on `stream` so far
	int fork_mark();  // remember this point of `stream` ...
	int fork_wait();  // ... and let aux wait for it (called after more work has been enqueued on `stream`)
	int join_aux();   // `stream` waits for everything enqueued on aux
	std::string err;
	int device = 0;
	int num_cu = 256;

	// generic scratch
	DevBuf pos4;          // float4[n] packed positions (xyz, 0)
	DevBuf pos4_alt;      // second buffer for gathers
	DevBuf part;          // partial results of reductions / direct j-splits
	DevBuf small;         // a few hundred bytes of device scalars
	DevBuf tmp3, tmp3b;   // float[3n] scratch for gathers of xyz triplets (tmp3b: second velocity copy of nbco_integrate_steps)
	// kd-tree build
	DevBuf keys, keys_alt, idx, idx_alt, unsort, unsort_alt, sort_tmp;
	DevBuf treebuf;
	KdTreeDev kd;
	// traversal and interaction lists
	DevBuf frontier_a, frontier_b, p2p_list, m2l_list, counters;
	DevBuf p2p_keys, p2p_keys_alt, m2l_keys, m2l_keys_alt, p2p_start, m2l_start;
	DevBuf p2p_chunk_off, p2p_chunks, p2p_desc;
	DevBuf order, order_alt;     // opts.track_order: position in the state -> particle number of the state the tracking started from
	long long order_n = -1;
	DevBuf p2p_sec, p2p_react;   // mutual near field: per-target range of entries delivered by other waves, reaction records
	const int *pc_mult = nullptr, *pc_total = nullptr;   // inputs of the on-demand directed pair count
	int pc_shift = 0;
	DevBuf list_cnt, trav_ctr;
	DevBuf prep_state;   // min / max accumulators + completion counter of the build prologue kernel
	DevBuf sel_hist, sel_nodes, sel_ties;   // selection build (k_kdselect.hip)
	// multi-GPU kd-domain sharding: boxes / split axes of the global levels 0 .. d, the assembled global tree
	DevBuf dist_top, dist_tree;
	// LET exchange: need masks + per-receiver selections, arrival flags, the (sparse) global position array
	DevBuf let_sel, let_have, dist_pos;
	// octree-traceless evaluator (k_fmm_oct.hip)
	DevBuf oct_tree, oct_groups;
	OctTreeDev oct;
	struct DistState
	{
		int world = 0, rank = 0, d = 0, L = 0;
		long long n_global = 0, n_local = 0;
		bool partitioned = false, build_done = false, local_done = false, rebuilt = false, traversed = false, let_selected = false, let_packed = false;
		const void *pos_all = nullptr;   // gathered positions, between the two halves of the finish stage
		// LET exchange with capped segments (nbco_dist_let_pack_capped): the attempt in flight is one, and its build turned out flagged
		bool let_capped = false, let_flagged = false;
		// attempts finished so far; the guard of attempt k reports into word pair k & 1, and a pair whose attempt the caller
		// declared void (nbco_dist_let_settle(ok = 0)) is dropped unread
		long long let_epoch = 0;
		bool let_ignore[2] = {false, false};
	} dist;
	// warm select (k_kdselect.hip): one histogram pass per level around the previous build's pivots.  used: this build ran it;
	// a flagged build that used it is repeated cold before anything is escalated, three misses in a row switch it off
	bool sel_warm_enabled = true, sel_warm_used = false;
	// coarsen: bucket width x 4^coarsen after misses; recent / recent_miss: the current window of 32 warm builds; cooldown: cold builds left
	int sel_warm_coarsen = 0, sel_warm_good = 0, sel_warm_recent = 0, sel_warm_recent_miss = 0, sel_warm_cooldown = 0, sel_warm_cool_len = 128;
	long long sel_warm_builds = 0, sel_warm_misses = 0;
	// A miss costs a whole evaluation, a warm build saves ~5 % of one: it pays while fewer than one build in twenty misses.
	// Drift-type misses are answered with coarser buckets (a window 4x as wide per miss, three times); a node whose parent
	// changed its split axis cannot be predicted at all, and while the cloud changes shape that happens every few builds:
	// two misses within 32 warm builds send the warm select into a cool-down (128 builds, doubling up to 4096).
	void note_warm_miss()
	{
		++sel_warm_misses;
		sel_warm_good = 0;
		if (sel_warm_coarsen < 3) ++sel_warm_coarsen;
		if (++sel_warm_recent_miss >= 2)
		{
			sel_warm_cooldown = sel_warm_cool_len;
			sel_warm_cool_len = sel_warm_cool_len < 2048 ? 2 * sel_warm_cool_len : 4096;
			sel_warm_recent = sel_warm_recent_miss = 0;
		}
	}
	void note_warm_ok()
	{
		if (++sel_warm_recent >= 32) sel_warm_recent = sel_warm_recent_miss = 0;
		if (++sel_warm_good >= 64)   // a quiet stretch: back towards fine buckets and short cool-downs
		{
			if (sel_warm_coarsen > 0) --sel_warm_coarsen;
			sel_warm_cool_len = sel_warm_cool_len > 256 ? sel_warm_cool_len / 2 : 128;
			sel_warm_good = 0;
		}
	}
	bool sel_three_pass = false;            // set after the first tie / bucket overflow: three radix passes per select
	bool force_sort_build = false;          // set after the second: use the sorting build from then on
	bool escalate_build()                   // next more conservative build; false when there is none left
	{
		if (!sel_three_pass) { sel_three_pass = true; return true; }
		if (!force_sort_build) { force_sort_build = true; return true; }
		return false;
	}
	long long list_cap = 0;
	// distributed re-partition in progress (k_dpart.hip)
	struct DPart
	{
		int stage = 0, level = 0, pass = 0, world = 0, rank = 0, d = 0;
		long long n_global = 0, n_local = 0;
		float *state = nullptr;
		void *work = nullptr;
		size_t tie_bytes = 0;
		bool seg_flip = false, pos_flip = false;
		long long rows_send[64] = {}, rows_recv[64] = {};
	} dpart;
	DevBuf dpart_buf;
	// the last kd-tree evaluation, for nbco_energy_fmm (pointers into the context's buffers; valid while tree_valid)
	struct LastEval
	{
		bool valid = false, have_p2p = false;
		const float *center = nullptr, *mpole = nullptr;
		const float4 *csz = nullptr, *pos = nullptr;
		const int *mult = nullptr, *index = nullptr;
		int L = 0, ntot = 0, order = 0, shift = 0, real_bytes = 4;   // real_bytes: element size of mpole (8 after an evaluation with opts.far_fp64)
		long long n = 0, own0 = 0, own_n = 0;
	} last_eval;
	// bookkeeping of the last evaluation
	nbco_kd_info info{};
	long long eval_counter = 0;
	bool tree_valid = false;
	long long tree_n = 0;
	int tree_order = 0;
	// profiling
	unsigned profiling = 0;   // bit i set: record events around phase i
	PhaseTimer timers[NBCO_PH_COUNT];

	int fail(int code, const std::string &msg) { err = msg; return code; }
	int fail_hip(hipError_t e, const char *what, const char *file, int line)
	{
		char buf[512];
		snprintf(buf, sizeof buf, "HIP error: %s (%s) at %s:%d", hipGetErrorString(e), what, file, line);
		err = buf;
		return NBCO_ERR_HIP;
	}
	int reserve(DevBuf &b, size_t bytes);
	void phase_begin(int ph);
	void phase_end(int ph);
};

// enqueue on the auxiliary stream for the lifetime of the scope (every launcher reads c->stream at call time)
struct StreamScope
{
	nbco_ctx *c;
	hipStream_t saved;
	StreamScope(nbco_ctx *c_, hipStream_t s) : c(c_), saved(c_->stream) { c->stream = s; }
	~StreamScope() { c->stream = saved; }
};

struct PhaseScope
{
	nbco_ctx *c;
	int ph;
	PhaseScope(nbco_ctx *c_, int ph_) : c(c_), ph(ph_) { c->phase_begin(ph); }
	~PhaseScope() { c->phase_end(ph); }
};

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- launchers implemented in the kernel translation units -------------------------------------
// k_axpy.hip
int launch_step(nbco_ctx *c, float *b, const float *a, float ds, long long n3);
int launch_add_elastic(nbco_ctx *c, const float *p, float *a, long long n, const float *k, bool assign);
int launch_rescale(nbco_ctx *c, float *a, long long n3, const float *param);
int launch_gather3(nbco_ctx *c, float *dst, const float *src, const int *map, long long n, bool inverse);
int launch_copy(nbco_ctx *c, float *dst, const float *src, long long n3);
int launch_pack4(nbco_ctx *c, float4 *dst, const float *src3, long long n);
// fused integrator pieces
int launch_kick_drift(nbco_ctx *c, float *x, float *v, const float *a, float ks, float ds, long long n3);
int launch_finish_kick(nbco_ctx *c, const float *x, const float *v_in, float *v, float *a, const float *param, float ks, long long n, bool elastic);
// k_direct.hip
int launch_direct(nbco_ctx *c, const float *p, float *a, long long n, const float *param, bool kahan);
// k_reduce.hip
int launch_minmax(nbco_ctx *c, const float *p3, long long n, float *out6_dev);
int launch_minmax4(nbco_ctx *c, const float4 *p4, long long n, float *out6_dev);
int launch_mean_relerr(nbco_ctx *c, const float *x, const float *ref, long long n, float *out_host);
int launch_pow_sum(nbco_ctx *c, const float *x, int expo, long long n, double *out3_host);
int launch_energy(nbco_ctx *c, const float *buf, long long n, const float *param, double *out3_host);
// k_fmm_kd.hip
int fmm_kdtree_eval(nbco_ctx *c, float *p, float *a, long long n, const float *param);
int kd_copy_out(nbco_ctx *c, int which, void *host_dst, long long host_bytes);
int kd_count_pairs(nbco_ctx *c, long long *out);
int kd_energy_fmm(nbco_ctx *c, long long n_own, double *half_phi_sum);
int launch_energy_kin_ela(nbco_ctx *c, const float *buf, long long n, const float *param, double *out2_host);
// k_fmm_oct.hip
int fmm_oct_traceless_eval(nbco_ctx *c, float *p, float *a, long long n, const float *param, bool symmetric = false, int world = 1, int rank = 0,
                           long long *pbounds_host = nullptr);
int oct_copy_out(nbco_ctx *c, int which, void *host_dst, long long host_bytes);
// multi-GPU kd-domain sharding (k_fmm_kd.hip)
int kd_dist_layout(nbco_ctx *c, long long n_global, int world, int rank, nbco_dist_layout *out);
int kd_dist_partition(nbco_ctx *c, const float *state_all, long long n_global, int world, int rank, float *state_local);
int dpart_workspace(nbco_ctx *c, long long n_global, int world, long long *bytes);
int dpart_begin(nbco_ctx *c, float *state_local, long long n_global, int world, int rank, void *work, long long work_bytes, nbco_dist_step *out);
int dpart_next(nbco_ctx *c, nbco_dist_step *out);
int kd_finish_pending_order(nbco_ctx *c, float *p, long long n);
int kd_turnaround(nbco_ctx *c, float *p, const float *v_in, const float **v_now, const float *param, float ks, float ds, bool elastic, long long n,
                  const float *root6 = nullptr);
int kd_dist_turnaround(nbco_ctx *c, float *buf_local, long long n_local, const float *param, float ks, float ds, bool elastic);
int kd_dist_let_select(nbco_ctx *c, const void *csz_all, long long *counts);
int kd_dist_let_pack(nbco_ctx *c, const long long *counts_all, void *pos_send, void *mpole_send);
int kd_dist_let_finish(nbco_ctx *c, const long long *counts_all, const void *pos_recv, const void *mpole_recv, float *buf_local, float *a_local, const float *param);
int kd_dist_let_check(nbco_ctx *c);
int kd_dist_let_pack_capped(nbco_ctx *c, const long long *caps_out, void *pos_send, void *mpole_send);
int kd_dist_let_finish_capped(nbco_ctx *c, const long long *caps_in, const void *pos_recv, const void *mpole_recv, float *buf_local, float *a_local, const float *param);
int kd_dist_let_settle(nbco_ctx *c, int ok);
int kd_dist_local(nbco_ctx *c, float *buf_local, long long n_local, void *nodes_send, void *pos_send, void *csz_send = nullptr,
                  void *mpole_send = nullptr, int let_stage = 0);
int kd_dist_finish(nbco_ctx *c, const void *nodes_all, const void *pos_all, float *buf_local, float *a_local, const float *param);
int kd_dist_finish_traverse(nbco_ctx *c, const void *csz_all, const void *pos_all);
int kd_dist_finish_rest(nbco_ctx *c, const void *mpole_all, float *buf_local, float *a_local, const float *param);
// k_kdselect.hip
int kd_select_begin(nbco_ctx *c, int l0, bool zero = true, long long *words_a = nullptr, long long *words_b = nullptr);
int kd_select_level(nbco_ctx *c, int l, long long n, const float4 *pos_in, const int *unsort_in, float4 *pos_out, int *unsort_out,
                    float *lbound, float *rbound, int *splitdim, int *index, int *flag, bool warm = false);
// k_farfield.hip
int launch_upward_gen(nbco_ctx *c, int P, const float4 *pos, float *center, void *mpole, int *mult, const int *index, int L, int write_geom, int f64 = 0);
int launch_kd_centres(nbco_ctx *c, float *center, int *mult, int L, const float *lbound, const float *rbound, float4 *csz);
int launch_downward_gen(nbco_ctx *c, int P, const float *center, void *local, int L, int dom_d, int dom_g, int f64 = 0);
// (L: depth of the whole tree -- the high orders pick their loop form by a level's height above the leaves)
int launch_m2m_top_gen(nbco_ctx *c, int P, float *center, void *mpole, int *mult, int ltop, int L, int write_geom, int f64 = 0);
int launch_kd_centres_top(nbco_ctx *c, float *center, int *mult, int ltop, const float *lbound, const float *rbound, float4 *csz);
int launch_l2p_gen(nbco_ctx *c, int P, const float4 *pos, const float *center, const void *local, const float4 *near, const int *chunk_off,
                   const int *index, int mlt_max, const int *unsort, int scatter, const float *param, float *a, int have_near, long long n, int L,
                   long long own0, long long own_n, const int2 *sec_range = nullptr, const float4 *react = nullptr, long long react_cap = 0,
                   int react_stride = 32, int f64 = 0);
// k_m2l.hip
// mstride: reals per multipole tuple in `mpole` (0 = the offM(P) of the kd-tree layout; the symmetric octree evaluator keeps orders 0..P)
int launch_m2l_lanes(nbco_ctx *c, int P, const float4 *csz, const float *mpole, float *local, const uint64_t *keys, const int *start,
                     int shift, int ntot, int mstride = 0);
int launch_m2l_lanes_f64(nbco_ctx *c, int P, const float4 *csz, const double *mpole, double *local, const uint64_t *keys, const int *start,
                         int shift, int ntot, int mstride = 0);
