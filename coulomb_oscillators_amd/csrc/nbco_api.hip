// nbco_api.hip -- the C ABI of include/nbco.h: context, option handling, the evaluator dispatch
// of compute_force (integrator.cuh:22-28 over main3.cu:47-69) and the symplectic integrators
// (integrator.cuh:32-167).
#include "nbco_internal.hpp"
#include <chrono>
#include <cstdio>
#include <cmath>
#include <new>
#include <random>

int nbco_ctx::reserve(DevBuf &b, size_t bytes)
{
	nbco_ctx *c = this;
	if (bytes <= b.bytes) return NBCO_OK;
	if (b.ptr)
	{
		// earlier launches on the stream may still use the old allocation
		NBCO_HIP(hipStreamSynchronize(stream));
		NBCO_HIP(hipFree(b.ptr));
		b.ptr = nullptr;
		b.bytes = 0;
	}
	size_t want = bytes + bytes / 8 + 256;
	NBCO_HIP(hipMalloc(&b.ptr, want));
	b.bytes = want;
	if (poison) NBCO_HIP(hipMemsetAsync(b.ptr, 0x7f, want, stream));
	return NBCO_OK;
}

void nbco_ctx::phase_begin(int ph)
{
	if (!(profiling & (1u << ph))) return;
	hipEvent_t a, b;
	if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
	hipEventRecord(a, stream);
	timers[ph].pending.push_back({a, b});
}

void nbco_ctx::phase_end(int ph)
{
	if (!(profiling & (1u << ph)) || timers[ph].pending.empty()) return;
	hipEventRecord(timers[ph].pending.back().second, stream);
}

int nbco_ctx::fork_mark()
{
	if (!aux && !aux_is_main)
	{
		// high priority: its kernels are small, latency-bound links of the far-field chain and must not queue up behind
		// the thousands of workgroups of the near-field kernel running on the main stream
		int prio_lo = 0, prio_hi = 0;
		NBCO_HIP_M(this, hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
		if (getenv("NBCO_AUX_PRIO") && atoi(getenv("NBCO_AUX_PRIO")) == 0) prio_hi = prio_lo;   // A/B switch (diagnostics)
		if (getenv("NBCO_AUX_SERIAL") && atoi(getenv("NBCO_AUX_SERIAL")) != 0) { aux = stream; aux_is_main = true; }   // diagnostics: one stream
		else NBCO_HIP_M(this, hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, prio_hi));
		NBCO_HIP_M(this, hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
		NBCO_HIP_M(this, hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
	}
	NBCO_HIP_M(this, hipEventRecord(ev_fork, stream));
	return NBCO_OK;
}

int nbco_ctx::fork_wait()
{
	NBCO_HIP_M(this, hipStreamWaitEvent(aux, ev_fork, 0));
	aux_pending = true;
	return NBCO_OK;
}

int nbco_ctx::fork_aux()
{
	NBCO_TRY(fork_mark());
	return fork_wait();
}

int nbco_ctx::flags_begin()
{
	if (!h_flags)
	{
		NBCO_HIP_M(this, hipHostMalloc((void **)&h_flags, 64 * sizeof(int), hipHostMallocDefault));
		memset(h_flags, 0, 64 * sizeof(int));
		NBCO_HIP_M(this, hipEventCreateWithFlags(&ev_flags, hipEventDisableTiming));
	}
	return NBCO_OK;
}

// The host looks at the traversal's counts and flags once per evaluation, after everything else has been enqueued.  A blocking
// hipEventSynchronize wakes up tens of microseconds after the kernel has finished -- time the next step's launches start
// late by -- so the host polls the sequence word that traverse_finish_kernel stores last in pinned memory, and asks the
// event only now and then (a failed launch or a lost device must not leave it spinning).
int nbco_ctx::wait_flags()
{
	nbco_ctx *c = this;
	volatile int *seq = h_flags + 4;
	for (unsigned spin = 1;; ++spin)
	{
		if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == flags_seq) return NBCO_OK;
		if ((spin & 0x3FFu) == 0)
		{
			const hipError_t e = hipEventQuery(ev_flags);
			if (e == hipSuccess) { if (__atomic_load_n(seq, __ATOMIC_ACQUIRE) == flags_seq) return NBCO_OK; NBCO_HIP(hipEventSynchronize(ev_flags)); return NBCO_OK; }
			if (e != hipErrorNotReady) return fail_hip(e, "hipEventQuery(ev_flags)", __FILE__, __LINE__);
		}
#if defined(__x86_64__)
		__builtin_ia32_pause();
#endif
	}
}

int nbco_ctx::join_aux()
{
	if (!aux_pending) return NBCO_OK;
	NBCO_HIP_M(this, hipEventRecord(ev_join, aux));
	NBCO_HIP_M(this, hipStreamWaitEvent(stream, ev_join, 0));
	aux_pending = false;
	return NBCO_OK;
}

static int check_opts(nbco_ctx *c, const nbco_opts *o)
{
	if (o->fmm_order < 1 || o->fmm_order > kMaxOrder) return c->fail(NBCO_ERR_ARG, "fmm_order must be in 1..10");
	if (!(o->tree_radius > 0)) return c->fail(NBCO_ERR_ARG, "tree_radius must be positive");
	if (!(o->eps2 > 0)) return c->fail(NBCO_ERR_ARG, "eps2 must be positive");
	if (!(o->dens_inhom > 0)) return c->fail(NBCO_ERR_ARG, "dens_inhom must be positive");
	if (o->tree_L < 0 || o->tree_L > 30) return c->fail(NBCO_ERR_ARG, "tree_L must be in 0..30");
	if (o->tree_steps < 1) return c->fail(NBCO_ERR_ARG, "tree_steps must be >= 1");
	if (o->list_factor < 1) return c->fail(NBCO_ERR_ARG, "list_factor must be >= 1");
	return NBCO_OK;
}

extern "C" {

int nbco_opts_default(nbco_opts *o)
{
	if (!o) return NBCO_ERR_ARG;
	o->fmm_order = 3;        // constants.cuh:42
	o->tree_radius = 1.f;    // :43
	o->eps2 = 1.e-18f;       // :39
	o->coll = 1;             // :48
	o->unsort = 1;           // :48
	o->dens_inhom = 1.f;     // :50
	o->tree_L = 0;           // :44
	o->tree_steps = 1;
	o->m2l_first = 0;
	o->sync = 1;
	o->list_factor = 48;
	o->list_grow = 1;
	o->far_fp64 = 0;
	o->p2p_mutual = 0;
	o->track_order = 0;
	o->stream = nullptr;
	return NBCO_OK;
}

int nbco_create(nbco_ctx **out, const nbco_opts *o)
{
	if (!out) return NBCO_ERR_ARG;
	*out = nullptr;
	nbco_ctx *c = new (std::nothrow) nbco_ctx();
	if (!c) return NBCO_ERR_HIP;
	nbco_opts def;
	nbco_opts_default(&def);
	c->o = o ? *o : def;
	int rc = check_opts(c, &c->o);
	if (rc != NBCO_OK) { delete c; return rc; }
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev <= 0)
	{
		// no CPU fallback: the engine is HIP-only
		fprintf(stderr, "nbco_create: no usable HIP device (%s)\n", e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
		delete c;
		return NBCO_ERR_HIP;
	}
	if (hipGetDevice(&c->device) != hipSuccess) { delete c; return NBCO_ERR_HIP; }
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
	c->stream = (hipStream_t)c->o.stream;
	c->poison = getenv("NBCO_POISON") && atoi(getenv("NBCO_POISON")) != 0;
	if (getenv("NBCO_SEL_WARM")) c->sel_warm_enabled = atoi(getenv("NBCO_SEL_WARM")) != 0;
	if (hipMalloc(&c->small.ptr, 4096) != hipSuccess) { delete c; return NBCO_ERR_HIP; }
	c->small.bytes = 4096;
	*out = c;
	return NBCO_OK;
}

int nbco_destroy(nbco_ctx *c)
{
	if (c && getenv("NBCO_HOST_TIMING") && c->host_calls)
		fprintf(stderr, "[nbco] nbco_integrate: %lld calls, %.1f us host time per call, of which %.1f us waiting for the traversal flags\n", c->host_calls,
		        1e6 * c->host_call_s / c->host_calls, 1e6 * c->host_wait_s / c->host_calls);
	if (!c) return NBCO_OK;
	hipStreamSynchronize(c->stream);
	DevBuf *bufs[] = {&c->pos4, &c->pos4_alt, &c->part, &c->small, &c->tmp3, &c->keys, &c->keys_alt, &c->idx, &c->idx_alt,
	                  &c->unsort, &c->unsort_alt, &c->sort_tmp, &c->treebuf, &c->frontier_a, &c->frontier_b, &c->p2p_list,
	                  &c->m2l_list, &c->counters, &c->p2p_keys, &c->p2p_keys_alt, &c->m2l_keys, &c->m2l_keys_alt,
	                  &c->p2p_start, &c->m2l_start, &c->p2p_chunk_off, &c->p2p_chunks, &c->sel_hist, &c->sel_nodes, &c->sel_ties, &c->list_cnt,
	                  &c->dist_top, &c->dist_tree, &c->oct_tree, &c->oct_groups, &c->scan_tmp_aux, &c->p2p_desc, &c->trav_ctr, &c->prep_state, &c->p2p_sec, &c->p2p_react, &c->order, &c->order_alt,
	                  &c->let_sel, &c->let_have, &c->dist_pos, &c->tmp3b, &c->dpart_buf};
	if (c->aux && !c->aux_is_main) { hipStreamSynchronize(c->aux); hipStreamDestroy(c->aux); }
	if (c->ev_fork) hipEventDestroy(c->ev_fork);
	if (c->ev_join) hipEventDestroy(c->ev_join);
	if (c->ev_flags) hipEventDestroy(c->ev_flags);
	if (c->h_flags) hipHostFree(c->h_flags);
	for (DevBuf *b : bufs)
		if (b->ptr) hipFree(b->ptr);
	for (auto &t : c->timers)
		for (auto &ev : t.pending) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
	delete c;
	return NBCO_OK;
}

int nbco_set_opts(nbco_ctx *c, const nbco_opts *o)
{
	if (!c || !o) return NBCO_ERR_ARG;
	NBCO_TRY(check_opts(c, o));
	if ((hipStream_t)o->stream != c->stream) NBCO_HIP(hipStreamSynchronize(c->stream));
	if (o->list_factor != c->o.list_factor) c->list_growth = 1;
	bool topo = o->fmm_order != c->o.fmm_order || o->dens_inhom != c->o.dens_inhom || o->tree_L != c->o.tree_L
	            || o->unsort != c->o.unsort || o->p2p_mutual != c->o.p2p_mutual || o->track_order != c->o.track_order
	            || o->tree_steps != c->o.tree_steps;   // (a new rebuild schedule starts with a rebuild)
	c->o = *o;
	c->stream = (hipStream_t)o->stream;
	if (topo) { c->tree_valid = false; c->eval_counter = 0; }
	if (!o->track_order) c->order_n = -1;
	return NBCO_OK;
}

int nbco_get_opts(const nbco_ctx *c, nbco_opts *o)
{
	if (!c || !o) return NBCO_ERR_ARG;
	*o = c->o;
	return NBCO_OK;
}

const char *nbco_last_error(const nbco_ctx *c) { return c ? c->err.c_str() : "null context"; }

int nbco_sync(nbco_ctx *c)
{
	if (!c) return NBCO_ERR_ARG;
	NBCO_HIP(hipStreamSynchronize(c->stream));
	return NBCO_OK;
}

static int maybe_sync(nbco_ctx *c)
{
	if (c->o.sync) NBCO_HIP(hipStreamSynchronize(c->stream));
	return NBCO_OK;
}

// ---- basic kernels -----------------------------------------------------------------------------
int nbco_step(nbco_ctx *c, float *b, const float *a, float ds, long long n)
{
	if (!c || !b || !a || n < 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_step: bad arguments") : NBCO_ERR_ARG;
	PhaseScope ph(c, NBCO_PH_AXPY);
	return launch_step(c, b, a, ds, 3 * n);
}
int nbco_add_elastic(nbco_ctx *c, const float *p, float *a, long long n, const float *k)
{
	if (!c || !p || !a || n < 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_add_elastic: bad arguments") : NBCO_ERR_ARG;
	PhaseScope ph(c, NBCO_PH_AXPY);
	return launch_add_elastic(c, p, a, n, k, false);
}
int nbco_elastic(nbco_ctx *c, const float *p, float *a, long long n, const float *k)
{
	if (!c || !p || !a || n < 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_elastic: bad arguments") : NBCO_ERR_ARG;
	PhaseScope ph(c, NBCO_PH_AXPY);
	return launch_add_elastic(c, p, a, n, k, true);
}
int nbco_rescale(nbco_ctx *c, float *a, long long n, const float *param)
{
	if (!c || !a || !param || n < 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_rescale: bad arguments") : NBCO_ERR_ARG;
	PhaseScope ph(c, NBCO_PH_AXPY);
	return launch_rescale(c, a, 3 * n, param);
}
int nbco_gather(nbco_ctx *c, float *dst, const float *src, const int *map, long long n)
{
	if (!c || !dst || !src || !map || n < 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_gather: bad arguments") : NBCO_ERR_ARG;
	return launch_gather3(c, dst, src, map, n, false);
}
int nbco_gather_inverse(nbco_ctx *c, float *dst, const float *src, const int *map, long long n)
{
	if (!c || !dst || !src || !map || n < 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_gather_inverse: bad arguments") : NBCO_ERR_ARG;
	return launch_gather3(c, dst, src, map, n, true);
}
int nbco_copy(nbco_ctx *c, float *dst, const float *src, long long n)
{
	if (!c || !dst || !src || n < 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_copy: bad arguments") : NBCO_ERR_ARG;
	return launch_copy(c, dst, src, 3 * n);
}

// ---- evaluators --------------------------------------------------------------------------------
int nbco_direct(nbco_ctx *c, const float *p, float *a, long long n, const float *param)
{
	if (!c || !p || !a) return c ? c->fail(NBCO_ERR_ARG, "nbco_direct: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(launch_direct(c, p, a, n, param, false));
	return maybe_sync(c);
}
int nbco_direct3(nbco_ctx *c, const float *p, float *a, long long n, const float *param)
{
	if (!c || !p || !a) return c ? c->fail(NBCO_ERR_ARG, "nbco_direct3: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(launch_direct(c, p, a, n, param, true));
	return maybe_sync(c);
}
int nbco_fmm_kdtree(nbco_ctx *c, float *p, float *a, long long n, const float *param)
{
	if (!c || !p || !a) return c ? c->fail(NBCO_ERR_ARG, "nbco_fmm_kdtree: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(fmm_kdtree_eval(c, p, a, n, param));
	return maybe_sync(c);
}
int nbco_fmm_traceless(nbco_ctx *c, float *p, float *a, long long n, const float *param)
{
	if (!c || !p || !a) return c ? c->fail(NBCO_ERR_ARG, "nbco_fmm_traceless: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(fmm_oct_traceless_eval(c, p, a, n, param));
	return maybe_sync(c);
}
int nbco_fmm_symmetric(nbco_ctx *c, float *p, float *a, long long n, const float *param)
{
	if (!c || !p || !a) return c ? c->fail(NBCO_ERR_ARG, "nbco_fmm_symmetric: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(fmm_oct_traceless_eval(c, p, a, n, param, true));
	return maybe_sync(c);
}
int nbco_oct_get_info(nbco_ctx *c, nbco_oct_info *out)
{
	if (!c || !out) return c ? c->fail(NBCO_ERR_ARG, "nbco_oct_get_info: null pointer") : NBCO_ERR_ARG;
	if (!c->oct.valid) return c->fail(NBCO_ERR_ARG, "nbco_oct_get_info: no octree evaluation has run");
	const OctTreeDev &o = c->oct;
	out->L = o.L; out->ntot = o.ntot; out->order = o.order; out->tpl = o.tpl; out->n = o.n;
	out->m2l_entries = o.m2l_entries; out->p2p_groups = o.p2p_groups; out->p2p_desc = o.p2p_desc; out->p2p_chunks = o.p2p_chunks;
	out->real_bytes = o.real_bytes;
	out->mpole_reals = o.mpole_reals;
	return NBCO_OK;
}
int nbco_oct_copy(nbco_ctx *c, int which, void *host_dst, long long host_bytes)
{
	if (!c || !host_dst) return c ? c->fail(NBCO_ERR_ARG, "nbco_oct_copy: null pointer") : NBCO_ERR_ARG;
	return oct_copy_out(c, which, host_dst, host_bytes);
}

int nbco_dist_layout_query(nbco_ctx *c, long long n_global, int world, int rank, nbco_dist_layout *out)
{
	if (!c || !out) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_layout_query: null pointer") : NBCO_ERR_ARG;
	return kd_dist_layout(c, n_global, world, rank, out);
}
int nbco_dist_partition(nbco_ctx *c, const float *state_all, long long n_global, int world, int rank, float *state_local)
{
	if (!c || !state_all || !state_local) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_partition: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_partition(c, state_all, n_global, world, rank, state_local));
	return maybe_sync(c);
}
int nbco_dist_local(nbco_ctx *c, float *buf_local, long long n_local, void *nodes_send, void *pos_send)
{
	if (!c || !buf_local || !nodes_send || !pos_send) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_local: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_local(c, buf_local, n_local, nodes_send, pos_send));
	return maybe_sync(c);
}
int nbco_dist_local_build(nbco_ctx *c, float *buf_local, long long n_local, void *pos_send)
{
	if (!c || !buf_local || !pos_send) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_local_build: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_local(c, buf_local, n_local, nullptr, pos_send));
	return maybe_sync(c);
}
int nbco_dist_local_upward(nbco_ctx *c, float *buf_local, long long n_local, void *nodes_send)
{
	if (!c || !buf_local || !nodes_send) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_local_upward: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_local(c, buf_local, n_local, nodes_send, nullptr));
	return maybe_sync(c);
}
int nbco_dist_local_geom(nbco_ctx *c, float *buf_local, long long n_local, void *pos_send, void *csz_send)
{
	if (!c || !buf_local || !pos_send || !csz_send) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_local_geom: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_local(c, buf_local, n_local, nullptr, pos_send, csz_send, nullptr));
	return maybe_sync(c);
}
int nbco_dist_local_mpole(nbco_ctx *c, float *buf_local, long long n_local, void *mpole_send)
{
	if (!c || !buf_local || !mpole_send) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_local_mpole: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_local(c, buf_local, n_local, nullptr, nullptr, nullptr, mpole_send));
	return maybe_sync(c);
}
int nbco_dist_let_local_geom(nbco_ctx *c, float *buf_local, long long n_local, void *csz_send)
{
	if (!c || !buf_local || !csz_send) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_let_local_geom: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_local(c, buf_local, n_local, nullptr, nullptr, csz_send, nullptr, 1));
	return maybe_sync(c);
}
int nbco_dist_let_local_mpole(nbco_ctx *c, float *buf_local, long long n_local)
{
	if (!c || !buf_local) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_let_local_mpole: null pointer") : NBCO_ERR_ARG;
	return kd_dist_local(c, buf_local, n_local, nullptr, nullptr, nullptr, nullptr, 2);
}
int nbco_dist_let_select(nbco_ctx *c, const void *csz_all, long long *counts_send)
{
	if (!c || !csz_all || !counts_send) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_let_select: null pointer") : NBCO_ERR_ARG;
	return kd_dist_let_select(c, csz_all, counts_send);
}
int nbco_dist_let_pack(nbco_ctx *c, const long long *counts_all, void *pos_send, void *mpole_send)
{
	if (!c || !counts_all) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_let_pack: null pointer") : NBCO_ERR_ARG;
	return kd_dist_let_pack(c, counts_all, pos_send, mpole_send);
}
int nbco_dist_let_finish(nbco_ctx *c, const long long *counts_all, const void *pos_recv, const void *mpole_recv, float *buf_local, float *a_local,
                         const float *param)
{
	if (!c || !counts_all || !buf_local || !a_local) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_let_finish: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_let_finish(c, counts_all, pos_recv, mpole_recv, buf_local, a_local, param));
	return maybe_sync(c);
}
int nbco_dist_turnaround(nbco_ctx *c, float *buf_local, long long n_local, const float *param, double dt_, double scale_, int elastic)
{
	if (!c || !buf_local || !param) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_turnaround: null pointer") : NBCO_ERR_ARG;
	const long double dt = dt_, scale = scale_;
	return kd_dist_turnaround(c, buf_local, n_local, param, (float)(dt * scale * 0.5L), (float)dt, elastic != 0);
}
int nbco_dist_let_check(nbco_ctx *c)
{
	if (!c) return NBCO_ERR_ARG;
	return kd_dist_let_check(c);
}
int nbco_dist_let_pack_capped(nbco_ctx *c, const long long *caps_out, void *pos_send, void *mpole_send)
{
	if (!c || !caps_out) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_let_pack_capped: null pointer") : NBCO_ERR_ARG;
	return kd_dist_let_pack_capped(c, caps_out, pos_send, mpole_send);
}
int nbco_dist_let_finish_capped(nbco_ctx *c, const long long *caps_in, const void *pos_recv, const void *mpole_recv, float *buf_local, float *a_local,
                                const float *param)
{
	if (!c || !caps_in || !buf_local || !a_local) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_let_finish_capped: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_let_finish_capped(c, caps_in, pos_recv, mpole_recv, buf_local, a_local, param));
	return maybe_sync(c);
}
int nbco_dist_let_settle(nbco_ctx *c, int ok)
{
	if (!c) return NBCO_ERR_ARG;
	return kd_dist_let_settle(c, ok);
}
int nbco_dist_repartition_workspace(nbco_ctx *c, long long n_global, int world, long long *bytes)
{
	if (!c || !bytes) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_repartition_workspace: null pointer") : NBCO_ERR_ARG;
	return dpart_workspace(c, n_global, world, bytes);
}
int nbco_dist_repartition_begin(nbco_ctx *c, float *state_local, long long n_global, int world, int rank, void *work, long long work_bytes, nbco_dist_step *next)
{
	if (!c || !state_local || !work || !next) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_repartition_begin: null pointer") : NBCO_ERR_ARG;
	return dpart_begin(c, state_local, n_global, world, rank, work, work_bytes, next);
}
int nbco_dist_repartition_next(nbco_ctx *c, nbco_dist_step *next)
{
	if (!c || !next) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_repartition_next: null pointer") : NBCO_ERR_ARG;
	return dpart_next(c, next);
}
int nbco_dist_finish_traverse(nbco_ctx *c, const void *csz_all, const void *pos_all)
{
	if (!c || !csz_all || !pos_all) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_finish_traverse: null pointer") : NBCO_ERR_ARG;
	return kd_dist_finish_traverse(c, csz_all, pos_all);   // (never syncs: its point is to leave the stream busy)
}
int nbco_dist_finish_rest(nbco_ctx *c, const void *mpole_all, float *buf_local, float *a_local, const float *param)
{
	if (!c || !mpole_all || !buf_local || !a_local) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_finish_rest: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_finish_rest(c, mpole_all, buf_local, a_local, param));
	return maybe_sync(c);
}
int nbco_aux_stream(nbco_ctx *c, void **stream_out)
{
	if (!c || !stream_out) return c ? c->fail(NBCO_ERR_ARG, "nbco_aux_stream: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(c->fork_mark());   // creates the stream on first use (the event it records is harmless)
	*stream_out = (void *)c->aux;
	return NBCO_OK;
}
int nbco_dist_finish(nbco_ctx *c, const void *nodes_all, const void *pos_all, float *buf_local, float *a_local, const float *param)
{
	if (!c || !nodes_all || !pos_all || !buf_local || !a_local) return c ? c->fail(NBCO_ERR_ARG, "nbco_dist_finish: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(kd_dist_finish(c, nodes_all, pos_all, buf_local, a_local, param));
	return maybe_sync(c);
}

static int eval_kind(nbco_ctx *c, int kind, float *p, float *a, long long n, const float *param)
{
	switch (kind)
	{
	case NBCO_EVAL_DIRECT: return launch_direct(c, p, a, n, param, false);
	case NBCO_EVAL_DIRECT_KAHAN: return launch_direct(c, p, a, n, param, true);
	case NBCO_EVAL_FMM_KDTREE: return fmm_kdtree_eval(c, p, a, n, param);
	case NBCO_EVAL_FMM_TRACELESS: return fmm_oct_traceless_eval(c, p, a, n, param, false);
	case NBCO_EVAL_FMM_SYMMETRIC: return fmm_oct_traceless_eval(c, p, a, n, param, true);
	default: return c->fail(NBCO_ERR_ARG, "unknown evaluator kind");
	}
}

// the uniform-octree evaluators on `world` GPUs: every rank holds the whole state and builds the whole tree, and evaluates the
// accelerations of its slab of the cell order only (k_fmm_oct.hip, oct_slab_kernel)
int nbco_fmm_oct_shard(nbco_ctx *c, float *p, float *a, long long n, const float *param, int symmetric, int world, int rank, long long *bounds_host)
{
	if (!c || !p || !a || !bounds_host) return c ? c->fail(NBCO_ERR_ARG, "nbco_fmm_oct_shard: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(fmm_oct_traceless_eval(c, p, a, n, param, symmetric != 0, world, rank, bounds_host));
	return maybe_sync(c);
}

int nbco_force(nbco_ctx *c, int kind, float *buf, long long n, const float *param, int elastic)
{
	if (!c || !buf || !param || n <= 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_force: bad arguments") : NBCO_ERR_ARG;
	float *x = buf, *a = buf + 6 * n;
	NBCO_TRY(eval_kind(c, kind, x, a, n, param));
	if (elastic)
	{
		PhaseScope ph(c, NBCO_PH_AXPY);
		NBCO_TRY(launch_add_elastic(c, x, a, n, param + 3, false));
	}
	return maybe_sync(c);
}

// integrator.cuh:32-167.  K(s): v += a*s, D(s): x += v*s, F: a = f(x) (+ elastic term).  The step
// coefficients are formed in long double and narrowed to float at the step call, as the
// reference does at its step_func call sites.
int nbco_integrate(nbco_ctx *c, int scheme, int kind, float *buf, long long n, const float *param, double dt_, double scale_,
                   int elastic)
{
	if (!c || !buf || !param || n <= 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_integrate: bad arguments") : NBCO_ERR_ARG;
	struct HostTimer
	{
		nbco_ctx *c;
		std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
		~HostTimer() { c->host_call_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++c->host_calls; }
	} host_timer{c};
	float *x = buf, *v = buf + 3 * n, *a = buf + 6 * n;
	const long long n3 = 3 * n;
	const long double dt = dt_, scale = scale_;
	auto K = [&](long double s) { PhaseScope ph(c, NBCO_PH_AXPY); return launch_step(c, v, a, (float)s, n3); };
	auto D = [&](long double s) { PhaseScope ph(c, NBCO_PH_AXPY); return launch_step(c, x, v, (float)s, n3); };
	auto F = [&]() {
		NBCO_TRY(eval_kind(c, kind, x, a, n, param));
		if (elastic)
		{
			PhaseScope ph(c, NBCO_PH_AXPY);
			NBCO_TRY(launch_add_elastic(c, x, a, n, param + 3, false));
		}
		return (int)NBCO_OK;
	};
	const long double th = 1.3512071919596576340476878089715L;   // integrator.cuh:98
	const long double xi = +0.1786178958448091E+00L, la = -0.2123418310626054E+00L, ch = -0.6626458266981849E-01L;  // :130-132
	switch (scheme)
	{
	case NBCO_INTEG_EULER:
		NBCO_TRY(K(dt * scale)); NBCO_TRY(D(dt)); NBCO_TRY(F());
		break;
	case NBCO_INTEG_PRE_EULER:
		NBCO_TRY(F()); NBCO_TRY(K(dt * scale)); NBCO_TRY(D(dt));
		break;
	case NBCO_INTEG_LEAPFROG:
	{
		long double ds = dt * scale * 0.5L;
		{
			// K(ds) D(dt) fused into one pass over x, v, a
			PhaseScope ph(c, NBCO_PH_AXPY);
			NBCO_TRY(launch_kick_drift(c, x, v, a, (float)ds, (float)dt, n3));
		}
		// F K(ds): the elastic term and the kick share one pass over x, v, a -- and when the evaluator has just re-ordered
		// the particles, the kick reads the tree-ordered velocities straight from the evaluator's scratch copy
		c->defer_v_copy = true;
		c->v_deferred = nullptr;
		const int rc = eval_kind(c, kind, x, a, n, param);
		c->defer_v_copy = false;
		const float *v_in = c->v_deferred ? c->v_deferred : v;
		c->v_deferred = nullptr;
		if (rc != NBCO_OK)
		{
			if (v_in != v) hipMemcpyAsync(v, v_in, sizeof(float) * (size_t)n3, hipMemcpyDeviceToDevice, c->stream);   // leave a whole state behind
			return rc;
		}
		{
			PhaseScope ph(c, NBCO_PH_AXPY);
			NBCO_TRY(launch_finish_kick(c, x, v_in, v, a, param, (float)ds, n, elastic != 0));
		}
		break;
	}
	case NBCO_INTEG_FORESTRUTH:
	{
		long double ds = dt * scale;
		NBCO_TRY(D(dt * th / 2)); NBCO_TRY(F());
		NBCO_TRY(K(ds * th)); NBCO_TRY(D(dt * (1 - th) / 2)); NBCO_TRY(F());
		NBCO_TRY(K(ds * (1 - 2 * th))); NBCO_TRY(D(dt * (1 - th) / 2)); NBCO_TRY(F());
		NBCO_TRY(K(ds * th)); NBCO_TRY(D(dt * th / 2));
		break;
	}
	case NBCO_INTEG_PEFRL:
	{
		long double ds = dt * scale;
		NBCO_TRY(D(dt * xi)); NBCO_TRY(F());
		NBCO_TRY(K(ds * (1 - 2 * la) / 2)); NBCO_TRY(D(dt * ch)); NBCO_TRY(F());
		NBCO_TRY(K(ds * la)); NBCO_TRY(D(dt * (1 - 2 * (ch + xi)))); NBCO_TRY(F());
		NBCO_TRY(K(ds * la)); NBCO_TRY(D(dt * ch)); NBCO_TRY(F());
		NBCO_TRY(K(ds * (1 - 2 * la) / 2)); NBCO_TRY(D(dt * xi));
		break;
	}
	default:
		return c->fail(NBCO_ERR_ARG, "unknown integrator scheme");
	}
	return maybe_sync(c);
}

// `steps` steps of the scheme in one call.  Leapfrog over the kd-tree evaluator (tree order kept, opts.unsort = 0) fuses what lies
// between two force evaluations -- tree order for x and v, elastic term, the two half kicks, the drift and the next build's
// prologue: four passes over the state -- into one (kd_turnaround); every particle sees the same operations with the same
// roundings as in `steps` calls of nbco_integrate, and the final state is bit-identical to theirs.  Everything else loops.
int nbco_integrate_steps(nbco_ctx *c, int scheme, int kind, float *buf, long long n, const float *param, double dt_, double scale_, int elastic,
                         int steps)
{
	if (!c || !buf || !param || n <= 0 || steps < 0) return c ? c->fail(NBCO_ERR_ARG, "nbco_integrate_steps: bad arguments") : NBCO_ERR_ARG;
	const bool fuse = scheme == NBCO_INTEG_LEAPFROG && kind == NBCO_EVAL_FMM_KDTREE && !c->o.unsort && !c->o.track_order && steps >= 2;
	if (!fuse)
	{
		const int sync = c->o.sync;
		c->o.sync = 0;
		int rc = NBCO_OK;
		for (int s = 0; s < steps && rc == NBCO_OK; ++s) rc = nbco_integrate(c, scheme, kind, buf, n, param, dt_, scale_, elastic);
		c->o.sync = sync;
		if (rc != NBCO_OK) return rc;
		return maybe_sync(c);
	}
	float *x = buf, *v = buf + 3 * n, *a = buf + 6 * n;
	const long long n3 = 3 * n;
	const long double dt = dt_, scale = scale_;
	const float ds = (float)(dt * scale * 0.5L), dtf = (float)dt;
	{
		PhaseScope ph(c, NBCO_PH_AXPY);
		NBCO_TRY(launch_kick_drift(c, x, v, a, ds, dtf, n3));
	}
	const float *v_now = v;
	auto home = [&]() {   // the velocities back into the caller's array
		if (v_now != v) hipMemcpyAsync(v, v_now, sizeof(float) * (size_t)n3, hipMemcpyDeviceToDevice, c->stream);
		v_now = v;
	};
	c->defer_order = true;
	for (int s = 0; s < steps; ++s)
	{
		int rc = eval_kind(c, kind, x, a, n, param);
		if (rc == NBCO_OK && s + 1 < steps) rc = kd_turnaround(c, buf, v_now, &v_now, param, ds, dtf, elastic != 0, n);
		if (rc != NBCO_OK)
		{
			c->defer_order = false;
			home();
			kd_finish_pending_order(c, x, n);   // leave a whole state behind
			return rc;
		}
	}
	c->defer_order = false;
	home();
	// tail of the last step, as in nbco_integrate: tree order, then a -= k o x, v += a ds
	c->defer_v_copy = true;
	c->v_deferred = nullptr;
	const int rc = kd_finish_pending_order(c, x, n);
	c->defer_v_copy = false;
	const float *v_in = c->v_deferred ? c->v_deferred : v;
	c->v_deferred = nullptr;
	if (rc != NBCO_OK) return rc;
	{
		PhaseScope ph(c, NBCO_PH_AXPY);
		NBCO_TRY(launch_finish_kick(c, x, v_in, v, a, param, ds, n, elastic != 0));
	}
	return maybe_sync(c);
}

// ---- reductions --------------------------------------------------------------------------------
int nbco_minmax(nbco_ctx *c, const float *p, long long n, float *minmax6_dev)
{
	if (!c || !p || !minmax6_dev) return c ? c->fail(NBCO_ERR_ARG, "nbco_minmax: null pointer") : NBCO_ERR_ARG;
	NBCO_TRY(launch_minmax(c, p, n, minmax6_dev));
	return maybe_sync(c);
}
int nbco_mean_relerr(nbco_ctx *c, const float *x, const float *ref, long long n, float *out_host)
{
	if (!c || !x || !ref || !out_host) return c ? c->fail(NBCO_ERR_ARG, "nbco_mean_relerr: null pointer") : NBCO_ERR_ARG;
	return launch_mean_relerr(c, x, ref, n, out_host);
}
int nbco_pow_sum(nbco_ctx *c, const float *x, int expo, long long n, double *out3_host)
{
	if (!c || !x || !out3_host) return c ? c->fail(NBCO_ERR_ARG, "nbco_pow_sum: null pointer") : NBCO_ERR_ARG;
	return launch_pow_sum(c, x, expo, n, out3_host);
}
int nbco_energy(nbco_ctx *c, const float *buf, long long n, const float *param, double *out3_host)
{
	if (!c || !buf || !param || !out3_host) return c ? c->fail(NBCO_ERR_ARG, "nbco_energy: null pointer") : NBCO_ERR_ARG;
	return launch_energy(c, buf, n, param, out3_host);
}

int nbco_energy_fmm(nbco_ctx *c, const float *buf, long long n, const float *param, double *out3_host)
{
	if (!c || !buf || !param || !out3_host) return c ? c->fail(NBCO_ERR_ARG, "nbco_energy_fmm: null pointer") : NBCO_ERR_ARG;
	double ke[2], half_phi = 0;
	NBCO_TRY(launch_energy_kin_ela(c, buf, n, param, ke));
	NBCO_TRY(kd_energy_fmm(c, n, &half_phi));
	float p0;
	NBCO_HIP(hipMemcpy(&p0, param, sizeof(float), hipMemcpyDeviceToHost));
	out3_host[0] = ke[0]; out3_host[1] = ke[1]; out3_host[2] = (double)p0 * half_phi;
	return NBCO_OK;
}

// ---- introspection -----------------------------------------------------------------------------
int nbco_kd_get_info(nbco_ctx *c, nbco_kd_info *info)
{
	if (!c || !info) return NBCO_ERR_ARG;
	if (c->info.directed_p2p < 0 && c->counters.ptr && c->tree_valid)
	{
		long long v = 0;
		NBCO_TRY(kd_count_pairs(c, &v));   // one small kernel over the sorted P2P list of the last evaluation
		c->info.directed_p2p = v;
	}
	*info = c->info;
	info->build_mode = c->force_sort_build ? 2 : (c->sel_three_pass ? 1 : 0);
	info->warm_builds = c->sel_warm_builds; info->warm_misses = c->sel_warm_misses;
	info->real_bytes = c->kd.real_bytes;
	return NBCO_OK;
}
int nbco_kd_copy(nbco_ctx *c, int which, void *host_dst, long long host_bytes)
{
	if (!c || !host_dst) return NBCO_ERR_ARG;
	return kd_copy_out(c, which, host_dst, host_bytes);
}

// ---- checked build -----------------------------------------------------------------------------
int nbco_debug_violations(nbco_ctx *c, long long *out8)
{
	if (!c || !out8) return c ? c->fail(NBCO_ERR_ARG, "nbco_debug_violations: null pointer") : NBCO_ERR_ARG;
	for (int i = 0; i < NBCO_CHK_SITES; ++i) out8[i] = 0;
#ifdef NBCO_CHECKED
	NBCO_HIP(hipDeviceSynchronize());
	unsigned v[NBCO_CHK_SITES] = {};
	NBCO_TRY(nbco_checked_collect_kd(v));
	NBCO_TRY(nbco_checked_collect_oct(v));
	NBCO_TRY(nbco_checked_collect_far(v));
	for (int i = 0; i < NBCO_CHK_SITES; ++i) out8[i] = v[i];
	return NBCO_OK;
#else
	return c->fail(NBCO_ERR_UNSUPPORTED, "nbco_debug_violations: this is not the checked build (libnbco_hip_checked.so)");
#endif
}

// ---- profiling ---------------------------------------------------------------------------------
int nbco_profile_enable(nbco_ctx *c, int on)
{
	if (!c) return NBCO_ERR_ARG;
	c->profiling = on < 0 ? 0xFFFFFFFFu : (unsigned)on;
	return NBCO_OK;
}
static int drain_timers(nbco_ctx *c)
{
	NBCO_HIP(hipStreamSynchronize(c->stream));
	for (auto &t : c->timers)
	{
		for (auto &ev : t.pending)
		{
			float ms = 0.f;
			if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { t.total_ms += ms; t.launches += 1; }
			hipEventDestroy(ev.first);
			hipEventDestroy(ev.second);
		}
		t.pending.clear();
	}
	return NBCO_OK;
}
int nbco_profile_reset(nbco_ctx *c)
{
	if (!c) return NBCO_ERR_ARG;
	NBCO_TRY(drain_timers(c));
	for (auto &t : c->timers) { t.total_ms = 0; t.launches = 0; }
	return NBCO_OK;
}
int nbco_profile_get(nbco_ctx *c, int phase, double *total_ms, long long *launches)
{
	if (!c || phase < 0 || phase >= NBCO_PH_COUNT) return NBCO_ERR_ARG;
	NBCO_TRY(drain_timers(c));
	if (total_ms) *total_ms = c->timers[phase].total_ms;
	if (launches) *launches = c->timers[phase].launches;
	return NBCO_OK;
}

// ---- initial condition: initGA / initU of main3.cu:71-137 over the reference's generator (host only) --------------------------
namespace {
struct V3 { float x, y, z; };
void centre_dist(V3 *d, long long n)   // main3.cu:71-80 (fp32 running sums, as there)
{
	V3 c{0, 0, 0};
	for (long long i = 0; i < n; ++i) { c.x += d[i].x; c.y += d[i].y; c.z += d[i].z; }
	c.x /= (float)n; c.y /= (float)n; c.z /= (float)n;
	for (long long i = 0; i < n; ++i) { d[i].x -= c.x; d[i].y -= c.y; d[i].z -= c.z; }
}
void adjust_rms(V3 *d, long long n, V3 adj)   // main3.cu:82-92
{
	V3 s{0, 0, 0};
	for (long long i = 0; i < n; ++i) { s.x += d[i].x * d[i].x; s.y += d[i].y * d[i].y; s.z += d[i].z * d[i].z; }
	s.x = std::sqrt(s.x / (float)n); s.y = std::sqrt(s.y / (float)n); s.z = std::sqrt(s.z / (float)n);
	const V3 f{adj.x / s.x, adj.y / s.y, adj.z / s.z};   // `data[i] *= adj / d`: the quotient first, then one product
	for (long long i = 0; i < n; ++i) { d[i].x *= f.x; d[i].y *= f.y; d[i].z *= f.z; }
}
} // namespace

int nbco_init_gaussian(float *host_state, long long n, const float *sx, const float *su, unsigned long long seed, unsigned long long discard,
                       int uniform_positions)
{
	if (!host_state || !sx || !su || n <= 0) return NBCO_ERR_ARG;
	std::mt19937_64 gen(seed);
	gen.discard(discard);
	std::normal_distribution<float> dist(0.f, 1.f);
	for (long long i = 0; i < 6 * n; ++i) host_state[i] = dist(gen);   // all position deviates, then all velocity deviates (:121-123)
	V3 *pos = reinterpret_cast<V3 *>(host_state), *vel = pos + n;
	const V3 x{sx[0], sx[1], sx[2]}, u{su[0], su[1], su[2]};
	for (long long i = 0; i < n; ++i) { pos[i].x *= x.x; pos[i].y *= x.y; pos[i].z *= x.z; }
	for (long long i = 0; i < n; ++i) { vel[i].x *= u.x; vel[i].y *= u.y; vel[i].z *= u.z; }
	centre_dist(pos, n); adjust_rms(pos, n, x);
	centre_dist(vel, n); adjust_rms(vel, n, u);
	if (uniform_positions)   // initU with a = -1, b = 1 (main3.cu:94-111)
	{
		std::uniform_real_distribution<float> dx(-1, 1), dy(-1, 1), dz(-1, 1);
		for (long long i = 0; i < n; ++i) { pos[i].x = dx(gen); pos[i].y = dy(gen); pos[i].z = dz(gen); }
		centre_dist(pos, n);
	}
	return NBCO_OK;
}

// The rows [first, first + count) of the state nbco_init_gaussian(n, ..) would produce, without ever holding more than the slice:
// the generator is run through the whole stream twice (the centring and the RMS rescaling need sums over ALL n particles, taken
// in the same order and precision as there), so the result equals the corresponding rows of the full state bit for bit.
int nbco_init_gaussian_slice(float *host_slice, long long n, long long first, long long count, const float *sx, const float *su,
                             unsigned long long seed, unsigned long long discard, int uniform_positions)
{
	if (!host_slice || !sx || !su || n <= 0 || first < 0 || count <= 0 || first + count > n) return NBCO_ERR_ARG;
	V3 *pos = reinterpret_cast<V3 *>(host_slice), *vel = pos + count;
	const V3 sc[2] = {{sx[0], sx[1], sx[2]}, {su[0], su[1], su[2]}};
	V3 mean[2], fac[2];
	// pass 1: the scaled deviates of the slice, and the running sums of centre_dist
	{
		std::mt19937_64 gen(seed);
		gen.discard(discard);
		std::normal_distribution<float> dist(0.f, 1.f);
		for (int part = 0; part < 2; ++part)
		{
			V3 *out = part ? vel : pos;
			V3 c{0, 0, 0};
			for (long long i = 0; i < n; ++i)
			{
				V3 v;
				v.x = dist(gen); v.y = dist(gen); v.z = dist(gen);
				v.x *= sc[part].x; v.y *= sc[part].y; v.z *= sc[part].z;
				c.x += v.x; c.y += v.y; c.z += v.z;
				if (i >= first && i < first + count) out[i - first] = v;
			}
			mean[part] = V3{c.x / (float)n, c.y / (float)n, c.z / (float)n};
		}
	}
	// pass 2: the sums of squares of the centred values (adjust_rms)
	{
		std::mt19937_64 gen(seed);
		gen.discard(discard);
		std::normal_distribution<float> dist(0.f, 1.f);
		for (int part = 0; part < 2; ++part)
		{
			V3 s{0, 0, 0};
			for (long long i = 0; i < n; ++i)
			{
				V3 v;
				v.x = dist(gen); v.y = dist(gen); v.z = dist(gen);
				v.x *= sc[part].x; v.y *= sc[part].y; v.z *= sc[part].z;
				v.x -= mean[part].x; v.y -= mean[part].y; v.z -= mean[part].z;
				s.x += v.x * v.x; s.y += v.y * v.y; s.z += v.z * v.z;
			}
			s.x = std::sqrt(s.x / (float)n); s.y = std::sqrt(s.y / (float)n); s.z = std::sqrt(s.z / (float)n);
			fac[part] = V3{sc[part].x / s.x, sc[part].y / s.y, sc[part].z / s.z};
		}
		for (int part = 0; part < 2; ++part)
		{
			V3 *out = part ? vel : pos;
			for (long long i = 0; i < count; ++i)
			{
				out[i].x -= mean[part].x; out[i].y -= mean[part].y; out[i].z -= mean[part].z;
				out[i].x *= fac[part].x; out[i].y *= fac[part].y; out[i].z *= fac[part].z;
			}
		}
		if (uniform_positions)   // initU: the generator goes on behind the 6 n normal deviates
		{
			std::uniform_real_distribution<float> dx(-1, 1), dy(-1, 1), dz(-1, 1);
			V3 c{0, 0, 0};
			for (long long i = 0; i < n; ++i)
			{
				V3 v;
				v.x = dx(gen); v.y = dy(gen); v.z = dz(gen);
				c.x += v.x; c.y += v.y; c.z += v.z;
				if (i >= first && i < first + count) pos[i - first] = v;
			}
			c.x /= (float)n; c.y /= (float)n; c.z /= (float)n;
			for (long long i = 0; i < count; ++i) { pos[i].x -= c.x; pos[i].y -= c.y; pos[i].z -= c.z; }
		}
	}
	return NBCO_OK;
}

} // extern "C"
