// nbco3_dist.cpp -- C++20 multi-GPU host over the C ABI (include/nbco.h, section "multi-GPU") and RCCL: the kd-tree FMM
// simulation loop of nbco3 (main3.cu:832-874) with the particles sharded by kd-domain over the GPUs of one node, one process
// per GPU.  The reference is single-GPU (fmm_cart3_kdtree.cuh:1529 hard-codes device 0); what is kept from it is the command
// line, the initial state and the snapshot format -- a snapshot of a G-GPU run is the same [pos | vel] file, in the tree order
// of the global kd-tree (rank r's particles are rows [r N/G, (r+1) N/G)).
//
//   nbco3_dist -gpus G [-n N] [-p order] [-ds dt] [-iters n] [-steps n] [-r radius] [-i dens] [-rebalance k] [-tree-steps k]
//              [-o folder] [-exchange let|let-exact|gather] [-partition dist|gather]
//
// The launcher process is a SUPERVISOR, not a rank: it forks all G ranks before anything touches the GPU, never makes a HIP or
// RCCL call itself, waits for the ranks, and when one of them exits with a failure (or is killed) it kills the others -- which
// would otherwise block in their next collective for ever -- and returns non-zero.  Nothing is restarted or re-exec'ed.  Rank
// 0's ncclUniqueId reaches the other ranks through pipes.  Between two snapshots the leapfrog steps run with ONE pass over the
// domain's state between two force evaluations (nbco_dist_turnaround), as nbco3 does on one GPU (nbco_integrate_steps).  Per evaluation: subtree build -> all-gather of positions and traversal records ->
// multipoles -> all-gather of the multipoles, on a communication stream of their own, under the traversal -> lists, near and
// far field, L2P.  Every `rebalance` evaluations the domains are cut again from the gathered state (nbco_dist_partition).
// Defaults since round 2 (INTEGRATION.md sections 4a, 4b): the locally-essential-tree exchange -- all-gather of the traversal
// records only, then grouped ncclSend / ncclRecv of exactly the multipoles and positions the other ranks' lists name -- and the
// re-partition without gathering the state (nbco_dist_repartition_*: the library names a collective, this host runs it).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/nbco.h"

namespace {

#define HIPCHK(call)                                                                                                          \
	do {                                                                                                                      \
		hipError_t e_ = (call);                                                                                               \
		if (e_ != hipSuccess) { std::cerr << "GPUassert: " << hipGetErrorString(e_) << ' ' << __FILE__ << ' ' << __LINE__ << std::endl; std::exit(2); } \
	} while (0)
#define NCCLCHK(call)                                                                                                         \
	do {                                                                                                                      \
		ncclResult_t r_ = (call);                                                                                             \
		if (r_ != ncclSuccess) { std::cerr << "RCCL error: " << ncclGetErrorString(r_) << ' ' << __FILE__ << ' ' << __LINE__ << std::endl; std::exit(3); } \
	} while (0)

struct Args
{
	int gpus = 1, n = 1 << 20, order = 3, iters = 30001, steps = 200, rebalance = 16, tree_steps = 1;
	int fail_rank = -1, fail_iter = 0;   // test hook (NBCO3_DIST_FAIL=rank:iteration): that rank exits with status 9 before that iteration
	int hang_rank = -1;                  // test hook (NBCO3_DIST_HANG=rank): that rank sleeps for ever instead of running
	bool let = true, capped = true, dist_partition = true;
	float dt = 5.e-4f, radius = 1.f, dens = 1.f, xi = 2.e-6f;
	std::string out = "out";
};

struct Rank
{
	int rank, world;
	nbco_ctx *ctx = nullptr;
	ncclComm_t comm{};
	hipStream_t comm_stream{};
	hipEvent_t ev_ready{}, ev_geom{}, ev_mpole{};
	nbco_dist_layout lay{};
	float *buf = nullptr, *state_all = nullptr, *par = nullptr;
	char *pos_send = nullptr, *nodes_send = nullptr, *pos_all = nullptr, *nodes_all = nullptr;
	long long evals = 0;
	int partition_fallbacks = 0;
	bool let = true, dist_partition = true;
	// LET exchange: count blocks (device + host), record buffers sized for the worst case (everything needed by everyone)
	long long *counts_send = nullptr, *counts_all_dev = nullptr;
	long long *counts_all = nullptr;   // pinned host memory, world x let_counts
	// capped form (nbco_dist_let_pack_capped): segments sized from the count matrix of the evaluation before, no host
	// synchronisation in the middle of the evaluation; `capped` off = always the exact form (-exchange let-exact)
	bool capped = true, have_prev = false;
	std::vector<long long> prev_counts;
	hipEvent_t ev_counts{};
	long long capped_evals = 0, let_redos = 0;
	char *let_pos_send = nullptr, *let_mp_send = nullptr, *let_pos_recv = nullptr, *let_mp_recv = nullptr;
	// re-partition without gathering the state: the library's workspace
	char *work = nullptr;
	long long work_bytes = 0;

	void check(int rc, const char *what)
	{
		if (rc != NBCO_OK) { std::cerr << "rank " << rank << ": " << what << ": " << nbco_last_error(ctx) << std::endl; std::exit(4); }
	}

	// all-gather on the communication stream, ordered behind everything enqueued on the compute (null) stream so far
	void gather(const void *send, void *recv, size_t bytes, hipEvent_t done)
	{
		HIPCHK(hipEventRecord(ev_ready, nullptr));
		HIPCHK(hipStreamWaitEvent(comm_stream, ev_ready, 0));
		NCCLCHK(ncclAllGather(send, recv, bytes, ncclChar, comm, comm_stream));
		HIPCHK(hipEventRecord(done, comm_stream));
	}

	// grouped point-to-point exchange: rows_send[r] records of `bytes` each go to rank r from consecutive segments of `send`,
	// rows_recv[s] arrive from rank s into consecutive segments of `recv` (the all-to-all with uneven splits of torch.distributed)
	void all_to_all(const char *send, char *recv, const long long *rows_send, const long long *rows_recv, size_t bytes, hipStream_t st)
	{
		NCCLCHK(ncclGroupStart());
		size_t so = 0, ro = 0;
		for (int r = 0; r < world; ++r)
		{
			if (rows_send[r] > 0) NCCLCHK(ncclSend(send + so, (size_t)rows_send[r] * bytes, ncclChar, r, comm, st));
			if (rows_recv[r] > 0) NCCLCHK(ncclRecv(recv + ro, (size_t)rows_recv[r] * bytes, ncclChar, r, comm, st));
			so += (size_t)rows_send[r] * bytes; ro += (size_t)rows_recv[r] * bytes;
		}
		NCCLCHK(ncclGroupEnd());
	}

	// nbco_dist_repartition_*: every collective the library names runs on the compute stream, in order.  Returns false when the
	// library reports more pivot ties on one rank than the distributed select resolves (NBCO_ERR_UNSUPPORTED: lattice, planar or
	// duplicated coordinates).  The tie counts are gathered, so every rank gets the same answer at the same stage, and the local
	// state has not been touched at that point.
	bool repartition()
	{
		// test hook: with one rank no pivot is ever selected, so the library cannot be made to report ties; NBCO3_DIST_FAKE_TIES=1
		// answers the first cut as the library would (the library's own report is covered by tests/test_gpu_dist.py, in lockstep)
		static bool faked = false;
		if (!faked && getenv("NBCO3_DIST_FAKE_TIES")) { faked = true; return false; }
		nbco_dist_step st{};
		check(nbco_dist_repartition_begin(ctx, buf, lay.n_global, world, rank, work, work_bytes, &st), "nbco_dist_repartition_begin");
		while (st.op != NBCO_COLL_DONE)
		{
			switch (st.op)
			{
			case NBCO_COLL_ALLREDUCE_MIN_I32: NCCLCHK(ncclAllReduce(work + st.send_off, work + st.send_off, (size_t)st.count, ncclInt32, ncclMin, comm, nullptr)); break;
			case NBCO_COLL_ALLREDUCE_SUM_I32: NCCLCHK(ncclAllReduce(work + st.send_off, work + st.send_off, (size_t)st.count, ncclInt32, ncclSum, comm, nullptr)); break;
			case NBCO_COLL_ALLGATHER: NCCLCHK(ncclAllGather(work + st.send_off, work + st.recv_off, (size_t)st.count, ncclChar, comm, nullptr)); break;
			case NBCO_COLL_ALLTOALL: all_to_all(work + st.send_off, work + st.recv_off, st.rows_send, st.rows_recv, (size_t)st.row_bytes, nullptr); break;
			default: std::cerr << "rank " << rank << ": unknown collective " << st.op << std::endl; std::exit(4);
			}
			const int rc = nbco_dist_repartition_next(ctx, &st);
			if (rc == NBCO_ERR_UNSUPPORTED) return false;
			check(rc, "nbco_dist_repartition_next");
		}
		evals = 0;
		return true;
	}

	// segment sizes of the capped exchange, from the count matrix of the evaluation before (a quarter of head room plus a
	// constant, never more than a domain holds): records rank s sizes for rank r.  Same matrix, same table on every rank.
	long long cap_nodes(int s, int r) const
	{
		if (s == r) return 0;
		const long long c = prev_counts[(size_t)s * lay.let_counts + 2 * r];
		return std::min<long long>(c + c / 4 + 64, lay.ntot_local);
	}
	long long cap_parts(int s, int r) const
	{
		if (s == r) return 0;
		const long long c = prev_counts[(size_t)s * lay.let_counts + 2 * r + 1];
		return std::min<long long>(c + c / 4 + 512, lay.n_local);
	}
	void keep_counts() { prev_counts.assign(counts_all, counts_all + (size_t)lay.let_counts * world); have_prev = true; }

	// one attempt in the capped form; false = void, the caller repeats the evaluation in the exact form
	bool force_let_capped()
	{
		const long long nl = lay.n_local;
		const int S = lay.let_counts;
		char *csz_send = nodes_send, *csz_all = nodes_all;
		std::vector<long long> caps_out(2 * world), caps_in(2 * world), ps(world), pr(world), ms(world), mr(world);
		for (int r = 0; r < world; ++r)
		{
			caps_out[2 * r] = ms[r] = cap_nodes(rank, r); caps_out[2 * r + 1] = ps[r] = cap_parts(rank, r);
			caps_in[2 * r] = mr[r] = cap_nodes(r, rank); caps_in[2 * r + 1] = pr[r] = cap_parts(r, rank);
		}
		check(nbco_dist_let_local_geom(ctx, buf, nl, csz_send), "nbco_dist_let_local_geom");
		NCCLCHK(ncclAllGather(csz_send, csz_all, (size_t)lay.csz_bytes, ncclChar, comm, nullptr));
		check(nbco_dist_let_local_mpole(ctx, buf, nl), "nbco_dist_let_local_mpole");
		check(nbco_dist_let_select(ctx, csz_all, counts_send), "nbco_dist_let_select");
		NCCLCHK(ncclAllGather(counts_send, counts_all_dev, (size_t)S, ncclInt64, comm, nullptr));
		HIPCHK(hipMemcpyAsync(counts_all, counts_all_dev, sizeof(long long) * (size_t)S * world, hipMemcpyDeviceToHost, nullptr));
		HIPCHK(hipEventRecord(ev_counts, nullptr));
		check(nbco_dist_let_pack_capped(ctx, caps_out.data(), let_pos_send, let_mp_send), "nbco_dist_let_pack_capped");
		all_to_all(let_pos_send, let_pos_recv, ps.data(), pr.data(), 16, nullptr);
		all_to_all(let_mp_send, let_mp_recv, ms.data(), mr.data(), (size_t)lay.let_node_bytes, nullptr);
		check(nbco_dist_let_finish_capped(ctx, caps_in.data(), let_pos_recv, let_mp_recv, buf, buf + 6 * nl, par), "nbco_dist_let_finish_capped");
		// the whole evaluation is queued: only now look at the counts
		HIPCHK(hipEventSynchronize(ev_counts));
		bool ok = true;
		for (int s = 0; s < world && ok; ++s)
		{
			ok = counts_all[(size_t)s * S + 2 * world] == 0 && counts_all[(size_t)s * S + 2 * world + 1] == 0;
			for (int r = 0; r < world && ok; ++r)
				ok = s == r || (counts_all[(size_t)s * S + 2 * r] <= cap_nodes(s, r) && counts_all[(size_t)s * S + 2 * r + 1] <= cap_parts(s, r));
		}
		if (getenv("NBCO3_DIST_VOID") && capped_evals + let_redos == atoll(getenv("NBCO3_DIST_VOID"))) ok = false;   // test hook: this attempt is void
		check(nbco_dist_let_settle(ctx, ok ? 1 : 0), "nbco_dist_let_settle");
		if (!ok) { ++let_redos; return false; }
		keep_counts();
		++capped_evals;
		return true;
	}

	// one force evaluation with the LET exchange
	void force_let(bool elastic)
	{
		const long long nl = lay.n_local;
		const int S = lay.let_counts;
		char *csz_send = nodes_send, *csz_all = nodes_all;
		if (capped && have_prev && force_let_capped())
		{
			if (elastic) check(nbco_add_elastic(ctx, buf, buf + 6 * nl, nl, par + 3), "nbco_add_elastic");
			++evals;
			return;
		}
		for (int attempt = 0;; ++attempt)
		{
		check(nbco_dist_let_local_geom(ctx, buf, nl, csz_send), "nbco_dist_let_local_geom");
		NCCLCHK(ncclAllGather(csz_send, csz_all, (size_t)lay.csz_bytes, ncclChar, comm, nullptr));
		check(nbco_dist_let_local_mpole(ctx, buf, nl), "nbco_dist_let_local_mpole");
		for (int round = 0;; ++round)
		{
			check(nbco_dist_let_select(ctx, csz_all, counts_send), "nbco_dist_let_select");
			NCCLCHK(ncclAllGather(counts_send, counts_all_dev, (size_t)S, ncclInt64, comm, nullptr));
			HIPCHK(hipMemcpyAsync(counts_all, counts_all_dev, sizeof(long long) * (size_t)S * world, hipMemcpyDeviceToHost, nullptr));
			HIPCHK(hipStreamSynchronize(nullptr));   // the evaluation's one host synchronisation
			bool overflow = false;
			for (int s = 0; s < world; ++s) overflow = overflow || counts_all[(size_t)s * S + 2 * world] != 0;
			if (!overflow) break;
			if (round == 8) { std::cerr << "rank " << rank << ": the traversal lists keep overflowing" << std::endl; std::exit(4); }
		}
		// a flagged tree build somewhere (its flag travels with the counts: no host round trip behind the build): everybody starts over
		bool flagged = false;
		for (int s = 0; s < world; ++s) flagged = flagged || counts_all[(size_t)s * S + 2 * world + 1] != 0;
		if (!flagged) break;
		if (attempt == 5) { std::cerr << "rank " << rank << ": a tree build keeps being flagged" << std::endl; std::exit(4); }
		}
		check(nbco_dist_let_pack(ctx, counts_all, let_pos_send, let_mp_send), "nbco_dist_let_pack");
		std::vector<long long> ps(world), pr(world), ms(world), mr(world);
		for (int r = 0; r < world; ++r)
		{
			ms[r] = counts_all[(size_t)rank * S + 2 * r]; ps[r] = counts_all[(size_t)rank * S + 2 * r + 1];
			mr[r] = counts_all[(size_t)r * S + 2 * rank]; pr[r] = counts_all[(size_t)r * S + 2 * rank + 1];
		}
		all_to_all(let_pos_send, let_pos_recv, ps.data(), pr.data(), 16, nullptr);
		all_to_all(let_mp_send, let_mp_recv, ms.data(), mr.data(), (size_t)lay.let_node_bytes, nullptr);
		check(nbco_dist_let_finish(ctx, counts_all, let_pos_recv, let_mp_recv, buf, buf + 6 * nl, par), "nbco_dist_let_finish");
		keep_counts();
		if (elastic) check(nbco_add_elastic(ctx, buf, buf + 6 * nl, nl, par + 3), "nbco_add_elastic");
		++evals;
	}

	void partition()
	{
		if (dist_partition)
		{
			if (repartition()) return;
			// pivot ties beyond the distributed select, on every rank alike: from here on the gathered form, which takes any input
			if (rank == 0) std::cerr << "nbco3_dist: distributed re-partition not possible (" << nbco_last_error(ctx) << "): switching to -partition gather" << std::endl;
			dist_partition = false;
			++partition_fallbacks;
		}
		const long long nl = lay.n_local, N = lay.n_global;
		if (!state_all) HIPCHK(hipMalloc((void **)&state_all, sizeof(float) * 6 * (size_t)N));
		gather(buf, state_all, sizeof(float) * 3 * nl, ev_geom);                   // positions
		gather(buf + 3 * nl, state_all + 3 * N, sizeof(float) * 3 * nl, ev_geom);   // velocities
		HIPCHK(hipStreamWaitEvent(nullptr, ev_geom, 0));
		check(nbco_dist_partition(ctx, state_all, N, world, rank, buf), "nbco_dist_partition");
		evals = 0;
	}

	void force(int rebalance, bool elastic)
	{
		if (rebalance > 0 && evals >= rebalance) partition();
		if (let) { force_let(elastic); return; }
		const long long nl = lay.n_local;
		char *csz_send = nodes_send, *mp_send = nodes_send + lay.csz_bytes;
		char *csz_all = nodes_all, *mp_all = nodes_all + (size_t)world * lay.csz_bytes;
		check(nbco_dist_local_geom(ctx, buf, nl, pos_send, csz_send), "nbco_dist_local_geom");
		gather(pos_send, pos_all, (size_t)lay.pos_bytes, ev_geom);
		gather(csz_send, csz_all, (size_t)lay.csz_bytes, ev_geom);
		check(nbco_dist_local_mpole(ctx, buf, nl, mp_send), "nbco_dist_local_mpole");
		gather(mp_send, mp_all, (size_t)lay.mpole_bytes, ev_mpole);
		HIPCHK(hipStreamWaitEvent(nullptr, ev_geom, 0));
		check(nbco_dist_finish_traverse(ctx, csz_all, pos_all), "nbco_dist_finish_traverse");
		// only the context's second stream reads the gathered multipoles
		void *aux = nullptr;
		check(nbco_aux_stream(ctx, &aux), "nbco_aux_stream");
		HIPCHK(hipStreamWaitEvent((hipStream_t)aux, ev_mpole, 0));
		check(nbco_dist_finish_rest(ctx, mp_all, buf, buf + 6 * nl, par), "nbco_dist_finish_rest");
		if (elastic) check(nbco_add_elastic(ctx, buf, buf + 6 * nl, nl, par + 3), "nbco_add_elastic");
		++evals;
	}

	// `steps` leapfrog steps (integrator.cuh:68-96).  Between two force evaluations ONE pass over the domain's state does the
	// elastic term, the half kick that ends a step, the half kick and the drift that begin the next, and the next local build's
	// prologue (nbco_dist_turnaround; bit-identical to nbco_add_elastic + three nbco_step calls).
	void leapfrog_steps(float dt, int rebalance, int steps)
	{
		const long long nl = lay.n_local;
		if (steps <= 0) return;
		check(nbco_step(ctx, buf + 3 * nl, buf + 6 * nl, 0.5f * dt, nl), "step");
		check(nbco_step(ctx, buf, buf + 3 * nl, dt, nl), "step");
		for (int s = 0; s < steps; ++s)
		{
			force(rebalance, false);
			if (s + 1 < steps) check(nbco_dist_turnaround(ctx, buf, nl, par, (double)dt, 1.0, 1), "nbco_dist_turnaround");
		}
		check(nbco_add_elastic(ctx, buf, buf + 6 * nl, nl, par + 3), "nbco_add_elastic");
		check(nbco_step(ctx, buf + 3 * nl, buf + 6 * nl, 0.5f * dt, nl), "step");
	}
};

int run_rank(const Args &a, int rank, const ncclUniqueId &id)
{
	int ndev = 0;
	HIPCHK(hipGetDeviceCount(&ndev));
	if (ndev < a.gpus) { std::cerr << "Error: " << a.gpus << " GPUs requested, " << ndev << " visible" << std::endl; return -1; }
	HIPCHK(hipSetDevice(rank));
	Rank r{rank, a.gpus};
	NCCLCHK(ncclCommInitRank(&r.comm, a.gpus, id, rank));
	HIPCHK(hipStreamCreateWithFlags(&r.comm_stream, hipStreamNonBlocking));
	HIPCHK(hipEventCreateWithFlags(&r.ev_ready, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&r.ev_geom, hipEventDisableTiming));
	HIPCHK(hipEventCreateWithFlags(&r.ev_mpole, hipEventDisableTiming));

	nbco_opts o;
	nbco_opts_default(&o);
	o.fmm_order = a.order; o.tree_radius = a.radius; o.dens_inhom = a.dens;
	o.unsort = 0; o.sync = 0; o.tree_steps = a.tree_steps;
	if (nbco_create(&r.ctx, &o) != NBCO_OK) { std::cerr << "nbco_create failed" << std::endl; return -1; }
	r.check(nbco_dist_layout_query(r.ctx, a.n, a.gpus, rank, &r.lay), "nbco_dist_layout_query");
	const long long N = a.n, nl = r.lay.n_local;

	// every rank draws ITS rows of the one initial state (the reference's stream; rank r starts out owning rows [r nl, (r + 1) nl)):
	// two passes through the generator, never more than the slice in memory (nbco_init_gaussian_slice)
	std::vector<float> host(6 * (size_t)nl);
	const float sx[3] = {0.003f, 0.001f, 0.01f}, su[3] = {1.095f * 0.003f, 0.001f, 0.01f};
	if (nbco_init_gaussian_slice(host.data(), N, (long long)rank * nl, nl, sx, su, NBCO_REF_SEED, NBCO_REF_DISCARD, 0) != NBCO_OK) return -1;
	if (const char *q = getenv("NBCO3_DIST_QUANTISE"))   // test hook: positions on a lattice (many exact ties with every pivot)
	{
		const float step = (float)atof(q);
		if (step > 0) for (size_t i = 0; i < 3 * (size_t)nl; ++i) host[i] = step * std::nearbyint(host[i] / step);
	}
	const float parh[6] = {a.xi / (float)N, 0, 0, 1.095f * 1.095f, 1.f, 1.f};
	HIPCHK(hipMalloc((void **)&r.buf, sizeof(float) * 9 * (size_t)nl));
	r.let = a.let; r.dist_partition = a.dist_partition && a.gpus <= 32;
	if (r.dist_partition)   // (the gathered form allocates its 24 N bytes on first use)
	{
		r.check(nbco_dist_repartition_workspace(r.ctx, N, a.gpus, &r.work_bytes), "nbco_dist_repartition_workspace");
		HIPCHK(hipMalloc((void **)&r.work, (size_t)r.work_bytes));
	}
	if (r.let)
	{
		const size_t S = (size_t)r.lay.let_counts, others = (size_t)(a.gpus > 1 ? a.gpus - 1 : 1);
		HIPCHK(hipMalloc((void **)&r.counts_send, sizeof(long long) * S));
		HIPCHK(hipMalloc((void **)&r.counts_all_dev, sizeof(long long) * S * a.gpus));
		HIPCHK(hipHostMalloc((void **)&r.counts_all, sizeof(long long) * S * a.gpus, hipHostMallocDefault));
		HIPCHK(hipEventCreateWithFlags(&r.ev_counts, hipEventDisableTiming));
		r.capped = a.capped;
		// worst case: every other rank needs every particle and every node of this one / this one needs all of theirs
		HIPCHK(hipMalloc((void **)&r.let_pos_send, 16 * (size_t)nl * others));
		HIPCHK(hipMalloc((void **)&r.let_pos_recv, 16 * (size_t)nl * others));
		HIPCHK(hipMalloc((void **)&r.let_mp_send, (size_t)r.lay.let_node_bytes * r.lay.ntot_local * others));
		HIPCHK(hipMalloc((void **)&r.let_mp_recv, (size_t)r.lay.let_node_bytes * r.lay.ntot_local * others));
	}
	HIPCHK(hipMalloc((void **)&r.par, sizeof parh));
	HIPCHK(hipMalloc((void **)&r.pos_send, (size_t)r.lay.pos_bytes));
	HIPCHK(hipMalloc((void **)&r.nodes_send, (size_t)r.lay.nodes_bytes));
	HIPCHK(hipMalloc((void **)&r.pos_all, (size_t)r.lay.pos_bytes * a.gpus));
	HIPCHK(hipMalloc((void **)&r.nodes_all, (size_t)r.lay.nodes_bytes * a.gpus));
	HIPCHK(hipMemset(r.buf, 0, sizeof(float) * 9 * (size_t)nl));
	HIPCHK(hipMemcpy(r.buf, host.data(), sizeof(float) * 6 * nl, hipMemcpyHostToDevice));   // [pos nl | vel nl]
	host = std::vector<float>();
	HIPCHK(hipMemcpy(r.par, parh, sizeof parh, hipMemcpyHostToDevice));

	r.partition();
	r.force(0, true);   // precompute accelerations (main3.cu:836-839)
	HIPCHK(hipDeviceSynchronize());
	std::vector<float> slice(6 * (size_t)nl);
	int *flag = nullptr;
	HIPCHK(hipMalloc((void **)&flag, sizeof(int)));
	auto loop_t0 = std::chrono::steady_clock::now();
	int loop_first = 0;
	// snapshots follow the iterations 0, steps, 2 steps, ..: the iterations in between run as one leapfrog_steps call (main3.cu:840-870)
	for (int iter = 0; iter < a.iters;)
	{
		const int run = iter % a.steps == 0 ? 1 : std::min(a.steps - iter % a.steps, a.iters - iter);
		if (rank == a.fail_rank && a.fail_iter >= iter && a.fail_iter < iter + run)
		{
			std::cerr << "rank " << rank << ": NBCO3_DIST_FAIL requested a failure before iteration " << a.fail_iter << std::endl;
			std::_Exit(9);
		}
		r.leapfrog_steps(a.dt, a.rebalance, run);
		iter += run;
		if ((iter - 1) % a.steps != 0) continue;
		const int snap = iter - 1;
		// every rank writes its rows of the one snapshot file (rank 0 creates it first; the barrier is a tiny all-reduce)
		HIPCHK(hipDeviceSynchronize());
		HIPCHK(hipMemcpy(slice.data(), r.buf, sizeof(float) * 6 * nl, hipMemcpyDeviceToHost));
		const std::string name = a.out + "/out" + std::to_string(snap) + '_' + std::to_string(a.dt) + ".bin";
		if (rank == 0)
		{
			std::cout << snap << ' ' << std::flush;
			FILE *f = std::fopen(name.c_str(), "wb");
			if (!f) { std::cerr << "Error: cannot write on output location. Check that \"" << a.out << "\" folder exists. Create it if not." << std::endl; return -1; }
			std::fclose(f);
		}
		HIPCHK(hipMemset(flag, 0, sizeof(int)));
		NCCLCHK(ncclAllReduce(flag, flag, 1, ncclInt, ncclSum, r.comm, r.comm_stream));
		HIPCHK(hipStreamSynchronize(r.comm_stream));
		FILE *f = std::fopen(name.c_str(), "r+b");
		if (!f) { std::cerr << "rank " << rank << ": cannot open " << name << std::endl; return -1; }
		std::fseek(f, (long)(sizeof(float) * 3 * (size_t)rank * nl), SEEK_SET);
		std::fwrite(slice.data(), sizeof(float), 3 * (size_t)nl, f);
		std::fseek(f, (long)(sizeof(float) * 3 * ((size_t)N + (size_t)rank * nl)), SEEK_SET);
		std::fwrite(slice.data() + 3 * (size_t)nl, sizeof(float), 3 * (size_t)nl, f);
		std::fclose(f);
		if (snap == 0) { loop_t0 = std::chrono::steady_clock::now(); loop_first = 1; }   // the timer below starts behind the first snapshot
	}
	HIPCHK(hipDeviceSynchronize());
	// all ranks have finished their loops when this all-reduce returns: rank 0's clock then covers the slowest rank
	HIPCHK(hipMemset(flag, 0, sizeof(int)));
	NCCLCHK(ncclAllReduce(flag, flag, 1, ncclInt, ncclSum, r.comm, r.comm_stream));
	HIPCHK(hipStreamSynchronize(r.comm_stream));
	HIPCHK(hipFree(flag));
	if (rank == 0)
	{
		std::cout << std::endl;
		// (not in the reference: wall time of the integration loop behind the first snapshot -- what bench.py's `cli_dist` leg reads)
		std::cout << "Loop time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - loop_t0).count() << " s, " << a.iters - loop_first
		          << " iterations, " << a.gpus << " ranks, partition fallbacks " << r.partition_fallbacks << ", capped evaluations " << r.capped_evals
		          << ", repeated " << r.let_redos << std::endl;
	}
	nbco_destroy(r.ctx);
	ncclCommDestroy(r.comm);
	return 0;
}

} // namespace

int main(int argc, char **argv)
{
	Args a;
	for (int i = 1; i < argc; ++i)
	{
		const std::string f(argv[i]);
		auto val = [&]() -> const char * { if (i + 1 >= argc) { std::cerr << "Error: missing argument to '" << f << "'\n"; std::exit(-1); } return argv[++i]; };
		if (f == "-gpus") a.gpus = atoi(val());
		else if (f == "-n") a.n = atoi(val());
		else if (f == "-p") a.order = atoi(val());
		else if (f == "-ds") a.dt = (float)atof(val());
		else if (f == "-iters") a.iters = atoi(val()) + 1;   // main3.cu:357
		else if (f == "-steps") a.steps = atoi(val());
		else if (f == "-r") a.radius = (float)atof(val());
		else if (f == "-i") a.dens = (float)atof(val());
		else if (f == "-xi") a.xi = (float)atof(val());
		else if (f == "-rebalance") a.rebalance = atoi(val());
		else if (f == "-tree-steps") a.tree_steps = atoi(val());
		else if (f == "-exchange") { const std::string v = val(); if (v != "let" && v != "let-exact" && v != "gather") { std::cerr << "Error: -exchange let|let-exact|gather\n"; return -1; } a.let = v != "gather"; a.capped = v == "let"; }
		else if (f == "-partition") { const std::string v = val(); if (v != "dist" && v != "gather") { std::cerr << "Error: -partition dist|gather\n"; return -1; } a.dist_partition = v == "dist"; }
		else if (f == "-o") a.out = val();
		else if (f == "-h" || f == "-help")
		{
			std::cout << "Usage: nbco3_dist -gpus G [-n N] [-p order] [-ds dt] [-iters n] [-steps n] [-r radius] [-i dens] [-xi v] [-rebalance k] [-o folder]\n"
			             "                  [-exchange let|let-exact|gather] [-partition dist|gather]\n"
			             "  kd-tree FMM simulation (the nbco3 loop) with the particles sharded by kd-domain over G GPUs of this node, one process\n"
			             "  per GPU, RCCL all-gathers in between; G a power of two, N a multiple of G with at least 4096 particles per GPU.\n";
			return 0;
		}
		else { std::cerr << "Error: unrecognised option '" << argv[i] << "'\n"; return -1; }
	}
	if (const char *e = getenv("NBCO3_DIST_HANG")) a.hang_rank = atoi(e);
	if (const char *e = getenv("NBCO3_DIST_FAIL")) { if (sscanf(e, "%d:%d", &a.fail_rank, &a.fail_iter) != 2) a.fail_rank = -1; }
	if (a.gpus < 1 || (a.gpus & (a.gpus - 1)) || a.n <= 0 || a.n % a.gpus || a.steps <= 0 || a.iters <= 0 || a.tree_steps < 1)
	{
		std::cerr << "Error: -gpus must be a power of two dividing -n\n";
		return -1;
	}
	// The supervisor forks ALL ranks before any HIP / RCCL call is made anywhere and makes none itself.  Rank 0 creates the
	// ncclUniqueId and writes it into one pipe per sibling (the pipes are created first, so every child inherits them).
	std::vector<int> rd(a.gpus, -1), wr(a.gpus, -1);
	for (int r = 1; r < a.gpus; ++r)
	{
		int fd[2];
		if (pipe(fd) != 0) { perror("pipe"); return -1; }
		rd[r] = fd[0]; wr[r] = fd[1];
	}
	std::vector<pid_t> kids(a.gpus, (pid_t)-1);
	auto kill_all = [&](int sig) { for (pid_t k : kids) if (k > 0) kill(k, sig); };
	for (int r = 0; r < a.gpus; ++r)
	{
		const pid_t pid = fork();
		if (pid < 0) { perror("fork"); kill_all(SIGKILL); return -1; }
		if (pid == 0)
		{
			if (r == a.hang_rank) for (;;) pause();   // test hook: a rank that never comes back (as one blocked in a collective)
			ncclUniqueId id;
			if (r == 0)
			{
				for (int q = 1; q < a.gpus; ++q) close(rd[q]);
				NCCLCHK(ncclGetUniqueId(&id));
				for (int q = 1; q < a.gpus; ++q)
				{
					if (write(wr[q], &id, sizeof id) != (ssize_t)sizeof id) { perror("write"); std::_Exit(5); }
					close(wr[q]);
				}
			}
			else
			{
				for (int q = 1; q < a.gpus; ++q) { close(wr[q]); if (q != r) close(rd[q]); }
				if (read(rd[r], &id, sizeof id) != (ssize_t)sizeof id) { std::cerr << "rank " << r << ": no RCCL id from rank 0" << std::endl; std::_Exit(5); }
				close(rd[r]);
			}
			const int rc = run_rank(a, r, id);
			std::cout << std::flush;
			std::_Exit(rc == 0 ? 0 : 1);
		}
		kids[r] = pid;
	}
	for (int r = 1; r < a.gpus; ++r) { close(rd[r]); close(wr[r]); }
	// supervise: the first rank that fails (non-zero status or a signal) ends the run -- the others would wait in a collective
	// for ever.  They are told to stop (SIGTERM), given two seconds, and then killed; each by its PID.
	int alive = a.gpus, rc = 0;
	while (alive > 0)
	{
		int st = 0;
		const pid_t pid = waitpid(-1, &st, 0);
		if (pid < 0) { if (errno == EINTR) continue; break; }
		int rank = -1;
		for (int r = 0; r < a.gpus; ++r) if (kids[r] == pid) rank = r;
		if (rank < 0) continue;
		kids[rank] = -1;
		--alive;
		const bool ok = WIFEXITED(st) && WEXITSTATUS(st) == 0;
		if (ok || rc != 0) continue;
		rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + WTERMSIG(st);
		std::cerr << "nbco3_dist: rank " << rank << (WIFEXITED(st) ? " exited with status " : " was killed by signal ") << (WIFEXITED(st) ? WEXITSTATUS(st) : WTERMSIG(st))
		          << "; stopping the other " << alive << " rank(s)" << std::endl;
		kill_all(SIGTERM);
		for (int waited = 0; waited < 20 && alive > 0; ++waited)
		{
			usleep(100000);
			for (int r = 0; r < a.gpus; ++r)
				if (kids[r] > 0 && waitpid(kids[r], &st, WNOHANG) == kids[r]) { kids[r] = -1; --alive; }
		}
		kill_all(SIGKILL);
	}
	return rc;
}
