// nbco3_dist.cpp -- C++20 multi-GPU host over the C ABI (include/nbco.h, section "multi-GPU") and RCCL: the kd-tree FMM
// simulation loop of nbco3 (main3.cu:832-874) with the particles sharded by kd-domain over the GPUs of one node, one process
// per GPU.  The reference is single-GPU (fmm_cart3_kdtree.cuh:1529 hard-codes device 0); what is kept from it is the command
// line, the initial state and the snapshot format -- a snapshot of a G-GPU run is the same [pos | vel] file, in the tree order
// of the global kd-tree (rank r's particles are rows [r N/G, (r+1) N/G)).
//
//   nbco3_dist -gpus G [-n N] [-p order] [-ds dt] [-iters n] [-steps n] [-r radius] [-i dens] [-rebalance k] [-o folder]
//              [-exchange let|gather] [-partition dist|gather]
//
// The launcher process forks the G ranks BEFORE anything touches the GPU; rank 0's ncclUniqueId reaches the others through
// pipes.  Per evaluation: subtree build -> all-gather of positions and traversal records ->
// multipoles -> all-gather of the multipoles, on a communication stream of their own, under the traversal -> lists, near and
// far field, L2P.  Every `rebalance` evaluations the domains are cut again from the gathered state (nbco_dist_partition).
// Defaults since round 2 (INTEGRATION.md sections 4a, 4b): the locally-essential-tree exchange -- all-gather of the traversal
// records only, then grouped ncclSend / ncclRecv of exactly the multipoles and positions the other ranks' lists name -- and the
// re-partition without gathering the state (nbco_dist_repartition_*: the library names a collective, this host runs it).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
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
	int gpus = 1, n = 1 << 20, order = 3, iters = 30001, steps = 200, rebalance = 16;
	bool let = true, dist_partition = true;
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
	bool let = true, dist_partition = true;
	// LET exchange: count blocks (device + host), record buffers sized for the worst case (everything needed by everyone)
	long long *counts_send = nullptr, *counts_all_dev = nullptr;
	std::vector<long long> counts_all;
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

	// nbco_dist_repartition_*: every collective the library names runs on the compute stream, in order
	void repartition()
	{
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
			check(nbco_dist_repartition_next(ctx, &st), "nbco_dist_repartition_next");
		}
		evals = 0;
	}

	// one force evaluation with the LET exchange
	void force_let()
	{
		const long long nl = lay.n_local;
		const int S = lay.let_counts;
		char *csz_send = nodes_send, *csz_all = nodes_all;
		for (int attempt = 0;; ++attempt)
		{
		check(nbco_dist_let_local_geom(ctx, buf, nl, csz_send), "nbco_dist_let_local_geom");
		NCCLCHK(ncclAllGather(csz_send, csz_all, (size_t)lay.csz_bytes, ncclChar, comm, nullptr));
		check(nbco_dist_let_local_mpole(ctx, buf, nl), "nbco_dist_let_local_mpole");
		for (int round = 0;; ++round)
		{
			check(nbco_dist_let_select(ctx, csz_all, counts_send), "nbco_dist_let_select");
			NCCLCHK(ncclAllGather(counts_send, counts_all_dev, (size_t)S, ncclInt64, comm, nullptr));
			HIPCHK(hipMemcpyAsync(counts_all.data(), counts_all_dev, sizeof(long long) * (size_t)S * world, hipMemcpyDeviceToHost, nullptr));
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
		check(nbco_dist_let_pack(ctx, counts_all.data(), let_pos_send, let_mp_send), "nbco_dist_let_pack");
		std::vector<long long> ps(world), pr(world), ms(world), mr(world);
		for (int r = 0; r < world; ++r)
		{
			ms[r] = counts_all[(size_t)rank * S + 2 * r]; ps[r] = counts_all[(size_t)rank * S + 2 * r + 1];
			mr[r] = counts_all[(size_t)r * S + 2 * rank]; pr[r] = counts_all[(size_t)r * S + 2 * rank + 1];
		}
		all_to_all(let_pos_send, let_pos_recv, ps.data(), pr.data(), 16, nullptr);
		all_to_all(let_mp_send, let_mp_recv, ms.data(), mr.data(), (size_t)lay.let_node_bytes, nullptr);
		check(nbco_dist_let_finish(ctx, counts_all.data(), let_pos_recv, let_mp_recv, buf, buf + 6 * nl, par), "nbco_dist_let_finish");
		check(nbco_add_elastic(ctx, buf, buf + 6 * nl, nl, par + 3), "nbco_add_elastic");
		++evals;
	}

	void partition()
	{
		if (dist_partition) { repartition(); return; }
		const long long nl = lay.n_local, N = lay.n_global;
		gather(buf, state_all, sizeof(float) * 3 * nl, ev_geom);                   // positions
		gather(buf + 3 * nl, state_all + 3 * N, sizeof(float) * 3 * nl, ev_geom);   // velocities
		HIPCHK(hipStreamWaitEvent(nullptr, ev_geom, 0));
		check(nbco_dist_partition(ctx, state_all, N, world, rank, buf), "nbco_dist_partition");
		evals = 0;
	}

	void force(int rebalance)
	{
		if (rebalance > 0 && evals >= rebalance) partition();
		if (let) { force_let(); return; }
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
		check(nbco_add_elastic(ctx, buf, buf + 6 * nl, nl, par + 3), "nbco_add_elastic");
		++evals;
	}

	void leapfrog(float dt, int rebalance)   // integrator.cuh:68-96
	{
		const long long nl = lay.n_local;
		check(nbco_step(ctx, buf + 3 * nl, buf + 6 * nl, 0.5f * dt, nl), "step");
		check(nbco_step(ctx, buf, buf + 3 * nl, dt, nl), "step");
		force(rebalance);
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
	o.unsort = 0; o.sync = 0; o.tree_steps = 1;
	if (nbco_create(&r.ctx, &o) != NBCO_OK) { std::cerr << "nbco_create failed" << std::endl; return -1; }
	r.check(nbco_dist_layout_query(r.ctx, a.n, a.gpus, rank, &r.lay), "nbco_dist_layout_query");
	const long long N = a.n, nl = r.lay.n_local;

	// every rank samples the same initial state (the reference's stream) and starts out owning a contiguous slice of it
	std::vector<float> host(6 * (size_t)N);
	const float sx[3] = {0.003f, 0.001f, 0.01f}, su[3] = {1.095f * 0.003f, 0.001f, 0.01f};
	if (nbco_init_gaussian(host.data(), N, sx, su, NBCO_REF_SEED, NBCO_REF_DISCARD, 0) != NBCO_OK) return -1;
	const float parh[6] = {a.xi / (float)N, 0, 0, 1.095f * 1.095f, 1.f, 1.f};
	HIPCHK(hipMalloc((void **)&r.buf, sizeof(float) * 9 * (size_t)nl));
	r.let = a.let; r.dist_partition = a.dist_partition && a.gpus <= 32;
	if (!r.dist_partition) HIPCHK(hipMalloc((void **)&r.state_all, sizeof(float) * 6 * (size_t)N));
	else
	{
		r.check(nbco_dist_repartition_workspace(r.ctx, N, a.gpus, &r.work_bytes), "nbco_dist_repartition_workspace");
		HIPCHK(hipMalloc((void **)&r.work, (size_t)r.work_bytes));
	}
	if (r.let)
	{
		const size_t S = (size_t)r.lay.let_counts, others = (size_t)(a.gpus > 1 ? a.gpus - 1 : 1);
		HIPCHK(hipMalloc((void **)&r.counts_send, sizeof(long long) * S));
		HIPCHK(hipMalloc((void **)&r.counts_all_dev, sizeof(long long) * S * a.gpus));
		r.counts_all.resize(S * a.gpus);
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
	HIPCHK(hipMemcpy(r.buf, host.data() + 3 * (size_t)rank * nl, sizeof(float) * 3 * nl, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(r.buf + 3 * nl, host.data() + 3 * (size_t)N + 3 * (size_t)rank * nl, sizeof(float) * 3 * nl, hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(r.par, parh, sizeof parh, hipMemcpyHostToDevice));

	r.partition();
	r.force(0);   // precompute accelerations (main3.cu:836-839)
	std::vector<float> slice(6 * (size_t)nl);
	for (int iter = 0; iter < a.iters; ++iter)
	{
		r.leapfrog(a.dt, a.rebalance);
		if (iter % a.steps != 0) continue;
		// every rank writes its rows of the one snapshot file (rank 0 creates it first; the barrier is a tiny all-reduce)
		HIPCHK(hipDeviceSynchronize());
		HIPCHK(hipMemcpy(slice.data(), r.buf, sizeof(float) * 6 * nl, hipMemcpyDeviceToHost));
		const std::string name = a.out + "/out" + std::to_string(iter) + '_' + std::to_string(a.dt) + ".bin";
		if (rank == 0)
		{
			std::cout << iter << ' ' << std::flush;
			FILE *f = std::fopen(name.c_str(), "wb");
			if (!f) { std::cerr << "Error: cannot write on output location. Check that \"" << a.out << "\" folder exists. Create it if not." << std::endl; return -1; }
			std::fclose(f);
		}
		int *flag = nullptr;
		HIPCHK(hipMalloc((void **)&flag, sizeof(int)));
		HIPCHK(hipMemset(flag, 0, sizeof(int)));
		NCCLCHK(ncclAllReduce(flag, flag, 1, ncclInt, ncclSum, r.comm, r.comm_stream));
		HIPCHK(hipStreamSynchronize(r.comm_stream));
		HIPCHK(hipFree(flag));
		FILE *f = std::fopen(name.c_str(), "r+b");
		if (!f) { std::cerr << "rank " << rank << ": cannot open " << name << std::endl; return -1; }
		std::fseek(f, (long)(sizeof(float) * 3 * (size_t)rank * nl), SEEK_SET);
		std::fwrite(slice.data(), sizeof(float), 3 * (size_t)nl, f);
		std::fseek(f, (long)(sizeof(float) * 3 * ((size_t)N + (size_t)rank * nl)), SEEK_SET);
		std::fwrite(slice.data() + 3 * (size_t)nl, sizeof(float), 3 * (size_t)nl, f);
		std::fclose(f);
	}
	HIPCHK(hipDeviceSynchronize());
	if (rank == 0) std::cout << std::endl;
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
		else if (f == "-exchange") { const std::string v = val(); if (v != "let" && v != "gather") { std::cerr << "Error: -exchange let|gather\n"; return -1; } a.let = v == "let"; }
		else if (f == "-partition") { const std::string v = val(); if (v != "dist" && v != "gather") { std::cerr << "Error: -partition dist|gather\n"; return -1; } a.dist_partition = v == "dist"; }
		else if (f == "-o") a.out = val();
		else if (f == "-h" || f == "-help")
		{
			std::cout << "Usage: nbco3_dist -gpus G [-n N] [-p order] [-ds dt] [-iters n] [-steps n] [-r radius] [-i dens] [-xi v] [-rebalance k] [-o folder]\n"
			             "                  [-exchange let|gather] [-partition dist|gather]\n"
			             "  kd-tree FMM simulation (the nbco3 loop) with the particles sharded by kd-domain over G GPUs of this node, one process\n"
			             "  per GPU, RCCL all-gathers in between; G a power of two, N a multiple of G with at least 4096 particles per GPU.\n";
			return 0;
		}
		else { std::cerr << "Error: unrecognised option '" << argv[i] << "'\n"; return -1; }
	}
	if (a.gpus < 1 || (a.gpus & (a.gpus - 1)) || a.n <= 0 || a.n % a.gpus || a.steps <= 0 || a.iters <= 0)
	{
		std::cerr << "Error: -gpus must be a power of two dividing -n\n";
		return -1;
	}
	// the ranks are forked before this process makes any HIP / RCCL call; rank 0 creates the id and passes it down the pipes
	std::vector<int> to_child(a.gpus, -1);
	std::vector<pid_t> kids;
	int my_rank = 0, from_parent = -1;
	for (int r = 1; r < a.gpus; ++r)
	{
		int fd[2];
		if (pipe(fd) != 0) { perror("pipe"); return -1; }
		const pid_t pid = fork();
		if (pid < 0) { perror("fork"); return -1; }
		if (pid == 0) { my_rank = r; from_parent = fd[0]; close(fd[1]); kids.clear(); break; }
		close(fd[0]);
		to_child[r] = fd[1];
		kids.push_back(pid);
	}
	ncclUniqueId id;
	if (my_rank == 0)
	{
		NCCLCHK(ncclGetUniqueId(&id));
		for (int r = 1; r < a.gpus; ++r)
			if (write(to_child[r], &id, sizeof id) != (ssize_t)sizeof id) { perror("write"); return -1; }
	}
	else if (read(from_parent, &id, sizeof id) != (ssize_t)sizeof id) { perror("read"); return -1; }
	int rc = run_rank(a, my_rank, id);
	if (my_rank == 0)
		for (pid_t k : kids)
		{
			int st = 0;
			waitpid(k, &st, 0);
			if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : -1;
		}
	return rc;
}
