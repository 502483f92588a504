// nbco3.cpp -- C++20 host driver over the C ABI (include/nbco.h): the reference's `nbco3` command-line
// interface, initial conditions and headerless binary state files, re-implemented for the MI355X
// engine.  Behaviour follows main3.cu:225-883 (flags :247-623, input :629-652, Gaussian / uniform
// init :71-137,662-666, parameter pack :685-692, -accuracy :737-788, -test :790-811, -test2 :812-831,
// simulation loop + snapshots :832-874).  `-cpu` runs the simulation loop on the host with the compensated direct sum
// (nbco_cpu.hpp: plumbing for small N; the reference's CPU twin of the FMM is out of scope); the test / tuning modes
// need the GPU.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>
#include <string_view>
#include <vector>

#include "nbco_reference_api.hpp"
#include "nbco_cpu.hpp"

using namespace nbco_ref;
using namespace std::chrono;

namespace {

#define HIPCHK(call)                                                                                  \
	do {                                                                                              \
		hipError_t e_ = (call);                                                                       \
		if (e_ != hipSuccess) { std::cerr << "GPUassert: " << hipGetErrorString(e_) << ' ' << __FILE__ << ' ' << __LINE__ << std::endl; std::exit((int)e_); } \
	} while (0)

struct V3 { SCAL x, y, z; };

const char *kHelp =
    "This program comes with ABSOLUTELY NO WARRANTY.\n\n"
    "Usage: nbco3 [options] [input]\n\n"
    "  [input] is the path to a binary file with the positions of all particles followed by their\n"
    "  velocities (fp32 xyz triplets, no header).  Without it the system is sampled from a gaussian.\n\n"
    "Other options:\n"
    "  -h or -help       Display this documentation.\n"
    "  -o <output>       Output folder (default './out', must exist).\n"
    "  -n <npart>        Number of particles (default 30001; ignored with [input]).\n"
    "  -ds <v>           Time step (default 5e-4).\n"
    "  -iters <n>        Number of simulation iterations (default 30000).\n"
    "  -steps <n>        Steps between snapshots (default 200).\n"
    "  -integ <name>     Integrator: eu, fr, pefrl (default: leapfrog).\n"
    "  -p <order>        FMM expansion order (default 3).\n"
    "  -r <radius>       Interaction radius (default 1).\n"
    "  -eps <v>          Smoothing length (default 1e-9).\n"
    "  -i <v>            Max FMM level is round(log2(n*i/p^2)) (default 1).\n"
    "  -maxlevel <n>     Maximum kd-tree level.\n"
    "  -ncoll            Skip the P2P pass.\n"
    "  -accuracy <v>     Search r, p for the fastest setting below this error.\n"
    "  -test             Print relative errors and the time of one evaluation.\n"
    "  -test2            Relative error while the tree is reused.\n"
    "  -xi <v>, -omega0 <wx> <wy>, -x <sx> <sy> <sz>, -u <ux> <uy> <uz>   physical parameters.\n"
    "  -snapshot-order <tree|input>   order of the particles in the snapshots: tree (default, as the reference: the order the\n"
    "                    last tree rebuild left them in) or input (every particle keeps the row it had in the initial state).\n"
    "  -cpu              Run the simulation on the host: compensated direct sum O(N^2) over C++20 threads (small N; the test and\n"
    "                    tuning modes need the GPU).  -cpu-threads <n> sets the number of threads (default 8); -cacheline <n> is\n"
    "                    accepted and ignored.\n";

// device allocation that frees itself
template <class T> struct DeviceArray
{
	T *ptr = nullptr;
	explicit DeviceArray(size_t count) { HIPCHK(hipMalloc((void **)&ptr, count * sizeof(T))); }
	~DeviceArray() { if (ptr) (void)hipFree(ptr); }
	DeviceArray(const DeviceArray &) = delete;
	DeviceArray &operator=(const DeviceArray &) = delete;
};

// One run of the program: the device state [pos | vel | acc], the parameter pack and the engine context, with the three
// modes of main3.cu:707-874 as member functions.
struct Session
{
	int n;
	DeviceArray<SCAL> state, par, ref_acc;
	bool have_ref = false;

	Session(const nbco_opts &o, int n_, const std::vector<float> &host, const SCAL (&p)[6]) : n(n_), state(9 * (size_t)n_), par(6), ref_acc(3 * (size_t)n_)
	{
		upload(host);
		HIPCHK(hipMemcpy(par.ptr, p, sizeof p, hipMemcpyHostToDevice));
		init(o);
	}
	~Session() { nbco_destroy(ctx()); ctx() = nullptr; }

	void upload(const std::vector<float> &host) { HIPCHK(hipMemcpy(state.ptr, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice)); }   // acc not copied
	SCAL *acc() { return state.ptr + 6 * (size_t)n; }
	nbco_opts opts() const { nbco_opts cur; nbco_get_opts(ctx(), &cur); return cur; }
	template <class F> void change(F &&f) { nbco_opts cur = opts(); f(cur); init(cur); }

	// seconds per evaluation: one untimed call, then batches of 1, 2, 4, ... calls until min_seconds have passed (main3.cu:707-735)
	SCAL seconds_per_evaluation(SCAL min_seconds)
	{
		compute_force(fmm_cart3_kdtree, state.ptr, n, par.ptr);
		long long calls = 0;
		SCAL elapsed = 0;
		const auto t0 = steady_clock::now();
		for (long long batch = 1; calls == 0 || elapsed < min_seconds; batch *= 2)
		{
			for (long long i = 0; i < batch; ++i) compute_force(fmm_cart3_kdtree, state.ptr, n, par.ptr);
			calls += batch;
			elapsed = duration_cast<microseconds>(steady_clock::now() - t0).count() * (SCAL)1.e-6;
		}
		return elapsed / (SCAL)calls;
	}

	// mean relative error of the FMM accelerations against the compensated direct sum (main3.cu:139-181).  With the caller's
	// order kept (unsort) the direct sum is cached until `refresh`; otherwise the FMM runs first because it permutes the state.
	SCAL mean_error(bool refresh)
	{
		float err = 0;
		if (opts().unsort)
		{
			if (refresh || !have_ref)
			{
				compute_force(direct3, state.ptr, n, par.ptr);
				check(nbco_copy(ctx(), ref_acc.ptr, acc(), n), "copy");
				have_ref = true;
			}
			compute_force(fmm_cart3_kdtree, state.ptr, n, par.ptr);
			check(nbco_mean_relerr(ctx(), acc(), ref_acc.ptr, n, &err), "mean_relerr");
		}
		else
		{
			compute_force(fmm_cart3_kdtree, state.ptr, n, par.ptr);
			check(nbco_copy(ctx(), ref_acc.ptr, acc(), n), "copy");
			compute_force(direct3, state.ptr, n, par.ptr);
			check(nbco_mean_relerr(ctx(), ref_acc.ptr, acc(), n, &err), "mean_relerr");
		}
		return (SCAL)err;
	}

	// -accuracy: fastest (r, p) of the grid of main3.cu:739-740 whose error stays below the bound (main3.cu:737-788)
	int search_parameters(SCAL bound)
	{
		struct Candidate { SCAL r; int p; SCAL seconds, error; };
		static constexpr SCAL radii[] = {1.11f, 1.25f, 1.43f, 1.67f, 2.f, 2.5f, 3.f};
		std::vector<Candidate> admissible;
		std::cout << "Parameter optimization in progress, please wait" << std::flush;
		for (SCAL r : radii)
			for (int p = 1; p <= 6; ++p)
			{
				change([&](nbco_opts &c) { c.coll = 1; c.unsort = 1; c.tree_radius = r; c.fmm_order = p; });
				const SCAL e = mean_error(false);
				if (e < bound) admissible.push_back({r, p, seconds_per_evaluation(0), e});
				std::cout << '.' << std::flush;
			}
		if (admissible.empty()) { std::cout << "\nOptimization failed!" << std::endl; return -1; }
		const Candidate *best = &admissible.front();
		for (const Candidate &c : admissible)
			if (c.seconds < best->seconds) best = &c;
		change([&](nbco_opts &c) { c.tree_radius = best->r; c.fmm_order = best->p; });
		std::cout << "\nBest parameters: r = " << best->r << ", p = " << best->p << ", time = " << best->seconds << ", error = " << best->error << std::endl;
		return 0;
	}

	// -test: time of one evaluation at the chosen order, then the error table for orders 1..10 (main3.cu:790-811)
	void print_error_table(const std::vector<float> &host)
	{
		change([](nbco_opts &c) { c.unsort = 0; });
		std::cout << opts().fmm_order << ": Average time: " << seconds_per_evaluation(1) << " [s]" << std::endl;
		upload(host);   // the timing loop permuted the state: the table is quoted in the caller's order
		for (int p = 1; p <= 10; ++p)
		{
			change([&](nbco_opts &c) { c.unsort = 1; c.fmm_order = p; });
			std::cout << p << ": Relative error: " << mean_error(p == 1) << std::endl;
		}
	}

	// -test2: error of successive evaluations while the particles move in the trap and the tree is reused (main3.cu:812-831)
	void print_reuse_errors(SCAL dt)
	{
		change([](nbco_opts &c) { c.unsort = 0; });
		const int evaluations = opts().tree_steps + 1;
		for (int i = 0; i < evaluations; ++i)
		{
			const SCAL e = mean_error(false);
			pre_symplectic_euler(add_elastic, state.ptr, n, par.ptr + 3, dt, step);
			std::cout << "Relative error after " << i << " steps: " << e << std::endl;
		}
	}

	// simulation: accelerations first, then nIters fused integrator steps; a snapshot [pos | vel] every nSteps iterations
	// (main3.cu:832-874).  Evaluations are enqueued without a drain; the copy of a snapshot is what waits for the device.
	int simulate(int scheme, SCAL dt, int nIters, int nSteps, const std::string &folder, std::vector<float> &host, size_t state_bytes, bool input_order)
	{
		change([&](nbco_opts &c) { c.unsort = 0; c.sync = 0; c.track_order = input_order ? 1 : 0; });
		std::vector<int> order(input_order ? (size_t)n : 0);
		std::vector<float> rows(input_order ? host.size() : 0);
		check(nbco_force(ctx(), NBCO_EVAL_FMM_KDTREE, state.ptr, n, par.ptr, 1), "compute_force");
		check(nbco_sync(ctx()), "sync");
		auto loop_t0 = std::chrono::steady_clock::now(), steady_t0 = loop_t0;
		int loop_first = 0, steady_first = -1;
		// (not in the reference: a second clock that starts kSteadyFrom iterations in, behind the first builds of the run -- cold
		// median selections, list buffers sized from nothing -- so that the figure is comparable with a library run in its stride)
		constexpr int kSteadyFrom = 9;
		// snapshots follow the iterations 0, nSteps, 2 nSteps, ..: the steps in between are ONE nbco_integrate_steps call (same final
		// state as step-by-step calls; leapfrog fuses what lies between two force evaluations into one pass)
		for (int iter = 0; iter < nIters;)
		{
			int run = iter % nSteps == 0 ? 1 : std::min(nSteps - iter % nSteps, nIters - iter);
			if (steady_first < 0 && nIters >= 3 * kSteadyFrom)
			{
				if (iter == kSteadyFrom) { check(nbco_sync(ctx()), "sync"); steady_t0 = std::chrono::steady_clock::now(); steady_first = iter; }
				else if (iter < kSteadyFrom) run = std::min(run, kSteadyFrom - iter);   // (K steps in one call == the same steps in two calls, bit for bit)
			}
			check(nbco_integrate_steps(ctx(), scheme, NBCO_EVAL_FMM_KDTREE, state.ptr, n, par.ptr, (double)dt, 1.0, 1, run), "integrate");
			iter += run;
			if ((iter - 1) % nSteps != 0) continue;
			const int snap = iter - 1;
			std::cout << snap << ' ' << std::flush;
			check(nbco_sync(ctx()), "sync");
			HIPCHK(hipMemcpy(host.data(), state.ptr, state_bytes, hipMemcpyDeviceToHost));   // acc not copied
			std::ofstream fout(folder + "/out" + std::to_string(snap) + '_' + std::to_string(dt) + ".bin", std::ios::out | std::ios::binary);
			if (!fout)
			{
				std::cerr << "Error: cannot write on output location. Check that \"" << folder << "\" folder exists. Create it if not." << std::endl;
				return -1;
			}
			if (input_order)
			{
				// the engine composes the permutations of all tree rebuilds: row i of the state is particle order[i] of the initial state
				check(nbco_kd_copy(ctx(), NBCO_KD_ORDER, order.data(), (long long)(order.size() * sizeof(int))), "kd_copy");
				for (size_t i = 0; i < (size_t)n; ++i)
					for (int k = 0; k < 3; ++k)
					{
						rows[3 * (size_t)order[i] + k] = host[3 * i + k];
						rows[3 * ((size_t)n + order[i]) + k] = host[3 * ((size_t)n + i) + k];
					}
				fout.write(reinterpret_cast<const char *>(rows.data()), (std::streamsize)state_bytes);
			}
			else fout.write(reinterpret_cast<const char *>(host.data()), (std::streamsize)state_bytes);
			if (snap == 0) { loop_t0 = std::chrono::steady_clock::now(); loop_first = 1; }   // the timer below starts behind the first snapshot
		}
		check(nbco_sync(ctx()), "sync");
		std::cout << std::endl;
		// (not in the reference: wall time of the integration loop behind the first snapshot -- what bench.py's `cli` leg reads)
		const auto loop_t1 = std::chrono::steady_clock::now();
		std::cout << "Loop time: " << std::chrono::duration<double>(loop_t1 - loop_t0).count() << " s, " << nIters - loop_first << " iterations" << std::endl;
		if (steady_first >= 0)
			std::cout << "Steady loop time: " << std::chrono::duration<double>(loop_t1 - steady_t0).count() << " s, " << nIters - steady_first
			          << " iterations (from iteration " << steady_first << " on)" << std::endl;
		return 0;
	}
};

} // namespace

int main(int argc, const char **argv)
{
	std::cout << "N-body coulomb oscillators -- MI355X engine (nbco3 interface of locuoco/coulomb_oscillators)\n\n"
	             "Type 'nbco3 -h' for a brief documentation.\n\n";

	int nBodies = 30001;
	SCAL dt = (SCAL)5.e-4;
	int nIters = 30001, nSteps = 200;
	std::string strout("out"), strin;
	bool in = false, test = false, test2 = false, b_accuracy = false, input_order = false, cpu = false;
	SCAL accuracy = (SCAL)0.001;
	int scheme = NBCO_INTEG_LEAPFROG;   // main3.cu:238
	SCAL xi = (SCAL)2.e-6;
	V3 omega0{(SCAL)1.095, (SCAL)1.0, (SCAL)1.0}, x{(SCAL)0.003, (SCAL)0.001, (SCAL)0.01};
	V3 u{omega0.x * x.x, omega0.y * x.y, omega0.z * x.z};
	nbco_opts o;
	nbco_opts_default(&o);
	o.tree_steps = 8;    // constants.cuh:45, the reference GPU driver's rebuild cadence
	o.m2l_first = 1;     // reference GPU traversal order (fmm_cart3_kdtree.cuh:520-534)

	auto need = [&](int i, int k, const char *flag) {
		if (i + k >= argc) { std::cerr << "Error: missing argument" << (k > 1 ? "(s)" : "") << " to '" << flag << "'\n"; return false; }
		return true;
	};
	for (int i = 1; i < argc; ++i)
	{
		std::string_view a(argv[i]);
		if (a.empty() || a[0] != '-') { strin = argv[i]; in = true; continue; }
		if (a == "-h" || a == "-help") { std::cout << kHelp; return 0; }
		else if (a == "-o") { if (!need(i, 1, "-o")) return -1; strout = argv[++i]; }
		else if (a == "-n")
		{
			if (!need(i, 1, "-n")) return -1;
			nBodies = atoi(argv[i + 1]);
			if (nBodies <= 0) { std::cerr << "Error: invalid argument to '-n': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-ds")
		{
			if (!need(i, 1, "-ds")) return -1;
			dt = (SCAL)atof(argv[i + 1]);
			if (dt <= 0) { std::cerr << "Error: invalid argument to '-ds': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-iters")
		{
			if (!need(i, 1, "-iters")) return -1;
			nIters = atoi(argv[i + 1]) + 1;   // main3.cu:357
			if (nIters <= 0) { std::cerr << "Error: invalid argument to '-iters': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-steps")
		{
			if (!need(i, 1, "-steps")) return -1;
			nSteps = atoi(argv[i + 1]);
			if (nSteps <= 0) { std::cerr << "Error: invalid argument to '-steps': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-integ")
		{
			if (!need(i, 1, "-integ")) return -1;
			// the reference compares from the second character on (main3.cu:389-395): "-fr", "xfr", ...;
			// the documented bare names are accepted as well
			std::string_view v(argv[i + 1]);
			std::string_view tail = v.size() > 1 ? v.substr(1) : std::string_view{};
			if (v == "eu" || tail == "eu") scheme = NBCO_INTEG_EULER;
			else if (v == "fr" || tail == "fr") scheme = NBCO_INTEG_FORESTRUTH;
			else if (v == "pefrl" || tail == "pefrl") scheme = NBCO_INTEG_PEFRL;
			else { std::cerr << "Error: invalid argument to '-integ': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-p")
		{
			if (!need(i, 1, "-p")) return -1;
			o.fmm_order = atoi(argv[i + 1]);
			if (o.fmm_order <= 0) { std::cerr << "Error: invalid argument to '-p': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-r")
		{
			if (!need(i, 1, "-r")) return -1;
			o.tree_radius = (float)atof(argv[i + 1]);
			if (o.tree_radius <= 0) { std::cerr << "Error: invalid argument to '-r': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-eps")
		{
			if (!need(i, 1, "-eps")) return -1;
			float e = (float)atof(argv[i + 1]);
			if (e <= 0) { std::cerr << "Error: invalid argument to '-eps': " << argv[i + 1] << '\n'; return -1; }
			o.eps2 = e * e;   // main3.cu:440-446
			if (o.eps2 == 0) { std::cerr << "Error: too small argument to '-eps': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-i")
		{
			if (!need(i, 1, "-i")) return -1;
			o.dens_inhom = (float)atof(argv[i + 1]);
			if (o.dens_inhom <= 0) { std::cerr << "Error: invalid argument to '-i': " << argv[i + 1] << " (should be greater than 0)\n"; return -1; }
			++i;
		}
		else if (a == "-maxlevel")
		{
			if (!need(i, 1, "-maxlevel")) return -1;
			o.tree_L = atoi(argv[i + 1]);
			if (o.tree_L <= 0) { std::cerr << "Error: invalid argument to '-maxlevel': " << argv[i + 1] << " (should be greater than 0)\n"; return -1; }
			++i;
		}
		else if (a == "-ncoll") o.coll = 0;
		else if (a == "-accuracy")
		{
			if (!need(i, 1, "-accuracy")) return -1;
			b_accuracy = true;
			accuracy = (SCAL)atof(argv[i + 1]);
			if (accuracy <= 0) { std::cerr << "Error: invalid argument to '-accuracy': " << argv[i + 1] << " (should be greater than 0)\n"; return -1; }
			++i;
		}
		else if (a == "-cpu") cpu = true;
		else if (a == "-cpu-threads")
		{
			if (!need(i, 1, "-cpu-threads")) return -1;
			nbco_cpu::threads() = atoi(argv[i + 1]);
			if (nbco_cpu::threads() <= 0) { std::cerr << "Error: invalid argument to '-cpu-threads': " << argv[i + 1] << " (should be greater than 0)\n"; return -1; }
			++i;
		}
		else if (a == "-cacheline")
		{
			if (!need(i, 1, "-cacheline")) return -1;   // (main3.cu: cache-line padding of the reference's CPU FMM; nothing to tune here)
			++i;
		}
		else if (a == "-snapshot-order")
		{
			if (!need(i, 1, "-snapshot-order")) return -1;
			const std::string_view v(argv[i + 1]);
			if (v == "input") input_order = true;
			else if (v == "tree") input_order = false;
			else { std::cerr << "Error: invalid argument to '-snapshot-order': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-test") test = true;
		else if (a == "-test2") test2 = true;
		else if (a == "-xi")
		{
			if (!need(i, 1, "-xi")) return -1;
			xi = (SCAL)atof(argv[i + 1]);
			if (xi < 0) { std::cerr << "Error: invalid argument to '-xi': " << argv[i + 1] << '\n'; return -1; }
			++i;
		}
		else if (a == "-omega0")
		{
			if (!need(i, 2, "-omega0")) return -1;
			omega0.x = (SCAL)atof(argv[i + 1]); omega0.y = (SCAL)atof(argv[i + 2]);   // main3.cu:568-569 (z is not read)
			if (omega0.x < 0 || omega0.y < 0) { std::cerr << "Error: invalid argument(s) to '-omega0': " << argv[i + 1] << ' ' << argv[i + 2] << '\n'; return -1; }
			i += 2;
		}
		else if (a == "-x" || a == "-u")
		{
			if (!need(i, 3, a == "-x" ? "-x" : "-u")) return -1;
			V3 v{(SCAL)atof(argv[i + 1]), (SCAL)atof(argv[i + 2]), (SCAL)atof(argv[i + 3])};
			if (v.x < 0 || v.y < 0 || v.z < 0) { std::cerr << "Error: invalid argument(s) to '" << a << "': " << argv[i + 1] << ' ' << argv[i + 2] << ' ' << argv[i + 3] << '\n'; return -1; }
			(a == "-x" ? x : u) = v;
			i += 3;
		}
		else { std::cerr << "Error: unrecognised option '" << argv[i] << "'\n"; return -1; }
	}

	// ---- state ---------------------------------------------------------------------------------------
	std::vector<float> host;   // [pos | vel] as the state files hold them; the accelerations never leave the device
	if (in)
	{
		std::ifstream fin(strin, std::ios::in | std::ios::binary | std::ios::ate);
		if (!fin) { std::cerr << "Error: cannot read from input location." << std::endl; return -1; }
		const std::streamoff len = fin.tellg();
		nBodies = (int)((size_t)len / 2 / sizeof(V3));       // main3.cu:636
		host.resize(6 * (size_t)nBodies);
		fin.seekg(0, std::ios::beg);
		fin.read(reinterpret_cast<char *>(host.data()), (std::streamsize)(host.size() * sizeof(float)));
	}
	else
	{
		host.resize(6 * (size_t)nBodies);
		const float sx[3]{x.x, x.y, x.z}, su[3]{u.x, u.y, u.z};
		// main3.cu:662-666: mt19937_64(5351550349027530206), discard(624 * 2), initGA, and initU over the same stream for -test
		if (nbco_init_gaussian(host.data(), nBodies, sx, su, NBCO_REF_SEED, NBCO_REF_DISCARD, test ? 1 : 0) != NBCO_OK)
		{
			std::cerr << "Error: cannot sample the initial state." << std::endl;
			return -1;
		}
	}
	if (nBodies <= 0) { std::cerr << "Error: no particles." << std::endl; return -1; }
	const size_t state_bytes = host.size() * sizeof(float);

	if (!test && !test2)
	{
		std::ofstream farg(strout + "/args.txt", std::ios::out);   // main3.cu:669-683
		if (!farg)
		{
			std::cerr << "Error: cannot write on output location. Check that \"" << strout << "\" folder exists. Create it if not." << std::endl;
			return -1;
		}
		for (int i = 0; i < argc; ++i) farg << argv[i] << ' ';
	}

	const SCAL par[6]{xi / (SCAL)nBodies, 0, 0, omega0.x * omega0.x, omega0.y * omega0.y, omega0.z * omega0.z};   // main3.cu:685-692

	if (cpu)
	{
		// BASELINE config 1: the same loop and files with the host evaluator (coulombOscillatorDirect_cpu, main3.cu:53-57, :844)
		if (test || test2 || b_accuracy) { std::cerr << "Error: '-cpu' runs the simulation only: -test, -test2 and -accuracy need the GPU\n"; return -1; }
		if (input_order) { std::cerr << "Error: '-snapshot-order input' is what '-cpu' writes anyway (the direct sum never permutes the state)\n"; return -1; }
		std::vector<nbco_cpu::V3> st(3 * (size_t)nBodies);
		std::memcpy(st.data(), host.data(), state_bytes);
		const nbco_cpu::Scheme sch = scheme == NBCO_INTEG_EULER ? nbco_cpu::Euler : scheme == NBCO_INTEG_FORESTRUTH ? nbco_cpu::ForestRuth
		                             : scheme == NBCO_INTEG_PEFRL ? nbco_cpu::Pefrl : nbco_cpu::Leapfrog;
		nbco_cpu::force(st.data(), nBodies, par, o.eps2);
		for (int iter = 0; iter < nIters; ++iter)
		{
			nbco_cpu::integrate(sch, st.data(), nBodies, par, o.eps2, (long double)dt);
			if (iter % nSteps != 0) continue;
			std::cout << iter << ' ' << std::flush;
			std::ofstream fout(strout + "/out" + std::to_string(iter) + '_' + std::to_string(dt) + ".bin", std::ios::out | std::ios::binary);
			if (!fout)
			{
				std::cerr << "Error: cannot write on output location. Check that \"" << strout << "\" folder exists. Create it if not." << std::endl;
				return -1;
			}
			fout.write(reinterpret_cast<const char *>(st.data()), (std::streamsize)state_bytes);
		}
		std::cout << std::endl;
		return 0;
	}

	Session s(o, nBodies, host, par);
	int rc = 0;
	if (b_accuracy) rc = s.search_parameters(accuracy);
	if (rc == 0)
	{
		if (test) s.print_error_table(host);
		else if (test2) s.print_reuse_errors(dt);
		else rc = s.simulate(scheme, dt, nIters, nSteps, strout, host, state_bytes, input_order);
	}
	return rc;
}
