// nbco3.cpp -- C++20 host driver over the C ABI (include/nbco.h): the reference's `nbco3` command-line
// interface, initial conditions and headerless binary state files, re-implemented for the MI355X
// engine.  Behaviour follows main3.cu:225-883 (flags :247-623, input :629-652, Gaussian / uniform
// init :71-137,662-666, parameter pack :685-692, -accuracy :737-788, -test :790-811, -test2 :812-831,
// simulation loop + snapshots :832-874).  The force path is GPU only: `-cpu`, `-cpu-threads` and
// `-cacheline` are recognised and rejected (the reference's CPU twin is out of scope).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <random>
#include <string>
#include <string_view>
#include <vector>

#include "nbco_reference_api.hpp"

using namespace nbco_ref;
using namespace std::chrono;

namespace {

#define HIPCHK(call)                                                                                  \
	do {                                                                                              \
		hipError_t e_ = (call);                                                                       \
		if (e_ != hipSuccess) { std::cerr << "GPUassert: " << hipGetErrorString(e_) << ' ' << __FILE__ << ' ' << __LINE__ << std::endl; std::exit((int)e_); } \
	} while (0)

struct V3 { SCAL x, y, z; };

void centerDist(V3 *d, int n)   // main3.cu:71-80
{
	V3 c{0, 0, 0};
	for (int i = 0; i < n; ++i) { c.x += d[i].x; c.y += d[i].y; c.z += d[i].z; }
	c.x /= (SCAL)n; c.y /= (SCAL)n; c.z /= (SCAL)n;
	for (int i = 0; i < n; ++i) { d[i].x -= c.x; d[i].y -= c.y; d[i].z -= c.z; }
}

void adjustRMS(V3 *d, int n, V3 adj)   // main3.cu:82-92
{
	V3 s{0, 0, 0};
	for (int i = 0; i < n; ++i) { s.x += d[i].x * d[i].x; s.y += d[i].y * d[i].y; s.z += d[i].z * d[i].z; }
	s.x = std::sqrt(s.x / (SCAL)n); s.y = std::sqrt(s.y / (SCAL)n); s.z = std::sqrt(s.z / (SCAL)n);
	for (int i = 0; i < n; ++i) { d[i].x = d[i].x * adj.x / s.x; d[i].y = d[i].y * adj.y / s.y; d[i].z = d[i].z * adj.z / s.z; }
}

void initGA(V3 *data, int nBodies, V3 x, V3 u, std::mt19937_64 &gen)   // main3.cu:113-137
{
	std::normal_distribution<SCAL> dist((SCAL)0, (SCAL)1);
	SCAL *s = reinterpret_cast<SCAL *>(data);
	for (long long i = 0; i < 6LL * nBodies; ++i) s[i] = dist(gen);
	for (int i = 0; i < nBodies; ++i) { data[i].x *= x.x; data[i].y *= x.y; data[i].z *= x.z; }
	for (int i = nBodies; i < 2 * nBodies; ++i) { data[i].x *= u.x; data[i].y *= u.y; data[i].z *= u.z; }
	centerDist(data, nBodies); adjustRMS(data, nBodies, x);
	centerDist(data + nBodies, nBodies); adjustRMS(data + nBodies, nBodies, u);
}

void initU(V3 *data, int nBodies, std::mt19937_64 &gen)   // main3.cu:94-111 with a = -1, b = 1
{
	std::uniform_real_distribution<SCAL> dx(-1, 1), dy(-1, 1), dz(-1, 1);
	for (int i = 0; i < nBodies; ++i) { data[i].x = dx(gen); data[i].y = dy(gen); data[i].z = dz(gen); }
	centerDist(data, nBodies);
}

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
    "  -cpu, -cpu-threads <n>, -cacheline <n>   not available: this build is GPU (MI355X) only.\n";

} // namespace

int main(int argc, const char **argv)
{
	std::cout << "N-body coulomb oscillators -- MI355X engine (nbco3 interface of locuoco/coulomb_oscillators)\n\n"
	             "Type 'nbco3 -h' for a brief documentation.\n\n";

	int nBodies = 30001;
	SCAL dt = (SCAL)5.e-4;
	int nIters = 30001, nSteps = 200;
	std::string strout("out"), strin;
	bool in = false, test = false, test2 = false, b_accuracy = false;
	SCAL accuracy = (SCAL)0.001;
	integrator_t symp_integ = leapfrog;
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
			if (v == "eu" || tail == "eu") symp_integ = symplectic_euler;
			else if (v == "fr" || tail == "fr") symp_integ = forestruth;
			else if (v == "pefrl" || tail == "pefrl") symp_integ = pefrl;
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
		else if (a == "-cpu" || a == "-cpu-threads" || a == "-cacheline")
		{
			std::cerr << "Error: '" << a << "' is not available: this build runs the force path on the GPU only\n";
			return -1;
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
	std::vector<char> c_buf;
	size_t cpyBytes = 0;
	if (in)
	{
		std::ifstream fin(strin, std::ios::in | std::ios::binary);
		if (!fin) { std::cerr << "Error: cannot read from input location." << std::endl; return -1; }
		fin.ignore(std::numeric_limits<std::streamsize>::max());
		cpyBytes = (size_t)fin.gcount();
		nBodies = (int)(cpyBytes / 2 / sizeof(V3));       // main3.cu:636
		cpyBytes = 2 * (size_t)nBodies * sizeof(V3);
		c_buf.resize(3 * (size_t)nBodies * sizeof(V3));
		fin.clear();
		fin.seekg(0, std::ios::beg);
		fin.read(c_buf.data(), (std::streamsize)cpyBytes);
	}
	else
	{
		cpyBytes = 2 * (size_t)nBodies * sizeof(V3);
		c_buf.resize(3 * (size_t)nBodies * sizeof(V3));
		std::mt19937_64 gen(5351550349027530206ULL);      // main3.cu:662-666
		gen.discard(624 * 2);
		initGA(reinterpret_cast<V3 *>(c_buf.data()), nBodies, x, u, gen);
		if (test) initU(reinterpret_cast<V3 *>(c_buf.data()), nBodies, gen);
	}
	if (nBodies <= 0) { std::cerr << "Error: no particles." << std::endl; return -1; }
	SCAL *buf = reinterpret_cast<SCAL *>(c_buf.data());

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

	SCAL par[6]{xi / (SCAL)nBodies, 0, 0, omega0.x * omega0.x, omega0.y * omega0.y, omega0.z * omega0.z};   // main3.cu:685-692

	SCAL *d_buf = nullptr, *d_par = nullptr, *d_tmp = nullptr;
	HIPCHK(hipMalloc((void **)&d_buf, c_buf.size()));
	HIPCHK(hipMalloc((void **)&d_par, sizeof par));
	HIPCHK(hipMemcpy(d_buf, buf, cpyBytes, hipMemcpyHostToDevice));   // acc not copied
	HIPCHK(hipMemcpy(d_par, par, sizeof par, hipMemcpyHostToDevice));
	init(o);

	auto set_opts = [&](auto &&mod) { nbco_opts cur; nbco_get_opts(ctx(), &cur); mod(cur); init(cur); };

	auto test_time = [&](SCAL min_loop = 0, int loop_n = 1) {   // main3.cu:707-735
		compute_force(fmm_cart3_kdtree, d_buf, nBodies, d_par);
		SCAL duration;
		int loop_counter = 0;
		auto begin = steady_clock::now();
		do
		{
			for (int i = 0; i < loop_n; ++i) compute_force(fmm_cart3_kdtree, d_buf, nBodies, d_par);
			auto end = steady_clock::now();
			loop_counter += loop_n;
			loop_n *= 2;
			duration = duration_cast<microseconds>(end - begin).count() * (SCAL)1.e-6;
		} while (duration < min_loop);
		return duration / loop_counter;
	};
	// mean relative error of the FMM against the compensated direct sum (main3.cu:139-181, b_unsort case)
	auto test_accuracy = [&](bool b_update) {
		static bool have_ref = false;
		VEC *acc = reinterpret_cast<VEC *>(d_buf) + 2 * (size_t)nBodies;
		if (!d_tmp) HIPCHK(hipMalloc((void **)&d_tmp, sizeof(V3) * (size_t)nBodies));
		nbco_opts cur;
		nbco_get_opts(ctx(), &cur);
		if (cur.unsort)
		{
			if (b_update || !have_ref)
			{
				compute_force(direct3, d_buf, nBodies, d_par);
				check(nbco_copy(ctx(), d_tmp, &acc->x, nBodies), "copy");
				have_ref = true;
			}
			compute_force(fmm_cart3_kdtree, d_buf, nBodies, d_par);
			float err = 0;
			check(nbco_mean_relerr(ctx(), &acc->x, d_tmp, nBodies, &err), "mean_relerr");
			return (SCAL)err;
		}
		compute_force(fmm_cart3_kdtree, d_buf, nBodies, d_par);
		check(nbco_copy(ctx(), d_tmp, &acc->x, nBodies), "copy");
		compute_force(direct3, d_buf, nBodies, d_par);
		float err = 0;
		check(nbco_mean_relerr(ctx(), d_tmp, &acc->x, nBodies, &err), "mean_relerr");
		return (SCAL)err;
	};

	if (b_accuracy)   // main3.cu:737-788
	{
		const int search_p[] = {1, 2, 3, 4, 5, 6};
		const SCAL search_r[] = {1.11f, 1.25f, 1.43f, 1.67f, 2.f, 2.5f, 3.f};
		SCAL best_r = 0, best_time = FLT_MAX, best_accuracy = 0;
		int best_p = 0;
		std::cout << "Parameter optimization in progress, please wait" << std::flush;
		for (SCAL r : search_r)
			for (int p : search_p)
			{
				set_opts([&](nbco_opts &c) { c.coll = 1; c.unsort = 1; c.tree_radius = r; c.fmm_order = p; });
				SCAL curr = test_accuracy(false);
				if (curr < accuracy)
				{
					SCAL t = test_time();
					if (t < best_time) { best_r = r; best_p = p; best_accuracy = curr; best_time = t; }
				}
				std::cout << '.' << std::flush;
			}
		if (best_time == FLT_MAX) { std::cout << "\nOptimization failed!" << std::endl; return -1; }
		set_opts([&](nbco_opts &c) { c.tree_radius = best_r; c.fmm_order = best_p; });
		std::cout << "\nBest parameters: r = " << best_r << ", p = " << best_p << ", time = " << best_time << ", error = " << best_accuracy << std::endl;
	}

	if (test)   // main3.cu:790-811
	{
		set_opts([](nbco_opts &c) { c.unsort = 0; });
		nbco_opts cur;
		nbco_get_opts(ctx(), &cur);
		std::cout << cur.fmm_order << ": Average time: " << test_time(1) << " [s]" << std::endl;
		HIPCHK(hipMemcpy(d_buf, buf, cpyBytes, hipMemcpyHostToDevice));   // restore the caller's order for the error table
		for (int p = 1; p <= 10; ++p)
		{
			set_opts([&](nbco_opts &c) { c.unsort = 1; c.fmm_order = p; });
			std::cout << p << ": Relative error: " << test_accuracy(p == 1) << std::endl;
		}
	}
	else if (test2)   // main3.cu:812-831
	{
		set_opts([](nbco_opts &c) { c.unsort = 0; });
		nbco_opts cur;
		nbco_get_opts(ctx(), &cur);
		for (int i = 0; i < cur.tree_steps + 1; ++i)
		{
			SCAL relerr = test_accuracy(false);
			pre_symplectic_euler(add_elastic, d_buf, nBodies, d_par + 3, dt, step);
			std::cout << "Relative error after " << i << " steps: " << relerr << std::endl;
		}
	}
	else   // main3.cu:832-874
	{
		set_opts([](nbco_opts &c) { c.unsort = 0; });
		compute_force(coulombOscillatorFMMKD3, d_buf, nBodies, d_par);
		for (int iter = 0; iter < nIters; ++iter)
		{
			symp_integ(coulombOscillatorFMMKD3, d_buf, nBodies, d_par, dt, step, 1);
			if (iter % nSteps == 0)
			{
				std::cout << iter << ' ' << std::flush;
				HIPCHK(hipMemcpy(buf, d_buf, cpyBytes, hipMemcpyDeviceToHost));   // acc not copied
				std::ofstream fout(strout + "/out" + std::to_string(iter) + '_' + std::to_string(dt) + ".bin", std::ios::out | std::ios::binary);
				if (!fout)
				{
					std::cerr << "Error: cannot write on output location. Check that \"" << strout << "\" folder exists. Create it if not." << std::endl;
					return -1;
				}
				fout.write(c_buf.data(), (std::streamsize)cpyBytes);
			}
		}
		std::cout << std::endl;
	}

	nbco_destroy(ctx());
	if (d_tmp) HIPCHK(hipFree(d_tmp));
	HIPCHK(hipFree(d_buf));
	HIPCHK(hipFree(d_par));
	return 0;
}
