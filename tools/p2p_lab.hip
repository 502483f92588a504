// p2p_lab.hip -- the near-field launch of one real evaluation, replayed outside the library (diagnostics only).
//
//   NBCO_P2P_DUMP=/tmp/p2p.bin python3 tools/p2p_dump.py          (writes the launch's inputs + partial sums, csrc/k_fmm_kd.hip: p2p_dump)
//   ./build/p2p_lab /tmp/p2p.bin [json-out]
//
// What it does: (1) runs the production kernel (csrc/k_p2p.hpp, included as is) on the dumped arrays and checks its partial
// sums against the dumped ones bit for bit; (2) times it the way the step sees it -- launches of ~0.2 ms with idle gaps -- by HIP
// events; (3) runs a copy with two clock reads per wave (s_memtime = shader clock, s_memrealtime = 100 MHz) and the hardware
// id of the SIMD it ran on, and prints what only a timeline shows: the clock the chip holds during the launch, how full the
// wave slots are over the launch, the cycles a SIMD spends per 64 pairs while it is busy, and how long the tail is;
// (4) runs the candidate kernels below on the same arrays, checks them against the production sums and times them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <map>
#include <array>
#include <string>
#include <algorithm>
#include <unistd.h>

#include "nbco_internal.hpp"
#include "k_p2p.hpp"
#include "p2p_lab_kernels.hpp"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Dump
{
	long long n = 0, entries = 0, chunks = 0, mlt_max = 0, stride = 0;
	std::vector<float4> pos, partial;
	std::vector<int2> desc;
	std::vector<int4> chunk;
};

static Dump load(const char *path)
{
	Dump d;
	FILE *f = fopen(path, "rb");
	if (!f) { perror(path); exit(1); }
	long long head[8];
	if (fread(head, sizeof head, 1, f) != 1) { fprintf(stderr, "short header\n"); exit(1); }
	d.n = head[0]; d.entries = head[1]; d.chunks = head[2]; d.mlt_max = head[3]; d.stride = head[4];
	d.pos.resize(d.n); d.desc.resize(d.entries); d.chunk.resize(d.chunks); d.partial.resize(d.chunks * d.stride);
	auto get = [&](void *p, size_t b) { if (fread(p, 1, b, f) != b) { fprintf(stderr, "short file\n"); exit(1); } };
	get(d.pos.data(), sizeof(float4) * d.pos.size());
	get(d.desc.data(), sizeof(int2) * d.desc.size());
	get(d.chunk.data(), sizeof(int4) * d.chunk.size());
	get(d.partial.data(), sizeof(float4) * d.partial.size());
	fclose(f);
	return d;
}

struct Timing { double median_ms, min_ms; };

template <class F> static Timing time_bursts(F launch, int reps = 30)
{
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	for (int i = 0; i < 3; ++i) launch();
	CHK(hipDeviceSynchronize());
	std::vector<float> t;
	for (int r = 0; r < reps; ++r)
	{
		usleep(600);   // the rest of a step: the chip idles between two near-field launches
		CHK(hipEventRecord(e0));
		launch();
		CHK(hipEventRecord(e1));
		CHK(hipEventSynchronize(e1));
		float ms = 0;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		t.push_back(ms);
	}
	std::sort(t.begin(), t.end());
	CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
	return {t[t.size() / 2], t[0]};
}

int main(int argc, char **argv)
{
	if (argc < 2) { fprintf(stderr, "usage: p2p_lab dump.bin [out.json]\n"); return 2; }
	CHK(hipSetDevice(0));
	Dump d = load(argv[1]);
	// directed pairs and 64-pair wave steps of the launch
	double pairs = 0;
	long long wave_steps = 0;
	std::vector<long long> steps_of(d.chunks);
	for (long long c = 0; c < d.chunks; ++c)
	{
		const int4 k = d.chunk[c];
		long long src = 0;
		for (int e = k.y; e < k.z; ++e) src += d.desc[e].y;
		pairs += (double)src * k.w;
		// the production kernel: two 32-lane groups walk the entries, a tile is 32 sources
		const long long tiles = ((k.z - k.y) + 1) / 2;
		steps_of[c] = tiles * 32;
		wave_steps += steps_of[c];
	}
	printf("dump: n %lld, entries %lld, chunks %lld, mlt_max %lld; directed pairs %.4e, wave steps (64 lane pairs each) %lld = %.4e lane pairs (%.1f %% useful)\n", d.n,
	       d.entries, d.chunks, d.mlt_max, pairs, wave_steps, 64.0 * wave_steps, 100.0 * pairs / (64.0 * wave_steps));

	// how much of their source lists do neighbouring target leaves share?  (a wave that serves K sibling leaves walks the UNION of
	// their lists; a source leaf that only some of them name leaves the others' lanes idle)
	{
		std::map<int, std::vector<int>> list_of;   // target first particle -> source first particles
		for (long long c = 0; c < d.chunks; ++c)
			for (int e = d.chunk[c].y; e < d.chunk[c].z; ++e) list_of[d.chunk[c].x].push_back(d.desc[e].x);
		std::vector<int> firsts;
		for (auto &kv : list_of) firsts.push_back(kv.first);
		for (int K : {2, 4, 8})
		{
			double sum_lists = 0, sum_union = 0;
			for (size_t i = 0; i + K <= firsts.size(); i += K)
			{
				std::vector<int> u;
				for (int k = 0; k < K; ++k) { const auto &l = list_of[firsts[i + k]]; sum_lists += (double)l.size(); u.insert(u.end(), l.begin(), l.end()); }
				std::sort(u.begin(), u.end());
				u.erase(std::unique(u.begin(), u.end()), u.end());
				sum_union += (double)u.size();
			}
			printf("groups of %d consecutive target leaves: mean list %.2f entries, mean union %.2f, lane use when walking the union %.3f\n", K,
			       sum_lists / ((double)firsts.size()), sum_union / ((double)firsts.size() / K), sum_lists / (K * sum_union));
		}
	}

	float4 *pos, *partial, *partial2;
	int2 *desc;
	int4 *chunk;
	int *total;
	CHK(hipMalloc(&pos, sizeof(float4) * d.n));
	CHK(hipMalloc(&desc, sizeof(int2) * (d.entries + 64)));
	CHK(hipMalloc(&chunk, sizeof(int4) * d.chunks));
	CHK(hipMalloc(&partial, sizeof(float4) * d.chunks * d.stride));
	CHK(hipMalloc(&partial2, sizeof(float4) * d.chunks * d.stride));
	CHK(hipMalloc(&total, sizeof(int)));
	CHK(hipMemcpy(pos, d.pos.data(), sizeof(float4) * d.n, hipMemcpyHostToDevice));
	CHK(hipMemset(desc, 0, sizeof(int2) * (d.entries + 64)));
	CHK(hipMemcpy(desc, d.desc.data(), sizeof(int2) * d.entries, hipMemcpyHostToDevice));
	CHK(hipMemcpy(chunk, d.chunk.data(), sizeof(int4) * d.chunks, hipMemcpyHostToDevice));
	const int hc = (int)d.chunks;
	CHK(hipMemcpy(total, &hc, sizeof(int), hipMemcpyHostToDevice));
	const float eps2 = 1e-18f;
	const int stride = (int)d.stride, src_max = (int)d.mlt_max, npos = (int)d.n;
	if (d.mlt_max > 32 || d.mlt_max <= 16) { fprintf(stderr, "the lab covers the 32-wide target group only (mlt_max %lld)\n", d.mlt_max); return 1; }

	std::vector<float4> ref(d.partial.size()), got(d.partial.size());
	auto fetch = [&](float4 *dev, std::vector<float4> &h) { CHK(hipMemcpy(h.data(), dev, sizeof(float4) * h.size(), hipMemcpyDeviceToHost)); };
	// which slots of `partial` are written: target ti < k.w of every chunk
	auto compare = [&](const std::vector<float4> &a, const std::vector<float4> &b, double &max_rel, long long &nbits) {
		max_rel = 0; nbits = 0;
		double scale = 0; long long cnt = 0;
		for (long long c = 0; c < d.chunks; ++c)
			for (int t = 0; t < d.chunk[c].w; ++t) { const float4 v = a[c * d.stride + t]; scale += std::sqrt((double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z); ++cnt; }
		scale /= (double)cnt;
		for (long long c = 0; c < d.chunks; ++c)
			for (int t = 0; t < d.chunk[c].w; ++t)
			{
				const float4 u = a[c * d.stride + t], v = b[c * d.stride + t];
				if (memcmp(&u, &v, 12) != 0) ++nbits;
				const double du = std::sqrt((double)(u.x - v.x) * (u.x - v.x) + (double)(u.y - v.y) * (u.y - v.y) + (double)(u.z - v.z) * (u.z - v.z));
				const double m = std::sqrt((double)u.x * u.x + (double)u.y * u.y + (double)u.z * u.z);
				max_rel = std::max(max_rel, du / (m + scale));
			}
	};

	FILE *js = argc > 2 ? fopen(argv[2], "w") : nullptr;
	if (js) fprintf(js, "{\"pairs\": %.6e, \"wave_steps\": %lld, \"chunks\": %lld, \"variants\": [\n", pairs, wave_steps, d.chunks);
	bool first = true;
	auto report = [&](const char *name, Timing t, double max_rel, long long nbits, const char *note) {
		const double tf = pairs * 20 / (t.median_ms * 1e-3) / 1e12;
		printf("%-28s median %.4f ms (min %.4f)  %.3e pairs/s  %.2f TFLOP/s = %.3f of 157.3   max rel diff %.2e, %lld sums differ in bits  %s\n", name, t.median_ms, t.min_ms,
		       pairs / (t.median_ms * 1e-3), tf, tf / 157.3, max_rel, nbits, note);
		if (js) { fprintf(js, "%s  {\"variant\": \"%s\", \"median_ms\": %.5f, \"min_ms\": %.5f, \"frac_of_157.3\": %.4f, \"max_rel_diff\": %.3e, \"sums_differing\": %lld, \"note\": \"%s\"}", first ? "" : ",\n", name, t.median_ms, t.min_ms, tf / 157.3, max_rel, nbits, note); first = false; }
	};

	// ---- (1) + (2) production kernel ------------------------------------------------------------------------------------
	const int grid = (int)((d.chunks + kP2PWaves - 1) / kP2PWaves);
	auto prod = [&]() { hipLaunchKernelGGL(p2p_kernel<32>, dim3(grid), dim3(64 * kP2PWaves), 0, 0, (const float4 *)pos, (const int2 *)desc, (const int4 *)chunk, (const int *)total, eps2, src_max, stride, partial, npos); };
	CHK(hipMemset(partial, 0, sizeof(float4) * d.chunks * d.stride));
	prod();
	CHK(hipDeviceSynchronize());
	fetch(partial, ref);
	double mr; long long nb;
	compare(d.partial, ref, mr, nb);
	printf("production kernel against the dumped partial sums: %lld sums differ, max rel %.2e\n", nb, mr);
	report("p2p_kernel<32> (production)", time_bursts(prod), mr, nb, "csrc/k_p2p.hpp as built into the library");

	// ---- (3) timeline of the production kernel -------------------------------------------------------------------------
	{
		lab::WaveStamp *st;
		CHK(hipMalloc(&st, sizeof(lab::WaveStamp) * d.chunks));
		CHK(hipMemset(st, 0, sizeof(lab::WaveStamp) * d.chunks));
		for (int rep = 0; rep < 3; ++rep)
			hipLaunchKernelGGL(lab::p2p_stamped<32>, dim3(grid), dim3(64 * kP2PWaves), 0, 0, (const float4 *)pos, (const int2 *)desc, (const int4 *)chunk, (const int *)total, eps2, src_max, stride, partial2, npos, st);
		CHK(hipDeviceSynchronize());
		std::vector<lab::WaveStamp> hs(d.chunks);
		CHK(hipMemcpy(hs.data(), st, sizeof(lab::WaveStamp) * d.chunks, hipMemcpyDeviceToHost));
		lab::timeline_report(hs, steps_of, js);
		CHK(hipFree(st));
	}

	// ---- (4) candidates ------------------------------------------------------------------------------------------------
	lab::Args a{pos, desc, chunk, total, eps2, src_max, stride, partial2, npos, (int)d.chunks, nullptr};
	lab::WaveStamp *cst;
	CHK(hipMalloc(&cst, sizeof(lab::WaveStamp) * d.chunks));
	for (const lab::Candidate &cand : lab::candidates())
	{
		CHK(hipMemset(partial2, 0, sizeof(float4) * d.chunks * d.stride));
		cand.launch(a);
		CHK(hipDeviceSynchronize());
		fetch(partial2, got);
		compare(ref, got, mr, nb);
		report(cand.name, time_bursts([&]() { cand.launch(a); }), mr, nb, cand.note);
		if (cand.stamps)
		{
			CHK(hipMemset(cst, 0, sizeof(lab::WaveStamp) * d.chunks));
			lab::Args as = a;
			as.stamp = cst;
			for (int rep = 0; rep < 3; ++rep) cand.launch(as);
			CHK(hipDeviceSynchronize());
			std::vector<lab::WaveStamp> hs(d.chunks);
			CHK(hipMemcpy(hs.data(), cst, sizeof(lab::WaveStamp) * d.chunks, hipMemcpyDeviceToHost));
			lab::timeline_report(hs, steps_of, nullptr, cand.name);
		}
	}
	// ---- (5) the order in which the work units are dispatched --------------------------------------------------------------
	// The library dispatches them in target order; the tail of the launch (timeline above) is as long as the last waves live.
	// Same kernel, same units, other orders: longest first (what a sort by size would give) and full units first, remainders
	// behind them in target order (what two output regions of the unit table would give, no sort).
	{
		int4 *chunk2;
		CHK(hipMalloc(&chunk2, sizeof(int4) * d.chunks));
		auto run_order = [&](const char *name, const char *note, const std::vector<int> &order) {
			std::vector<int4> perm(d.chunks);
			for (long long i = 0; i < d.chunks; ++i) perm[i] = d.chunk[order[i]];
			CHK(hipMemcpy(chunk2, perm.data(), sizeof(int4) * d.chunks, hipMemcpyHostToDevice));
			auto go = [&]() { hipLaunchKernelGGL(p2p_kernel<32>, dim3(grid), dim3(64 * kP2PWaves), 0, 0, (const float4 *)pos, (const int2 *)desc, (const int4 *)chunk2, (const int *)total, eps2, src_max, stride, partial2, npos); };
			CHK(hipMemset(partial2, 0, sizeof(float4) * d.chunks * d.stride));
			go();
			CHK(hipDeviceSynchronize());
			fetch(partial2, got);
			std::vector<float4> back(got.size());
			for (long long i = 0; i < d.chunks; ++i)
				for (long long t = 0; t < d.stride; ++t) back[(size_t)order[i] * d.stride + t] = got[(size_t)i * d.stride + t];
			compare(ref, back, mr, nb);
			report(name, time_bursts(go), mr, nb, note);
		};
		std::vector<int> order(d.chunks);
		for (long long i = 0; i < d.chunks; ++i) order[i] = (int)i;
		std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return steps_of[x] > steps_of[y]; });
		run_order("units longest first", "production kernel, unit table sorted by size", order);
		std::vector<int> two;
		long long full = 0;
		for (long long i = 0; i < d.chunks; ++i) full = std::max(full, steps_of[i]);
		for (long long i = 0; i < d.chunks; ++i) if (steps_of[i] == full) two.push_back((int)i);
		for (long long i = 0; i < d.chunks; ++i) if (steps_of[i] != full) two.push_back((int)i);
		run_order("full units first", "production kernel, full units then remainders in target order", two);
		std::vector<int> rev(d.chunks);
		for (long long i = 0; i < d.chunks; ++i) rev[i] = order[d.chunks - 1 - i];
		run_order("units shortest first", "the wrong way round, for contrast", rev);
		CHK(hipFree(chunk2));
	}
	// ---- (6) equal shares: a target's entries dealt evenly over its units (17 entries -> 9 + 8 instead of 16 + 1) ----------------
	{
		// per target (first particle): entry range and particle count, from the dumped units (a target's units are consecutive)
		std::vector<int4> eq;
		for (long long c = 0; c < d.chunks;)
		{
			long long e = c;
			while (e + 1 < d.chunks && d.chunk[e + 1].x == d.chunk[c].x) ++e;
			const int y = d.chunk[c].y, z = d.chunk[e].z, cnt = z - y, n = (int)(e - c + 1);
			const int per = (cnt + n - 1) / n;
			for (int k = 0; k < n; ++k) eq.push_back(make_int4(d.chunk[c].x, std::min(y + k * per, z), std::min(y + (k + 1) * per, z), d.chunk[c].w));
			c = e + 1;
		}
		int4 *chunk3;
		CHK(hipMalloc(&chunk3, sizeof(int4) * eq.size()));
		CHK(hipMemcpy(chunk3, eq.data(), sizeof(int4) * eq.size(), hipMemcpyHostToDevice));
		auto go = [&]() { hipLaunchKernelGGL(p2p_kernel<32>, dim3(grid), dim3(64 * kP2PWaves), 0, 0, (const float4 *)pos, (const int2 *)desc, (const int4 *)chunk3, (const int *)total, eps2, src_max, stride, partial2, npos); };
		CHK(hipMemset(partial2, 0, sizeof(float4) * d.chunks * d.stride));
		go();
		CHK(hipDeviceSynchronize());
		fetch(partial2, got);
		// per-target totals (double) of both unit tables
		std::map<int, std::array<double, 3>> tot_ref, tot_got;
		double scale = 0, worst = 0;
		for (long long c = 0; c < d.chunks; ++c)
			for (int t = 0; t < d.chunk[c].w; ++t)
			{
				auto &a = tot_ref[d.chunk[c].x + t]; const float4 v = ref[c * d.stride + t]; a[0] += v.x; a[1] += v.y; a[2] += v.z;
				auto &b = tot_got[eq[c].x + t]; const float4 w = got[c * d.stride + t]; b[0] += w.x; b[1] += w.y; b[2] += w.z;
			}
		for (auto &kv : tot_ref) scale += std::sqrt(kv.second[0] * kv.second[0] + kv.second[1] * kv.second[1] + kv.second[2] * kv.second[2]);
		scale /= (double)tot_ref.size();
		for (auto &kv : tot_ref)
		{
			const auto &b = tot_got[kv.first];
			const double du = std::sqrt((kv.second[0] - b[0]) * (kv.second[0] - b[0]) + (kv.second[1] - b[1]) * (kv.second[1] - b[1]) + (kv.second[2] - b[2]) * (kv.second[2] - b[2]));
			const double m = std::sqrt(kv.second[0] * kv.second[0] + kv.second[1] * kv.second[1] + kv.second[2] * kv.second[2]);
			worst = std::max(worst, du / (m + scale));
		}
		report("equal shares", time_bursts(go), worst, -1, "production kernel, a target's entries dealt evenly over its units (per-particle totals compared)");
		CHK(hipFree(chunk3));
	}
	if (js) { fprintf(js, "\n]}\n"); fclose(js); }
	return 0;
}
