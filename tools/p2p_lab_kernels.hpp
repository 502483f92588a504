// p2p_lab_kernels.hpp -- kernels and host analysis of tools/p2p_lab.hip (diagnostics only; nothing here is linked into the library)
#pragma once
#include <functional>

namespace lab {

struct WaveStamp { unsigned long long c0, c1, r0, r1; unsigned hw, xcc; };

__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }     // HW_REG_HW_ID, 32 bits
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20); }   // HW_REG_XCC_ID

// The production kernel (csrc/k_p2p.hpp: p2p_kernel) with two clock reads per work unit.  The body is the same text; a wave
// that takes several units (never, with the library's grid) stamps each of them.
template <int TPL>
__global__ __launch_bounds__(64 * kP2PWaves) NBCO_P2P_ATTR void p2p_stamped(const float4 *__restrict__ pos, const int2 *__restrict__ desc, const int4 *__restrict__ chunk,
                                                                           const int *__restrict__ nchunks_total, float eps2, int src_max, int stride,
                                                                           float4 *__restrict__ partial, int npos, WaveStamp *__restrict__ stamp)
{
	constexpr int G = 64 / TPL;
	__shared__ __attribute__((aligned(16))) float tile_all[kP2PWaves][2][G][3 * TPL];
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane / TPL, li = lane % TPL;
	float(*tile)[G][3 * TPL] = tile_all[wv];
	const int total = *nchunks_total;
	const int cstride = gridDim.x * kP2PWaves;
	const int nchunk = (src_max + TPL - 1) / TPL;
	const float4 far = make_float4(1.e18f, 1.e18f, 1.e18f, 0.f);
	int cid = blockIdx.x * kP2PWaves + wv;
	if (cid >= total) return;
	const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	int4 ck = chunk[cid];
	float4 pt = pos[ck.x + min(li, ck.w - 1)];
	int2 dsc = lane < ck.z - ck.y ? desc[ck.y + lane] : make_int2(0, 0);
	for (; cid < total; cid += cstride)
	{
		const int it = ck.x, mt = ck.w, d0 = ck.y, d1 = ck.z;
		const int nid = cid + cstride;
		int4 nk = make_int4(0, 0, 0, 1);
		if (nid < total) nk = chunk[nid];
		float4 npt = pt;
		int2 ndsc = make_int2(0, 0);
		bool next_loaded = false;
		for (int tb = 0; tb < mt; tb += TPL)
		{
			const int ti = tb + li;
			const float4 pi = tb == 0 ? pt : pos[it + (ti < mt ? ti : mt - 1)];
			float ax = 0.f, ay = 0.f, az = 0.f;
			for (int eb = d0; eb < d1; eb += 64)
			{
				const int nent = min(64, d1 - eb);
				const int2 mine = (eb == d0) ? dsc : ((lane < nent) ? desc[eb + lane] : make_int2(0, 0));
				const int ntile = (nent + G - 1) / G;
				auto fetch = [&](int et, int jc) -> float4 {
					const int ent = et * G + g;
					const int is = __shfl(mine.x, ent), ms = __shfl(mine.y, ent);
					const int j = jc * TPL + li;
					return (ent < nent && j < ms) ? pos[is + j] : far;
				};
				float4 cur = fetch(0, 0);
				int b = 0;
				for (int et = 0; et < ntile; ++et)
					for (int jc = 0; jc < nchunk; ++jc)
					{
						tile[b][g][3 * li] = cur.x; tile[b][g][3 * li + 1] = cur.y; tile[b][g][3 * li + 2] = cur.z;
						int jn = jc + 1, en = et;
						if (jn == nchunk) { jn = 0; ++en; }
						if (en < ntile) cur = fetch(en, jn);
						else if (!next_loaded && nid < total)
						{
							npt = pos[nk.x + min(li, nk.w - 1)];
							ndsc = lane < nk.z - nk.y ? desc[nk.y + lane] : make_int2(0, 0);
							next_loaded = true;
						}
						wave_lds_sync();
						const float4 *t4 = reinterpret_cast<const float4 *>(tile[b][g]);
						float tx = 0.f, ty = 0.f, tz = 0.f;
#pragma unroll 2
						for (int q4 = 0; q4 < TPL / 4; ++q4)
						{
							const float4 A = t4[3 * q4], B = t4[3 * q4 + 1], C = t4[3 * q4 + 2];
							P2P_PAIR(A.x, A.y, A.z)
							P2P_PAIR(A.w, B.x, B.y)
							P2P_PAIR(B.z, B.w, C.x)
							P2P_PAIR(C.y, C.z, C.w)
						}
						ax += tx; ay += ty; az += tz;
						wave_lds_sync();
						b ^= 1;
					}
			}
#pragma unroll
			for (int o = TPL; o < 64; o <<= 1)
			{
				ax += __shfl_xor(ax, o);
				ay += __shfl_xor(ay, o);
				az += __shfl_xor(az, o);
			}
			if (g == 0 && ti < mt) partial[(size_t)cid * stride + ti] = make_float4(ax, ay, az, 0.f);
		}
		if (!next_loaded && nid < total)
		{
			npt = pos[nk.x + min(li, nk.w - 1)];
			ndsc = lane < nk.z - nk.y ? desc[nk.y + lane] : make_int2(0, 0);
		}
		if (lane == 0)
		{
			WaveStamp s;
			s.c0 = c0; s.r0 = r0; s.c1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
			s.hw = hw_id(); s.xcc = xcc_id();
			stamp[cid] = s;
		}
		ck = nk; pt = npt; dsc = ndsc;
	}
}

// what the stamps say
static void timeline_report(const std::vector<WaveStamp> &st, const std::vector<long long> &steps_of, FILE *js, const char *what = "production")
{
	unsigned long long rmin = ~0ull, rmax = 0;
	double dc = 0, dr = 0;
	long long steps = 0;
	std::map<unsigned long long, std::vector<size_t>> by_simd;
	for (size_t i = 0; i < st.size(); ++i)
	{
		const WaveStamp &s = st[i];
		if (s.r1 == 0) continue;
		rmin = std::min(rmin, s.r0); rmax = std::max(rmax, s.r1);
		dc += (double)(s.c1 - s.c0); dr += (double)(s.r1 - s.r0);
		steps += steps_of[i];
		// gfx9 HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
		const unsigned long long key = ((unsigned long long)(s.xcc & 0xf) << 32) | (s.hw & 0xfff0u & ~0xc0u);
		by_simd[key].push_back(i);
	}
	const double span_us = (double)(rmax - rmin) / 100.0, clock_ghz = dc / dr * 0.1;
	printf("timeline of %s:\n", what);
	printf("timeline: span %.1f us (first wave start to last wave end), %zu SIMDs seen, clock held %.3f GHz (sum of wave cycles / sum of wave 100 MHz ticks)\n", span_us,
	       by_simd.size(), clock_ghz);
	// slot occupancy over the span and cycles per step while busy
	double wave_us = 0;
	for (const WaveStamp &s : st) if (s.r1) wave_us += (double)(s.r1 - s.r0) / 100.0;
	const double slots = (double)by_simd.size() * 6;
	printf("timeline: sum of wave lifetimes %.0f us = %.1f %% of %zu SIMDs x 6 slots x span; mean wave lifetime %.2f us, %.1f steps per wave\n", wave_us,
	       100.0 * wave_us / (slots * span_us), by_simd.size(), wave_us / (double)st.size(), (double)steps / (double)st.size());
	// per SIMD: busy time = union of its waves' intervals; cycles per step = busy cycles / steps run there
	double busy_us = 0, last_end_sum = 0;
	std::vector<double> ends, cps;
	for (auto &kv : by_simd)
	{
		std::vector<std::pair<unsigned long long, unsigned long long>> iv;
		long long ssteps = 0;
		for (size_t i : kv.second) { iv.push_back({st[i].r0, st[i].r1}); ssteps += steps_of[i]; }
		std::sort(iv.begin(), iv.end());
		unsigned long long cur0 = iv[0].first, cur1 = iv[0].second, tot = 0;
		for (auto &p : iv) { if (p.first > cur1) { tot += cur1 - cur0; cur0 = p.first; cur1 = p.second; } else cur1 = std::max(cur1, p.second); }
		tot += cur1 - cur0;
		busy_us += (double)tot / 100.0;
		ends.push_back((double)(cur1 - rmin) / 100.0);
		if (ssteps > 0) cps.push_back((double)tot / 100.0 * 1e-6 * clock_ghz * 1e9 / (double)ssteps);
	}
	std::sort(ends.begin(), ends.end());
	std::sort(cps.begin(), cps.end());
	printf("timeline: a SIMD is busy (>= 1 wave resident) %.1f %% of the span on average; SIMDs finish at %.1f / %.1f / %.1f us (10 %% / median / last)\n",
	       100.0 * busy_us / ((double)by_simd.size() * span_us), ends[ends.size() / 10], ends[ends.size() / 2], ends.back());
	printf("timeline: shader cycles per 64-pair step while a SIMD is busy: %.1f / %.1f / %.1f (10 %% / median / 90 %% of SIMDs); ideal 26 (11 x 2 + v_rsq 4)\n",
	       cps[cps.size() / 10], cps[cps.size() / 2], cps[cps.size() * 9 / 10]);
	// how many waves are resident over time (20 bins)
	const int bins = 20;
	std::vector<double> res(bins, 0.0);
	for (const WaveStamp &s : st)
	{
		if (!s.r1) continue;
		const double a = (double)(s.r0 - rmin) / (double)(rmax - rmin) * bins, b = (double)(s.r1 - rmin) / (double)(rmax - rmin) * bins;
		for (int k = 0; k < bins; ++k) res[k] += std::max(0.0, std::min(b, k + 1.0) - std::max(a, (double)k));
	}
	printf("timeline: resident waves per SIMD over the span (20 bins):");
	for (int k = 0; k < bins; ++k) printf(" %.1f", res[k] / (double)by_simd.size());
	printf("\n");
	if (js)
	{
		fprintf(js, "  {\"timeline\": {\"span_us\": %.2f, \"simds\": %zu, \"clock_ghz\": %.4f, \"slot_occupancy\": %.4f, \"simd_busy_frac\": %.4f, \"cycles_per_step_median\": %.2f, \"resident_waves_per_simd\": [",
		        span_us, by_simd.size(), clock_ghz, wave_us / (slots * span_us), busy_us / ((double)by_simd.size() * span_us), cps[cps.size() / 2]);
		for (int k = 0; k < bins; ++k) fprintf(js, "%s%.2f", k ? ", " : "", res[k] / (double)by_simd.size());
		fprintf(js, "]}},\n");
	}
}

struct Args { const float4 *pos; const int2 *desc; const int4 *chunk; const int *total; float eps2; int src_max, stride; float4 *partial; int npos, nchunks; WaveStamp *stamp; };
struct Candidate { const char *name; const char *note; std::function<void(const Args &)> launch; bool stamps; };

// ---- candidates ------------------------------------------------------------------------------------------------------------
#include "p2p_gen_lab.inc"   // csrc/gen_p2p.py <file> lab: the library's blocks + the diagnostic ones

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p; }

// (v1) the production kernel with the pair loop of a tile as the generated, parity-aware instruction stream (csrc/gen_p2p.py)
// MODE (timing experiments with WRONG sums): 1 = no LDS reads behind the first four sources of a tile, 2 = no source fetch / staging
template <int WAVES_PER_EU, bool WIDE, int MODE = 0>
__global__ __launch_bounds__(64 * kP2PWaves) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU))) void p2p_v1(
	const float4 *__restrict__ pos, const int2 *__restrict__ desc, const int4 *__restrict__ chunk, const int *__restrict__ nchunks_total, float eps2, int src_max,
	int stride, float4 *__restrict__ partial, int npos, WaveStamp *__restrict__ stamp)
{
	constexpr int TPL = 32, G = 2;
	__shared__ __attribute__((aligned(16))) float tile_all[kP2PWaves][2][G][3 * TPL];
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane / TPL, li = lane % TPL;
	float(*tile)[G][3 * TPL] = tile_all[wv];
	const int total = *nchunks_total;
	const int cstride = gridDim.x * kP2PWaves;
	const float4 far = make_float4(1.e18f, 1.e18f, 1.e18f, 0.f);
	int cid = blockIdx.x * kP2PWaves + wv;
	if (cid >= total) return;
	unsigned long long c0 = 0, r0 = 0;
	if (stamp) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
	int4 ck = chunk[cid];
	float4 pt = pos[ck.x + min(li, ck.w - 1)];
	int2 dsc = lane < ck.z - ck.y ? desc[ck.y + lane] : make_int2(0, 0);
	for (; cid < total; cid += cstride)
	{
		const int it = ck.x, mt = ck.w, d0 = ck.y, d1 = ck.z;
		const int nid = cid + cstride;
		int4 nk = make_int4(0, 0, 0, 1);
		if (nid < total) nk = chunk[nid];
		float4 npt = pt;
		int2 ndsc = make_int2(0, 0);
		bool next_loaded = false;
		const float4 pi = pt;
		float ax = 0.f, ay = 0.f, az = 0.f;
		for (int eb = d0; eb < d1; eb += 64)
		{
			const int nent = min(64, d1 - eb);
			const int2 mine = (eb == d0) ? dsc : ((lane < nent) ? desc[eb + lane] : make_int2(0, 0));
			const int ntile = (nent + G - 1) / G;
			auto fetch = [&](int et) -> float4 {
				const int ent = et * G + g;
				const int is = __shfl(mine.x, ent), ms = __shfl(mine.y, ent);
				return (ent < nent && li < ms) ? pos[is + li] : far;
			};
			float4 cur = fetch(0);
			int b = 0;
			for (int et = 0; et < ntile; ++et)
			{
				if (MODE != 2 || et == 0)
				{
					tile[b][g][3 * li] = cur.x; tile[b][g][3 * li + 1] = cur.y; tile[b][g][3 * li + 2] = cur.z;
				}
				if (et + 1 < ntile) { if (MODE != 2) cur = fetch(et + 1); }
				else if (!next_loaded && nid < total)
				{
					npt = pos[nk.x + min(li, nk.w - 1)];
					ndsc = lane < nk.z - nk.y ? desc[nk.y + lane] : make_int2(0, 0);
					next_loaded = true;
				}
				float tx = 0.f, ty = 0.f, tz = 0.f;
				if constexpr (MODE == 1) p2p_tile_lab_noread(lds_addr(&tile[b][g][0]), pi.x, pi.y, pi.z, eps2, tx, ty, tz);
				else if constexpr (WIDE) p2p_tile_gen32_wide(lds_addr(&tile[b][g][0]), pi.x, pi.y, pi.z, eps2, tx, ty, tz);
				else p2p_tile_gen32(lds_addr(&tile[b][g][0]), pi.x, pi.y, pi.z, eps2, tx, ty, tz);
				ax += tx; ay += ty; az += tz;
				b ^= 1;
			}
		}
		ax += __shfl_xor(ax, 32);
		ay += __shfl_xor(ay, 32);
		az += __shfl_xor(az, 32);
		if (g == 0 && li < mt) partial[(size_t)cid * stride + li] = make_float4(ax, ay, az, 0.f);
		if (!next_loaded && nid < total)
		{
			npt = pos[nk.x + min(li, nk.w - 1)];
			ndsc = lane < nk.z - nk.y ? desc[nk.y + lane] : make_int2(0, 0);
		}
		if (stamp && lane == 0)
		{
			WaveStamp s;
			s.c0 = c0; s.r0 = r0; s.c1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
			s.hw = hw_id(); s.xcc = xcc_id();
			stamp[cid] = s;
		}
		ck = nk; pt = npt; dsc = ndsc;
	}
}

template <class K> static void launch_like_production(K kernel, const Args &a)
{
	const int grid = (a.nchunks + kP2PWaves - 1) / kP2PWaves;
	hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * kP2PWaves), 0, 0, a.pos, a.desc, a.chunk, a.total, a.eps2, a.src_max, a.stride, a.partial, a.npos, a.stamp);
}

static std::vector<Candidate> candidates()
{
	std::vector<Candidate> v;
	v.push_back({"v1 2 in flight, 6 w/SIMD", "production structure, pair loop = csrc/gen_p2p.py", [](const Args &a) { launch_like_production(p2p_v1<6, false>, a); }, true});
	v.push_back({"v1 2 in flight, 5 w/SIMD", "", [](const Args &a) { launch_like_production(p2p_v1<5, false>, a); }, true});
	v.push_back({"v1 6 w/SIMD, NO LDS READS", "wrong sums: timing experiment", [](const Args &a) { launch_like_production(p2p_v1<6, false, 1>, a); }, true});
	v.push_back({"v1 6 w/SIMD, NO FETCH", "wrong sums: timing experiment", [](const Args &a) { launch_like_production(p2p_v1<6, false, 2>, a); }, true});
	v.push_back({"v1 4 in flight, 5 w/SIMD", "", [](const Args &a) { launch_like_production(p2p_v1<5, true>, a); }, false});
	v.push_back({"v1 4 in flight, 4 w/SIMD", "", [](const Args &a) { launch_like_production(p2p_v1<4, true>, a); }, false});
	return v;
}

} // namespace lab
