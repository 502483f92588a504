// pair_ceiling.hip -- measured VALU issue ceiling of the Coulomb pair body on gfx950 (MI355X).
//
// The near-field kernels (csrc/k_p2p.hpp, csrc/k_direct.hip) evaluate one directed pair with 13 vector instructions:
// 3 v_sub, 3 v_fma (r^2 + eps^2), v_rsq, 2 v_mul (r^-3), 3 v_fma (accumulate).  This program times that body on
// REGISTER operands only (no LDS, no global loads inside the loop) at 1..8 waves per SIMD, next to the raw issue
// rates of v_fma_f32 and v_rsq_f32, so that DESIGN.md can quote a measured ceiling instead of an assumed v_rsq cost.
// In-kernel cycles come from s_memtime (shader clock), wall time from HIP events; both are printed.
//
//   hipcc --offload-arch=gfx950 -O3 -o pair_ceiling tools/pair_ceiling.hip && ./pair_ceiling [json-out]
//
// Variants:
//   fma      48 independent v_fma_f32 per iteration
//   rsq      16 independent v_rsq_f32 per iteration
//   fma+rsq  12 v_fma : 1 v_rsq, independent (does the transcendental unit overlap the FMA pipe?)
//   pair     the 13-instruction body, 8 sources per iteration, one target per lane (production shape)
//   pair4    the same with 4 targets per lane (direct_tiles shape: more independent chains per wave)
//   pair_pk, pair4_pk   the packed form of that body (two sources per v_pk_* instruction): what the kernels run since round 3
//   mutual   Newton-III form: r^-3 once per unordered pair, +d*w to the lane's target and -d*w to the source's
//            accumulator held in registers (16 instructions per 2 directed pairs; the cross-lane traffic a real
//            mutual kernel needs is NOT included: this is its upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <unistd.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kSrc = 8;

// the compiler must believe the value changed (so nothing is hoisted out of the loop) without emitting an instruction
#define OPAQUE(v) asm volatile("" : "+v"(v))

#define PAIR(PX, PY, PZ, SX, SY, SZ, AX, AY, AZ)                       \
	{                                                                  \
		float dx = (PX) - (SX), dy = (PY) - (SY), dz = (PZ) - (SZ);    \
		float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));     \
		float ri = __builtin_amdgcn_rsqf(r2);                          \
		float w = ri * ri * ri;                                        \
		AX = fmaf(dx, w, AX); AY = fmaf(dy, w, AY); AZ = fmaf(dz, w, AZ); \
	}

struct Stamp { unsigned long long cyc; };

__global__ __launch_bounds__(256) void k_fma(const float *in, float *out, Stamp *st, int iters)
{
	float a[48];
	const float x = in[threadIdx.x & 63], y = in[64 + (threadIdx.x & 63)];
#pragma unroll
	for (int k = 0; k < 48; ++k) a[k] = x + k;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it)
	{
#pragma unroll
		for (int k = 0; k < 48; ++k) a[k] = fmaf(a[k], x, y);
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
#pragma unroll
	for (int k = 0; k < 48; ++k) s += a[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)].cyc = t1 - t0;
}

__global__ __launch_bounds__(256) void k_rsq(const float *in, float *out, Stamp *st, int iters)
{
	float a[16];
	const float x = in[threadIdx.x & 63];
#pragma unroll
	for (int k = 0; k < 16; ++k) a[k] = x + k + 1.f;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it)
	{
#pragma unroll
		for (int k = 0; k < 16; ++k) a[k] = __builtin_amdgcn_rsqf(a[k]);
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
#pragma unroll
	for (int k = 0; k < 16; ++k) s += a[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)].cyc = t1 - t0;
}

// 12 independent FMAs for every independent rsq, 4 groups per iteration
__global__ __launch_bounds__(256) void k_mix(const float *in, float *out, Stamp *st, int iters)
{
	float a[48], r[4];
	const float x = in[threadIdx.x & 63], y = in[64 + (threadIdx.x & 63)];
#pragma unroll
	for (int k = 0; k < 48; ++k) a[k] = x + k;
#pragma unroll
	for (int k = 0; k < 4; ++k) r[k] = x + k + 1.f;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it)
	{
#pragma unroll
		for (int g = 0; g < 4; ++g)
		{
#pragma unroll
			for (int k = 0; k < 12; ++k) a[12 * g + k] = fmaf(a[12 * g + k], x, y);
			r[g] = __builtin_amdgcn_rsqf(r[g]);
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
#pragma unroll
	for (int k = 0; k < 48; ++k) s += a[k];
#pragma unroll
	for (int k = 0; k < 4; ++k) s += r[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)].cyc = t1 - t0;
}

template <int TGT>
__global__ __launch_bounds__(256) void k_pair(const float *in, float *out, Stamp *st, int iters, float eps2)
{
	float sx[kSrc], sy[kSrc], sz[kSrc];
	float px[TGT], py[TGT], pz[TGT], ax[TGT], ay[TGT], az[TGT];
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int k = 0; k < kSrc; ++k) { sx[k] = in[128 + 3 * k]; sy[k] = in[129 + 3 * k]; sz[k] = in[130 + 3 * k]; }
#pragma unroll
	for (int t = 0; t < TGT; ++t) { px[t] = in[lane] + t; py[t] = in[64 + lane] - t; pz[t] = in[lane] * 0.5f + t; ax[t] = ay[t] = az[t] = 0.f; }
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it)
	{
#pragma unroll
		for (int k = 0; k < kSrc; ++k)
		{
			OPAQUE(sx[k]); OPAQUE(sy[k]); OPAQUE(sz[k]);   // a new source every time, as far as the compiler can tell
#pragma unroll
			for (int t = 0; t < TGT; ++t) PAIR(px[t], py[t], pz[t], sx[k], sy[k], sz[k], ax[t], ay[t], az[t])
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
#pragma unroll
	for (int t = 0; t < TGT; ++t) s += ax[t] + ay[t] + az[t];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)].cyc = t1 - t0;
}

// The packed form of the pair body (two sources per v_pk_* instruction; what p2p_kernel and direct_tiles run since round 3)
typedef float v2f __attribute__((ext_vector_type(2)));
#define PAIR2(PX, PY, PZ, SX, SY, SZ, AX, AY, AZ)                                                                       \
	{                                                                                                                  \
		const v2f dx = (PX) - (SX), dy = (PY) - (SY), dz = (PZ) - (SZ);                                                \
		const v2f r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, e2))); \
		v2f ri;                                                                                                        \
		ri.x = __builtin_amdgcn_rsqf(r2.x);                                                                            \
		ri.y = __builtin_amdgcn_rsqf(r2.y);                                                                            \
		const v2f ri3 = ri * ri * ri;                                                                                  \
		AX = __builtin_elementwise_fma(dx, ri3, AX); AY = __builtin_elementwise_fma(dy, ri3, AY); AZ = __builtin_elementwise_fma(dz, ri3, AZ); \
	}
template <int TGT>
__global__ __launch_bounds__(256) void k_pair_pk(const float *in, float *out, Stamp *st, int iters, float eps2)
{
	static_assert(kSrc % 2 == 0, "sources in pairs");
	v2f sx[kSrc / 2], sy[kSrc / 2], sz[kSrc / 2];
	v2f px[TGT], py[TGT], pz[TGT], ax[TGT], ay[TGT], az[TGT];
	const v2f e2 = {eps2, eps2};
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int k = 0; k < kSrc / 2; ++k)
	{
		sx[k] = v2f{in[128 + 6 * k], in[131 + 6 * k]}; sy[k] = v2f{in[129 + 6 * k], in[132 + 6 * k]}; sz[k] = v2f{in[130 + 6 * k], in[133 + 6 * k]};
	}
#pragma unroll
	for (int t = 0; t < TGT; ++t)
	{
		const float a = in[lane] + t, b = in[64 + lane] - t, c = in[lane] * 0.5f + t;
		px[t] = v2f{a, a}; py[t] = v2f{b, b}; pz[t] = v2f{c, c};
		ax[t] = ay[t] = az[t] = v2f{0.f, 0.f};
	}
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it)
	{
#pragma unroll
		for (int k = 0; k < kSrc / 2; ++k)
		{
			OPAQUE(sx[k].x); OPAQUE(sx[k].y); OPAQUE(sy[k].x); OPAQUE(sy[k].y); OPAQUE(sz[k].x); OPAQUE(sz[k].y);
#pragma unroll
			for (int t = 0; t < TGT; ++t) PAIR2(px[t], py[t], pz[t], sx[k], sy[k], sz[k], ax[t], ay[t], az[t])
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = 0;
#pragma unroll
	for (int t = 0; t < TGT; ++t) s += ax[t].x + ax[t].y + ay[t].x + ay[t].y + az[t].x + az[t].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)].cyc = t1 - t0;
}

// Newton-III: one r^-3 serves the pair (target of this lane, source k) in both directions; the source accumulators live in
// this lane's registers (a real kernel rotates them across lanes)
__global__ __launch_bounds__(256) void k_mutual(const float *in, float *out, Stamp *st, int iters, float eps2)
{
	float sx[kSrc], sy[kSrc], sz[kSrc], bx[kSrc], by[kSrc], bz[kSrc];
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int k = 0; k < kSrc; ++k) { sx[k] = in[128 + 3 * k]; sy[k] = in[129 + 3 * k]; sz[k] = in[130 + 3 * k]; bx[k] = by[k] = bz[k] = 0.f; }
	float px = in[lane], py = in[64 + lane], pz = in[lane] * 0.5f, ax = 0.f, ay = 0.f, az = 0.f;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it)
	{
#pragma unroll
		for (int k = 0; k < kSrc; ++k)
		{
			OPAQUE(sx[k]); OPAQUE(sy[k]); OPAQUE(sz[k]);
			float dx = px - sx[k], dy = py - sy[k], dz = pz - sz[k];
			float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));
			float ri = __builtin_amdgcn_rsqf(r2);
			float w = ri * ri * ri;
			ax = fmaf(dx, w, ax); ay = fmaf(dy, w, ay); az = fmaf(dz, w, az);
			bx[k] = fmaf(-dx, w, bx[k]); by[k] = fmaf(-dy, w, by[k]); bz[k] = fmaf(-dz, w, bz[k]);
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	float s = ax + ay + az;
#pragma unroll
	for (int k = 0; k < kSrc; ++k) s += bx[k] + by[k] + bz[k];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)].cyc = t1 - t0;
}

// The mutual body as a real kernel issues it: lanes of a 16-lane row hold one target and one source each; at step s a lane
// meets the source of the lane s places away (DPP row_ror on the operands, no data movement instructions), and its
// contribution to that source travels back the same way into the owner's accumulator (r^-3 rotated once, the distance on the DPP operand
// of the multiply-adds): 16 instructions per 2 directed pairs.  This is the step of csrc/k_p2p.hpp's p2p_mutual_kernel.
template <int S> __device__ __forceinline__ float row_ror(float v)
{
	if constexpr (S == 0) return v;
	else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + S, 0xF, 0xF, false));
}
template <int S>
__device__ __forceinline__ void mutual_step(float px, float py, float pz, float sx, float sy, float sz, float eps2, float &ax, float &ay, float &az,
                                            float &bx, float &by, float &bz)
{
	const float dx = row_ror<S>(sx) - px, dy = row_ror<S>(sy) - py, dz = row_ror<S>(sz) - pz;   // source - target
	const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));
	const float ri = __builtin_amdgcn_rsqf(r2);
	const float w = ri * ri * ri;
	ax = fmaf(-dx, w, ax); ay = fmaf(-dy, w, ay); az = fmaf(-dz, w, az);
	if constexpr (S == 0) { bx = fmaf(dx, w, bx); by = fmaf(dy, w, by); bz = fmaf(dz, w, bz); }
	else
	{
		float wr;   // (see csrc/k_p2p.hpp: one asm block, the s_nop covers the DPP read-after-write wait states)
		asm volatile("s_nop 1\n\t"
		             "v_mov_b32_dpp %3, %7 row_ror:%8 row_mask:0xf bank_mask:0xf\n\t"
		             "v_fmac_f32_dpp %0, %4, %3 row_ror:%8 row_mask:0xf bank_mask:0xf\n\t"
		             "v_fmac_f32_dpp %1, %5, %3 row_ror:%8 row_mask:0xf bank_mask:0xf\n\t"
		             "v_fmac_f32_dpp %2, %6, %3 row_ror:%8 row_mask:0xf bank_mask:0xf"
		             : "+v"(bx), "+v"(by), "+v"(bz), "=&v"(wr)
		             : "v"(dx), "v"(dy), "v"(dz), "v"(w), "n"((16 - S) % 16));
	}
}
template <int S>
__device__ __forceinline__ void mutual_steps(float px, float py, float pz, float sx, float sy, float sz, float eps2, float &ax, float &ay, float &az,
                                             float &bx, float &by, float &bz)
{
	mutual_step<S>(px, py, pz, sx, sy, sz, eps2, ax, ay, az, bx, by, bz);
	if constexpr (S + 1 < 16) mutual_steps<S + 1>(px, py, pz, sx, sy, sz, eps2, ax, ay, az, bx, by, bz);
}
__global__ __launch_bounds__(256) void k_mutual_dpp(const float *in, float *out, Stamp *st, int iters, float eps2)
{
	const int lane = threadIdx.x & 63;
	float px = in[lane], py = in[64 + lane], pz = in[lane] * 0.5f, ax = 0.f, ay = 0.f, az = 0.f;
	float sx = in[128 + lane], sy = in[192 + lane], sz = in[256 + lane], bx = 0.f, by = 0.f, bz = 0.f;
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it)
	{
		OPAQUE(sx); OPAQUE(sy); OPAQUE(sz);   // a new source block every iteration
		mutual_steps<0>(px, py, pz, sx, sy, sz, eps2, ax, ay, az, bx, by, bz);
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * blockDim.x + threadIdx.x] = ax + ay + az + bx + by + bz;
	if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)].cyc = t1 - t0;
}

struct Result { std::string name, mode; int waves; double ms, ticks_per_iter, rate; std::string unit; };

int main(int argc, char **argv)
{
	int dev = 0;
	CHK(hipSetDevice(dev));
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, dev));
	const int cus = prop.multiProcessorCount;
	const int max_blocks = cus * 8;
	float *in, *out;
	Stamp *st;
	CHK(hipMalloc(&in, 4096));
	CHK(hipMalloc(&out, sizeof(float) * 256 * (size_t)max_blocks));
	CHK(hipMalloc(&st, sizeof(Stamp) * 4 * (size_t)max_blocks));
	std::vector<float> h(1024);
	srand(7);
	for (auto &v : h) v = 0.5f + (float)(rand() & 0xFFFF) / 65536.f;
	CHK(hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0));
	CHK(hipEventCreate(&e1));
	const float eps2 = 1e-18f;
	std::vector<Result> res;
	std::vector<Stamp> hs(4 * (size_t)max_blocks);

	// sustained: one launch of `iters` iterations (tens of milliseconds: the chip settles at its clock under load);
	// burst: launches of about 0.2 ms -- the length of the near-field kernel inside a step -- one per millisecond, median time
	auto run = [&](const char *name, int waves, double units_per_iter_per_lane, const char *unit, int iters_sustained, auto launch) {
		const int grid = cus * waves;   // 256-thread blocks: one wave on each of the CU's four SIMDs, `waves` blocks per CU
		launch(grid, 64);               // warm-up
		CHK(hipDeviceSynchronize());
		CHK(hipEventRecord(e0));
		launch(grid, iters_sustained);
		CHK(hipEventRecord(e1));
		CHK(hipEventSynchronize(e1));
		float ms = 0;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		CHK(hipMemcpy(hs.data(), st, sizeof(Stamp) * 4 * (size_t)grid, hipMemcpyDeviceToHost));
		std::vector<unsigned long long> cyc(4 * (size_t)grid);
		for (size_t i = 0; i < cyc.size(); ++i) cyc[i] = hs[i].cyc;
		std::nth_element(cyc.begin(), cyc.begin() + cyc.size() / 2, cyc.end());
		const double ticks = (double)cyc[cyc.size() / 2] / iters_sustained;   // s_memtime ticks per loop iteration of one wave
		double rate = (double)grid * 256 * units_per_iter_per_lane * iters_sustained / (ms * 1e-3);
		res.push_back({name, "sustained", waves, ms, ticks, rate, unit});
		printf("%-10s sustained waves/SIMD %d  %8.3f ms  %.3e %s/s\n", name, waves, ms, rate, unit);
		// burst
		const int iters_burst = std::max(8, (int)(iters_sustained * 0.2 / ms));
		std::vector<float> t;
		for (int rep = 0; rep < 25; ++rep)
		{
			usleep(800);
			CHK(hipEventRecord(e0));
			launch(grid, iters_burst);
			CHK(hipEventRecord(e1));
			CHK(hipEventSynchronize(e1));
			float b = 0;
			CHK(hipEventElapsedTime(&b, e0, e1));
			if (rep >= 5) t.push_back(b);
		}
		std::nth_element(t.begin(), t.begin() + t.size() / 2, t.end());
		const double bms = t[t.size() / 2];
		rate = (double)grid * 256 * units_per_iter_per_lane * iters_burst / (bms * 1e-3);
		res.push_back({name, "burst", waves, bms, 0, rate, unit});
		printf("%-10s burst     waves/SIMD %d  %8.3f ms  %.3e %s/s\n", name, waves, bms, rate, unit);
	};

	for (int w : {1, 2, 4, 6, 8})
	{
		run("fma", w, 48, "fma", 20000, [&](int g, int it) { hipLaunchKernelGGL(k_fma, dim3(g), dim3(256), 0, 0, in, out, st, it); });
		run("rsq", w, 16, "rsq", 20000, [&](int g, int it) { hipLaunchKernelGGL(k_rsq, dim3(g), dim3(256), 0, 0, in, out, st, it); });
		run("fma+rsq", w, 4, "(12 fma + 1 rsq)", 20000, [&](int g, int it) { hipLaunchKernelGGL(k_mix, dim3(g), dim3(256), 0, 0, in, out, st, it); });
		run("pair", w, kSrc, "pairs", 20000, [&](int g, int it) { hipLaunchKernelGGL(k_pair<1>, dim3(g), dim3(256), 0, 0, in, out, st, it, eps2); });
		run("pair4", w, 4 * kSrc, "pairs", 5000, [&](int g, int it) { hipLaunchKernelGGL(k_pair<4>, dim3(g), dim3(256), 0, 0, in, out, st, it, eps2); });
		run("pair_pk", w, kSrc, "pairs", 20000, [&](int g, int it) { hipLaunchKernelGGL(k_pair_pk<1>, dim3(g), dim3(256), 0, 0, in, out, st, it, eps2); });
		run("pair4_pk", w, 4 * kSrc, "pairs", 5000, [&](int g, int it) { hipLaunchKernelGGL(k_pair_pk<4>, dim3(g), dim3(256), 0, 0, in, out, st, it, eps2); });
		run("mutual", w, 2 * kSrc, "pairs", 20000, [&](int g, int it) { hipLaunchKernelGGL(k_mutual, dim3(g), dim3(256), 0, 0, in, out, st, it, eps2); });
		run("mutual_dpp", w, 2 * 16, "pairs", 10000, [&](int g, int it) { hipLaunchKernelGGL(k_mutual_dpp, dim3(g), dim3(256), 0, 0, in, out, st, it, eps2); });
	}

	if (argc > 1)
	{
		FILE *f = fopen(argv[1], "w");
		if (!f) { perror(argv[1]); return 1; }
		fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"flop_per_pair\": 20, \"peak_tflops\": 157.3,\n"
		           " \"note\": \"rates are wall-clock (HIP events); sustained = one launch of tens of ms, burst = 0.2 ms launches once per ms (median of 20)\",\n \"results\": [\n",
		        prop.gcnArchName, cus, prop.clockRate / 1000);
		for (size_t i = 0; i < res.size(); ++i)
		{
			const Result &r = res[i];
			const bool is_pair = r.unit == "pairs";
			fprintf(f, "  {\"variant\": \"%s\", \"mode\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"unit\": \"%s\", \"rate_per_s\": %.4e", r.name.c_str(), r.mode.c_str(), r.waves,
			        r.ms, r.unit.c_str(), r.rate);
			if (r.unit == "fma") fprintf(f, ", \"tflops\": %.2f, \"frac_of_157.3\": %.4f", r.rate * 2 / 1e12, r.rate * 2 / 157.3e12);
			if (is_pair) fprintf(f, ", \"tflops_at_20\": %.2f, \"frac_of_157.3\": %.4f", r.rate * 20 / 1e12, r.rate * 20 / 157.3e12);
			fprintf(f, "}%s\n", i + 1 < res.size() ? "," : "");
		}
		fprintf(f, "]}\n");
		fclose(f);
	}
	return 0;
}
