// k_p2p.hpp -- chunked near-field kernel shared by the kd-tree and the octree evaluators.
// (pair body of fmm_p2p_interaction, fmm_cart3_kdtree.cuh:767-795 / p2p3_krnl, appel.cuh:320-366)
#pragma once
#include "nbco_internal.hpp"
#include <algorithm>

namespace {

// LDS hand-off between lanes of ONE wave: DS operations of a wave execute in order, so only the compiler
// has to be kept from reordering them (no s_barrier: the waves of a block work on different items; a
// fence or __syncthreads would also drain vmcnt and with it any prefetched global data).
__device__ __forceinline__ void wave_lds_sync()
{
	asm volatile("" ::: "memory");
	__builtin_amdgcn_wave_barrier();
	asm volatile("" ::: "memory");
}

// A work unit ("chunk") is int4{first target particle, y, z, number of targets}: a target group of up to TPL consecutive
// particles (a kd leaf, or a slice of an octree cell) against the source descriptors [y, z); a source descriptor is
// (first particle, count <= src_max).  One wave per chunk at a time.  TPL lanes cover the targets, the 64/TPL lane groups walk
// different source descriptors concurrently.  Descriptors are fetched 64 at a time (one per lane) and handed out
// with shuffles; each group's source tile is prefetched into registers while the previous tile is being
// consumed, staged in a double-buffered LDS tile and read back as group-uniform ds_read_b128
// broadcasts.  Slots beyond a source range hold a far point whose r^-3 underflows to exactly 0 (3e36 <
// FLT_MAX, (3e36)^-3/2 ~ 2e-55 -> 0): the pair loop needs no predicate.  The wave stores the partial sums of
// its targets at partial[chunk * stride + target]; the consumer adds a group's chunks in list order (fixed
// order: bit-reproducible, no atomics).
//
// A chunk starts with a chain of dependent loads (record -> target position + descriptors -> first source tile): the
// record carries the target range itself, and a wave that gets more than one chunk fetches the next chunk's record,
// target and descriptors while it works on the current one.
#define P2P_PAIR(PX, PY, PZ)                                               \
	{                                                                      \
		float dx = pi.x - (PX), dy = pi.y - (PY), dz = pi.z - (PZ);        \
		float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));         \
		float ri = __builtin_amdgcn_rsqf(r2);                              \
		float ri3 = ri * ri * ri;                                          \
		tx = fmaf(dx, ri3, tx);                                            \
		ty = fmaf(dy, ri3, ty);                                            \
		tz = fmaf(dz, ri3, tz);                                            \
	}

// Two sources per lane and step in PACKED fp32 instructions (v_pk_add / v_pk_fma / v_pk_mul_f32: two fp32 operations per lane
// and issue slot).  A wave of plain v_fma_f32 reaches 0.64-0.68 of the nominal fp32 peak on this part and pulls the clock down
// to 1.9 GHz; v_pk_fma_f32 reaches 0.955 at 2.39 GHz (`tools/valu_probe.py` pk_*, `profiles/r03z_valu_probe_pk.txt`) -- the
// nominal peak IS the packed rate.  The tile keeps x, y and z of its sources in three rows, so one ds_read_b128 per coordinate
// delivers four sources as two aligned register pairs; 11 packed instructions and two v_rsq_f32 per two pairs instead of 22 + 2.
// Per pair the arithmetic is what P2P_PAIR does, bit for bit; a tile's even and odd sources are summed apart and then added.
typedef float p2p_v2f __attribute__((ext_vector_type(2)));
#define P2P_PAIR2(XS, YS, ZS)                                                                                          \
	{                                                                                                                  \
		const p2p_v2f dx = pix - (XS), dy = piy - (YS), dz = piz - (ZS);                                               \
		const p2p_v2f r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, eps2v))); \
		p2p_v2f ri;                                                                                                    \
		ri.x = __builtin_amdgcn_rsqf(r2.x);                                                                            \
		ri.y = __builtin_amdgcn_rsqf(r2.y);                                                                            \
		const p2p_v2f ri3 = ri * ri * ri;                                                                              \
		tx2 = __builtin_elementwise_fma(dx, ri3, tx2);                                                                 \
		ty2 = __builtin_elementwise_fma(dy, ri3, ty2);                                                                 \
		tz2 = __builtin_elementwise_fma(dz, ri3, tz2);                                                                 \
	}
#ifndef NBCO_P2P_PACKED
#define NBCO_P2P_PACKED 1
#endif

#ifndef NBCO_P2P_UNROLL
#define NBCO_P2P_UNROLL 2   // source quads per trip of the tile loop
#endif
#ifndef NBCO_P2P_WAVES
#define NBCO_P2P_WAVES 4
#endif
#ifndef NBCO_P2P_ATTR
// six waves per SIMD (<= 80 VGPRs instead of the 95 the allocator takes when left alone): +8 % pair rate on the kd-tree
// lists, whose chunks start with a chain of dependent loads that more resident waves hide
#define NBCO_P2P_ATTR __attribute__((amdgpu_waves_per_eu(6, 6)))
#endif
constexpr int kP2PWaves = NBCO_P2P_WAVES;   // waves per 256-thread block; 64-thread blocks would cap a CU at 8 waves

template <int TPL>
__global__ __launch_bounds__(64 * kP2PWaves) NBCO_P2P_ATTR void p2p_kernel(const float4 *__restrict__ pos, const int2 *__restrict__ desc,
                                                             const int4 *__restrict__ chunk, const int *__restrict__ nchunks_total,
                                                             float eps2, int src_max, int stride, float4 *__restrict__ partial, int npos)
{
	constexpr int G = 64 / TPL;
	// checked build: a work unit must lie inside the particle array and hold an ordered entry range, a source descriptor inside
	// the particle array; anything else is counted and replaced by an empty one
	auto sane_chunk = [&](int4 k) { return NBCO_CHECKED_OK(k.x >= 0 && k.w >= 1 && k.x + k.w <= npos && k.y <= k.z, NBCO_CHK_CHUNK) ? k : make_int4(0, 0, 0, 1); };
	auto sane_desc = [&](int2 d) { return NBCO_CHECKED_OK(d.x >= 0 && d.y >= 0 && d.x + d.y <= npos, NBCO_CHK_DESC) ? d : make_int2(0, 0); };
	// source tiles as packed xyz triplets: four sources are read with three ds_read_b128 and every loaded
	// dword is used (a float4-per-source tile is narrowed to ds_read_b96 by hipcc, twice the LDS cycles)
	__shared__ __attribute__((aligned(16))) float tile_all[kP2PWaves][2][G][3 * TPL];
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane / TPL, li = lane % TPL;
	float(*tile)[G][3 * TPL] = tile_all[wv];
	const int total = *nchunks_total;
	const int cstride = gridDim.x * kP2PWaves;
	const int nchunk = (src_max + TPL - 1) / TPL;
	const float4 far = make_float4(1.e18f, 1.e18f, 1.e18f, 0.f);

	int cid = blockIdx.x * kP2PWaves + wv;
	if (cid >= total) return;
	// header of the current chunk: record, target position (first target block), first batch of descriptors
	int4 ck = sane_chunk(chunk[cid]);
	float4 pt = pos[ck.x + min(li, ck.w - 1)];
	int2 dsc = lane < ck.z - ck.y ? sane_desc(desc[ck.y + lane]) : make_int2(0, 0);
	for (; cid < total; cid += cstride)
	{
		const int it = ck.x, mt = ck.w, d0 = ck.y, d1 = ck.z;
		// the next chunk's record now, its target and descriptors once the record has arrived (below)
		const int nid = cid + cstride;
		int4 nk = make_int4(0, 0, 0, 1);
		if (nid < total) nk = sane_chunk(chunk[nid]);
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
				const int2 mine = (eb == d0) ? dsc : ((lane < nent) ? sane_desc(desc[eb + lane]) : make_int2(0, 0));
				const int ntile = (nent + G - 1) / G;
				// tile (et, jc) covers descriptor et * G + g, source particles jc * TPL + li
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
#if NBCO_P2P_PACKED
						tile[b][g][li] = cur.x; tile[b][g][TPL + li] = cur.y; tile[b][g][2 * TPL + li] = cur.z;   // rows x | y | z
#else
						tile[b][g][3 * li] = cur.x; tile[b][g][3 * li + 1] = cur.y; tile[b][g][3 * li + 2] = cur.z;
#endif
						int jn = jc + 1, en = et;
						if (jn == nchunk) { jn = 0; ++en; }
						if (en < ntile) cur = fetch(en, jn);
						else if (!next_loaded && nid < total)
						{
							// last tile of a descriptor batch: the next chunk's record has long arrived
							npt = pos[nk.x + min(li, nk.w - 1)];
							ndsc = lane < nk.z - nk.y ? sane_desc(desc[nk.y + lane]) : make_int2(0, 0);
							next_loaded = true;
						}
						wave_lds_sync();
						const float4 *t4 = reinterpret_cast<const float4 *>(tile[b][g]);
						// a tile's contributions are summed on their own and then added to the running total: the rounding error
						// of a long fp32 sum grows with its length, and a lane sees thousands of sources per chunk when the leaves
						// are large (the reference's CPU path also accumulates per leaf pair first)
#if NBCO_P2P_PACKED
						p2p_v2f tx2 = {0.f, 0.f}, ty2 = {0.f, 0.f}, tz2 = {0.f, 0.f};
						const p2p_v2f pix = {pi.x, pi.x}, piy = {pi.y, pi.y}, piz = {pi.z, pi.z}, eps2v = {eps2, eps2};
#pragma unroll NBCO_P2P_UNROLL
						for (int q4 = 0; q4 < TPL / 4; ++q4)
						{
							const float4 X = t4[q4], Y = t4[TPL / 4 + q4], Z = t4[TPL / 2 + q4];
							P2P_PAIR2((p2p_v2f{X.x, X.y}), (p2p_v2f{Y.x, Y.y}), (p2p_v2f{Z.x, Z.y}))
							P2P_PAIR2((p2p_v2f{X.z, X.w}), (p2p_v2f{Y.z, Y.w}), (p2p_v2f{Z.z, Z.w}))
						}
						ax += tx2.x + tx2.y; ay += ty2.x + ty2.y; az += tz2.x + tz2.y;
#else
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
#endif
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
			ndsc = lane < nk.z - nk.y ? sane_desc(desc[nk.y + lane]) : make_int2(0, 0);
		}
		ck = nk; pt = npt; dsc = ndsc;
	}
}

// Grid: one wave per chunk of the host-side estimate (the hardware dispatcher balances the very uneven chunks better than a
// persistent grid with a static stride did: measured 0.20 ms against 0.24 ms); the stride loop only runs more than once
// when the estimate was too small.
template <int TPL>
static void launch_p2p(nbco_ctx *c, const float4 *pos, const int2 *desc, const int4 *chunk, const int *ntotal, long long chunks_hint, int src_max,
                       int stride, float4 *partial, long long npos)
{
	const int grid = (int)std::max<long long>(1, (chunks_hint + kP2PWaves - 1) / kP2PWaves);
	// (Six workgroups per CU, i.e. six waves per SIMD: capping them at five / four / three with unused LDS -- to leave the far-field
	// chain on the second stream more of a SIMD's registers -- made the step 1 / 2 / 5 % slower, profiles/r03c_ab_ldspad.txt)
	hipLaunchKernelGGL(p2p_kernel<TPL>, dim3(grid), dim3(64 * kP2PWaves), 0, c->stream, pos, desc, chunk, ntotal, c->o.eps2, src_max, stride, partial, (int)npos);
}


// ---- mutual (Newton III) form for leaves of 17..32 particles ----------------------------------------------------------------
// The reference's GPU pair kernel evaluates each unordered leaf pair once and applies +-d (fmm_cart3_kdtree.cuh:874-959, smem
// partner accumulators + global atomics).  Here one wave takes a leaf pair (A, B) as four 16 x 16 blocks, one per 16-lane row:
// row r holds targets A[16 (r >> 1) ..] and sources B[16 (r & 1) ..], one of each per lane, in registers.  At step s a lane
// meets the source held by the lane s places away in its row -- the rotation is a DPP modifier (row_ror) on the subtraction's
// operand, no instruction moves data -- and the contribution u = d r^-3 goes to the lane's own target (a -= u) and, rotated
// back by the same modifier, to the owner of the source (b += d w): 16 vector instructions for two directed
// pairs instead of 26, one v_rsq_f32 instead of two, and no LDS traffic at all.  Sums stay in a fixed order (bit-reproducible,
// no atomics): the wave keeps the target sums of a work unit in registers and stores them as the unit's partial sums, and it
// stores the source sums of every (A, B) as a 32-particle "reaction" record that the L2P kernel adds to B's particles in
// list order.  Entry codes (desc.w): 0 = one direction only (the leaf with itself; a source leaf of another kd-domain),
// 1 = both directions, reaction stored at react[desc.z] (desc.z = sorted position of the mirror entry, p2p_link_kernel),
// 2 = the other leaf's wave delivers this entry's sum (skipped here).
template <int S> __device__ __forceinline__ float row_ror16(float v)
{
	if constexpr (S == 0) return v;
	else return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + S, 0xF, 0xF, false));
}
template <int S, bool REACT>
__device__ __forceinline__ void mutual_steps(float px, float py, float pz, float sx, float sy, float sz, float eps2, float &ax, float &ay, float &az,
                                             float &bx, float &by, float &bz)
{
	const float dx = row_ror16<S>(sx) - px, dy = row_ror16<S>(sy) - py, dz = row_ror16<S>(sz) - pz;   // source - target
	const float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));
	const float ri = __builtin_amdgcn_rsqf(r2);
	const float w = ri * ri * ri;
	ax = fmaf(-dx, w, ax); ay = fmaf(-dy, w, ay); az = fmaf(-dz, w, az);
	if constexpr (REACT)
	{
		// the owner of the source adds d * w of the lane that met it: w is rotated once (v_mov_b32_dpp), d rides on the DPP operand
		// of three v_fmac_f32_dpp -- 16 vector instructions per step and lane for two directed pairs.  Written out as one asm
		// block: hipcc folds a DPP move into v_sub / v_add but not into the tied v_fmac.  The leading s_nop covers the two wait
		// states a DPP read needs after a VALU write of the same register (the compiler cannot see the hazard inside asm).
		if constexpr (S == 0)
		{
			bx = fmaf(dx, w, bx); by = fmaf(dy, w, by); bz = fmaf(dz, w, bz);
		}
		else
		{
			float wr;
			asm volatile("s_nop 1\n\t"
			             "v_mov_b32_dpp %3, %7 row_ror:%8 row_mask:0xf bank_mask:0xf\n\t"
			             "v_fmac_f32_dpp %0, %4, %3 row_ror:%8 row_mask:0xf bank_mask:0xf\n\t"
			             "v_fmac_f32_dpp %1, %5, %3 row_ror:%8 row_mask:0xf bank_mask:0xf\n\t"
			             "v_fmac_f32_dpp %2, %6, %3 row_ror:%8 row_mask:0xf bank_mask:0xf"
			             : "+v"(bx), "+v"(by), "+v"(bz), "=&v"(wr)
			             : "v"(dx), "v"(dy), "v"(dz), "v"(w), "n"((16 - S) % 16));
		}
	}
	if constexpr (S + 1 < 16) mutual_steps<S + 1, REACT>(px, py, pz, sx, sy, sz, eps2, ax, ay, az, bx, by, bz);
}

// Leaves of more than 32 particles are taken as TH halves of up to 32 (TH = 1, 2, 4 for leaves of up to 32, 64, 128): a leaf
// pair is TH x TH passes of the four-block scheme; a wave keeps TH target positions and TH sets of target sums in registers and
// stores one 32-particle reaction record per source half (react[(pair * TH + half) * 32 + lane]).
template <int TH, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void p2p_mutual_kernel(const float4 *__restrict__ pos, const int4 *__restrict__ desc, const int4 *__restrict__ chunk,
                                                         const int *__restrict__ nchunks_total, float eps2, int stride, float4 *__restrict__ partial,
                                                         float4 *__restrict__ react, int react_cap, int npos)
{
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4, k = lane & 15;
	auto sane_desc = [&](int4 d) { return NBCO_CHECKED_OK(d.x >= 0 && d.y >= 0 && d.x + d.y <= npos && (unsigned)d.w <= 2u, NBCO_CHK_DESC) ? d : make_int4(0, 0, 0, 2); };
	const int tsub = 16 * (row >> 1) + k, ssub = 16 * (row & 1) + k;   // this lane's particle inside a target / source half
	const int total = *nchunks_total;
	const int cstride = gridDim.x * WAVES;
	const float4 far = make_float4(1.e18f, 1.e18f, 1.e18f, 0.f);
	for (int cid = blockIdx.x * WAVES + wv; cid < total; cid += cstride)
	{
		int4 ck = chunk[cid];
		if (!NBCO_CHECKED_OK(ck.x >= 0 && ck.w >= 1 && ck.x + ck.w <= npos && ck.y <= ck.z, NBCO_CHK_CHUNK)) ck = make_int4(0, 0, 0, 1);
		const int it = __builtin_amdgcn_readfirstlane(ck.x), e0 = __builtin_amdgcn_readfirstlane(ck.y), e1 = __builtin_amdgcn_readfirstlane(ck.z),
		          mt = __builtin_amdgcn_readfirstlane(ck.w);
		// (loads always go to a valid particle and the value is replaced afterwards: a select between a load and a constant makes
		// the compiler select between ADDRESSES, with the constant parked in scratch and a flat load in the loop)
		float4 pt[TH];
		float ax[TH], ay[TH], az[TH];
#pragma unroll
		for (int h = 0; h < TH; ++h)
		{
			pt[h] = pos[it + min(32 * h + tsub, mt - 1)];
			if (32 * h + tsub >= mt) pt[h] = far;
			ax[h] = ay[h] = az[h] = 0.f;
		}
		// Software pipeline over the entries, two deep: the descriptor of entry e + 2 (scalar load) and the source positions of entry
		// e + 1 (vector load, raw: the out-of-range lanes are replaced by the far point only when the values are taken over)
		// are requested BEFORE entry e is evaluated and first touched after it, so neither latency sits in front of the 16 steps.
		// (Selecting the far point right behind the load put an s_waitcnt vmcnt(0) ahead of every entry's arithmetic.)
		auto request_sources = [&](const int4 &d, float4 (&raw)[TH]) {
#pragma unroll
			for (int h = 0; h < TH; ++h) raw[h] = pos[d.x + min(32 * h + ssub, max(d.y - 1, 0))];
		};
		auto take_sources = [&](const int4 &d, const float4 (&raw)[TH], float4 (&ps)[TH]) {
#pragma unroll
			for (int h = 0; h < TH; ++h) ps[h] = (32 * h + ssub < d.y) ? raw[h] : far;
		};
		const int4 none = make_int4(0, 0, 0, 2);
		int4 d = e0 < e1 ? sane_desc(desc[e0]) : none;
		int4 dn = e0 + 1 < e1 ? sane_desc(desc[e0 + 1]) : none;
		float4 ps[TH], raw[TH];
		request_sources(d, raw);
		take_sources(d, raw, ps);
		for (int e = e0; e < e1; ++e)
		{
			const int4 dnn = e + 2 < e1 ? sane_desc(desc[e + 2]) : none;
			request_sources(dn, raw);
			const int code = __builtin_amdgcn_readfirstlane(d.w), ms = __builtin_amdgcn_readfirstlane(d.y);
			if (code == 1)
			{
				const int pid = __builtin_amdgcn_readfirstlane(d.z);
#pragma unroll
				for (int hs = 0; hs < TH; ++hs)
				{
					if (32 * hs >= ms) break;
					float bx = 0.f, by = 0.f, bz = 0.f;
#pragma unroll
					for (int ht = 0; ht < TH; ++ht)
					{
						if (32 * ht >= mt) break;
						float tx = 0.f, ty = 0.f, tz = 0.f;
						mutual_steps<0, true>(pt[ht].x, pt[ht].y, pt[ht].z, ps[hs].x, ps[hs].y, ps[hs].z, eps2, tx, ty, tz, bx, by, bz);
						ax[ht] += tx; ay[ht] += ty; az[ht] += tz;
					}
					// rows 0 and 2 (1 and 3) hold the two halves of the sums of this source half's lower (upper) 16 particles
					bx += __shfl_xor(bx, 32); by += __shfl_xor(by, 32); bz += __shfl_xor(bz, 32);
					if (lane < 32 && pid >= 0 && pid < react_cap) react[((size_t)pid * TH + hs) * 32 + lane] = make_float4(bx, by, bz, 0.f);
				}
			}
			else if (code == 0)
			{
#pragma unroll
				for (int hs = 0; hs < TH; ++hs)
				{
					if (32 * hs >= ms) break;
#pragma unroll
					for (int ht = 0; ht < TH; ++ht)
					{
						if (32 * ht >= mt) break;
						float tx = 0.f, ty = 0.f, tz = 0.f, bx, by, bz;
						mutual_steps<0, false>(pt[ht].x, pt[ht].y, pt[ht].z, ps[hs].x, ps[hs].y, ps[hs].z, eps2, tx, ty, tz, bx, by, bz);
						ax[ht] += tx; ay[ht] += ty; az[ht] += tz;
					}
				}
			}
			take_sources(dn, raw, ps);
			d = dn; dn = dnn;
		}
		// rows 0 and 1 (2 and 3) hold the two halves of the sums of a target half's lower (upper) 16 particles
#pragma unroll
		for (int h = 0; h < TH; ++h)
		{
			const float sx = ax[h] + __shfl_xor(ax[h], 16), sy = ay[h] + __shfl_xor(ay[h], 16), sz = az[h] + __shfl_xor(az[h], 16);
			if ((row & 1) == 0 && 32 * h + tsub < mt) partial[(size_t)cid * stride + 32 * h + tsub] = make_float4(sx, sy, sz, 0.f);
		}
	}
}

// Where a reaction record goes: the wave of (A, B), A before B, stores B's sums at the SORTED position of the mirror entry
// (B <- A) in B's list, so that the records of a leaf lie side by side in list order and the L2P kernel reads them with
// independent, contiguous loads (a per-leaf reduction over records scattered by pair number cost 0.06-0.1 ms).  One thread per
// entry of code 1 finds that position by binary search over B's sorted sources and leaves it in desc.z.
__global__ __launch_bounds__(256) void p2p_link_kernel(int4 *__restrict__ desc, const uint64_t *__restrict__ keys, const int *__restrict__ start,
                                                       const int *__restrict__ total_ptr, int shift)
{
	const int total = *total_ptr;
	const uint64_t mask = (1ull << shift) - 1;
	for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256)
	{
		if (desc[e].w != 1) continue;
		const uint64_t key = keys[e];
		const int a = (int)(key >> shift), b = (int)(key & mask);   // target, source: look for source a in the list of target b
		int lo = start[b], hi = start[b + 1];
		while (lo < hi)
		{
			const int mid = (lo + hi) >> 1;
			if ((int)(keys[mid] & mask) < a) lo = mid + 1; else hi = mid;
		}
		desc[e].z = lo;
	}
}

// Does the mutual kernel beat the one-directional one for leaves of up to mlt_max particles?  It evaluates (32 TH)^2 lane
// pairs per leaf pair at 1.5x the pair rate (measured ratio of the two bodies, tools/pair_ceiling.hip); the one-directional
// kernel wastes the lanes beyond mlt_max of its target group.
static int p2p_mutual_halves(int mlt_max)
{
	if (mlt_max <= 16 || mlt_max > 128) return 0;
	const int th = mlt_max <= 32 ? 1 : (mlt_max <= 64 ? 2 : 4);
	const int tpl = mlt_max <= 32 ? 32 : 64, groups = (mlt_max + tpl - 1) / tpl;
	const double mutual = 1.5 * (double)mlt_max * mlt_max / ((32.0 * th) * (32.0 * th)), one_way = (double)mlt_max / (tpl * groups);
	return mutual > one_way ? th : 0;
}

static void launch_p2p_link(nbco_ctx *c, int4 *desc, const uint64_t *keys, const int *start, const int *total_ptr, int shift, long long entries_hint)
{
	const int grid = (int)std::min<long long>(std::max<long long>(1, (entries_hint + 255) / 256), 4096);
	hipLaunchKernelGGL(p2p_link_kernel, dim3(grid), dim3(256), 0, c->stream, desc, keys, start, total_ptr, shift);
}

template <int WAVES>
static void launch_p2p_mutual_w(nbco_ctx *c, int th, const float4 *pos, const int4 *desc, const int4 *chunk, const int *ntotal, long long chunks_hint, int stride,
                                float4 *partial, float4 *react, long long react_cap, long long npos)
{
	const int grid = (int)std::max<long long>(1, (chunks_hint + WAVES - 1) / WAVES);
	const int cap = (int)std::min<long long>(react_cap, 0x7fffffff);
	if (th == 1)
		hipLaunchKernelGGL((p2p_mutual_kernel<1, WAVES>), dim3(grid), dim3(64 * WAVES), 0, c->stream, pos, desc, chunk, ntotal, c->o.eps2, stride, partial, react, cap, (int)npos);
	else if (th == 2)
		hipLaunchKernelGGL((p2p_mutual_kernel<2, WAVES>), dim3(grid), dim3(64 * WAVES), 0, c->stream, pos, desc, chunk, ntotal, c->o.eps2, stride, partial, react, cap, (int)npos);
	else
		hipLaunchKernelGGL((p2p_mutual_kernel<4, WAVES>), dim3(grid), dim3(64 * WAVES), 0, c->stream, pos, desc, chunk, ntotal, c->o.eps2, stride, partial, react, cap, (int)npos);
}
// Workgroups of two waves: a workgroup keeps its wave slots until its longest work unit is done, and the units are uneven
// (1..16 entries), while four waves per SIMD already saturate this register-only body (tools/pair_ceiling.hip)
static void launch_p2p_mutual(nbco_ctx *c, int th, const float4 *pos, const int4 *desc, const int4 *chunk, const int *ntotal, long long chunks_hint, int stride,
                              float4 *partial, float4 *react, long long react_cap, long long npos)
{
	static const int waves = getenv("NBCO_MUT_WAVES") ? atoi(getenv("NBCO_MUT_WAVES")) : 2;
	if (waves == 1) launch_p2p_mutual_w<1>(c, th, pos, desc, chunk, ntotal, chunks_hint, stride, partial, react, react_cap, npos);
	else if (waves == 2) launch_p2p_mutual_w<2>(c, th, pos, desc, chunk, ntotal, chunks_hint, stride, partial, react, react_cap, npos);
	else launch_p2p_mutual_w<4>(c, th, pos, desc, chunk, ntotal, chunks_hint, stride, partial, react, react_cap, npos);
}

} // namespace
