// k_direct.hip -- direct O(N^2) Coulomb sum, a_i = k * sum_j d_ij / (|d_ij|^2 + EPS2)^(3/2).
// Reference behaviour: direct.cuh:51-256 (`direct` smem-tiled, `direct2` naive, `direct3` Kahan);
// the j = i term is included and contributes exactly zero (d = 0), as in the reference.
//
// gfx950 design: positions are packed once to float4 (one coalesced 16-byte load per particle),
// a 256-thread workgroup owns 256*IB targets held in registers (IB per lane), source particles are
// staged through a double-buffered LDS tile of packed xyz triplets and read back as wave-uniform
// ds_read_b128 broadcasts (three reads per four sources), each source feeding IB pair evaluations.  The j range is split over gridDim.y so that every CU holds
// several workgroups (>= 2 waves per SIMD are needed to saturate VALU issue on CDNA4); partial
// sums are combined in a fixed order by a second tiny kernel, so results are bit-reproducible.
// Per TWO pairs and lane: 3 v_pk_add, 3 v_pk_fma (r^2), 2 v_rsq_f32, 2 v_pk_mul, 3 v_pk_fma = 13 VALU issues.
#include "nbco_internal.hpp"

namespace {

constexpr int kBlock = 256;
constexpr int IB = 4;                 // targets per lane
constexpr int kTile = 256;            // sources per LDS tile

#define NBCO_PAIR(PX, PY, PZ)                                             \
	_Pragma("unroll") for (int k = 0; k < IB; ++k)                         \
	{                                                                      \
		float dx = xi[k] - (PX), dy = yi[k] - (PY), dz = zi[k] - (PZ);     \
		float r2 = fmaf(dx, dx, fmaf(dy, dy, fmaf(dz, dz, eps2)));         \
		float ri = __builtin_amdgcn_rsqf(r2);                              \
		float ri3 = ri * ri * ri;                                          \
		ax2[k].x = fmaf(dx, ri3, ax2[k].x);                                \
		ay2[k].x = fmaf(dy, ri3, ay2[k].x);                                \
		az2[k].x = fmaf(dz, ri3, az2[k].x);                                \
	}
// Two SOURCES per packed fp32 instruction (v_pk_add / v_pk_fma / v_pk_mul_f32: the nominal fp32 peak of this part is the packed
// rate -- tools/valu_probe.py pk_*).  The tile keeps x, y and z in three rows, so one ds_read_b128 per coordinate
// delivers four sources as two aligned register pairs; a target's coordinates sit in both halves of a pair (set up once per
// lane).  Per pair the arithmetic is the scalar form's; a target's even and odd sources are summed apart between two folds.
typedef float dir_v2f __attribute__((ext_vector_type(2)));
#define NBCO_PAIR2(XS, YS, ZS)                                                                                         \
	_Pragma("unroll") for (int k = 0; k < IB; ++k)                                                                     \
	{                                                                                                                  \
		const dir_v2f dx = xi2[k] - (XS), dy = yi2[k] - (YS), dz = zi2[k] - (ZS);                                      \
		const dir_v2f r2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, __builtin_elementwise_fma(dz, dz, eps2v))); \
		dir_v2f ri;                                                                                                    \
		ri.x = __builtin_amdgcn_rsqf(r2.x);                                                                            \
		ri.y = __builtin_amdgcn_rsqf(r2.y);                                                                            \
		const dir_v2f ri3 = ri * ri * ri;                                                                              \
		ax2[k] = __builtin_elementwise_fma(dx, ri3, ax2[k]);                                                           \
		ay2[k] = __builtin_elementwise_fma(dy, ri3, ay2[k]);                                                           \
		az2[k] = __builtin_elementwise_fma(dz, ri3, az2[k]);                                                           \
	}

template <bool KAHAN>
__global__ __launch_bounds__(kBlock) void direct_tiles(const float4 *__restrict__ pos, float4 *__restrict__ part, long long n,
                                                       float eps2, int tiles_per_split)
{
	// source tile as three rows x | y | z: four sources are fetched with three ds_read_b128, each delivering two aligned pairs
	__shared__ __attribute__((aligned(16))) float tile[2][3 * kTile];
	const int tid = threadIdx.x;
	const long long i0 = (long long)blockIdx.x * (kBlock * IB) + tid;

	float xi[IB], yi[IB], zi[IB];
	float sx[IB], sy[IB], sz[IB], cx[IB], cy[IB], cz[IB];
#pragma unroll
	for (int k = 0; k < IB; ++k)
	{
		long long i = i0 + (long long)k * kBlock;
		float4 p = pos[i < n ? i : n - 1];
		xi[k] = p.x; yi[k] = p.y; zi[k] = p.z;
		sx[k] = sy[k] = sz[k] = 0.f;
		cx[k] = cy[k] = cz[k] = 0.f;
	}
	dir_v2f xi2[IB], yi2[IB], zi2[IB];
#pragma unroll
	for (int k = 0; k < IB; ++k) { xi2[k] = dir_v2f{xi[k], xi[k]}; yi2[k] = dir_v2f{yi[k], yi[k]}; zi2[k] = dir_v2f{zi[k], zi[k]}; }
	const dir_v2f eps2v = {eps2, eps2};

	const long long ntiles = (n + kTile - 1) / kTile;
	const long long t_beg = (long long)blockIdx.y * tiles_per_split;
	long long t_end = t_beg + tiles_per_split;
	if (t_end > ntiles) t_end = ntiles;

	if (t_beg < t_end)
	{
		long long j = t_beg * kTile + tid;
		float4 q = pos[j < n ? j : n - 1];
		tile[0][tid] = q.x; tile[0][kTile + tid] = q.y; tile[0][2 * kTile + tid] = q.z;
	}
	__syncthreads();

	for (long long t = t_beg; t < t_end; ++t)
	{
		const int cur = (int)((t - t_beg) & 1);
		// prefetch the next tile into registers while this one is consumed
		float4 nxt = make_float4(0.f, 0.f, 0.f, 0.f);
		const bool has_next = (t + 1 < t_end);
		if (has_next)
		{
			long long j = (t + 1) * kTile + tid;
			nxt = pos[j < n ? j : n - 1];
		}
		long long rem = n - t * kTile;
		const int jcount = rem < kTile ? (int)rem : kTile;

		dir_v2f ax2[IB], ay2[IB], az2[IB];   // .x: even sources (and the tail's single ones), .y: odd sources
#pragma unroll
		for (int k = 0; k < IB; ++k) ax2[k] = ay2[k] = az2[k] = dir_v2f{0.f, 0.f};

		const float4 *t4 = reinterpret_cast<const float4 *>(tile[cur]);
		const int quads = jcount >> 2;
		// direct3 folds every 16 sources into the compensated running sums (the error of the plain partial sums grows with
		// their length: 256-source partial sums left it at 1.7e-6 on clustered inputs, where the reference's per-term
		// compensation reaches 2e-7)
		auto fold = [&]() {
#pragma unroll
			for (int k = 0; k < IB; ++k)
			{
				float y, s;
				y = (ax2[k].x + ax2[k].y) - cx[k]; s = sx[k] + y; cx[k] = (s - sx[k]) - y; sx[k] = s; ax2[k] = dir_v2f{0.f, 0.f};
				y = (ay2[k].x + ay2[k].y) - cy[k]; s = sy[k] + y; cy[k] = (s - sy[k]) - y; sy[k] = s; ay2[k] = dir_v2f{0.f, 0.f};
				y = (az2[k].x + az2[k].y) - cz[k]; s = sz[k] + y; cz[k] = (s - sz[k]) - y; sz[k] = s; az2[k] = dir_v2f{0.f, 0.f};
			}
		};
#pragma unroll 2
		for (int q = 0; q < quads; ++q)
		{
			const float4 X = t4[q], Y = t4[kTile / 4 + q], Z = t4[kTile / 2 + q];
			NBCO_PAIR2((dir_v2f{X.x, X.y}), (dir_v2f{Y.x, Y.y}), (dir_v2f{Z.x, Z.y}))
			NBCO_PAIR2((dir_v2f{X.z, X.w}), (dir_v2f{Y.z, Y.w}), (dir_v2f{Z.z, Z.w}))
			if (KAHAN && (q & 3) == 3) fold();
		}
		for (int j = quads << 2; j < jcount; ++j)
		{
			const float px = tile[cur][j], py = tile[cur][kTile + j], pz = tile[cur][2 * kTile + j];
			NBCO_PAIR(px, py, pz)
		}
		// fold the tile sums into the running sums (two-level summation; compensated for direct3)
#pragma unroll
		for (int k = 0; k < IB; ++k)
		{
			if (KAHAN)
			{
				float y, s;
				y = (ax2[k].x + ax2[k].y) - cx[k]; s = sx[k] + y; cx[k] = (s - sx[k]) - y; sx[k] = s;
				y = (ay2[k].x + ay2[k].y) - cy[k]; s = sy[k] + y; cy[k] = (s - sy[k]) - y; sy[k] = s;
				y = (az2[k].x + az2[k].y) - cz[k]; s = sz[k] + y; cz[k] = (s - sz[k]) - y; sz[k] = s;
			}
			else { sx[k] += ax2[k].x + ax2[k].y; sy[k] += ay2[k].x + ay2[k].y; sz[k] += az2[k].x + az2[k].y; }
		}
		if (has_next) { tile[cur ^ 1][tid] = nxt.x; tile[cur ^ 1][kTile + tid] = nxt.y; tile[cur ^ 1][2 * kTile + tid] = nxt.z; }
		__syncthreads();
	}

#pragma unroll
	for (int k = 0; k < IB; ++k)
	{
		long long i = i0 + (long long)k * kBlock;
		if (i < n) part[(long long)blockIdx.y * n + i] = make_float4(sx[k], sy[k], sz[k], 0.f);
	}
}

// a[i] = k * sum_s part[s][i], fixed order (direct.cuh:101 `a[i] = k*atmp`)
template <bool KAHAN>
__global__ __launch_bounds__(kBlock) void direct_combine(const float4 *__restrict__ part, float *__restrict__ a, long long n, int splits,
                                                         const float *__restrict__ param)
{
	const float k = param ? param[0] : 1.f;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		float sx = 0.f, sy = 0.f, sz = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
		for (int s = 0; s < splits; ++s)
		{
			float4 v = part[(long long)s * n + i];
			if (KAHAN)
			{
				float y, t;
				y = v.x - cx; t = sx + y; cx = (t - sx) - y; sx = t;
				y = v.y - cy; t = sy + y; cy = (t - sy) - y; sy = t;
				y = v.z - cz; t = sz + y; cz = (t - sz) - y; sz = t;
			}
			else { sx += v.x; sy += v.y; sz += v.z; }
		}
		a[3 * i] = k * sx; a[3 * i + 1] = k * sy; a[3 * i + 2] = k * sz;
	}
}

} // namespace

int launch_direct(nbco_ctx *c, const float *p, float *a, long long n, const float *param, bool kahan)
{
	if (n <= 0) return c->fail(NBCO_ERR_ARG, "nbco_direct: n must be positive");
	PhaseScope ph(c, NBCO_PH_DIRECT);
	NBCO_TRY(c->reserve(c->pos4, sizeof(float4) * (size_t)n));
	c->last_eval.valid = false;   // the tree-ordered positions of the last kd evaluation are overwritten
	NBCO_TRY(launch_pack4(c, c->pos4.as<float4>(), p, n));

	const long long iblocks = (n + kBlock * IB - 1) / (kBlock * IB);
	const long long ntiles = (n + kTile - 1) / kTile;
	// aim for >= 4 workgroups per CU
	long long splits = (4LL * c->num_cu + iblocks - 1) / iblocks;
	if (splits > ntiles) splits = ntiles;
	if (splits > 64) splits = 64;
	if (splits < 1) splits = 1;
	int tiles_per_split = (int)((ntiles + splits - 1) / splits);
	splits = (ntiles + tiles_per_split - 1) / tiles_per_split;

	NBCO_TRY(c->reserve(c->part, sizeof(float4) * (size_t)n * (size_t)splits));
	dim3 grid((unsigned)iblocks, (unsigned)splits);
	if (kahan)
		hipLaunchKernelGGL(direct_tiles<true>, grid, dim3(kBlock), 0, c->stream, c->pos4.as<float4>(), c->part.as<float4>(), n, c->o.eps2, tiles_per_split);
	else
		hipLaunchKernelGGL(direct_tiles<false>, grid, dim3(kBlock), 0, c->stream, c->pos4.as<float4>(), c->part.as<float4>(), n, c->o.eps2, tiles_per_split);
	NBCO_HIP(hipGetLastError());
	int cgrid = ceil_div(n, kBlock);
	if (cgrid > 2048) cgrid = 2048;
	if (kahan)
		hipLaunchKernelGGL(direct_combine<true>, dim3(cgrid), dim3(kBlock), 0, c->stream, c->part.as<float4>(), a, n, (int)splits, param);
	else
		hipLaunchKernelGGL(direct_combine<false>, dim3(cgrid), dim3(kBlock), 0, c->stream, c->part.as<float4>(), a, n, (int)splits, param);
	NBCO_HIP(hipGetLastError());
	return NBCO_OK;
}
