// kd_energy_kernels.hpp -- part of k_fmm_kd.hip (included there, in this place: one translation unit, one anonymous namespace)
// FMM potential energy: multipole-to-particle potential and the per-particle pass
// (no include guard on purpose: this is a section of that file, not a header)
// ---- FMM potential energy (no reference driver computes an energy) -------------------------------
// phi_i = sum_{j != i} (|x_i - x_j|^2 + eps2)^(-1/2) with the interaction lists of the last evaluation: the leaf's P2P list
// pair by pair, and for the M2L list of the leaf and of each of its ancestors the source node's multipole expansion evaluated
// AT THE PARTICLE (the reference's m2p_pot3, fmm_cart_base3.cuh:1474-1490, instead of an order-0 local: no second truncation).
// With b_K = d^K f / K!, f = 1/|d|:  phi = sum_K M[K] |K|! b_K(d), d = x_i - c_s, and the Taylor coefficients follow
//   k R^2 b_K + (2k - 1) sum_a d_a b_{K - e_a} + (k - 1) sum_a b_{K - 2 e_a} = 0,   k = |K|, R^2 = |d|^2
// (tests/test_oracle_closed_form.py derives the same numbers from the polynomial form of the derivatives).  fp64 throughout:
// this is a diagnostic that runs once per snapshot, and energy drifts are read at the 1e-6 level.
__host__ __device__ constexpr int sym_index(int x, int z, int n) { return (n * (n + 1) - (n - z) * (n - z + 1)) / 2 + n - x; }
__host__ __device__ constexpr int sym_offset(int n) { return n * (n + 1) * (n + 2) / 6; }

template <int P, typename T>
__device__ inline double m2p_potential(const T *__restrict__ M, double dx, double dy, double dz, double eps2)
{
	constexpr int offM = sym_offset(P);
	double B[offM > 0 ? offM : 1];
	const double R2 = dx * dx + dy * dy + dz * dz + eps2, iR2 = 1.0 / R2;
	B[0] = sqrt(iR2);
	double phi = (double)M[0] * B[0], fact = 1.0;
#pragma unroll
	for (int k = 1; k < P; ++k)
	{
		fact *= (double)k;
		const double c1 = -(double)(2 * k - 1) * iR2 / (double)k, c2 = -(double)(k - 1) * iR2 / (double)k;
		double s = 0.0;
#pragma unroll
		for (int z = 0; z <= k; ++z)
#pragma unroll
			for (int x = k - z; x >= 0; --x)
			{
				const int y = k - x - z;
				double t1 = 0.0, t2 = 0.0;
				if (x >= 1) t1 += dx * B[sym_offset(k - 1) + sym_index(x - 1, z, k - 1)];
				if (y >= 1) t1 += dy * B[sym_offset(k - 1) + sym_index(x, z, k - 1)];
				if (z >= 1) t1 += dz * B[sym_offset(k - 1) + sym_index(x, z - 1, k - 1)];
				if (k >= 2)
				{
					if (x >= 2) t2 += B[sym_offset(k - 2) + sym_index(x - 2, z, k - 2)];
					if (y >= 2) t2 += B[sym_offset(k - 2) + sym_index(x, z, k - 2)];
					if (z >= 2) t2 += B[sym_offset(k - 2) + sym_index(x, z - 2, k - 2)];
				}
				const double b = c1 * t1 + c2 * t2;
				B[sym_offset(k) + sym_index(x, z, k)] = b;
				s += (double)M[sym_offset(k) + sym_index(x, z, k)] * b;
			}
		phi += fact * s;
	}
	return phi;
}

template <int P, typename T>
__global__ __launch_bounds__(kBlock) void kd_potential_kernel(nbco_ctx::LastEval le, const uint64_t *__restrict__ m2l_keys, const int *__restrict__ m2l_start,
                                                              const uint64_t *__restrict__ p2p_keys, const int *__restrict__ p2p_start, float eps2f,
                                                              double *__restrict__ part)
{
	constexpr int offM = sym_offset(P);
	const long long io = (long long)blockIdx.x * kBlock + threadIdx.x;
	double phi = 0.0;
	if (io < le.own_n)
	{
		const long long i = le.own0 + io;
		const int beg = kd_beg(le.L), lf = (int)(((1LL << le.L) * i) / le.n);
		const uint64_t mask = (1ull << le.shift) - 1;
		const float4 p = le.pos[i];
		const double eps2 = (double)eps2f;
		if (le.have_p2p)
			for (int e = p2p_start[lf]; e < p2p_start[lf + 1]; ++e)
			{
				const int src = beg + (int)(p2p_keys[e] & mask);
				const int is = le.index[src], ms = le.mult[src];
				for (int j = 0; j < ms; ++j)
				{
					if (is + j == i) continue;
					const float4 q = le.pos[is + j];
					const double dx = (double)p.x - (double)q.x, dy = (double)p.y - (double)q.y, dz = (double)p.z - (double)q.z;
					phi += 1.0 / sqrt(dx * dx + dy * dy + dz * dz + eps2);
				}
			}
		for (int node = beg + lf;; node = (node - 1) >> 1)
		{
			for (int e = m2l_start[node]; e < m2l_start[node + 1]; ++e)
			{
				const int sn = (int)(m2l_keys[e] & mask);
				const float4 cs = le.csz[sn];
				phi += m2p_potential<P, T>(reinterpret_cast<const T *>(le.mpole) + (size_t)sn * offM, (double)p.x - (double)cs.x, (double)p.y - (double)cs.y, (double)p.z - (double)cs.z, eps2);
			}
			if (node == 0) break;
		}
	}
	// block sum -> one partial per block (summed in a fixed order on the host)
	__shared__ double sh[kBlock / 64];
	for (int o = 32; o > 0; o >>= 1) phi += __shfl_xor(phi, o);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = phi;
	__syncthreads();
	if (threadIdx.x == 0)
	{
		double t = 0.0;
		for (int k = 0; k < kBlock / 64; ++k) t += sh[k];
		part[blockIdx.x] = t;
	}
}

template <int P> static void launch_potential(nbco_ctx *c, int grid, double *part)
{
	// (the multipoles of the last evaluation are doubles when it ran with opts.far_fp64: LastEval::real_bytes)
	if (c->last_eval.real_bytes == 8)
		hipLaunchKernelGGL((kd_potential_kernel<P, double>), dim3(grid), dim3(kBlock), 0, c->stream, c->last_eval, (const uint64_t *)c->m2l_keys_alt.as<uint64_t>(),
		                   (const int *)c->m2l_start.as<int>(), (const uint64_t *)c->p2p_keys_alt.as<uint64_t>(), (const int *)c->p2p_start.as<int>(), c->o.eps2,
		                   part);
	else
		hipLaunchKernelGGL((kd_potential_kernel<P, float>), dim3(grid), dim3(kBlock), 0, c->stream, c->last_eval, (const uint64_t *)c->m2l_keys_alt.as<uint64_t>(),
		                   (const int *)c->m2l_start.as<int>(), (const uint64_t *)c->p2p_keys_alt.as<uint64_t>(), (const int *)c->p2p_start.as<int>(), c->o.eps2,
		                   part);
}

