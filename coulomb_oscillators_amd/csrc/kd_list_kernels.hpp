// kd_list_kernels.hpp -- part of k_fmm_kd.hip (included there, in this place: one translation unit, one anonymous namespace)
// directed lists by counting sort: fill, per-target sort with source descriptors and work units, pair count; state reorder kernels
// (no include guard on purpose: this is a section of that file, not a header)
// ---- directed lists by counting sort ---------------------------------------------------------------
// key = target << shift | source.  P2P works on leaf numbers (node - kd_beg(L)) and gets one (leaf, leaf) self entry per
// leaf (fmm_cart3_kdtree.cuh:1059-1071); M2L works on node numbers.
// count (during the traversal, which also hands every entry its slot) -> exclusive scan -> scatter ->
// per-target sort of the (short) source ranges.  The last step makes the lists, and with them every
// floating-point sum downstream, identical from run to run.

__global__ __launch_bounds__(kBlock) void list_fill_kernel(const int2 *__restrict__ pairs, const int2 *__restrict__ ranks, const int *__restrict__ pref,
                                                           long long capR, int sub, int self0, int nself, int shift, const int *__restrict__ start,
                                                           uint64_t *__restrict__ keys, const int *__restrict__ chunk_off, int ntargets,
                                                           const int *__restrict__ leaf_index, const int *__restrict__ leaf_mult,
                                                           int4 *__restrict__ chunk, int pidmode)
{
	// pidmode (mutual near field): an entry is (source << 32 | index of its unordered pair), so that after the per-target sort the
	// two directions of a pair still know each other; self entries carry 0xFFFFFFFF
	// P2P list: the work-unit table of the pair kernel only needs the two prefix sums, like the fill: same launch.
	// chunk record: {first particle of the target leaf, first entry, end entry, particles of the target leaf}
	if (chunk)
		for (int i = blockIdx.x * kBlock + threadIdx.x; i < ntargets; i += gridDim.x * kBlock)
		{
			const int b = start[i], e = start[i + 1], o = chunk_off[i], n = chunk_off[i + 1] - o;
			const int ind = leaf_index[i], mlt = leaf_mult[i];
			// equal shares (a list of 17 becomes 9 + 8, not 16 + 1: a wave that only gets one source leaf spends its life
			// in the chain of dependent loads at the head of a chunk)
			const int per = n > 0 ? (e - b + n - 1) / n : 0;
			for (int k = 0; k < n; ++k) chunk[o + k] = make_int4(ind, min(b + k * per, e), min(b + (k + 1) * per, e), mlt);
		}
	const long long npairs = pref[kTravK];   // the pair count never leaves the device (regions are clamped to their capacity)
	const long long total = npairs + nself;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock)
	{
		if (i < npairs)
		{
			const long long slot = region_slot(pref, capR, i);
			if (!NBCO_CHECKED_OK(slot >= 0 && slot < capR * kTravK, NBCO_CHK_FILL)) continue;
			const int2 p = pairs[slot], r = ranks[slot];
			const uint64_t a = (uint64_t)(p.x - sub), b = (uint64_t)(p.y - sub);
			if (!NBCO_CHECKED_OK(a < (uint64_t)ntargets && b < (uint64_t)ntargets && (r.x < 0 || start[a] + r.x < start[a + 1])
			                         && (r.y < 0 || start[b] + r.y < start[b + 1]), NBCO_CHK_FILL)) continue;
			if (pidmode)
			{
				if (r.x >= 0) keys[start[a] + r.x] = (b << 32) | (uint64_t)(uint32_t)i;
				if (r.y >= 0) keys[start[b] + r.y] = (a << 32) | (uint64_t)(uint32_t)i;
			}
			else
			{
				if (r.x >= 0) keys[start[a] + r.x] = (a << shift) | b;
				if (r.y >= 0) keys[start[b] + r.y] = (b << shift) | a;
			}
		}
		else
		{
			// self entries of the domain's own leaves: counted first (traverse_init_kernel), so they own slot 0
			const uint64_t t = (uint64_t)(self0 + (i - npairs));
			keys[start[t]] = pidmode ? ((t << 32) | 0xFFFFFFFFull) : ((t << shift) | t);
		}
	}
}

// one wave per target: rank sort of its source range (distinct keys) in registers up to 512 entries (the BASELINE ball
// has 17 on average, 254..300 at most; a bitonic network in LDS for the 257..512 class cost 30 us of a 45 us kernel), and
// beyond that (wide opening radii, outliers late in a long run) a radix sort of the range through global memory
#ifndef NBCO_P2P_CHUNK
#define NBCO_P2P_CHUNK 16
#endif
constexpr int kP2PChunk = NBCO_P2P_CHUNK;   // source leaves per near-field work unit (see the P2P section)

// diagnostics only: the near-field launch's arrays as one binary file {header 8 x int64: n, entries, chunks, mlt_max, stride, 0, 0, 0;
// float4 pos[n]; int2 desc[entries]; int4 chunk[chunks]; float4 partial[chunks * stride]} for tools/p2p_lab.hip.  Synchronises.
static int p2p_dump(nbco_ctx *c, const char *path, const float4 *pos, const int2 *desc, const int4 *chunk, const int *nchunks, const int *nentries, int mlt_max,
                    const float4 *partial, long long n)
{
	NBCO_HIP(hipStreamSynchronize(c->stream));
	int hc = 0, he = 0;
	NBCO_HIP(hipMemcpy(&hc, nchunks, sizeof(int), hipMemcpyDeviceToHost));
	NBCO_HIP(hipMemcpy(&he, nentries, sizeof(int), hipMemcpyDeviceToHost));
	FILE *f = std::fopen(path, "wb");
	if (!f) return c->fail(NBCO_ERR_ARG, "NBCO_P2P_DUMP: cannot open the file");
	const long long head[8] = {n, he, hc, mlt_max, mlt_max, 0, 0, 0};
	std::fwrite(head, sizeof head, 1, f);
	auto put = [&](const void *dev, size_t bytes) {
		std::vector<char> h(bytes);
		if (hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
		return std::fwrite(h.data(), 1, bytes, f) == bytes;
	};
	const bool ok = put(pos, sizeof(float4) * (size_t)n) && put(desc, sizeof(int2) * (size_t)he) && put(chunk, sizeof(int4) * (size_t)hc) &&
	                put(partial, sizeof(float4) * (size_t)hc * (size_t)mlt_max);
	std::fclose(f);
	return ok ? NBCO_OK : c->fail(NBCO_ERR_HIP, "NBCO_P2P_DUMP: copy failed");
}
// DESC: the list is the P2P list -- also emit the source descriptor (first particle, multiplicity) of every sorted entry,
// so the pair kernel does no dependent index -> mult -> position loads
// what the per-target sort of the P2P list also produces for the mutual near-field kernel (k_p2p.hpp); desc4 == nullptr: off
struct MutualLists
{
	int4 *desc4 = nullptr;            // per sorted entry: {first particle of the source leaf, its multiplicity, pair index, code}
	int4 *chunk = nullptr;            // work units {first particle of the target leaf, first entry, end entry, multiplicity}
	const int *chunk_off = nullptr;   // chunk slots of target t: [chunk_off[t], chunk_off[t + 1])
	int2 *sec_range = nullptr;        // per target: its entries whose sums other waves deliver (code 2)
	int self0 = 0, nself = 0;         // leaves of the own kd-domain
};
template <bool DESC>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void list_segsort_kernel(const int *__restrict__ start, int ntargets, uint64_t *in, uint64_t *out, int shift,
                                                              const int *__restrict__ leaf_index, const int *__restrict__ leaf_mult,
                                                              int2 *__restrict__ desc, MutualLists mu, int long_from)
{
	__shared__ unsigned digit_off[kBlock / 64][256];   // long ranges only: per-wave digit offsets of the radix passes
	const uint64_t smask = (1ull << shift) - 1;
	// mutual near field (mu.desc4 set): entries are (source << 32 | pair index); the sorted list is written in the usual
	// (target << shift | source) form, and every entry gets a descriptor {first particle, multiplicity, pair index, code}
	// with code 1 = this target's wave evaluates the pair for both leaves (source after target), 2 = the source's wave does
	// (source before target), 0 = one direction only (the leaf itself, or a source outside the kd-domain [self0, self0 + nself))
	const int lowbit = (DESC && mu.desc4) ? 32 : 0;
	// (li, lm: first particle and multiplicity of the source leaf, fetched by the caller as soon as it knows the source -- beside the
	// ranking, not behind it)
	auto emit = [&](int t, int slot, uint64_t key, int li, int lm) {
		const int src = (int)((key >> lowbit) & smask);
		if (!NBCO_CHECKED_OK(src >= 0 && src < ntargets && slot >= start[t] && slot < start[t + 1], NBCO_CHK_SORT)) return;
		if (DESC && mu.desc4)
		{
			out[slot] = ((uint64_t)t << shift) | (uint64_t)src;
			const int code = (src == t || src < mu.self0 || src >= mu.self0 + mu.nself) ? 0 : (src > t ? 1 : 2);
			mu.desc4[slot] = make_int4(li, lm, (int)(uint32_t)key, code);
			return;
		}
		out[slot] = key;
		if (DESC) desc[slot] = make_int2(li, lm);
	};
	// mutual near field: work units of target t once its list is sorted.  Sorted order = [sources of lower kd-domains: nf
	// entries][own sources before t: delivered by their waves][t itself][sources after t]; the work entries (all but the
	// second group) are dealt to the target's chunk slots in equal shares; a unit that spans the gap skips it by code.
	auto finish_target = [&](int t, int s, int cnt, int nf, int nq, int lane) {
		const int o = mu.chunk_off[t], n = mu.chunk_off[t + 1] - o;
		const int w = nf + (cnt - nq), per = n > 0 ? (w + n - 1) / n : 0;
		const int ind = leaf_index[t], mlt = leaf_mult[t];
		for (int i = lane; i < n; i += 64)
		{
			const int jb = min(i * per, w), je = min((i + 1) * per, w);
			mu.chunk[o + i] = make_int4(ind, s + (jb < nf ? jb : jb - nf + nq), s + (je <= nf ? je : je - nf + nq), mlt);
		}
		if (lane == 0) mu.sec_range[t] = make_int2(s + nf, s + nq);
	};
	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	// ranges of up to 64 * J entries: every lane keeps J entries in registers and ranks them against all entries of the
	// range, which are broadcast one by one with v_readlane (an SGPR lane index: the loop is scalar, no LDS or memory
	// latency inside).  All entries of a range share the target, so comparing the 32-bit source indices orders the keys.
	auto rank_in_registers = [&](auto jtag, int t, int s, int cnt) {
		constexpr int J = decltype(jtag)::value;
		uint64_t key[J];
		unsigned src[J];
		// (the descriptors are fetched beside the ranking only while they fit the register budget of 8 waves per SIMD: with more
		// than four entries per lane they would spill to scratch, and a kernel with a scratch segment pays for it on every launch)
		constexpr bool PRE = DESC && J <= 4;
		int rank[J], li[PRE ? J : 1], lm[PRE ? J : 1];
#pragma unroll
		for (int j = 0; j < J; ++j)
		{
			key[j] = lane + 64 * j < cnt ? in[s + lane + 64 * j] : ~0ull;
			src[j] = (unsigned)((key[j] >> lowbit) & smask);
			rank[j] = 0;
		}
#pragma unroll
		for (int j = 0; j < J; ++j)
			if (PRE)
			{
				const int sidx = (int)min(src[j], (unsigned)(ntargets - 1));   // (idle lanes hold the all-ones key)
				li[j] = leaf_index[sidx]; lm[j] = leaf_mult[sidx];
			}
#pragma unroll
		for (int jb = 0; jb < J; ++jb)
		{
			const int lim = min(64, cnt - 64 * jb);   // wave-uniform
			for (int q = 0; q < lim; ++q)
			{
				const unsigned other = (unsigned)__builtin_amdgcn_readlane((int)src[jb], q);
#pragma unroll
				for (int j = 0; j < J; ++j) rank[j] += other < src[j] ? 1 : 0;
			}
		}
		int nf = 0, nq = 0;
		if (DESC && mu.desc4)
		{
#pragma unroll
			for (int j = 0; j < J; ++j)
			{
				const bool valid = lane + 64 * j < cnt;
				nf += __popcll(__ballot(valid && (int)src[j] < mu.self0));
				nq += __popcll(__ballot(valid && (int)src[j] < t));
			}
		}
#pragma unroll
		for (int j = 0; j < J; ++j)
			if (lane + 64 * j < cnt)
			{
				if (PRE) emit(t, s + rank[j], key[j], li[j], lm[j]);
				else
				{
					const int sidx = DESC ? (int)min(src[j], (unsigned)(ntargets - 1)) : 0;
					emit(t, s + rank[j], key[j], DESC ? leaf_index[sidx] : 0, DESC ? leaf_mult[sidx] : 0);
				}
			}
		if (DESC && mu.desc4) finish_target(t, s, cnt, nf, nq, lane);
	};
	for (int t = blockIdx.x * (kBlock / 64) + wv; t < ntargets; t += gridDim.x * (kBlock / 64))
	{
		// wave-uniform by construction; telling the compiler so keeps the loops scalar
		const int s = __builtin_amdgcn_readfirstlane(start[t]), cnt = __builtin_amdgcn_readfirstlane(start[t + 1]) - s;
		if (long_from > 0 && cnt > long_from) continue;   // (list_longsort_kernel, launched behind this one, takes the long ranges)
		if (cnt <= 64) rank_in_registers(std::integral_constant<int, 1>{}, t, s, cnt);
		else if (cnt <= 128) rank_in_registers(std::integral_constant<int, 2>{}, t, s, cnt);
		else if (cnt <= 192) rank_in_registers(std::integral_constant<int, 3>{}, t, s, cnt);
		else if (cnt <= 256) rank_in_registers(std::integral_constant<int, 4>{}, t, s, cnt);
		else if (cnt <= 320) rank_in_registers(std::integral_constant<int, 5>{}, t, s, cnt);
		else if (cnt <= 384) rank_in_registers(std::integral_constant<int, 6>{}, t, s, cnt);
		else if (cnt <= 512) rank_in_registers(std::integral_constant<int, 8>{}, t, s, cnt);
		else
		{
			// Long range (an outlier's leaf can be paired with most of the tree: tens of thousands of entries): the wave sorts
			// it by source index with a stable LSD radix sort, 8 bits per pass, ping-ponging between its slices of the
			// unsorted and the sorted key arrays.  O(cnt) per pass, where ranking would be O(cnt^2).
			unsigned *off = digit_off[wv];
			uint64_t *src = in + s, *dst = out + s;
			const uint64_t below = (1ull << lane) - 1ull;
			for (int bit = lowbit; bit < lowbit + shift; bit += 8)
			{
				for (int b = lane; b < 256; b += 64) off[b] = 0u;
				wave_lds_sync();
				for (int i = lane; i < cnt; i += 64) atomicAdd(&off[(unsigned)(src[i] >> bit) & 255u], 1u);
				wave_lds_sync();
				// exclusive scan of the 256 digit counts (four per lane)
				unsigned v0 = off[4 * lane], v1 = off[4 * lane + 1], v2 = off[4 * lane + 2], v3 = off[4 * lane + 3];
				const unsigned sum = v0 + v1 + v2 + v3;
				const unsigned incl = wave_scan_add(sum);
				const unsigned base = incl - sum;
				wave_lds_sync();
				off[4 * lane] = base; off[4 * lane + 1] = base + v0; off[4 * lane + 2] = base + v0 + v1; off[4 * lane + 3] = base + v0 + v1 + v2;
				wave_lds_sync();
				// stable scatter, 64 entries at a time in range order
				for (int i0 = 0; i0 < cnt; i0 += 64)
				{
					const int i = i0 + lane;
					const bool valid = i < cnt;
					const uint64_t key = valid ? src[i] : 0ull;
					const unsigned dg = (unsigned)(key >> bit) & 255u;
					uint64_t same = __ballot(valid);   // lanes holding the same digit
#pragma unroll
					for (int b = 0; b < 8; ++b)
					{
						const uint64_t bal = __ballot((dg >> b) & 1u);
						same &= ((dg >> b) & 1u) ? bal : ~bal;
					}
					const unsigned before = (unsigned)__popcll(same & below);
					const unsigned o = valid ? off[dg] : 0u;
					if (valid) dst[o + before] = key;
					wave_lds_sync();
					if (valid && before == 0u) off[dg] = o + (unsigned)__popcll(same);
					wave_lds_sync();
				}
				// the next pass reads what this one wrote (same wave, through memory)
				__threadfence_block();
				uint64_t *t2 = src; src = dst; dst = t2;
			}
			// `src` holds the sorted range now
			int nf = 0, nq = 0;
			for (int i0 = 0; i0 < cnt; i0 += 64)
			{
				const int i = i0 + lane;
				const uint64_t key = i < cnt ? src[i] : 0ull;
				const int sv = (int)((key >> lowbit) & smask);
				if (i < cnt) emit(t, s + i, key, DESC ? leaf_index[min(sv, ntargets - 1)] : 0, DESC ? leaf_mult[min(sv, ntargets - 1)] : 0);
				nf += __popcll(__ballot(i < cnt && sv < mu.self0));
				nq += __popcll(__ballot(i < cnt && sv < t));
			}
			if (DESC && mu.desc4) finish_target(t, s, cnt, nf, nq, lane);
		}
	}
}

// ---- long ranges of the P2P list --------------------------------------------------------------------------------------------
// A thousand steps into the benchmark's run a few leaves have been stretched by ejected particles and are partners of most of
// the tree: 140 targets hold a quarter of all entries, the longest 16 000 (`profiles/r03o_late_lists.txt`).  One wave per
// target -- list_segsort_kernel's mapping -- then means the whole sort waits for the one wave with the longest range (0.54 of
// a 3.0 ms step).  When the previous evaluation's lists say that long ranges are to be expected the host launches this kernel
// behind the sort and tells the sort to leave ranges above `long_from` entries alone.  A workgroup collects the long targets
// a hash assigns to it and sorts them one by one with all its threads.  The sources of one target are distinct, so an entry's
// place in the sorted range is the number of set bits below its source in a bitmap of the range's sources (trees of up to
// 2^16 leaves): two passes over the range with nothing carried from one entry to the next.  Same output as the sort's own
// long-range path, entry for entry.
constexpr int kLongBlock = 256, kLongWords = 2048;
__global__ __launch_bounds__(kLongBlock) void list_longsort_kernel(const int *__restrict__ start, int ntargets, const uint64_t *__restrict__ in,
                                                                    uint64_t *__restrict__ out, int shift, const int *__restrict__ leaf_index,
                                                                    const int *__restrict__ leaf_mult, int2 *__restrict__ desc, MutualLists mu, int long_from)
{
	__shared__ unsigned words[kLongWords], pref[kLongWords];   // bitmap of the sources; set bits in front of every word
	__shared__ unsigned wsum[kLongBlock / 64];
	__shared__ int todo[kLongBlock * 8], ntodo;   // (room for every target of a round)
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const uint64_t smask = (1ull << shift) - 1;
	const int lowbit = mu.desc4 ? 32 : 0;
	const int nw = 1 << (shift - 5);                             // words in use (shift <= 16, checked by the host)
	const int per = (nw + kLongBlock - 1) / kLongBlock;          // consecutive words per thread in the prefix pass
	// every workgroup looks at all targets and keeps the long ones a hash of their number assigns to it: stretched leaves are
	// neighbours in the tree, a workgroup that took a contiguous share would get all of them
	constexpr int kPerThread = 8;   // targets per thread and round (the list has room for all of a round's)
	for (int t0 = 0; t0 < ntargets; t0 += kLongBlock * kPerThread)
	{
		if (tid == 0) ntodo = 0;
		__syncthreads();
#pragma unroll
		for (int q = 0; q < kPerThread; ++q)
		{
			const int mine = t0 + q * kLongBlock + tid;
			if (mine < ntargets && start[mine + 1] - start[mine] > long_from && (((unsigned)mine * 2654435761u) >> 12) % gridDim.x == blockIdx.x)
			{
				todo[atomicAdd(&ntodo, 1)] = mine;
			}
		}
		__syncthreads();
		const int n = ntodo;
		for (int e = 0; e < n; ++e)
		{
			const int t = todo[e], s = start[t], cnt = start[t + 1] - s;
			for (int w = tid; w < nw; w += kLongBlock) words[w] = 0u;
			__syncthreads();
#pragma unroll 8
			for (int i = tid; i < cnt; i += kLongBlock)
			{
				const unsigned sv = (unsigned)((in[s + i] >> lowbit) & smask);
				atomicOr(&words[sv >> 5], 1u << (sv & 31u));
			}
			__syncthreads();
			// set bits in front of every word: per thread a run of `per` words, then a scan over the threads
			unsigned sum = 0;
			for (int q = 0; q < per; ++q)
			{
				const int w = tid * per + q;
				if (w < nw) { pref[w] = sum; sum += (unsigned)__popc(words[w]); }
			}
			const unsigned incl = wave_scan_add(sum);
			if (lane == 63) wsum[wv] = incl;
			__syncthreads();
			unsigned base = incl - sum;
			for (int q = 0; q < wv; ++q) base += wsum[q];
			for (int q = 0; q < per; ++q)
			{
				const int w = tid * per + q;
				if (w < nw) pref[w] += base;
			}
			__syncthreads();
			auto below = [&](unsigned sv) {
				const unsigned w = sv >> 5;
				return (int)(pref[w] + (unsigned)__popc(words[w] & ((1u << (sv & 31u)) - 1u)));
			};
#pragma unroll 8
			for (int i = tid; i < cnt; i += kLongBlock)
			{
				const uint64_t key = in[s + i];
				const int sv = (int)((key >> lowbit) & smask), slot = s + below((unsigned)sv);
				if (!NBCO_CHECKED_OK(sv >= 0 && sv < ntargets && slot >= s && slot < s + cnt, NBCO_CHK_SORT)) continue;
				if (mu.desc4)
				{
					out[slot] = ((uint64_t)t << shift) | (uint64_t)sv;
					const int code = (sv == t || sv < mu.self0 || sv >= mu.self0 + mu.nself) ? 0 : (sv > t ? 1 : 2);
					mu.desc4[slot] = make_int4(leaf_index[sv], leaf_mult[sv], (int)(uint32_t)key, code);
				}
				else
				{
					out[slot] = key;
					desc[slot] = make_int2(leaf_index[sv], leaf_mult[sv]);
				}
			}
			if (mu.desc4)
			{
				// work units of the mutual near field: list_segsort_kernel's finish_target
				const int nf = below((unsigned)min(max(mu.self0, 0), (1 << shift) - 1)), nq = below((unsigned)t);
				const int o = mu.chunk_off[t], nu = mu.chunk_off[t + 1] - o;
				const int w = nf + (cnt - nq), share = nu > 0 ? (w + nu - 1) / nu : 0;
				const int ind = leaf_index[t], mlt = leaf_mult[t];
				for (int i = tid; i < nu; i += kLongBlock)
				{
					const int jb = min(i * share, w), je = min((i + 1) * share, w);
					mu.chunk[o + i] = make_int4(ind, s + (jb < nf ? jb : jb - nf + nq), s + (je <= nf ? je : je - nf + nq), mlt);
				}
				if (tid == 0) mu.sec_range[t] = make_int2(s + nf, s + nq);
			}
			__syncthreads();
		}
	}
}

// directed pair interactions = sum over the directed P2P entries of mult[target] * mult[source]
// (the self entries contribute mult^2), SURVEY 8(d)
__global__ __launch_bounds__(kBlock) void pair_count_kernel(const int *__restrict__ leaf_mult, const uint64_t *__restrict__ keys,
                                                            const int *__restrict__ count_ptr, int shift, unsigned long long *__restrict__ out)
{
	const long long count = *count_ptr;
	const uint64_t mask = (1ull << shift) - 1;
	unsigned long long s = 0;
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < count; i += (long long)gridDim.x * kBlock)
		s += (unsigned long long)leaf_mult[(int)(keys[i] >> shift)] * (unsigned long long)leaf_mult[(int)(keys[i] & mask)];
	for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
	if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// ---- P2P ---------------------------------------------------------------------------------------------
// Work units.  The number of source leaves per target leaf is very uneven (Gaussian ball, N = 1M: mean 17,
// max > 250), so one wave per target leaf leaves a tail as long as the rest of the kernel.  Every
// target leaf's sorted source range is therefore cut into chunks of at most kP2PChunk entries; a wave
// evaluates one chunk and stores the partial sums of its 32 targets, and the L2P kernel adds a leaf's
// chunks in list order (fixed order: still bit-reproducible, still no atomics).  Source descriptors and chunk
// counts come out of the per-target sort (list_segsort_kernel<true>).
// tree order for the caller's state in one pass: positions unpacked to xyz triplets, velocities gathered into a scratch
// copy (the gather cannot run in place)
__global__ __launch_bounds__(kBlock) void reorder_state_kernel(const float4 *__restrict__ pos, const int *__restrict__ unsort,
                                                               const float *__restrict__ v_in, float *__restrict__ p_out, float *__restrict__ v_tmp,
                                                               long long n, const int *__restrict__ order_in, int *__restrict__ order_out)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		const float4 q = pos[i];
		p_out[3 * i] = q.x; p_out[3 * i + 1] = q.y; p_out[3 * i + 2] = q.z;
		const long long s = unsort[i];
		v_tmp[3 * i] = v_in[3 * s]; v_tmp[3 * i + 1] = v_in[3 * s + 1]; v_tmp[3 * i + 2] = v_in[3 * s + 2];
		// opts.track_order: the particle now at position i was at position s before this rebuild, i.e. it is particle
		// order_in[s] of the state the tracking started from (order_in == nullptr: this is the first permutation)
		if (order_out) order_out[i] = order_in ? order_in[s] : (int)s;
	}
}

// sorted positions back to xyz triplets
__global__ __launch_bounds__(kBlock) void unpack4_kernel(const float4 *__restrict__ src, float *__restrict__ dst, long long n)
{
	for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock)
	{
		float4 p = src[i];
		dst[3 * i] = p.x; dst[3 * i + 1] = p.y; dst[3 * i + 2] = p.z;
	}
}

static int grid1d(long long n, int cap = 2048)
{
	long long b = (n + kBlock - 1) / kBlock;
	if (b < 1) b = 1;
	if (b > cap) b = cap;
	return (int)b;
}


