// Stage kernel K2: index probe -- the HBM-roofline kernel -- gfx950.
//
// Replaces mm_idx_get() + the occurrence filter of collect_matches() that run inside every
// index.map(seq) call of monica/genomes/aligner.py:193,215.
//
// One wave per read.  Per 64-minimizer step every lane streams one 8-byte query
// {hash, pos<<1|strand} (coalesced), gathers one 16-byte slot of the open-addressed table
// resident in HBM (linear probing; ~1.3 slots per query at load <= 0.5), and the wave
// writes the probe hits (0 < occurrences < mid_occ) as one compacted, ordered run of
// 16-byte records.  Queries that are too frequent only feed rep_len (A.4).
#include "device.h"

namespace mnc {

constexpr int PR_THREADS = 256;

__global__ __launch_bounds__(PR_THREADS) void mnc_probe_index(Batch B)
{
	const uint32_t r = (blockIdx.x * PR_THREADS + threadIdx.x) >> 6;
	if (r >= B.n_reads) return;
	const int lane = lane_id();
	const unsigned long long lt = (1ULL << lane) - 1ULL;
	const int64_t off = B.offsets[r];
	const int n = B.mz_cnt[r];
	const uint2 *mz = B.mz + off;
	HitRec *out = B.hits + off;
	const uint32_t mid_occ = (uint32_t)B.mid_occ;

	int n_hit = 0, rep = 0, last_en = 0;
	bool have_last = false;
	long long n_anchor = 0;
	uint32_t prev_hash = 0xffffffffu;               // hash of minimizer i0 - 1

	for (int i0 = 0; i0 < n; i0 += 64) {
		const int i = i0 + lane;
		const bool valid = i < n;
		uint2 q = valid ? mz[i] : make_uint2(0xffffffffu, 0);
		uint32_t cnt = 0;
		uint64_t val = 0;
		if (valid) {
			uint64_t slot = (uint64_t)q.x & B.table_mask;
			const uint32_t want = q.x + 1;
			for (;;) {
				const TableSlot s = B.table[slot];
				if (s.key == want) { cnt = s.cnt, val = s.val; break; }
				if (s.key == 0) break;
				slot = (slot + 1) & B.table_mask;
			}
		}
		// tandem flag: same hash as the neighbouring query minimizer (A.4)
		uint32_t left = __shfl_up(q.x, 1), right = __shfl_down(q.x, 1);
		if (lane == 0) left = prev_hash;
		if (lane == 63 || i + 1 >= n) right = (i + 1 < n) ? mz[i + 1].x : 0xffffffffu;
		const bool tandem = valid && (q.x == left || q.x == right);
		prev_hash = __shfl(q.x, 63);

		const bool high = valid && cnt >= mid_occ;
		const bool hit = valid && cnt > 0 && !high;

		// rep_len: union length of the k-mer intervals of too-frequent minimizers;
		// each contributes min(KMER, distance to the previous one)
		const unsigned long long hm = __ballot(high);
		if (hm) {
			const int en = (int)(q.y >> 1) + 1;
			const unsigned long long below = hm & lt;
			const int src = below ? 63 - __clzll((long long)below) : 0;
			const int prev_en = __shfl(en, src);
			int c = 0;
			if (high) {
				if (below) c = min(KMER, en - prev_en);
				else c = have_last ? min(KMER, en - last_en) : KMER;
			}
#pragma unroll
			for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
			rep += c;
			last_en = __shfl(en, 63 - __clzll((long long)hm));
			have_last = true;
		}

		const unsigned long long m = __ballot(hit);
		if (hit) {
			HitRec h;
			h.val = val, h.qinfo = q.y, h.cnt = cnt | (tandem ? 0x80000000u : 0u);
			out[n_hit + __popcll(m & lt)] = h;
		}
		n_hit += __popcll(m);
		long long c = hit ? (long long)cnt : 0;
#pragma unroll
		for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
		n_anchor += c;
	}
	if (lane == 0) {
		B.hit_cnt[r] = n_hit;
		B.an_cnt[r] = n_anchor;
		B.rep_len[r] = rep;
	}
}

void launch_probe(const Batch &B, hipStream_t st)
{
	if (B.n_reads == 0) return;
	const unsigned blocks = (B.n_reads + PR_THREADS / 64 - 1) / (PR_THREADS / 64);
	hipLaunchKernelGGL(mnc_probe_index, dim3(blocks), dim3(PR_THREADS), 0, st, B);
}

} // namespace mnc
