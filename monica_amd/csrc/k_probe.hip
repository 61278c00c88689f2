// Stage kernels K2a (partition queries by table region), K2b (index probe -- the HBM-roofline
// kernel) and K2c (collect probe hits per read) -- gfx950.
//
// Replaces mm_idx_get() + the occurrence filter of collect_matches() that run inside every
// index.map(seq) call of monica/genomes/aligner.py:193,215.
//
// A random 16-byte gather into a 0.5 GB table costs a whole 64-byte HBM request (measured:
// 6.8 GB fetched for 2.6 GB of algorithmic traffic, profiles/r01b).  So the batch's query
// minimizers are first partitioned by table region (256 .. 1 024 regions of contiguous slots,
// 2 MiB each for the 20-genome index); the probe of one region then runs out of one XCD's
// L2, and HBM only sees the streams: query records in, the table once, hit records out.
//
//   K1 (sketch)  per tile of 4 reads: histogram of its minimizers over the 256 regions
//   scan         bucket-major exclusive scan of the histogram -> run offsets
//   K2a          scatter 8-byte query records into bucket-major order
//   K2b          one wave per (bucket, super-tile of 256 reads) run: probe, compact hits
//   K2c          one workgroup per super-tile: gather its 256 runs, split hits per read
#include "device.h"

namespace mnc {

// ================================================================ K2a: partition
// One workgroup per tile (4 reads, one wave each).  Records are first laid out in LDS in
// bucket order (the tile's histogram row gives the local offsets), then copied out so that
// consecutive lanes write consecutive addresses of a run.  A tile with more records than the
// LDS stage holds (very long reads) writes its records directly.
constexpr int PA_THREADS = 64 * PT_READS;
constexpr int PA_STAGE = 4096;                      // records staged per tile
constexpr int PA_PRE = 16;                          // minimizers a lane fetches ahead (1024 per wave)

// BK: the type that holds a region number in the stage (a byte with 256 regions), OFF: a run's start in the record array
// (32 bits whenever the batch's records fit) -- the kernel's LDS sets its occupancy: 39 KB (four workgroups a CU) with
// 256 regions, 52 KB (three) with 1 024
template <class BK, class OFF>
__global__ __launch_bounds__(PA_THREADS) void mnc_partition_queries(Batch B)
{
	extern __shared__ __align__(16) uint8_t pa_smem[];
	const int pb_n = (int)B.pb_n, pb_bits = B.pb_bits;
	uint64_t *s_rec = reinterpret_cast<uint64_t*>(pa_smem);                   // [PA_STAGE]
	OFF *s_off = reinterpret_cast<OFF*>(s_rec + PA_STAGE);                    // [pb_n]
	uint32_t *s_cur = reinterpret_cast<uint32_t*>(s_off + pb_n);              // [pb_n]
	uint32_t *s_loc = s_cur + pb_n;                                           // [pb_n + 1] local exclusive offsets of the buckets
	BK *s_bkt = reinterpret_cast<BK*>(s_loc + pb_n + 4);                      // [PA_STAGE]
	// Workgroups are dealt round-robin over the 8 XCDs: give each XCD a contiguous range of
	// tiles, so that the short runs of neighbouring tiles (adjacent in a bucket's query array)
	// are merged into full lines by one L2 instead of being written piecemeal by eight
	const uint32_t per_xcd = (B.n_tiles + 7) / 8;
	const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	if (tile >= B.n_tiles) return;
	const int tid = threadIdx.x, lane = lane_id();
	// this wave's read: fetch its first minimizers before anything else waits on memory
	const uint32_t r = tile * PT_READS + (tid >> 6);
	const bool has = r < B.n_reads;
	const int64_t off = has ? B.offsets[r] : 0;
	const int n = has ? B.mz_cnt[r] : 0;
	const uint2 *mz = B.mz + off;
	uint2 pre[PA_PRE];
#pragma unroll
	for (int k = 0; k < PA_PRE; ++k) pre[k] = k * 64 + lane < n ? mz[k * 64 + lane] : make_uint2(0xffffffffu, 0);

	for (int k = tid; k < pb_n; k += PA_THREADS) {
		s_cur[k] = 0;
		s_off[k] = (OFF)B.q_off[(size_t)tile * pb_n + k];
		s_loc[k + 1] = B.hist_tm[(size_t)tile * pb_n + k];
	}
	if (tid == 0) s_loc[0] = 0;
	__syncthreads();
	if (tid < 64) {                                  // inclusive scan of the region counts by one wave: pb_n / 64 per lane
		const int per = pb_n >> 6;                      // 4, 8 or 16
		uint32_t sum = 0;
		for (int k = 0; k < per; ++k) sum += s_loc[1 + tid * per + k];
		uint32_t inc = sum;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (tid >= d) inc += o; }
		uint32_t run = inc - sum;
		for (int k = 0; k < per; ++k) { run += s_loc[1 + tid * per + k]; s_loc[1 + tid * per + k] = run; }
	}
	__syncthreads();
	const uint32_t total = s_loc[pb_n];
	const bool staged = total <= (uint32_t)PA_STAGE;
	uint32_t prev_hash = 0xffffffffu;               // hash of minimizer i0 - 1
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int i = i0 + lane;
		const bool valid = i < n;
		uint2 q = make_uint2(0xffffffffu, 0);
		if (i0 < PA_PRE * 64) {
#pragma unroll
			for (int k = 0; k < PA_PRE; ++k) if (i0 == k * 64) q = pre[k];
		} else if (valid) q = mz[i];
		// tandem flag: same hash as the neighbouring query minimizer (A.4)
		uint32_t left = __shfl_up(q.x, 1), right = __shfl_down(q.x, 1);
		if (lane == 0) left = prev_hash;
		if (lane == 63 || i + 1 >= n) right = (i + 1 < n) ? mz[i + 1].x : 0xffffffffu;
		const bool tandem = valid && (q.x == left || q.x == right);
		prev_hash = __shfl(q.x, 63);
		if (valid) {
			const uint32_t b = pb_bucket(q.x, pb_bits);
			const uint32_t rank = atomicAdd(&s_cur[b], 1u);
			const int64_t dst = (int64_t)s_off[b] + rank;
			const uint64_t rec = (uint64_t)pb_rest(q.x, pb_bits) | (uint64_t)(q.y & 1u) << 22 | (uint64_t)(tandem ? 1u : 0u) << 23 |
			                     (uint64_t)(q.y >> 1) << 24 | (uint64_t)r << 44;
			if (dst >= B.q_cap) *B.overflow = 1u;
			else if (staged) { const uint32_t at = s_loc[b] + rank; s_rec[at] = rec; s_bkt[at] = (BK)b; }
			else B.qrec[dst] = rec;
		}
	}
	__syncthreads();
	if (staged) {
		for (uint32_t i = tid; i < total; i += PA_THREADS) {
			const uint32_t b = s_bkt[i];
			const int64_t dst = (int64_t)s_off[b] + (int64_t)(i - s_loc[b]);
			if (dst < B.q_cap) B.qrec[dst] = s_rec[i];
		}
	}
}

// ---- K2a for an index with more than 256 table regions: G consecutive tiles a workgroup
// With 512 / 1 024 regions a tile's runs shrink to 7 / 3.6 records, and the copy-out above writes eighteen address
// segments per wave store instead of five (0.85 instead of 0.56 ms at 62 genomes).  The runs of consecutive tiles are
// adjacent in a region's record array, so a workgroup over G = regions / 256 tiles (8 / 16 reads, a wave each) writes
// runs of 14.5 records again.  The stage holds a record in six bytes -- the read is one of 4 G, the rest is rebuilt on
// the way out -- and the copy-out goes run by run (sixteen lanes a run), so no region number is kept per record:
// 60 KB of LDS for G = 2 (two workgroups a CU), 108 KB for G = 4 (one: sixteen waves a CU either way).
template <int G, class OFF>
__global__ __launch_bounds__(PA_THREADS * G) void mnc_partition_group(Batch B)
{
	constexpr int THREADS = PA_THREADS * G, STAGE = PA_STAGE * G;
	extern __shared__ __align__(16) uint8_t pa_smem[];
	const int pb_n = (int)B.pb_n, pb_bits = B.pb_bits;
	uint32_t *s_lo = reinterpret_cast<uint32_t*>(pa_smem);                    // [STAGE] rest | strand | tandem | low byte of the position
	OFF *s_off = reinterpret_cast<OFF*>(s_lo + STAGE);                        // [pb_n] start of the group's run in the record array
	uint32_t *s_cur = reinterpret_cast<uint32_t*>(s_off + pb_n);              // [pb_n]
	uint32_t *s_loc = s_cur + pb_n;                                           // [pb_n + 1] local exclusive offsets of the runs
	uint16_t *s_hi = reinterpret_cast<uint16_t*>(s_loc + pb_n + 4);           // [STAGE] position >> 8 | read of the group << 12
	const uint32_t n_groups = (B.n_tiles + G - 1) / G;
	const uint32_t per_xcd = (n_groups + 7) / 8;
	const uint32_t grp = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
	if (grp >= n_groups) return;
	const uint32_t tile0 = grp * G;
	const int tid = threadIdx.x, lane = lane_id();
	const uint32_t r0 = tile0 * PT_READS, r = r0 + (tid >> 6);
	const bool has = r < B.n_reads;
	const int64_t off = has ? B.offsets[r] : 0;
	const int n = has ? B.mz_cnt[r] : 0;
	const uint2 *mz = B.mz + off;
	uint2 pre[PA_PRE];
#pragma unroll
	for (int k = 0; k < PA_PRE; ++k) pre[k] = k * 64 + lane < n ? mz[k * 64 + lane] : make_uint2(0xffffffffu, 0);

	for (int k = tid; k < pb_n; k += THREADS) {
		uint32_t c = 0;
#pragma unroll
		for (int g = 0; g < G; ++g) if (tile0 + g < B.n_tiles) c += B.hist_tm[(size_t)(tile0 + g) * pb_n + k];
		s_cur[k] = 0;
		s_off[k] = (OFF)B.q_off[(size_t)tile0 * pb_n + k];
		s_loc[k + 1] = c;
	}
	if (tid == 0) s_loc[0] = 0;
	__syncthreads();
	if (tid < 64) {
		const int per = pb_n >> 6;
		uint32_t sum = 0;
		for (int k = 0; k < per; ++k) sum += s_loc[1 + tid * per + k];
		uint32_t inc = sum;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (tid >= d) inc += o; }
		uint32_t run = inc - sum;
		for (int k = 0; k < per; ++k) { run += s_loc[1 + tid * per + k]; s_loc[1 + tid * per + k] = run; }
	}
	__syncthreads();
	const uint32_t total = s_loc[pb_n];
	const bool staged = total <= (uint32_t)STAGE;
	const uint32_t lr = (uint32_t)(tid >> 6);
	uint32_t prev_hash = 0xffffffffu;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int i = i0 + lane;
		const bool valid = i < n;
		uint2 q = make_uint2(0xffffffffu, 0);
		if (i0 < PA_PRE * 64) {
#pragma unroll
			for (int k = 0; k < PA_PRE; ++k) if (i0 == k * 64) q = pre[k];
		} else if (valid) q = mz[i];
		uint32_t left = __shfl_up(q.x, 1), right = __shfl_down(q.x, 1);
		if (lane == 0) left = prev_hash;
		if (lane == 63 || i + 1 >= n) right = (i + 1 < n) ? mz[i + 1].x : 0xffffffffu;
		const bool tandem = valid && (q.x == left || q.x == right);
		prev_hash = __shfl(q.x, 63);
		if (valid) {
			const uint32_t b = pb_bucket(q.x, pb_bits);
			const uint32_t rank = atomicAdd(&s_cur[b], 1u);
			const int64_t dst = (int64_t)s_off[b] + rank;
			const uint32_t lo24 = pb_rest(q.x, pb_bits) | (q.y & 1u) << 22 | (tandem ? 1u : 0u) << 23, qpos = q.y >> 1;
			if (dst >= B.q_cap) *B.overflow = 1u;
			else if (staged) { const uint32_t at = s_loc[b] + rank; s_lo[at] = lo24 | qpos << 24; s_hi[at] = (uint16_t)((qpos >> 8 & 0xfffu) | lr << 12); }
			else B.qrec[dst] = (uint64_t)lo24 | (uint64_t)qpos << 24 | (uint64_t)r << 44;
		}
	}
	__syncthreads();
	if (staged) {
		// sixteen lanes a run: consecutive lanes write consecutive records of the record array
		const int sub = tid & 15;
		for (int b = tid >> 4; b < pb_n; b += THREADS / 16) {
			const uint32_t lo = s_loc[b], cnt = s_loc[b + 1] - lo;
			const int64_t dst = (int64_t)s_off[b];
			for (uint32_t i = sub; i < cnt; i += 16) {
				const uint32_t a = s_lo[lo + i], h = s_hi[lo + i];
				if (dst + i < B.q_cap) B.qrec[dst + i] = (uint64_t)a | (uint64_t)(h & 0xfffu) << 32 | (uint64_t)(r0 + (h >> 12)) << 44;
			}
		}
	}
}

// ================================================================ K2b: probe
// One wave per run = (bucket, super-tile of 256 reads), about 930 queries.  Workgroups are
// dealt round-robin over the 8 XCDs, so workgroup g takes bucket (g/8 / W)*8 + g%8: every XCD
// walks its own buckets in order, and the ~256 workgroups resident on it at any time span two
// or three buckets, whose 2 MiB table regions stay hot in its 4 MiB L2 (speed only, never
// correctness).
// THREADS: the workgroup; PR_U: queries per lane in flight; RUNS: consecutive runs (super-tiles) of the region a wave takes.
// What decides the kernel's speed at a large index is how many regions an XCD has open at once: its resident workgroups
// (LDS: two or three a CU) times the queries of a workgroup, over the queries of a region.  With 256 regions a region
// has 363 k queries and a workgroup of 8 waves x 930 takes 7.4 k: two regions open, 2 x 2 MiB of table in the 4 MiB L2.
// With 1 024 regions a region has 91 k and a run 232: workgroups that kept their 7.4 k queries (four runs a wave) had
// EIGHT regions open per XCD -- 16 MiB -- and 64 % of the table reads missed the L2 (profiles/r03l_pmc, 62 genomes).
// So a workgroup takes fewer queries, not more: one run a wave, 16 waves per copy of the region's filter into LDS.
template <int THREADS, int PR_U, int RUNS>
__global__ __launch_bounds__(THREADS) void mnc_probe_buckets(Batch B, uint32_t wgs_per_bucket)
{
	constexpr int PR_THREADS = THREADS;
	const uint32_t g = blockIdx.x, x = g & 7u, seq = g >> 3;
	const uint32_t bucket = (seq / wgs_per_bucket) * 8u + x;
	const uint32_t T0 = ((seq % wgs_per_bucket) * (PR_THREADS / 64) + (threadIdx.x >> 6)) * RUNS;
	// the region's presence filter and displacement table into LDS: 3 of 4 queries are absent
	// from the table and most of them stop at the filter, at LDS speed; a survivor costs one
	// 8-bit LDS read and exactly one 16-byte gather
	__shared__ __align__(16) uint32_t s_filter[PF_WORDS];
	__shared__ __align__(16) uint8_t s_disp[1 << PD_MAX_BITS];
	const int nb = 1 << B.disp_bits;
	if (bucket < B.pb_n) {
		const uint4 *src = reinterpret_cast<const uint4*>(B.filter + (size_t)bucket * PF_WORDS);
		uint4 *dst = reinterpret_cast<uint4*>(s_filter);
#pragma unroll
		for (int k = 0; k < PF_WORDS / 4 / PR_THREADS; ++k) dst[k * PR_THREADS + threadIdx.x] = src[k * PR_THREADS + threadIdx.x];
		const uint8_t *ds = B.disp + (size_t)bucket * nb;
		if (!B.disp_in_lds) {
			// a very large index: more displacement buckets per region than LDS holds; read them in place
		} else if (nb >= 16 * PR_THREADS) {
			for (int k = threadIdx.x; k < nb / 16; k += PR_THREADS)
				reinterpret_cast<uint4*>(s_disp)[k] = reinterpret_cast<const uint4*>(ds)[k];
		} else for (int k = threadIdx.x; k < nb; k += PR_THREADS) s_disp[k] = ds[k];
	}
	__syncthreads();
	if (bucket >= B.pb_n || T0 >= B.n_super) return;
	const int lane = lane_id();
	const unsigned long long lt = (1ULL << lane) - 1ULL;
	const uint32_t mid_occ = (uint32_t)B.mid_occ;
	const int rbits = B.region_bits;
	const uint32_t salt = B.salt[bucket];
	const bool disp_in_lds = B.disp_in_lds != 0;
	const uint8_t *disp_hbm = B.disp + (size_t)bucket * nb;
	const TableSlot *table = B.table + ((size_t)bucket << rbits);
	for (uint32_t T = T0; T < T0 + RUNS && T < B.n_super; ++T) {
		const uint32_t t0 = T * B.ps_tiles, t1 = min(t0 + B.ps_tiles, B.n_tiles);
		const int64_t q0 = q_start(B.q_off, B.n_tiles, bucket, t0, B.pb_n), q1 = q_start(B.q_off, B.n_tiles, bucket, t1, B.pb_n);
		const uint32_t read0 = T * (B.ps_tiles * PT_READS);
		int n_out = 0;
		if (q1 <= B.q_cap) {
			// PR_U queries per lane in flight: their probe chains overlap; the records of the
			// next step are fetched before this step's probes wait on memory
			uint64_t nxt[PR_U];
#pragma unroll
			for (int u = 0; u < PR_U; ++u) { const int64_t i = q0 + u * 64 + lane; nxt[u] = i < q1 ? B.qrec[i] : 0; }
			for (int64_t i0 = q0; i0 < q1; i0 += 64 * PR_U) {
				uint64_t rec[PR_U], val[PR_U];
				uint32_t cnt[PR_U], want[PR_U], slot[PR_U];
				bool pend[PR_U];
#pragma unroll
				for (int u = 0; u < PR_U; ++u) {
					pend[u] = i0 + u * 64 + lane < q1;
					rec[u] = nxt[u];
					cnt[u] = 0, val[u] = 0;
					const int64_t j = i0 + 64 * PR_U + u * 64 + lane;
					nxt[u] = j < q1 ? B.qrec[j] : 0;
				}
#pragma unroll
				for (int u = 0; u < PR_U; ++u) {
					const uint32_t rest = (uint32_t)rec[u] & 0x3fffffu;
					const uint32_t fm = pf_mask(rest);
					pend[u] = pend[u] && (s_filter[pf_word(rest)] & fm) == fm;
					want[u] = pb_hash(rest, bucket, B.pb_bits) + 1;
					const uint32_t db = rest & (uint32_t)(nb - 1);
					slot[u] = pd_slot(rest, disp_in_lds ? s_disp[db] : (pend[u] ? disp_hbm[db] : 0), rbits, salt);
				}
				TableSlot sl[PR_U];
#pragma unroll
				for (int u = 0; u < PR_U; ++u) if (pend[u]) sl[u] = table[slot[u]];
#pragma unroll
				for (int u = 0; u < PR_U; ++u) if (pend[u] && sl[u].key == want[u]) cnt[u] = sl[u].cnt, val[u] = sl[u].val;
#pragma unroll
				for (int u = 0; u < PR_U; ++u) {
					const bool emit = cnt[u] > 0;       // a hit, or a too-frequent minimizer (for rep_len)
					const unsigned long long m = __ballot(emit);
					if (emit) {
						HitRec h;
						const uint32_t qpos = (uint32_t)(rec[u] >> 24) & 0xfffffu, rd = (uint32_t)(rec[u] >> 44);
						h.val = val[u];
						h.qinfo = qpos << 1 | ((uint32_t)(rec[u] >> 22) & 1u) | (rd - read0) << 21;
						h.cnt = cnt[u] >= mid_occ ? HIT_HIGH : (cnt[u] | ((uint32_t)(rec[u] >> 23) & 1u) << 31);
						B.bhits[q0 + n_out + __popcll(m & lt)] = h;
					}
					n_out += __popcll(m);
				}
			}
		}
		if (lane == 0) B.bhit_cnt[(size_t)T * B.pb_n + bucket] = (uint32_t)n_out;
	}
}

// ================================================================ K2c: collect
// One workgroup per super-tile: its 256 runs of hits are read back and dealt to the 256 reads
// (hits from the bottom of the read's slot range, too-frequent minimizers from the top), then
// one wave per read finishes hit_cnt / an_cnt / rep_len.
constexpr int CO_THREADS = 1024;

__global__ __launch_bounds__(CO_THREADS) void mnc_collect_hits(Batch B)
{
	__shared__ int64_t s_start[PB_N_MAX];
	__shared__ uint32_t s_pre[PB_N_MAX + 1];
	__shared__ uint32_t s_cur[PS_TILES_MIN * PT_READS], s_hi[PS_TILES_MIN * PT_READS];
	__shared__ unsigned long long s_an[PS_TILES_MIN * PT_READS];
	const uint32_t T = blockIdx.x;
	const uint32_t t0 = T * B.ps_tiles;
	const int tid = threadIdx.x;
	const int PB_N = (int)B.pb_n, SUPER_READS = (int)(B.ps_tiles * PT_READS);
	for (int b = tid; b < PB_N; b += CO_THREADS) {
		s_start[b] = B.q_off[(size_t)t0 * PB_N + b];
		s_pre[b + 1] = B.bhit_cnt[(size_t)T * PB_N + b];
	}
	if (tid < SUPER_READS) s_cur[tid] = 0, s_hi[tid] = 0, s_an[tid] = 0;
	if (tid == 0) s_pre[0] = 0;
	__syncthreads();
	const uint32_t read0 = T * SUPER_READS;
	// a wave takes whole runs (their hits are contiguous), lanes take the hits of a run
	for (int b = tid >> 6; b < PB_N; b += CO_THREADS / 64) {
		const uint32_t cnt = s_pre[b + 1];
		const HitRec *src = B.bhits + s_start[b];
		for (uint32_t f = tid & 63; f < cnt; f += 64) {
			HitRec h = src[f];
			const uint32_t lr = (h.qinfo >> 21) & (SUPER_READS - 1);
			const uint32_t r = read0 + lr;
			const int64_t off = B.offsets[r];
			h.qinfo &= 0x1fffffu;
			if (h.cnt == HIT_HIGH) {
				const int64_t cap = B.offsets[r + 1] - off;
				B.hits[off + cap - 1 - atomicAdd(&s_hi[lr], 1u)] = h;
			} else {
				B.hits[off + atomicAdd(&s_cur[lr], 1u)] = h;
				atomicAdd(&s_an[lr], (unsigned long long)(h.cnt & 0x7fffffffu));
			}
		}
	}
	__syncthreads();
	// per read: counts and rep_len = sum over too-frequent minimizers, in position order, of
	// min(k, distance to the previous one)
	const int lane = lane_id();
	for (uint32_t lr = tid >> 6; lr < (uint32_t)SUPER_READS; lr += CO_THREADS / 64) {
		const uint32_t r = read0 + lr;
		if (r >= B.n_reads) break;
		const int64_t off = B.offsets[r];
		const int64_t cap = B.offsets[r + 1] - off;
		const int nh = (int)s_hi[lr];
		int rep = 0;
		for (int i0 = 0; i0 < nh; i0 += 64) {
			const int i = i0 + lane;
			int c = 0;
			if (i < nh) {
				const int en = (int)(B.hits[off + cap - 1 - i].qinfo >> 1) + 1;
				int prev = -1;
				for (int j = 0; j < nh; ++j) {
					const int o = (int)(B.hits[off + cap - 1 - j].qinfo >> 1) + 1;
					if (o < en && o > prev) prev = o;
				}
				c = prev < 0 ? KMER : min(KMER, en - prev);
			}
#pragma unroll
			for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
			rep += c;
		}
		if (lane == 0) {
			B.hit_cnt[r] = (int32_t)s_cur[lr];
			B.an_cnt[r] = (int64_t)s_an[lr];
			B.rep_len[r] = rep;
		}
	}
}

template <int G, class OFF>
static void launch_partition_group(const Batch &B, hipStream_t st)
{
	const size_t lds = (size_t)PA_STAGE * G * 6 + (size_t)B.pb_n * (sizeof(OFF) + 8) + 16;
	const uint32_t n_groups = (B.n_tiles + G - 1) / G;
	hipLaunchKernelGGL((mnc_partition_group<G, OFF>), dim3((n_groups + 7) / 8 * 8), dim3(PA_THREADS * G), lds, st, B);
}

// dynamic LDS above 64 KiB is opted into once per function (and device: called when an engine is made)
int partition_prepare()
{
	const int lds = PA_STAGE * 4 * 6 + PB_N_MAX * 16 + 16;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_partition_group<4, uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_partition_group<4, int64_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	// (two tiles a workgroup at 1 024 regions -- MNC_PARTITION_G=2 -- with 64-bit run offsets is 65 552 bytes: above 64 KiB too)
	if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_partition_group<2, uint32_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_partition_group<2, int64_t>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
	if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}

void launch_partition(const Batch &B, hipStream_t st)
{
	if (B.n_tiles == 0) return;
	const bool small = B.q_cap < (1LL << 32);
	// more than 256 regions: a workgroup over regions / 256 tiles (MNC_PARTITION_TILES=1: the one-tile form, for comparison)
	static const bool one_tile = getenv("MNC_PARTITION_TILES") && atoi(getenv("MNC_PARTITION_TILES")) == 1;
	static const int force_g = getenv("MNC_PARTITION_G") ? atoi(getenv("MNC_PARTITION_G")) : 0;   // (2: two tiles a workgroup at 1 024 regions too, for comparison)
	if (B.pb_n > 256 && !one_tile) {
		if (B.pb_n <= 512 || force_g == 2) { if (small) launch_partition_group<2, uint32_t>(B, st); else launch_partition_group<2, int64_t>(B, st); }
		else { if (small) launch_partition_group<4, uint32_t>(B, st); else launch_partition_group<4, int64_t>(B, st); }
		return;
	}
	const size_t fixed = (size_t)PA_STAGE * 8 + (size_t)B.pb_n * (small ? 12 : 16) + 16, bk = (size_t)PA_STAGE * (B.pb_n <= 256 ? 1 : 2);
	const dim3 grid((B.n_tiles + 7) / 8 * 8);
	if (B.pb_n <= 256 && small) hipLaunchKernelGGL((mnc_partition_queries<uint8_t, uint32_t>), grid, dim3(PA_THREADS), fixed + bk, st, B);
	else if (B.pb_n <= 256) hipLaunchKernelGGL((mnc_partition_queries<uint8_t, int64_t>), grid, dim3(PA_THREADS), fixed + bk, st, B);
	else if (small) hipLaunchKernelGGL((mnc_partition_queries<uint16_t, uint32_t>), grid, dim3(PA_THREADS), fixed + bk, st, B);
	else hipLaunchKernelGGL((mnc_partition_queries<uint16_t, int64_t>), grid, dim3(PA_THREADS), fixed + bk, st, B);
}

template <int THREADS, int PR_U, int RUNS>
static void launch_probe_as(const Batch &B, hipStream_t st)
{
	const uint32_t per_wg = (THREADS / 64) * RUNS;                              // super-tiles a workgroup takes
	const uint32_t W = (B.n_super + per_wg - 1) / per_wg;
	hipLaunchKernelGGL((mnc_probe_buckets<THREADS, PR_U, RUNS>), dim3(B.pb_n * W), dim3(THREADS), 0, st, B, W);
}

void launch_probe(const Batch &B, hipStream_t st)
{
	if (B.n_super == 0) return;
	// (1 024 regions: 16 waves x one run measured against 8 x 1 and 4 x 1 -- 0.73 / 0.79 / 1.02 ms at 62 genomes: fewer
	// queries per copy of the filter cost more than the locality gives, profiles/r03n_probe_shapes.txt)
	if (B.pb_n <= 512) launch_probe_as<512, 8, 1>(B, st);
	else launch_probe_as<1024, 4, 1>(B, st);
}

void launch_collect(const Batch &B, hipStream_t st)
{
	if (B.n_super == 0) return;
	hipLaunchKernelGGL(mnc_collect_hits, dim3(B.n_super), dim3(CO_THREADS), 0, st, B);
}

} // namespace mnc
