// Stage kernel K3: expand probe hits into anchors and sort them -- gfx950.
//
// Replaces the anchor generation and radix sort of collect_seed_hits() inside index.map(seq)
// (monica/genomes/aligner.py:193,215).  Upstream sorts by x only
// with an unstable radix sort; here the order is the total order (x, y), as in the oracle.
//
// One 256-thread workgroup per read; reads come in size classes (anchors per read) so the
// LDS tile is as small as the class allows.  Anchors are generated straight into LDS, sorted
// there with a bitonic network whose compare-exchanges are all ascending (so the +inf
// padding never moves), and written out once, sorted.  Reads with more anchors than the
// LDS tile fall back to the same network on the read's HBM segment.
#include "device.h"

namespace mnc {

constexpr int SO_THREADS = 128;

__device__ __forceinline__ bool anchor_less(const Anchor &a, const Anchor &b)
{
	return a.x < b.x || (a.x == b.x && a.y < b.y);
}

// ascending-only bitonic network over s[0..n) (n arbitrary; indices >= n act as +inf)
template <class Ptr>
__device__ void bitonic_sort(Ptr s, int n, int tid)
{
	int lg = 1;
	while ((1 << lg) < n) ++lg;
	const int half_n = 1 << (lg - 1);
	for (int lh = 1; lh <= lg; ++lh) {                              // h = 1 << lh
		{
			const int lhh = lh - 1, hh = 1 << lhh;                     // flip: i <-> block_end - (i - block_start)
			for (int t = tid; t < half_n; t += SO_THREADS) {
				const int q = (t >> lhh) << lh, o = t & (hh - 1);
				const int x = q + o, y = q + (hh << 1) - 1 - o;
				if (y < n) {
					Anchor ax = s[x], ay = s[y];
					if (anchor_less(ay, ax)) s[x] = ay, s[y] = ax;
				}
			}
			__syncthreads();
		}
		for (int lhh = lh - 2; lhh >= 0; --lhh) {                      // disperse: i <-> i + hh
			const int hh = 1 << lhh;
			for (int t = tid; t < half_n; t += SO_THREADS) {
				const int q = (t >> lhh) << (lhh + 1), o = t & (hh - 1);
				const int x = q + o, y = x + hh;
				if (y < n) {
					Anchor ax = s[x], ay = s[y];
					if (anchor_less(ay, ax)) s[x] = ay, s[y] = ax;
				}
			}
			__syncthreads();
		}
	}
}

// ---------------------------------------------------------------- packed anchors
// (x, y) order = (strand, contig, reference position, tandem flag, query position): with few
// enough contigs all of it fits one 64-bit word (Batch::rid_bits / rpos_bits), which halves
// the LDS traffic of the network and makes a compare-exchange one 64-bit compare.
__device__ __forceinline__ uint64_t pack_anchor(const Anchor &e, int rid_bits, int rpos_bits)
{
	const uint64_t strand = e.x >> 63, rid = e.x << 1 >> 33, rpos = (uint32_t)e.x;
	const uint64_t flag = e.y >> 42 & 1ULL, qpos = (uint32_t)e.y & 0xfffffu;
	return (((strand << rid_bits | rid) << rpos_bits | rpos) << 21) | flag << 20 | qpos;
}

__device__ __forceinline__ Anchor unpack_anchor(uint64_t w, int rid_bits, int rpos_bits)
{
	Anchor e;
	const uint64_t hi = w >> 21;
	const uint64_t rpos = hi & ((1ULL << rpos_bits) - 1ULL), rid = hi >> rpos_bits & ((1ULL << rid_bits) - 1ULL);
	const uint64_t strand = hi >> (rpos_bits + rid_bits) & 1ULL;
	e.x = strand << 63 | rid << 32 | rpos;
	e.y = (w >> 20 & 1ULL) << 42 | (uint64_t)KMER << 32 | (w & 0xfffffULL);
	return e;
}

__device__ __forceinline__ void cx(uint64_t &a, uint64_t &b)
{
	const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
	a = lo, b = hi;
}

// The same ascending-only network, with every run of compare-exchanges at distances 2 and 1
// (and the whole of the first two stages) done on four consecutive elements in registers: one
// 32-byte read and write per thread instead of two LDS rounds.  Slots >= n act as +inf.
__device__ void bitonic_sort_u64(uint64_t *s, int n, int tid)
{
	int lg = 1;
	while ((1 << lg) < n) ++lg;
	const int half_n = 1 << (lg - 1);
	const int quads = (n + 3) >> 2;
	const uint64_t INF = ~0ULL;
	auto load4 = [&](int g, uint64_t (&r)[4]) {
#pragma unroll
		for (int k = 0; k < 4; ++k) r[k] = 4 * g + k < n ? s[4 * g + k] : INF;
	};
	auto store4 = [&](int g, const uint64_t (&r)[4]) {
#pragma unroll
		for (int k = 0; k < 4; ++k) if (4 * g + k < n) s[4 * g + k] = r[k];
	};
	// stages h = 2 and h = 4
	for (int g = tid; g < quads; g += SO_THREADS) {
		uint64_t r[4];
		load4(g, r);
		cx(r[0], r[1]), cx(r[2], r[3]);                             // h = 2: flip
		if (lg >= 2) { cx(r[0], r[3]), cx(r[1], r[2]); cx(r[0], r[1]), cx(r[2], r[3]); }   // h = 4: flip, then distance 1
		store4(g, r);
	}
	__syncthreads();
	for (int lh = 3; lh <= lg; ++lh) {                              // h = 1 << lh
		{
			const int lhh = lh - 1, hh = 1 << lhh;                     // flip: i <-> block_end - (i - block_start)
			for (int t = tid; t < half_n; t += SO_THREADS) {
				const int q = (t >> lhh) << lh, o = t & (hh - 1);
				const int x = q + o, y = q + (hh << 1) - 1 - o;
				if (y < n) {
					const uint64_t ax = s[x], ay = s[y];
					if (ay < ax) s[x] = ay, s[y] = ax;
				}
			}
			__syncthreads();
		}
		for (int lhh = lh - 2; lhh >= 2; --lhh) {                      // disperse at distances >= 4: i <-> i + hh
			const int hh = 1 << lhh;
			for (int t = tid; t < half_n; t += SO_THREADS) {
				const int q = (t >> lhh) << (lhh + 1), o = t & (hh - 1);
				const int x = q + o, y = x + hh;
				if (y < n) {
					const uint64_t ax = s[x], ay = s[y];
					if (ay < ax) s[x] = ay, s[y] = ax;
				}
			}
			__syncthreads();
		}
		for (int g = tid; g < quads; g += SO_THREADS) {                // distances 2 and 1 in registers
			uint64_t r[4];
			load4(g, r);
			cx(r[0], r[2]), cx(r[1], r[3]);
			cx(r[0], r[1]), cx(r[2], r[3]);
			store4(g, r);
		}
		__syncthreads();
	}
}

// `list`/`count` = the reads of one size class (at most NM anchors each; NM == 0: the class
// of reads too large for LDS, sorted in their HBM segment); LDS tile = NM anchors (PACKED: NM
// 64-bit words).
template <bool PACKED>
__global__ __launch_bounds__(SO_THREADS) void mnc_expand_sort(Batch B, const uint32_t *lists, ClassSpans spans, int NM)
{
	extern __shared__ __align__(16) uint8_t so_smem[];
	Anchor *s_a = reinterpret_cast<Anchor*>(so_smem);
	uint64_t *s_w = reinterpret_cast<uint64_t*>(so_smem);
	__shared__ int s_scan[SO_THREADS / 64];

	// the reads of one or several adjacent size classes: block -> (class, index in its list)
	if (blockIdx.x >= spans.start[spans.n]) return;
	int cls = 0;
	while (blockIdx.x >= spans.start[cls + 1]) ++cls;
	const uint32_t r = lists[(size_t)cls * spans.stride + (blockIdx.x - spans.start[cls])];
	const int tid = threadIdx.x;
	const int64_t off = B.offsets[r];
	const int qlen = (int)(B.offsets[r + 1] - off);
	const int nh = B.hit_cnt[r];
	const int64_t a_off = B.an_off[r];
	const int64_t n64 = B.an_off[r + 1] - a_off;
	if (n64 <= 0) return;
	const int n = (int)n64;
	const bool in_lds = n <= NM;
	const int rid_bits = B.rid_bits, rpos_bits = B.rpos_bits;
	Anchor *g = B.a + a_off;
	const HitRec *hits = B.hits + off;

	// ---- expand: exclusive prefix of per-hit occurrence counts, tile by tile
	int base = 0;
	for (int i0 = 0; i0 < nh; i0 += SO_THREADS) {
		const int i = i0 + tid;
		HitRec h;
		h.val = 0, h.qinfo = 0, h.cnt = 0;
		if (i < nh) h = hits[i];
		const int cnt = (int)(h.cnt & 0x7fffffffu);
		// inclusive scan inside the wave, then across the waves
		int inc = cnt;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			int o = __shfl_up(inc, d);
			if ((tid & 63) >= d) inc += o;
		}
		if ((tid & 63) == 63) s_scan[tid >> 6] = inc;
		__syncthreads();
		int wbase = 0, all = 0;
#pragma unroll
		for (int w = 0; w < SO_THREADS / 64; ++w) {
			if (w < (tid >> 6)) wbase += s_scan[w];
			all += s_scan[w];
		}
		const int pos = base + wbase + inc - cnt;
		if (cnt > 0) {
			const uint32_t q_pos = h.qinfo >> 1, q_strand = h.qinfo & 1u;
			const uint64_t flags = (h.cnt >> 31) ? (1ULL << 42) : 0ULL;
			for (int k = 0; k < cnt; ++k) {
				const uint64_t rw = cnt == 1 ? h.val : B.positions[h.val + (uint64_t)k];
				const uint32_t rpos = (uint32_t)rw >> 1;
				Anchor e;
				if ((rw & 1ULL) == (uint64_t)q_strand) {
					e.x = (rw & 0xffffffff00000000ULL) | rpos;
					e.y = (uint64_t)KMER << 32 | q_pos;
				} else {
					e.x = 1ULL << 63 | (rw & 0xffffffff00000000ULL) | rpos;
					e.y = (uint64_t)KMER << 32 | (uint32_t)(qlen - ((int)q_pos + 1 - KMER) - 1);
				}
				e.y |= flags;
				if (!in_lds) g[pos + k] = e;
				else if (PACKED) s_w[pos + k] = pack_anchor(e, rid_bits, rpos_bits);
				else s_a[pos + k] = e;
			}
		}
		base += all;
		__syncthreads();
	}
	if (!in_lds) __threadfence_block();
	__syncthreads();

	// ---- sort by (x, y)
	if (!in_lds) {
		bitonic_sort(g, n, tid);
	} else if (PACKED) {
		bitonic_sort_u64(s_w, n, tid);
		for (int i = tid; i < n; i += SO_THREADS) g[i] = unpack_anchor(s_w[i], rid_bits, rpos_bits);
	} else {
		bitonic_sort(s_a, n, tid);
		for (int i = tid; i < n; i += SO_THREADS) g[i] = s_a[i];
	}
}

// `lists` + `spans`: the read lists of adjacent size classes, all sorted with one tile of NM anchors
void launch_expand_sort(const Batch &B, const uint32_t *lists, const ClassSpans &spans, int NM, hipStream_t st)
{
	const uint32_t count = spans.start[spans.n];
	if (count == 0) return;
	if (B.rid_bits > 0 && NM > 0)
		hipLaunchKernelGGL(mnc_expand_sort<true>, dim3(count), dim3(SO_THREADS), (size_t)NM * sizeof(uint64_t), st, B, lists, spans, NM);
	else
		hipLaunchKernelGGL(mnc_expand_sort<false>, dim3(count), dim3(SO_THREADS), (size_t)NM * sizeof(Anchor), st, B, lists, spans, NM);
}

int expand_sort_prepare(int max_nm)
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_expand_sort<false>),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)max_nm * sizeof(Anchor)));
	if (e == hipSuccess)
		e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_expand_sort<true>),
		                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)max_nm * sizeof(uint64_t)));
	if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}

} // namespace mnc
