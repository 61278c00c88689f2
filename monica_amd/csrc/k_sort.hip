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
	int npad = 2;
	while (npad < n) npad <<= 1;
	const int half_n = npad >> 1;
	for (int h = 2; h <= npad; h <<= 1) {
		// flip: i <-> block_end - (i - block_start)
		{
			const int hh = h >> 1;
			for (int t = tid; t < half_n; t += SO_THREADS) {
				const int q = (t / hh) * h, o = t % hh;
				const int x = q + o, y = q + h - 1 - o;
				if (y < n) {
					Anchor ax = s[x], ay = s[y];
					if (anchor_less(ay, ax)) s[x] = ay, s[y] = ax;
				}
			}
			__syncthreads();
		}
		for (int hh = h >> 2; hh > 0; hh >>= 1) {       // disperse: i <-> i + hh
			for (int t = tid; t < half_n; t += SO_THREADS) {
				const int q = (t / hh) * (hh << 1), o = t % hh;
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

// `list`/`count` = the reads of one size class (at most NM anchors each; NM == 0: the class
// of reads too large for LDS, sorted in their HBM segment); LDS tile = NM anchors.
__global__ __launch_bounds__(SO_THREADS) void mnc_expand_sort(Batch B, const uint32_t *list, uint32_t count, int NM)
{
	extern __shared__ __align__(16) uint8_t so_smem[];
	Anchor *s_a = reinterpret_cast<Anchor*>(so_smem);
	__shared__ int s_scan[SO_THREADS / 64];

	if (blockIdx.x >= count) return;
	const uint32_t r = list[blockIdx.x];
	const int tid = threadIdx.x;
	const int64_t off = B.offsets[r];
	const int qlen = (int)(B.offsets[r + 1] - off);
	const int nh = B.hit_cnt[r];
	const int64_t a_off = B.an_off[r];
	const int64_t n64 = B.an_off[r + 1] - a_off;
	if (n64 <= 0) return;
	const int n = (int)n64;
	const bool in_lds = n <= NM;
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
		// inclusive scan inside the wave, then across the 4 waves
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
				if (in_lds) s_a[pos + k] = e; else g[pos + k] = e;
			}
		}
		base += all;
		__syncthreads();
	}
	if (!in_lds) __threadfence_block();
	__syncthreads();

	// ---- sort by (x, y)
	if (in_lds) {
		bitonic_sort(s_a, n, tid);
		for (int i = tid; i < n; i += SO_THREADS) g[i] = s_a[i];
	} else {
		bitonic_sort(g, n, tid);
	}
}

void launch_expand_sort(const Batch &B, const uint32_t *list, uint32_t count, int NM, hipStream_t st)
{
	if (count == 0) return;
	hipLaunchKernelGGL(mnc_expand_sort, dim3(count), dim3(SO_THREADS), (size_t)NM * sizeof(Anchor), st, B, list, count, NM);
}

int expand_sort_prepare(int max_nm)
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_expand_sort),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)max_nm * sizeof(Anchor)));
	if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}

} // namespace mnc
