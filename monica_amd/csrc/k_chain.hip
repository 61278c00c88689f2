// Stage kernel K5, sequential form (backtrack -> chains for reads too large for the LDS
// classes of k_chain_ring.hip) and the size-class binning -- gfx950.
//
// Replaces mm_chain_dp() inside index.map(seq) (monica/genomes/aligner.py:193,215;
// SURVEY.md Appendix A.5).  All arithmetic is integer: the one floating-point term of the
// gap cost, (int)(dd * .01 * avg_span), is a host-computed look-up (avg_span == k exactly).
#include "device.h"

namespace mnc {

// ================================================================ K5: backtrack
__device__ void heapsort_u64(uint64_t *a, int n)
{
	for (int start = n / 2 - 1; start >= 0; --start) {
		int root = start;
		for (;;) {
			int c = 2 * root + 1;
			if (c >= n) break;
			if (c + 1 < n && a[c] < a[c + 1]) ++c;
			if (a[root] >= a[c]) break;
			uint64_t x = a[root]; a[root] = a[c], a[c] = x;
			root = c;
		}
	}
	for (int end = n - 1; end > 0; --end) {
		uint64_t x = a[0]; a[0] = a[end], a[end] = x;
		int root = 0;
		for (;;) {
			int c = 2 * root + 1;
			if (c >= end) break;
			if (c + 1 < end && a[c] < a[c + 1]) ++c;
			if (a[root] >= a[c]) break;
			uint64_t y = a[root]; a[root] = a[c], a[c] = y;
			root = c;
		}
	}
}

// One thread per read: chain ends -> peak walk -> best-first backtrack -> chain records
// ordered by (first anchor x, rank in backtrack order).
__global__ __launch_bounds__(64) void mnc_chain_backtrack(Batch B, const uint32_t *read_list, uint32_t n_list)
{
	const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
	if (li >= n_list) return;
	const uint32_t r = read_list ? read_list[li] : li;
	const int64_t a_off = B.an_off[r];
	const int n = (int)(B.an_off[r + 1] - a_off);
	B.n_chain[r] = 0;
	if (n <= 0) return;
	const Anchor *a = B.a + a_off;
	const int32_t *f = B.f + a_off, *p = B.p + a_off;
	int32_t *v = B.v + a_off, *t = B.t + a_off;
	uint64_t *u = B.u + a_off;
	ChainRec *out = B.chains_tmp + a_off / 3;              // backtrack order; K6 orders by first anchor
	const int min_sc = B.min_sc, min_cnt = B.min_cnt;

	for (int i = 0; i < n; ++i) t[i] = 0;
	for (int i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
	int n_u = 0;
	for (int i = 0; i < n; ++i) {
		if (t[i] == 0 && v[i] >= min_sc) {
			int j = i;
			while (j >= 0 && f[j] < v[j]) j = p[j];
			if (j < 0) j = i;
			u[n_u++] = (uint64_t)(uint32_t)f[j] << 32 | (uint32_t)j;
		}
	}
	if (n_u == 0) return;
	heapsort_u64(u, n_u);                              // ascending; walk it from the top

	for (int i = 0; i < n; ++i) t[i] = 0;
	int n_v = 0, k = 0;
	for (int ii = n_u - 1; ii >= 0; --ii) {            // best peak first (ties: larger index first)
		const uint64_t ui = u[ii];
		const int n_v0 = n_v;
		int j = (int32_t)(uint32_t)ui;
		do {
			v[n_v++] = j;
			t[j] = 1;
			j = p[j];
		} while (j >= 0 && t[j] == 0);
		int score = -1;
		if (j < 0) score = (int32_t)(ui >> 32);
		else if ((int32_t)(ui >> 32) - f[j] >= min_sc) score = (int32_t)(ui >> 32) - f[j];
		const int cnt = n_v - n_v0;
		if (score >= 0 && cnt >= min_cnt) {
			// chain record: anchors are v[n_v0 .. n_v) in last-to-first order
			ChainRec c;
			const Anchor first = a[v[n_v - 1]], last = a[v[n_v0]];
			c.x0 = first.x, c.y0 = first.y, c.x1 = last.x, c.y1 = last.y;
			c.score = score, c.cnt = cnt;
			int32_t span = (int32_t)(first.y >> 32 & 0xff);
			c.mlen = c.blen = span;
			Anchor prev = first;
			for (int m = n_v - 2; m >= n_v0; --m) {
				const Anchor cur = a[v[m]];
				const int sp = (int)(cur.y >> 32 & 0xff);
				const int tl = (int32_t)cur.x - (int32_t)prev.x;
				const int ql = (int32_t)cur.y - (int32_t)prev.y;
				c.blen += tl > ql ? tl : ql;
				c.mlen += tl > sp && ql > sp ? sp : tl < ql ? tl : ql;
				prev = cur;
			}
			c.as = v[n_v0], c.pad = k;
			out[k++] = c;
		} else n_v = n_v0;
	}
	B.n_chain[r] = k;
}

void launch_backtrack(const Batch &B, const uint32_t *read_list, uint32_t n_list, hipStream_t st)
{
	if (n_list == 0) return;
	hipLaunchKernelGGL(mnc_chain_backtrack, dim3((n_list + 63) / 64), dim3(64), 0, st, B, read_list, n_list);
}

// ================================================================ size classes for the row kernel
// One thread per read: smallest class whose LDS tile holds the read's anchors, or the
// "large" class (sequential kernels) for > 4096 anchors or >= 65 536 bases.
__global__ __launch_bounds__(256) void mnc_bin_reads(Batch B, ChainClasses C, uint32_t *cls_count, uint32_t *cls_list)
{
	// block-level histogram in LDS, then one global atomic per class and block
	__shared__ uint32_t s_cnt[MAX_CHAIN_CLASSES + 1], s_base[MAX_CHAIN_CLASSES + 1];
	if (threadIdx.x <= MAX_CHAIN_CLASSES) s_cnt[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	int c = -1;
	uint32_t local = 0;
	if (r < B.n_reads) {
		const int64_t n = B.an_cnt[r];
		B.n_chain[r] = 0;
		if (n > 0) {
			const int64_t qlen = B.offsets[r + 1] - B.offsets[r];
			c = C.n;                                   // large
			if (qlen < 65536) for (int k = 0; k < C.n; ++k) if (n <= C.nm[k]) { c = k; break; }
			local = atomicAdd(&s_cnt[c], 1u);
		}
	}
	__syncthreads();
	if (threadIdx.x <= (unsigned)C.n && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cls_count[threadIdx.x], s_cnt[threadIdx.x]);
	__syncthreads();
	if (c >= 0) cls_list[(size_t)c * B.n_reads + s_base[c] + local] = r;
}

void launch_bin_reads(const Batch &B, const ChainClasses &C, uint32_t *cls_count, uint32_t *cls_list, hipStream_t st)
{
	if (B.n_reads == 0) return;
	hipLaunchKernelGGL(mnc_bin_reads, dim3((B.n_reads + 255) / 256), dim3(256), 0, st, B, C, cls_count, cls_list);
}

} // namespace mnc
