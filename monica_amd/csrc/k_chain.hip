// Stage kernels K4 (chaining DP) and K5 (backtrack -> chains) -- gfx950.
//
// Replaces mm_chain_dp() inside index.map(seq) (monica/genomes/aligner.py:193,215;
// SURVEY.md Appendix A.5).  All arithmetic is integer: the one floating-point term of the
// gap cost, (int)(dd * .01 * avg_span), is a host-computed look-up (avg_span == k exactly).
#include "device.h"

namespace mnc {

// ================================================================ K4, form 1: one thread per read
// Straight sequential evaluation; used as the in-library cross-check of the wave kernel
// and for reads whose anchor count exceeds what the wave kernel keeps in LDS.
__global__ __launch_bounds__(64) void mnc_chain_dp_serial(Batch B, const uint32_t *read_list, uint32_t n_list)
{
	const uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
	if (li >= n_list) return;
	const uint32_t r = read_list ? read_list[li] : li;
	const int64_t a_off = B.an_off[r];
	const int n = (int)(B.an_off[r + 1] - a_off);
	if (n <= 0) return;
	const Anchor *a = B.a + a_off;
	int32_t *f = B.f + a_off, *p = B.p + a_off, *v = B.v + a_off, *t = B.t + a_off;
	const uint64_t max_dist = (uint64_t)B.max_gap;
	for (int i = 0; i < n; ++i) t[i] = 0;
	int st = 0;
	for (int i = 0; i < n; ++i) {
		const uint64_t ri = a[i].x;
		const int32_t qi = (int32_t)a[i].y, q_span = (int32_t)(a[i].y >> 32 & 0xff);
		int32_t max_f = q_span, max_j = -1, n_skip = 0;
		while (st < i && ri > a[st].x + max_dist) ++st;
		if (i - st > B.max_iter) st = i - B.max_iter;
		for (int j = i - 1; j >= st; --j) {
			const int64_t dr = (int64_t)(ri - a[j].x);
			const int32_t dq = qi - (int32_t)a[j].y;
			if (dr == 0 || dq <= 0 || dq > B.max_gap) continue;
			const int32_t dd = dr > dq ? (int32_t)(dr - dq) : (int32_t)(dq - dr);
			if (dd > B.bw) continue;
			const int32_t min_d = dq < dr ? dq : (int32_t)dr;
			int32_t sc = min_d > q_span ? q_span : min_d;
			sc -= B.gap_lut[dd];
			sc += f[j];
			if (sc > max_f) {
				max_f = sc, max_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == i) {
				if (++n_skip > B.max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = i;
		}
		f[i] = max_f, p[i] = max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;
	}
}

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
__global__ __launch_bounds__(64) void mnc_chain_backtrack(Batch B)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= B.n_reads) return;
	const int64_t a_off = B.an_off[r];
	const int n = (int)(B.an_off[r + 1] - a_off);
	B.n_chain[r] = 0;
	if (n <= 0) return;
	const Anchor *a = B.a + a_off;
	const int32_t *f = B.f + a_off, *p = B.p + a_off;
	int32_t *v = B.v + a_off, *t = B.t + a_off;
	uint64_t *u = B.u + a_off;
	const int64_t slot = a_off / 3;
	ChainRec *tmp = B.chains_tmp + slot;
	ChainRec *out = B.chains + slot;
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
			c.as = 0, c.pad = k;
			tmp[k++] = c;
		} else n_v = n_v0;
	}
	if (k == 0) return;
	// order by (x0, rank): keys are distinct.  Insertion sort on the (small) record list
	// for short lists, heap sort on an index key otherwise.
	if (k <= 16) {
		for (int i = 0; i < k; ++i) {
			ChainRec c = tmp[i];
			int j = i - 1;
			while (j >= 0 && (out[j].x0 > c.x0 || (out[j].x0 == c.x0 && out[j].pad > c.pad))) { out[j + 1] = out[j]; --j; }
			out[j + 1] = c;
		}
	} else {
		// sort ranks by x0 with a stable two-key trick: x0 may use all 64 bits, so sort an
		// index array by repeated selection through a heap on (x0, rank)
		int32_t *idx = B.tmp_i32 + slot * 4;           // >= k ints available (4 per slot)
		for (int i = 0; i < k; ++i) idx[i] = i;
		auto less = [&](int x, int y) { return tmp[x].x0 < tmp[y].x0 || (tmp[x].x0 == tmp[y].x0 && x < y); };
		for (int start = k / 2 - 1; start >= 0; --start) {
			int root = start;
			for (;;) {
				int c = 2 * root + 1;
				if (c >= k) break;
				if (c + 1 < k && less(idx[c], idx[c + 1])) ++c;
				if (!less(idx[root], idx[c])) break;
				int x = idx[root]; idx[root] = idx[c], idx[c] = x;
				root = c;
			}
		}
		for (int end = k - 1; end > 0; --end) {
			int x = idx[0]; idx[0] = idx[end], idx[end] = x;
			int root = 0;
			for (;;) {
				int c = 2 * root + 1;
				if (c >= end) break;
				if (c + 1 < end && less(idx[c], idx[c + 1])) ++c;
				if (!less(idx[root], idx[c])) break;
				int y = idx[root]; idx[root] = idx[c], idx[c] = y;
				root = c;
			}
		}
		for (int i = 0; i < k; ++i) out[i] = tmp[idx[i]];
	}
	int as = 0;
	for (int i = 0; i < k; ++i) { out[i].as = as; as += out[i].cnt; }
	B.n_chain[r] = k;
}

void launch_chain_dp_serial(const Batch &B, const uint32_t *read_list, uint32_t n_list, hipStream_t st)
{
	if (n_list == 0) return;
	hipLaunchKernelGGL(mnc_chain_dp_serial, dim3((n_list + 63) / 64), dim3(64), 0, st, B, read_list, n_list);
}

void launch_backtrack(const Batch &B, hipStream_t st)
{
	if (B.n_reads == 0) return;
	hipLaunchKernelGGL(mnc_chain_backtrack, dim3((B.n_reads + 63) / 64), dim3(64), 0, st, B);
}

} // namespace mnc
