// Stage kernel K6: chains -> regions -> hierarchy -> long-join -> [base-level alignment, k_align.hip]
// -> second hierarchy pass -> MAPQ -> monica's per-read decision and taxon counts -- gfx950.
//
// Replaces, for one index part: mm_gen_regs / mm_set_parent / mm_select_sub / mm_join_long /
// mm_set_mapq inside index.map(seq) and the Python that
// consumes the hits: the gate `hit.is_primary and hit.mapq >= mapping_quality`
// (monica/genomes/aligner.py:194,216), best_hit (aligner.py:328-339), the ambiguous /
// unmapped split (aligner.py:225-233, 264-265) and the count accumulation
// (aligner.py:247-263).
//
// One thread per read: a read has a handful of regions, so this stage is a few hundred
// scalar operations per read.  The float32 islands of minimap2 (overlap ratio, pri_ratio,
// MAPQ) are evaluated with IEEE single-precision operations in the reference's order
// (the library is built with -ffp-contract=off and correctly rounded division); logf()
// comes from a table the host fills with its own libm, so values are bit-identical to a
// host evaluation.
#include "device.h"

namespace mnc {

__device__ __forceinline__ uint64_t mix64(uint64_t key)
{
	key = ~key + (key << 21);
	key = key ^ key >> 24;
	key = (key + (key << 3)) + (key << 8);
	key = key ^ key >> 14;
	key = (key + (key << 2)) + (key << 4);
	key = key ^ key >> 28;
	key = key + (key << 31);
	return key;
}

__device__ __forceinline__ uint32_t wang32(uint32_t key)
{
	key += ~(key << 15);
	key ^=  (key >> 10);
	key +=  (key << 3);
	key ^=  (key >> 6);
	key += ~(key << 11);
	key ^=  (key >> 16);
	return key;
}

// element i of a per-thread array laid out [i][thread] in LDS (stride 64 elements: lanes of a
// wave touch consecutive addresses)
template <class T>
struct Strided {
	T *p;
	__device__ __forceinline__ T &operator[](int i) const { return p[(size_t)i * 64]; }
};

template <class K64P>
__device__ void sort_u64(K64P a, int n)       // ascending; insertion for the common tiny case
{
	if (n <= 24) {
		for (int i = 1; i < n; ++i) {
			uint64_t x = a[i];
			int j = i - 1;
			while (j >= 0 && a[j] > x) { a[j + 1] = a[j]; --j; }
			a[j + 1] = x;
		}
		return;
	}
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

// coordinates of a region from its first / last anchor (mm_reg_set_coor)
__device__ __forceinline__ void set_coor(mnc_reg_t &r, const RegX &e, int qlen)
{
	const int32_t q_span = (int32_t)(e.y0 >> 32 & 0xff);
	r.rev = (int32_t)(e.x0 >> 63);
	r.rid = (int32_t)(e.x0 << 1 >> 33);
	r.rs = (int32_t)e.x0 + 1 > q_span ? (int32_t)e.x0 + 1 - q_span : 0;
	r.re = (int32_t)e.x1 + 1;
	if (!r.rev) {
		r.qs = (int32_t)e.y0 + 1 - q_span;
		r.qe = (int32_t)e.y1 + 1;
	} else {
		r.qs = qlen - ((int32_t)e.y1 + 1);
		r.qe = qlen - ((int32_t)e.y0 + 1 - q_span);
	}
}

// re-number ids after a compaction and re-point parents (mm_sync_regs)
template <class RegP, class I32P>
__device__ void sync_regs(int n_regs, RegP regs, I32P tmp)
{
	if (n_regs <= 0) return;
	int max_id = -1;
	for (int i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id ? max_id : regs[i].id;
	for (int i = 0; i <= max_id; ++i) tmp[i] = -1;
	for (int i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
	for (int i = 0; i < n_regs; ++i) {
		const int pa = regs[i].parent;
		regs[i].id = i;
		if (pa == -2) regs[i].parent = i;
		else if (pa >= 0 && tmp[pa] >= 0) regs[i].parent = tmp[pa];
		else regs[i].parent = -1;
	}
}

// parent / secondary assignment, subsc, n_sub (mm_set_parent).  It runs twice when base-level
// alignment is on (chain_post, then align_regs): on the second pass both regions carry a DP
// result, dp_max2 of the parent is raised and a near-equal DP score counts as sub-optimal too.
template <class RegP, class K64P, class I32P>
__device__ __forceinline__ void set_parent(const Batch &B, int n_regs, RegP r, K64P cov, I32P w)
{
	if (n_regs <= 0) return;
	const int sub_diff = B.sc_a * 2 + B.sc_b;
	for (int i = 0; i < n_regs; ++i) r[i].id = i;
	int k = 1;
	w[0] = 0, r[0].parent = 0;
	for (int i = 1; i < n_regs; ++i) {
		const int si = r[i].qs, ei = r[i].qe;
		int n_cov = 0, uncov_len = 0, j;
		for (j = 0; j < k; ++j) {
			const int pj = w[j];
			int sj = r[pj].qs, ej = r[pj].qe;
			if (ej <= si || sj >= ei) continue;
			if (sj < si) sj = si;
			if (ej > ei) ej = ei;
			cov[n_cov++] = (uint64_t)(uint32_t)sj << 32 | (uint32_t)ej;
		}
		j = k;
		if (n_cov > 0) {
			int x = si;
			sort_u64(cov, n_cov);
			for (int jj = 0; jj < n_cov; ++jj) {
				if ((int)(cov[jj] >> 32) > x) uncov_len += (int)(cov[jj] >> 32) - x;
				x = (int32_t)cov[jj] > x ? (int32_t)cov[jj] : x;
			}
			if (ei > x) uncov_len += ei - x;
			for (j = 0; j < k; ++j) {
				const int pj = w[j];
				const int sj = r[pj].qs, ej = r[pj].qe;
				if (ej <= si || sj >= ei) continue;
				const int mn = ej - sj < ei - si ? ej - sj : ei - si;
				const int mx = ej - sj > ei - si ? ej - sj : ei - si;
				const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj)
				                       : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
				const float lhs = __fsub_rn(__fdiv_rn((float)ol, (float)mn), __fdiv_rn((float)uncov_len, (float)mx));
				if (lhs > B.mask_level) {
					int cnt_sub = 0;
					r[i].parent = r[pj].parent;
					r[pj].subsc = r[pj].subsc > r[i].score ? r[pj].subsc : r[i].score;
					if (r[i].cnt >= r[pj].cnt) cnt_sub = 1;
					if ((r[pj].flags & REG_HAS_DP) && (r[i].flags & REG_HAS_DP) &&
					    (r[pj].rid != r[i].rid || r[pj].rs != r[i].rs || r[pj].re != r[i].re || ol != mn)) {   // not the same hit twice
						const int sc = r[i].dp_max;
						r[pj].dp_max2 = r[pj].dp_max2 > sc ? r[pj].dp_max2 : sc;
						if (r[pj].dp_max - r[i].dp_max <= sub_diff) cnt_sub = 1;
					}
					if (cnt_sub) ++r[pj].n_sub;
					break;
				}
			}
		}
		if (j == k) w[k++] = i, r[i].parent = i, r[i].n_sub = 0;
	}
}

// keep primaries and the best secondaries (mm_select_sub); the in-place compaction reads r[p]
// after earlier slots may have been overwritten.  `ex` travels with `r`.
template <class RegP, class ExP, class I32P>
__device__ __forceinline__ int select_sub(const Batch &B, int n_regs, RegP r, ExP ex, I32P tmp)
{
	if (!(B.pri_ratio > 0.0f) || n_regs <= 0) return n_regs;
	const int min_diff = KMER * 2;
	int k = 0, n_2nd = 0;
	for (int i = 0; i < n_regs; ++i) {
		const int p = r[i].parent;
		if (p == i || (r[i].flags & REG_INV)) {                // a primary, or an inversion
			r[k] = r[i], ex[k] = ex[i], ++k;
		} else if (((float)r[i].score >= __fmul_rn((float)r[p].score, B.pri_ratio) || r[i].score + min_diff >= r[p].score) && n_2nd < B.best_n) {
			if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) {
				r[k] = r[i], ex[k] = ex[i], ++k, ++n_2nd;
			}
		}
	}
	if (k != n_regs) sync_regs(k, r, tmp);
	return k;
}

// mm_set_mapq: the branch a region takes depends on whether it carries a DP result
template <class RegP>
__device__ __forceinline__ void set_mapq(const Batch &B, uint32_t rd, int n_regs, RegP r)
{
	long long sum_sc = 0;
	for (int i = 0; i < n_regs; ++i) if (r[i].parent == r[i].id) sum_sc += r[i].score;
	const float uniq_ratio = __fdiv_rn((float)sum_sc, (float)(sum_sc + (long long)B.rep_len[rd]));
	for (int i = 0; i < n_regs; ++i) {
		const mnc_reg_t x = r[i];
		int mapq = 0;
		if (!(x.flags & REG_INV) && x.parent == x.id) {        // an inversion: 0
			const bool has_dp = (x.flags & REG_HAS_DP) != 0;
			const float pen_s1 = __fmul_rn(x.score > 100 ? 1.0f : __fmul_rn(0.01f, (float)x.score), uniq_ratio);
			float pen_cm = x.cnt > 10 ? 1.0f : __fmul_rn(0.1f, (float)x.cnt);
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			const int subsc = x.subsc > B.min_sc ? x.subsc : B.min_sc;
			const int ls = x.n_sub + 1 < B.logf_n ? x.n_sub + 1 : B.logf_n - 1;
			if (has_dp && x.dp_max2 > 0 && x.dp_max > 0) {
				// identity * pen_cm * 40 * (1 - x*x) * logf(dp_max / a); dp_max / a is a float division:
				// its logf is not a table entry, so logf((float)dp_max / a) = logf(dp_max) - logf(a)
				// would round differently -- the table is indexed by dp_max and holds logf((float)i / a)
				const float identity = __fdiv_rn((float)x.mlen, (float)x.blen);
				const float xr = __fdiv_rn(__fdiv_rn(__fmul_rn((float)x.dp_max2, (float)subsc), (float)x.dp_max), (float)x.score0);
				const int li = x.dp_max < B.logf_n ? x.dp_max : B.logf_n - 1;
				float q = __fmul_rn(__fmul_rn(__fmul_rn(__fmul_rn(identity, pen_cm), 40.0f), __fsub_rn(1.0f, __fmul_rn(xr, xr))), B.logf_a_lut[li]);
				mapq = (int)q;
				const float alt = __fadd_rn(__fdiv_rn(__fmul_rn(__fmul_rn(__fmul_rn(6.02f, identity), identity), (float)(x.dp_max - x.dp_max2)), (float)B.sc_a), .499f);
				const int mapq_alt = (int)alt;
				mapq = mapq < mapq_alt ? mapq : mapq_alt;
			} else {
				const float xr = __fdiv_rn((float)subsc, (float)x.score0);
				if (has_dp) {
					const float identity = __fdiv_rn((float)x.mlen, (float)x.blen);
					const int li = x.dp_max < B.logf_n ? (x.dp_max > 0 ? x.dp_max : 0) : B.logf_n - 1;
					float q = __fmul_rn(__fmul_rn(__fmul_rn(__fmul_rn(identity, pen_cm), 40.0f), __fsub_rn(1.0f, xr)), B.logf_a_lut[li]);
					mapq = (int)q;
				} else {
					const int li = x.score < B.logf_n ? x.score : B.logf_n - 1;
					float q = __fmul_rn(__fmul_rn(__fmul_rn(pen_cm, 40.0f), __fsub_rn(1.0f, xr)), B.logf_lut[li]);
					mapq = (int)q;
				}
			}
			mapq -= (int)__fadd_rn(__fmul_rn(4.343f, B.logf_lut[ls]), .499f);
			mapq = mapq > 0 ? mapq : 0;
			mapq = mapq < 60 ? mapq : 60;
			if (has_dp && x.dp_max > x.dp_max2 && mapq == 0) mapq = 1;
		}
		r[i].mapq = mapq;
	}
}

// monica: gate, best_hit, decision (aligner.py:216-233)
template <class RegP>
__device__ __forceinline__ void gate_and_decide(const Batch &B, int n_regs, RegP r, mnc_hit_t *gated,
                                                int32_t &assign, mnc_hit_t &best, int32_t &nhits)
{
	int ties = 0;
	for (int i = 0; i < n_regs; ++i) {
		const mnc_reg_t x = r[i];
		if (x.id == x.parent && x.mapq >= B.min_mapq) {
			mnc_hit_t h;
			h.rid = x.rid, h.mapq = x.mapq, h.nm = x.blen - x.mlen + x.n_ambi, h.mlen = x.mlen;
			gated[nhits] = h;
			if (nhits == 0) best = h, ties = 1;
			else {
				const long long l = (long long)h.nm * best.mlen, rr = (long long)best.nm * h.mlen;
				if (l < rr) best = h, ties = 1;
				else if (l == rr) best = h, ++ties;
			}
			++nhits;
		}
	}
	if (nhits > 0) assign = (nhits == 1 || ties == 1) ? best.rid : MNC_AMBIGUOUS;   // best = the minimal hit, also when it is tied
}

// The work of one read on a workspace of n slots per array: `r`, `ex`, `ka`, `kb`, `w`, `tmp`
// live in LDS for reads with few chains and in HBM scratch otherwise (the function is inlined
// once per address space).  Returns the number of regions kept; fills assign / best / nhits.
// With `chain_dst` (base-level alignment follows) it stops after long-join and tells every chain
// where its anchors go in the squeezed anchor array of the read: position | LONG_JOIN flag (bit
// 30) for a chain fused behind another one, -1 for a chain that is not kept.
template <class RegP, class ExP, class K64P, class I32P>
__device__ __forceinline__ int regions_of_read(const Batch &B, uint32_t rd, int qlen, int n, const ChainRec *ch,
                                               RegP r, ExP ex, K64P ka, K64P kb, I32P w, I32P tmp, mnc_hit_t *gated,
                                               int32_t &assign, mnc_hit_t &best, int32_t &nhits, int32_t *chain_dst)
{
	int n_regs = 0;
	// ---------------- chains ordered by (first anchor x, rank); `as` = running anchor count
	// in that order (mm_chain_dp's final ordering; total order instead of an unstable sort)
	for (int i = 0; i < n; ++i) w[i] = i, ka[i] = ch[i].x0, tmp[i] = ch[i].cnt;
	for (int i = 1; i < n; ++i) {
		const int c = w[i];
		const uint64_t xc = ka[c];
		int j = i - 1;
		while (j >= 0 && (ka[w[j]] > xc || (ka[w[j]] == xc && w[j] > c))) { w[j + 1] = w[j]; --j; }
		w[j + 1] = c;
	}
	{
		int as = 0;
		for (int i = 0; i < n; ++i) { const int c = w[i]; kb[c] = (uint64_t)(uint32_t)as << 32 | (uint32_t)c; as += tmp[c]; }   // `as` is unique: total order
	}
	// ---------------- regions, sorted by score (desc) with the pseudo-random tie-break
	uint32_t hash = wang32((uint32_t)qlen) + wang32((uint32_t)B.seed);
	hash = wang32(hash);
	for (int i = 0; i < n; ++i) {
		const uint32_t h = (uint32_t)mix64((mix64(ka[i]) + mix64(ch[i].y0)) ^ (uint64_t)hash);
		ka[i] = ((uint64_t)(uint32_t)ch[i].score << 32 | (uint32_t)tmp[i]) ^ (uint64_t)h;
	}
	// sort chain indices by (ka, as) descending: insertion (n is small)
	for (int i = 0; i < n; ++i) w[i] = i;
	for (int i = 1; i < n; ++i) {
		const int c = w[i];
		const uint64_t kc = ka[c], bc = kb[c];
		int j = i - 1;
		while (j >= 0 && (ka[w[j]] < kc || (ka[w[j]] == kc && kb[w[j]] < bc))) { w[j + 1] = w[j]; --j; }
		w[j + 1] = c;
	}
	for (int i = 0; i < n; ++i) {
		const int ci = w[i];
		const ChainRec c = ch[ci];
		mnc_reg_t x;
		x.id = i, x.parent = -1;
		x.score = x.score0 = (int32_t)(ka[ci] >> 32);
		x.hash = (uint32_t)ka[ci];
		x.cnt = c.cnt, x.as = (int32_t)(kb[ci] >> 32), x.mlen = c.mlen, x.blen = c.blen;
		x.subsc = 0, x.n_sub = 0, x.mapq = 0;
		x.dp_score = ci, x.dp_max = x.dp_max2 = x.n_ambi = x.n_cigar = x.flags = 0;     // dp_score: the chain, until the end
		if (chain_dst) chain_dst[ci] = -1;
		RegX e;
		e.x0 = c.x0, e.y0 = c.y0, e.x1 = c.x1, e.y1 = c.y1;
		set_coor(x, e, qlen);
		r[i] = x, ex[i] = e;
	}
	n_regs = n;

	set_parent(B, n_regs, r, ka, w);

	n_regs = select_sub(B, n_regs, r, ex, tmp);

	// ---------------- long-join of adjacent co-linear primaries (mm_join_long)
	if (chain_dst && n_regs == 1) r[0].as = 0, chain_dst[r[0].dp_score] = 0;
	if (n_regs >= 2) {
		K64P aux = ka;
		// squeeze: `as` becomes the running anchor count in original-`as` order
		for (int i = 0; i < n_regs; ++i) aux[i] = (uint64_t)(uint32_t)r[i].as << 32 | (uint32_t)i;
		sort_u64(aux, n_regs);
		int as = 0;
		for (int i = 0; i < n_regs; ++i) {
			const int ri = (int32_t)(uint32_t)aux[i];
			r[ri].as = as;
			if (chain_dst) chain_dst[r[ri].dp_score] = as;
			as += r[ri].cnt;
		}
		int n_aux = 0, n_drop = 0;
		for (int i = 0; i < n_regs; ++i)
			if (r[i].parent == i || r[i].parent < 0) aux[n_aux++] = (uint64_t)(uint32_t)r[i].as << 32 | (uint32_t)i;
		sort_u64(aux, n_aux);
		for (int i = n_aux - 1; i >= 1; --i) {
			const int i0 = (int32_t)(uint32_t)aux[i - 1], i1 = (int32_t)(uint32_t)aux[i];
			mnc_reg_t r0 = r[i0];
			const mnc_reg_t r1 = r[i1];
			RegX e0 = ex[i0];
			const RegX e1 = ex[i1];
			if (r0.as + r0.cnt != r1.as) continue;
			if (r0.rid != r1.rid || r0.rev != r1.rev) continue;
			if (e1.x0 <= e0.x1 || (int32_t)e1.y0 <= (int32_t)e0.y1) continue;
			const int64_t dx = (int64_t)(e1.x0 - e0.x1);
			int max_gap = (int32_t)e1.y0 - (int32_t)e0.y1, min_gap = max_gap;
			max_gap = max_gap > dx ? max_gap : (int)dx;
			min_gap = min_gap < dx ? min_gap : (int)dx;
			if (max_gap > B.max_join_long || min_gap > B.max_join_short) continue;
			const float per = __fdiv_rn((float)B.min_join_flank_sc, (float)B.max_join_long);
			const int sc_thres = (int)((double)__fmul_rn(per, (float)max_gap) + .499);
			if (r0.score < sc_thres || r1.score < sc_thres) continue;
			const int min_flank_len = (int)__fmul_rn((float)max_gap, B.min_join_flank_ratio);
			if (r0.re - r0.rs < min_flank_len || r0.qe - r0.qs < min_flank_len) continue;
			if (r1.re - r1.rs < min_flank_len || r1.qe - r1.qs < min_flank_len) continue;
			// join: r0 absorbs r1
			{
				const int sp = (int)(e1.y0 >> 32 & 0xff);
				const int tl = (int32_t)e1.x0 - (int32_t)e0.x1;
				const int ql = (int32_t)e1.y0 - (int32_t)e0.y1;
				r0.blen += (tl > ql ? tl : ql) + (r1.blen - sp);
				r0.mlen += (tl > sp && ql > sp ? sp : tl < ql ? tl : ql) + (r1.mlen - sp);
			}
			r0.cnt += r1.cnt, r0.score += r1.score;
			e0.x1 = e1.x1, e0.y1 = e1.y1;
			{ const int32_t m = r0.mlen, b = r0.blen; set_coor(r0, e0, qlen); r0.mlen = m, r0.blen = b; }
			r[i0] = r0, ex[i0] = e0;
			r[i1].cnt = 0;
			r[i1].parent = r0.id;
			if (chain_dst) chain_dst[r1.dp_score] |= 1 << 30;        // MM_SEED_LONG_JOIN on its first anchor
			++n_drop;
		}
		if (n_drop > 0) {
			for (int i = 0; i < n_regs; ++i) {
				const int pa = r[i].parent;
				if (pa >= 0 && r[i].id != pa)
					if (r[pa].parent >= 0 && r[pa].parent != pa) r[i].parent = r[pa].parent;
			}
			int k = 0;
			for (int i = 0; i < n_regs; ++i) {          // mm_filter_regs: cnt < min_cnt
				if (r[i].cnt < B.min_cnt) continue;
				if (k < i) r[k] = r[i], ex[k] = ex[i];
				++k;
			}
			n_regs = k;
			sync_regs(n_regs, r, tmp);
		}
	}

	for (int i = 0; i < n_regs; ++i) r[i].dp_score = 0;
	if (chain_dst) return n_regs;                               // base-level alignment comes next (k_align.hip)
	set_mapq(B, rd, n_regs, r);
	gate_and_decide(B, n_regs, r, gated, assign, best, nhits);
	return n_regs;
}

constexpr int RG_LDS_CHAINS = 4;               // reads with at most this many chains work in LDS

__global__ __launch_bounds__(64) void mnc_regions_decide(Batch B, RegX *regx_all, uint64_t *k64a_all,
                                                         uint64_t *k64b_all, mnc_hit_t *gated_all)
{
	__shared__ mnc_reg_t s_r[RG_LDS_CHAINS][64];
	__shared__ RegX s_ex[RG_LDS_CHAINS][64];
	__shared__ uint64_t s_ka[RG_LDS_CHAINS][64], s_kb[RG_LDS_CHAINS][64];
	__shared__ int32_t s_w[RG_LDS_CHAINS][64], s_tmp[RG_LDS_CHAINS][64];
	const uint32_t rd = blockIdx.x * blockDim.x + threadIdx.x;
	if (rd >= B.n_reads) return;
	const bool dp = B.contract == MNC_CONTRACT_DP;
	const int qlen = (int)(B.offsets[rd + 1] - B.offsets[rd]);
	int32_t assign = MNC_UNMAPPED, nhits = 0;
	mnc_hit_t best;
	best.rid = best.mapq = best.nm = best.mlen = 0;
	const int n = B.n_chain[rd];
	int n_regs = 0;
	if (n > 0) {
		const int64_t cslot = B.an_off[rd] / 3, slot = reg_slot(B, rd);      // chain slots, region slots
		const ChainRec *ch = B.chains_tmp + cslot;              // backtrack order (pad = rank)
		mnc_hit_t *gated = gated_all + slot;
		int32_t *chain_dst = dp ? B.chain_dst + cslot : nullptr;
		if (n <= RG_LDS_CHAINS) {
			const int t = threadIdx.x;
			n_regs = regions_of_read(B, rd, qlen, n, ch, Strided<mnc_reg_t>{&s_r[0][t]}, Strided<RegX>{&s_ex[0][t]},
			                         Strided<uint64_t>{&s_ka[0][t]}, Strided<uint64_t>{&s_kb[0][t]},
			                         Strided<int32_t>{&s_w[0][t]}, Strided<int32_t>{&s_tmp[0][t]}, gated, assign, best, nhits, chain_dst);
			mnc_reg_t *out = B.regs + slot;
			for (int i = 0; i < n_regs; ++i) out[i] = s_r[i][t];
		} else {
			int32_t *w = B.tmp_i32 + slot * 4;                  // n ints each (4 per slot available)
			n_regs = regions_of_read(B, rd, qlen, n, ch, B.regs + slot, regx_all + slot, k64a_all + slot, k64b_all + slot,
			                         w, w + n, gated, assign, best, nhits, chain_dst);
		}
		if (dp) {
			// every kept region goes to the base-level alignment stage, in the order mm_align_skeleton
			// walks them (a Z-drop split inserts its tail right behind its head: order = index << 8 | depth)
			const mnc_reg_t *rg = B.regs + slot;
			int total = 0;
			for (int i = 0; i < n_regs; ++i) total += rg[i].cnt;
			B.ca_cnt[rd] = total;
			if (n_regs > 0) {
				const unsigned long long w0 = atomicAdd(&B.dp_ctr[5], (unsigned long long)n_regs);
				for (int i = 0; i < n_regs; ++i) {
					RegDP d;
					memset(&d, 0, sizeof(d));
					d.read = (int32_t)rd, d.order = i << 8, d.state = 1;
					B.regdp[slot + i] = d;
					B.next_list[w0 + i] = (int32_t)(slot + i);
				}
			}
		}
	} else if (dp) B.ca_cnt[rd] = 0;
	if (dp) { B.reg_cnt[rd] = n_regs; return; }
	B.n_reg[rd] = n_regs;
	B.assign[rd] = assign;
	if (B.best) B.best[rd] = best;
	B.nhits[rd] = nhits;
	B.best_mlen[rd] = best.mlen;
}

// After the base-level alignment stage (k_align.hip): mm_filter_regs, mm_hit_sort, the second
// mm_set_parent / mm_select_sub pass (align_regs), mm_set_mapq with the DP branch, then monica's
// gate and decision.  One thread per read; the regions sit in the read's slots of B.regs in
// arrival order (split tails appended), RegDP.order gives the skeleton's order.
__global__ __launch_bounds__(64) void mnc_regions_post(Batch B, mnc_reg_t *work_all, RegX *regx_all, uint64_t *k64a_all,
                                                       int32_t *tmp_all, mnc_hit_t *gated_all)
{
	const uint32_t rd = blockIdx.x * blockDim.x + threadIdx.x;
	if (rd >= B.n_reads) return;
	int32_t assign = MNC_UNMAPPED, nhits = 0;
	mnc_hit_t best;
	best.rid = best.mapq = best.nm = best.mlen = 0;
	const int n0 = B.reg_cnt[rd];
	int n_regs = 0;
	if (n0 > 0) {
		const int64_t slot = reg_slot(B, rd);
		mnc_reg_t *src = B.regs + slot, *r = work_all + slot;
		RegX *ex = regx_all + slot;                          // x0: CIGAR offset, x1: n_cigar -- travels with the region
		uint64_t *ka = k64a_all + slot;
		int32_t *w = tmp_all + slot * 4, *tmp = w + n0;
		mnc_hit_t *gated = gated_all + slot;
		// skeleton order
		for (int i = 0; i < n0; ++i) w[i] = i;
		for (int i = 1; i < n0; ++i) {
			const int c = w[i], oc = B.regdp[slot + c].order;
			int j = i - 1;
			while (j >= 0 && B.regdp[slot + w[j]].order > oc) { w[j + 1] = w[j]; --j; }
			w[j + 1] = c;
		}
		// mm_filter_regs
		int k = 0;
		for (int i = 0; i < n0; ++i) {
			const mnc_reg_t x = src[w[i]];
			bool flt = !(x.flags & REG_INV) && x.cnt < B.min_cnt;
			if ((x.flags & REG_INV) && !(x.flags & REG_HAS_DP)) flt = true;   // its extension found nothing: mm_align1_inv returns no region then
			if (x.flags & REG_HAS_DP) {
				if (x.mlen < B.min_sc) flt = true;
				else if (x.dp_max < B.min_dp_max) flt = true;
				// max_clip_ratio = 1.0: (qs > qlen && qlen - qe > qlen) never holds
			}
			if (flt) continue;
			RegX e;
			e.x0 = (uint64_t)B.regdp[slot + w[i]].cig_off, e.y0 = 0, e.x1 = (uint64_t)(uint32_t)x.n_cigar, e.y1 = 0;
			r[k] = x, ex[k] = e, ++k;
		}
		n_regs = k;
		// mm_hit_sort: by DP score, hash as the tie-break (then position: a total order); descending
		if (n_regs > 1) {
			for (int i = 0; i < n_regs; ++i) {
				const int sc = (r[i].flags & REG_HAS_DP) ? r[i].dp_max : r[i].score;
				ka[i] = (uint64_t)(uint32_t)sc << 32 | r[i].hash;
				tmp[i] = i;
			}
			for (int i = 1; i < n_regs; ++i) {
				const int c = tmp[i];
				const uint64_t kc = ka[c];
				int j = i - 1;
				while (j >= 0 && (ka[tmp[j]] < kc || (ka[tmp[j]] == kc && tmp[j] < c))) { tmp[j + 1] = tmp[j]; --j; }
				tmp[j + 1] = c;
			}
			// permute through the source slots (free now)
			for (int i = 0; i < n_regs; ++i) src[i] = r[tmp[i]], B.regdp[slot + i].cig_off = (int64_t)ex[tmp[i]].x0;
			for (int i = 0; i < n_regs; ++i) { r[i] = src[i]; RegX e; e.x0 = (uint64_t)B.regdp[slot + i].cig_off, e.y0 = 0, e.x1 = (uint64_t)(uint32_t)src[i].n_cigar, e.y1 = 0; ex[i] = e; }
		}
		set_parent(B, n_regs, r, ka, w);
		n_regs = select_sub(B, n_regs, r, ex, tmp);
		set_mapq(B, rd, n_regs, r);
		gate_and_decide(B, n_regs, r, gated, assign, best, nhits);
		for (int i = 0; i < n_regs; ++i) src[i] = r[i], B.regdp[slot + i].cig_off = (int64_t)ex[i].x0, B.regdp[slot + i].n_cigar = r[i].n_cigar;
	}
	if (B.skip[rd]) {                                         // a kernel call of this read outgrew every workspace class: no decision
		assign = MNC_SKIPPED, nhits = 0;
		memset(&best, 0, sizeof(best));
	}
	B.n_reg[rd] = n_regs;
	B.assign[rd] = assign;
	if (B.best) B.best[rd] = best;
	B.nhits[rd] = nhits;
	B.best_mlen[rd] = best.mlen;
}

// ---------------------------------------------------------------- taxon counts
// aligner.py:247-263, all three modes at once: per genome {reads, bases, matching bases}.
// With a handful of genomes every read of the batch would hit the same few HBM words, so a
// block first adds its reads up in LDS and then flushes the bins it touched.
constexpr int CT_THREADS = 256, CT_READS = 4096, CT_LDS_BINS = 3 * 1024;

__global__ __launch_bounds__(CT_THREADS) void mnc_count_taxa(Batch B)
{
	__shared__ unsigned long long s_bins[CT_LDS_BINS];
	const int n_bins = B.n_genomes * 3;
	const bool in_lds = n_bins <= CT_LDS_BINS;
	if (in_lds) {
		for (int k = threadIdx.x; k < n_bins; k += CT_THREADS) s_bins[k] = 0;
		__syncthreads();
	}
	unsigned long long *bins = in_lds ? s_bins : (unsigned long long*)B.counts;
	const uint32_t lo = blockIdx.x * CT_READS;
	const uint32_t hi = min(B.n_reads, lo + CT_READS);
	for (uint32_t rd = lo + threadIdx.x; rd < hi; rd += CT_THREADS) {
		const int a = B.assign[rd];
		if (a < 0) continue;
		const int g = B.contig_genome[a];
		atomicAdd(&bins[g * 3 + 0], 1ULL);
		atomicAdd(&bins[g * 3 + 1], (unsigned long long)(B.offsets[rd + 1] - B.offsets[rd]));
		atomicAdd(&bins[g * 3 + 2], (unsigned long long)B.best_mlen[rd]);
	}
	if (in_lds) {
		__syncthreads();
		for (int k = threadIdx.x; k < n_bins; k += CT_THREADS)
			if (s_bins[k]) atomicAdd((unsigned long long*)&B.counts[k], s_bins[k]);
	}
}

// ---------------------------------------------------------------- gather gated hits -> CSR
__global__ __launch_bounds__(256) void mnc_gather_hits(Batch B, const mnc_hit_t *gated_all,
                                                       const int64_t *hit_off, mnc_hit_t *out)
{
	const uint32_t rd = blockIdx.x * blockDim.x + threadIdx.x;
	if (rd >= B.n_reads) return;
	const int n = B.nhits[rd];
	if (n == 0) return;
	const mnc_hit_t *src = gated_all + reg_slot(B, rd);
	mnc_hit_t *dst = out + hit_off[rd];
	for (int i = 0; i < n; ++i) dst[i] = src[i];
}

void launch_regions(const Batch &B, void *regx, uint64_t *k64a, uint64_t *k64b, mnc_hit_t *gated, hipStream_t st)
{
	if (B.n_reads == 0) return;
	hipLaunchKernelGGL(mnc_regions_decide, dim3((B.n_reads + 63) / 64), dim3(64), 0, st, B,
	                   reinterpret_cast<RegX*>(regx), k64a, k64b, gated);
	if (B.counts && B.contract != MNC_CONTRACT_DP)
		hipLaunchKernelGGL(mnc_count_taxa, dim3((B.n_reads + CT_READS - 1) / CT_READS), dim3(CT_THREADS), 0, st, B);
}

void launch_regions_post(const Batch &B, mnc_reg_t *work, void *regx, uint64_t *k64a, int32_t *tmp, mnc_hit_t *gated, hipStream_t st)
{
	if (B.n_reads == 0) return;
	hipLaunchKernelGGL(mnc_regions_post, dim3((B.n_reads + 63) / 64), dim3(64), 0, st, B, work,
	                   reinterpret_cast<RegX*>(regx), k64a, tmp, gated);
	if (B.counts) hipLaunchKernelGGL(mnc_count_taxa, dim3((B.n_reads + CT_READS - 1) / CT_READS), dim3(CT_THREADS), 0, st, B);
}

void launch_gather_hits(const Batch &B, const mnc_hit_t *gated, const int64_t *hit_off, mnc_hit_t *out, hipStream_t st)
{
	if (B.n_reads == 0) return;
	hipLaunchKernelGGL(mnc_gather_hits, dim3((B.n_reads + 255) / 256), dim3(256), 0, st, B, gated, hit_off, out);
}

} // namespace mnc
