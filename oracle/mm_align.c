/*
 * mm_align.c -- CPU ORACLE (test infrastructure, NOT product code).  See mm_oracle.h.
 *
 * The base-level alignment stage mappy 2.17 always runs inside index.map(seq)
 * (monica/genomes/aligner.py:193, 215; mappy ORs MM_F_CIGAR into the map options), restated
 * from the published minimap2 v2.17 algorithm:
 *
 *   per region      trim unreliable chain ends, drop seeds around long indels, pick the DP
 *                   window from neighbouring seeds, extend to the left, fill the gaps between
 *                   seeds at least min_ksw_len apart (two passes: approximate maximum first,
 *                   exact Z-drop only when a walk over the CIGAR shows a large drop), extend to
 *                   the right; a Z-drop inside a gap splits the region
 *   per region      left-align indels, merge I/D runs, strip a leading gap, then one walk over
 *                   the CIGAR gives mlen, blen, n_ambi and dp_max (mm_update_extra)
 *   per read        drop regions with mlen < min_chain_score or dp_max < min_dp_max, order by
 *                   dp_max, second parent / secondary pass (dp_max2, n_sub), select
 *
 * monica reads hit.mapq, hit.NM (= blen - mlen + n_ambi) and hit.mlen from the result
 * (aligner.py:194-195, 216-217) and counts mlen in 'matching' mode (aligner.py:259-263).
 *
 * The inversion alignment between the two halves of a region split by an inversion-like Z-drop
 * (mm_align1_inv) is restated too (align1_inv below): its hit has MAPQ 0 and cannot pass monica's
 * gate of 60 itself, but it takes part in the second hierarchy pass (it may become a secondary of
 * another primary and raise that one's dp_max2 / n_sub).
 *
 * PARITY UNPINNED (mm_oracle.h): none of this could be run against the real library here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mm_oracle.h"
#include "mm_internal.h"

typedef struct { uint32_t *c; int n, m; } cig_t;

static void append_cigar(cig_t *p, int n_cigar, const uint32_t *cigar)
{
	if (n_cigar == 0) return;
	if (p->n + n_cigar > p->m) {
		p->m = p->n + n_cigar;
		p->m += (p->m >> 1) + 8;
		p->c = (uint32_t*)realloc(p->c, (size_t)p->m * 4);
	}
	if (p->n > 0 && (p->c[p->n - 1] & 0xf) == (cigar[0] & 0xf)) {   /* same operation across the boundary */
		p->c[p->n - 1] += (cigar[0] >> 4) << 4;
		if (n_cigar > 1) memcpy(p->c + p->n, cigar + 1, (size_t)(n_cigar - 1) * 4);
		p->n += n_cigar - 1;
	} else {
		memcpy(p->c + p->n, cigar, (size_t)n_cigar * 4);
		p->n += n_cigar;
	}
}

static void seq_rev(int len, uint8_t *seq)
{
	int i;
	for (i = 0; i < len >> 1; ++i) { uint8_t t = seq[i]; seq[i] = seq[len - 1 - i], seq[len - 1 - i] = t; }
}

/* ------------------------------------------------------------------ CIGAR clean-up + region statistics */

static void fix_cigar(orc_reg_t *r, cig_t *p, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift)
{
	int32_t toff = 0, qoff = 0, to_shrink = 0;
	int k;
	*qshift = *tshift = 0;
	if (p->n <= 1) return;
	for (k = 0; k < p->n; ++k) {                              /* indel left alignment */
		uint32_t op = p->c[k] & 0xf, len = p->c[k] >> 4;
		if (len == 0) to_shrink = 1;
		if (op == 0) {
			toff += len, qoff += len;
		} else if (op == 1 || op == 2) {
			if (k > 0 && k < p->n - 1 && (p->c[k-1] & 0xf) == 0 && (p->c[k+1] & 0xf) == 0) {
				int l, prev_len = (int)(p->c[k-1] >> 4);
				if (op == 1) {
					for (l = 0; l < prev_len; ++l)
						if (qseq[qoff - 1 - l] != qseq[qoff + len - 1 - l]) break;
				} else {
					for (l = 0; l < prev_len; ++l)
						if (tseq[toff - 1 - l] != tseq[toff + len - 1 - l]) break;
				}
				if (l > 0) p->c[k-1] -= (uint32_t)l << 4, p->c[k+1] += (uint32_t)l << 4, qoff -= l, toff -= l;
				if (l == prev_len) to_shrink = 1;
			}
			if (op == 1) qoff += len;
			else toff += len;
		}
	}
	for (k = 0; k < p->n - 2; ++k) {                          /* runs like 5I6D7I become one I and one D */
		if ((p->c[k] & 0xf) > 0 && (p->c[k] & 0xf) + (p->c[k+1] & 0xf) == 3) {
			uint32_t s[3] = { 0, 0, 0 };
			int l;
			for (l = k; l < p->n; ++l) {
				uint32_t op = p->c[l] & 0xf;
				if (op == 1 || op == 2 || p->c[l] >> 4 == 0) s[op] += p->c[l] >> 4;
				else break;
			}
			if (s[1] > 0 && s[2] > 0 && l - k > 2) {
				p->c[k] = s[1] << 4 | 1;
				p->c[k+1] = s[2] << 4 | 2;
				for (k += 2; k < l; ++k) p->c[k] &= 0xf;
				to_shrink = 1;
			}
			k = l;
		}
	}
	if (to_shrink) {                                          /* squeeze out empty operations, merge equal neighbours */
		int l = 0;
		for (k = 0; k < p->n; ++k)
			if (p->c[k] >> 4 != 0) p->c[l++] = p->c[k];
		p->n = l;
		for (k = l = 0; k < p->n; ++k)
			if (k == p->n - 1 || (p->c[k] & 0xf) != (p->c[k+1] & 0xf)) p->c[l++] = p->c[k];
			else p->c[k+1] += p->c[k] >> 4 << 4;
		p->n = l;
	}
	if ((p->c[0] & 0xf) == 1 || (p->c[0] & 0xf) == 2) {       /* no leading I or D */
		int32_t l = (int32_t)(p->c[0] >> 4);
		if ((p->c[0] & 0xf) == 1) {
			if (r->rev) r->qe -= l;
			else r->qs += l;
			*qshift = l;
		} else r->rs += l, *tshift = l;
		--p->n;
		memmove(p->c, p->c + 1, (size_t)p->n * 4);
	}
}

static void update_extra(orc_reg_t *r, cig_t *p, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e)
{
	int k, l;
	int32_t s = 0, max = 0, qshift, tshift, toff = 0, qoff = 0;
	if (!(r->flags & ORC_REG_HAS_DP)) return;
	fix_cigar(r, p, qseq, tseq, &qshift, &tshift);
	qseq += qshift, tseq += tshift;
	r->blen = r->mlen = 0;
	for (k = 0; k < p->n; ++k) {
		uint32_t op = p->c[k] & 0xf, len = p->c[k] >> 4;
		if (op == 0) {
			int n_ambi = 0, n_diff = 0;
			for (l = 0; l < (int)len; ++l) {
				int cq = qseq[qoff + l], ct = tseq[toff + l];
				if (ct > 3 || cq > 3) ++n_ambi;
				else if (ct != cq) ++n_diff;
				s += mat[ct * 5 + cq];
				if (s < 0) s = 0;
				else max = max > s ? max : s;
			}
			r->blen += len - n_ambi, r->mlen += len - (n_ambi + n_diff), r->n_ambi += n_ambi;
			toff += len, qoff += len;
		} else if (op == 1) {
			int n_ambi = 0;
			for (l = 0; l < (int)len; ++l) if (qseq[qoff + l] > 3) ++n_ambi;
			r->blen += len - n_ambi, r->n_ambi += n_ambi;
			s -= q + e * len;
			if (s < 0) s = 0;
			qoff += len;
		} else if (op == 2) {
			int n_ambi = 0;
			for (l = 0; l < (int)len; ++l) if (tseq[toff + l] > 3) ++n_ambi;
			r->blen += len - n_ambi, r->n_ambi += n_ambi;
			s -= q + e * len;
			if (s < 0) s = 0;
			toff += len;
		}
	}
	r->dp_max = max;
	r->n_cigar = p->n;
}

/* ------------------------------------------------------------------ Z-drop test on a finished CIGAR */

static inline void update_max_zdrop(int32_t score, int i, int j, int32_t *max, int *max_i, int *max_j, int e, int *max_zdrop, int pos[2][2])
{
	if (score < *max) {
		int li = i - *max_i, lj = j - *max_j;
		int diff = li > lj ? li - lj : lj - li;
		int z = *max - score - diff * e;
		if (z > *max_zdrop) {
			*max_zdrop = z;
			pos[0][0] = *max_i, pos[0][1] = i + 1;
			pos[1][0] = *max_j, pos[1][1] = j + 1;
		}
	} else *max = score, *max_i = i, *max_j = j;
}

/* 0: fine; 1: the score drops by more than zdrop somewhere; 2: and the dropped stretch aligns
 * to its own reverse complement (an inversion) */
static int test_zdrop(const orc_opt_t *opt, const uint8_t *qseq, const uint8_t *tseq, int n_cigar, const uint32_t *cigar, const int8_t *mat)
{
	int k;
	int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
	int pos[2][2] = { { -1, -1 }, { -1, -1 } }, q_len, t_len;
	for (k = 0, score = 0; k < n_cigar; ++k) {
		uint32_t l, op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			for (l = 0; l < len; ++l) {
				score += mat[tseq[i + l] * 5 + qseq[j + l]];
				update_max_zdrop(score, i + l, j + l, &max, &max_i, &max_j, opt->e, &max_zdrop, pos);
			}
			i += len, j += len;
		} else if (op == 1 || op == 2) {
			score -= opt->q + opt->e * len;
			if (op == 1) j += len;
			else i += len;
			update_max_zdrop(score, i, j, &max, &max_i, &max_j, opt->e, &max_zdrop, pos);
		}
	}
	q_len = pos[1][1] - pos[1][0], t_len = pos[0][1] - pos[0][0];
	if (max_zdrop > opt->zdrop_inv && q_len < opt->max_gap && t_len < opt->max_gap) {
		uint8_t *qseq2 = (uint8_t*)malloc((size_t)(q_len > 0 ? q_len : 1));
		for (i = 0; i < q_len; ++i) {
			int c = qseq[pos[1][1] - i - 1];
			qseq2[i] = c >= 4 ? 4 : 3 - c;
		}
		score = orc_local_score(q_len, qseq2, t_len, tseq + pos[0][0], mat, opt->q, opt->e);
		free(qseq2);
		if (score >= opt->min_chain_score * opt->a && score >= opt->min_dp_max) return 2;
	}
	return max_zdrop > opt->zdrop ? 1 : 0;
}

/* ------------------------------------------------------------------ seed filters */

/* trim a chain end whose first (last) seeds sit on a different diagonal than what follows */
static void fix_bad_ends(const orc_reg_t *r, const orc128_t *a, int bw, int min_match, int32_t *as, int32_t *cnt)
{
	int32_t i, l, m;
	*as = r->as, *cnt = r->cnt;
	if (r->cnt < 3) return;
	m = l = a[r->as].y >> 32 & 0xff;
	for (i = r->as + 1; i < r->as + r->cnt - 1; ++i) {
		int32_t lq, lr, min, max;
		int32_t q_span = a[i].y >> 32 & 0xff;
		if (a[i].y & ORC_SEED_LONG_JOIN) break;
		lr = (int32_t)a[i].x - (int32_t)a[i-1].x;
		lq = (int32_t)a[i].y - (int32_t)a[i-1].y;
		min = lr < lq ? lr : lq;
		max = lr > lq ? lr : lq;
		if (max - min > l >> 1) *as = i;
		l += min;
		m += min < q_span ? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
	*cnt = r->as + r->cnt - *as;
	m = l = a[r->as + r->cnt - 1].y >> 32 & 0xff;
	for (i = r->as + r->cnt - 2; i > *as; --i) {
		int32_t lq, lr, min, max;
		int32_t q_span = a[i+1].y >> 32 & 0xff;
		if (a[i+1].y & ORC_SEED_LONG_JOIN) break;
		lr = (int32_t)a[i+1].x - (int32_t)a[i].x;
		lq = (int32_t)a[i+1].y - (int32_t)a[i].y;
		min = lr < lq ? lr : lq;
		max = lr > lq ? lr : lq;
		if (max - min > l >> 1) *cnt = i + 1 - *as;
		l += min;
		m += min < q_span ? min : q_span;
		if (l >= bw << 1 || (m >= min_match && m >= bw) || m >= r->mlen >> 1) break;
	}
}

static inline int seed_gap(const orc128_t *a, int i)       /* query step minus reference step between seed i-1 and i */
{
	return ((int32_t)a[i].y - (int32_t)a[i-1].y) - ((int32_t)a[i].x - (int32_t)a[i-1].x);
}

static int *collect_long_gaps(int as1, int cnt1, const orc128_t *a, int min_gap, int *n_)
{
	int i, n, *K;
	*n_ = 0;
	for (i = 1, n = 0; i < cnt1; ++i) {
		int gap = seed_gap(a + as1, i);
		if (gap < -min_gap || gap > min_gap) ++n;
	}
	if (n <= 1) return 0;
	K = (int*)malloc((size_t)n * sizeof(int));
	for (i = 1, n = 0; i < cnt1; ++i) {
		int gap = seed_gap(a + as1, i);
		if (gap < -min_gap || gap > min_gap) K[n++] = i;
	}
	*n_ = n;
	return K;
}

/* an insertion and a deletion close to each other that cancel: ignore the seeds between them */
static void filter_bad_seeds(int as1, int cnt1, orc128_t *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt)
{
	int max_st, max_en, n, i, k, max, *K;
	K = collect_long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	max = 0, max_st = max_en = -1;
	for (k = 0;; ++k) {
		int gap, l, n_ins = 0, n_del = 0, qs, rs, max_diff = 0, max_diff_l = -1;
		if (k == n || k >= max_en) {
			if (max_en > 0)
				for (i = K[max_st]; i < K[max_en]; ++i) a[as1 + i].y |= ORC_SEED_IGNORE;
			max = 0, max_st = max_en = -1;
			if (k == n) break;
		}
		i = K[k];
		gap = seed_gap(a + as1, i);
		if (gap > 0) n_ins += gap;
		else n_del += -gap;
		qs = (int32_t)a[as1 + i - 1].y;
		rs = (int32_t)a[as1 + i - 1].x;
		for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
			int j = K[l], diff;
			if ((int32_t)a[as1 + j].y - qs > max_ext_len || (int32_t)a[as1 + j].x - rs > max_ext_len) break;
			gap = seed_gap(a + as1, j);
			if (gap > 0) n_ins += gap;
			else n_del += -gap;
			diff = n_ins + n_del - abs(n_ins - n_del);
			if (max_diff < diff) max_diff = diff, max_diff_l = l;
		}
		if (max_diff > diff_thres && max_diff > max) max = max_diff, max_st = k, max_en = max_diff_l;
	}
	free(K);
}

/* a run of long gaps closer to each other than they are long: one long-join gap instead */
static void filter_bad_seeds_alt(int as1, int cnt1, orc128_t *a, int min_gap, int max_ext)
{
	int n, k, *K;
	K = collect_long_gaps(as1, cnt1, a, min_gap, &n);
	if (K == 0) return;
	for (k = 0; k < n;) {
		int i = K[k], l;
		int gap1 = seed_gap(a + as1, i);
		int re1 = (int32_t)a[as1 + i].x;
		int qe1 = (int32_t)a[as1 + i].y;
		gap1 = gap1 > 0 ? gap1 : -gap1;
		for (l = k + 1; l < n; ++l) {
			int j = K[l], gap2, q_span_pre, rs2, qs2, m;
			if ((int32_t)a[as1 + j].y - qe1 > max_ext || (int32_t)a[as1 + j].x - re1 > max_ext) break;
			gap2 = seed_gap(a + as1, j);
			q_span_pre = a[as1 + j - 1].y >> 32 & 0xff;
			rs2 = (int32_t)a[as1 + j - 1].x + q_span_pre;
			qs2 = (int32_t)a[as1 + j - 1].y + q_span_pre;
			m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
			gap2 = gap2 > 0 ? gap2 : -gap2;
			if (m > gap1 + gap2) break;
			re1 = (int32_t)a[as1 + j].x;
			qe1 = (int32_t)a[as1 + j].y;
			gap1 = gap2;
		}
		if (l > k + 1) {
			int j, end = K[l - 1];
			for (j = K[k]; j < end; ++j) a[as1 + j].y |= ORC_SEED_IGNORE;
			a[as1 + end].y |= ORC_SEED_LONG_JOIN;
		}
		k = l;
	}
	free(K);
}

/* test hook: what mm_align1 does to a region's seeds before any base is aligned -- trims the chain's ends
 * (mm_fix_bad_ends) and flags seeds around long indels (mm_filter_bad_seeds, mm_filter_bad_seeds_alt), with the
 * constants mm_align1 passes.  a[0 .. n) are the seeds of ONE region (x = reference end position, y = span << 32 |
 * query end position); the flags come back in a[].y (bit 40 LONG_JOIN, bit 41 IGNORE), the kept range in as1 / cnt1. */
void orc_test_seed_filters(const orc_opt_t *opt, int n, orc128_t *a, int mlen, int32_t *as1, int32_t *cnt1)
{
	orc_reg_t r;
	memset(&r, 0, sizeof(r));
	r.as = 0, r.cnt = n, r.mlen = mlen;
	fix_bad_ends(&r, a, opt->bw, opt->min_chain_score * 2, as1, cnt1);
	filter_bad_seeds(*as1, *cnt1, a, 10, 40, opt->max_gap >> 1, 10);
	filter_bad_seeds_alt(*as1, *cnt1, a, 30, opt->max_gap >> 1);
}

/* ------------------------------------------------------------------ one region */

static void split_reg(orc_reg_t *r, orc_reg_t *r2, int n, int qlen, const orc128_t *a)
{
	if (n <= 0 || n >= r->cnt) return;
	*r2 = *r;
	r2->id = -1;
	r2->flags = 0;                                           /* no CIGAR yet, split / split_inv cleared */
	r2->dp_score = r2->dp_max = r2->dp_max2 = r2->n_ambi = r2->n_cigar = 0;
	r2->cnt = r->cnt - n;
	r2->score = (int32_t)(r->score * ((float)r2->cnt / r->cnt) + .499);
	r2->as = r->as + n;
	if (r->parent == r->id) r2->parent = ORC_PARENT_TMP_PRI;
	orc_reg_set_coor(r2, qlen, a);
	r->cnt -= r2->cnt;
	r->score -= r2->score;
	orc_reg_set_coor(r, qlen, a);
	r->flags |= ORC_REG_SPLIT_L, r2->flags |= ORC_REG_SPLIT_R;
}

static void align_pair(const orc_opt_t *opt, int qlen, const uint8_t *qseq, int tlen, const uint8_t *tseq,
                       const int8_t *mat, int w, int end_bonus, int zdrop, int flag, orc_extz_t *ez)
{
	if (opt->max_sw_mat > 0 && (int64_t)tlen * qlen > opt->max_sw_mat) {
		orc_extz_reset(ez);
		ez->zdropped = 1;
	} else
		orc_ksw_extd2(qlen, qseq, tlen, tseq, 5, mat, (int8_t)opt->q, (int8_t)opt->e, (int8_t)opt->q2, (int8_t)opt->e2,
		              w, zdrop, end_bonus, flag, ez);
}

static void align1(const orc_opt_t *opt, const orc_index *mi, int qlen, uint8_t *const qseq0[2], orc_reg_t *r, orc_reg_t *r2,
                   cig_t *cig, int n_a, orc128_t *a, orc_extz_t *ez)
{
	const int k = orc_index_k(mi);
	int32_t rid = (int32_t)(a[r->as].x << 1 >> 33), rev = (int32_t)(a[r->as].x >> 63), as1, cnt1;
	uint8_t *tseq, *qseq;
	int32_t i, l, bw, dropped = 0, rs0, re0, qs0, qe0;
	int32_t rs, re, qs, qe;
	int32_t rs1, qs1, re1, qe1;
	int8_t mat[25];
	const int32_t ref_len = orc_index_len(mi, rid);

	r2->cnt = 0;
	if (r->cnt == 0) return;
	orc_gen_simple_mat(5, mat, (int8_t)opt->a, (int8_t)opt->b, (int8_t)opt->sc_ambi);
	bw = (int)(opt->bw * 1.5 + 1.);

	fix_bad_ends(r, a, opt->bw, opt->min_chain_score * 2, &as1, &cnt1);
	filter_bad_seeds(as1, cnt1, a, 10, 40, opt->max_gap >> 1, 10);
	filter_bad_seeds_alt(as1, cnt1, a, 30, opt->max_gap >> 1);
	/* the DP starts and ends in the middle of a seed */
	rs = (int32_t)a[as1].x - (k >> 1), qs = (int32_t)a[as1].y - (k >> 1);
	re = (int32_t)a[as1 + cnt1 - 1].x - (k >> 1), qe = (int32_t)a[as1 + cnt1 - 1].y - (k >> 1);

	/* how far the left extension may reach: up to nearby seeds of other chains */
	rs0 = (int32_t)a[r->as].x + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
	qs0 = (int32_t)a[r->as].y + 1 - (int32_t)(a[r->as].y >> 32 & 0xff);
	if (rs0 < 0) rs0 = 0;
	rs1 = qs1 = 0;
	for (i = r->as - 1, l = 0; i >= 0 && a[i].x >> 32 == a[r->as].x >> 32; --i) {
		int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		int32_t y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		if (x < rs0 && y < qs0) {
			if (++l > opt->min_cnt) {
				l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
				rs1 = rs0 - l, qs1 = qs0 - l;
				if (rs1 < 0) rs1 = 0;
				break;
			}
		}
	}
	if (qs > 0 && rs > 0) {
		l = qs < opt->max_gap ? qs : opt->max_gap;
		qs1 = qs1 > qs - l ? qs1 : qs - l;
		qs0 = qs0 < qs1 ? qs0 : qs1;
		l += l * opt->a > opt->q ? (l * opt->a - opt->q) / opt->e : 0;
		l = l < opt->max_gap ? l : opt->max_gap;
		l = l < rs ? l : rs;
		rs1 = rs1 > rs - l ? rs1 : rs - l;
		rs0 = rs0 < rs1 ? rs0 : rs1;
		rs0 = rs0 < rs ? rs0 : rs;
	} else rs0 = rs, qs0 = qs;
	/* and the right one */
	re0 = (int32_t)a[r->as + r->cnt - 1].x + 1;
	qe0 = (int32_t)a[r->as + r->cnt - 1].y + 1;
	re1 = ref_len, qe1 = qlen;
	for (i = r->as + r->cnt, l = 0; i < n_a && a[i].x >> 32 == a[r->as].x >> 32; ++i) {
		int32_t x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		int32_t y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		if (x > re0 && y > qe0) {
			if (++l > opt->min_cnt) {
				l = x - re0 > y - qe0 ? x - re0 : y - qe0;
				re1 = re0 + l, qe1 = qe0 + l;
				break;
			}
		}
	}
	if (qe < qlen && re < ref_len) {
		l = qlen - qe < opt->max_gap ? qlen - qe : opt->max_gap;
		qe1 = qe1 < qe + l ? qe1 : qe + l;
		qe0 = qe0 > qe1 ? qe0 : qe1;
		l += l * opt->a > opt->q ? (l * opt->a - opt->q) / opt->e : 0;
		l = l < opt->max_gap ? l : opt->max_gap;
		l = l < ref_len - re ? l : ref_len - re;
		re1 = re1 < re + l ? re1 : re + l;
		re0 = re0 > re1 ? re0 : re1;
	} else re0 = re, qe0 = qe;

	tseq = (uint8_t*)malloc((size_t)(re0 - rs0 > 0 ? re0 - rs0 : 1));

	if (qs > 0 && rs > 0) {                                   /* left extension: reversed sequences, gaps right-aligned */
		qseq = &qseq0[rev][qs0];
		orc_index_getseq(mi, (uint32_t)rid, (uint32_t)rs0, (uint32_t)rs, tseq);
		seq_rev(qs - qs0, qseq);
		seq_rev(rs - rs0, tseq);
		align_pair(opt, qs - qs0, qseq, rs - rs0, tseq, mat, bw, opt->end_bonus,
		           (r->flags & ORC_REG_SPLIT_INV) ? opt->zdrop_inv : opt->zdrop,
		           ORC_EZ_EXTZ_ONLY | ORC_EZ_RIGHT | ORC_EZ_REV_CIGAR, ez);
		if (ez->n_cigar > 0) {
			append_cigar(cig, ez->n_cigar, ez->cigar);
			r->flags |= ORC_REG_HAS_DP;
			r->dp_score += ez->max;
		}
		rs1 = rs - (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
		qs1 = qs - (ez->reach_end ? qs - qs0 : ez->max_q + 1);
		seq_rev(qs - qs0, qseq);
	} else rs1 = rs, qs1 = qs;
	re1 = rs, qe1 = qs;

	for (i = 1; i < cnt1; ++i) {                              /* gap filling */
		if ((a[as1 + i].y & (ORC_SEED_IGNORE | ORC_SEED_TANDEM)) && i != cnt1 - 1) continue;
		re = (int32_t)a[as1 + i].x - (k >> 1), qe = (int32_t)a[as1 + i].y - (k >> 1);
		re1 = re, qe1 = qe;
		if (i == cnt1 - 1 || (a[as1 + i].y & ORC_SEED_LONG_JOIN) || (qe - qs >= opt->min_ksw_len && re - rs >= opt->min_ksw_len)) {
			int j, bw1 = bw, zdrop_code;
			if (a[as1 + i].y & ORC_SEED_LONG_JOIN) bw1 = qe - qs > re - rs ? qe - qs : re - rs;
			qseq = &qseq0[rev][qs];
			orc_index_getseq(mi, (uint32_t)rid, (uint32_t)rs, (uint32_t)re, tseq);
			align_pair(opt, qe - qs, qseq, re - rs, tseq, mat, bw1, -1, opt->zdrop, ORC_EZ_APPROX_MAX, ez);   /* first pass */
			if ((zdrop_code = test_zdrop(opt, qseq, tseq, ez->n_cigar, ez->cigar, mat)) != 0)
				align_pair(opt, qe - qs, qseq, re - rs, tseq, mat, bw1, -1, zdrop_code == 2 ? opt->zdrop_inv : opt->zdrop, 0, ez);   /* exact */
			if (ez->n_cigar > 0) {
				append_cigar(cig, ez->n_cigar, ez->cigar);
				r->flags |= ORC_REG_HAS_DP;
			}
			if (ez->zdropped) {                                   /* truncated: the rest becomes a region of its own */
				for (j = i - 1; j >= 0; --j)
					if ((int32_t)a[as1 + j].x <= rs + ez->max_t) break;
				dropped = 1;
				if (j < 0) j = 0;
				r->dp_score += ez->max;
				re1 = rs + (ez->max_t + 1);
				qe1 = qs + (ez->max_q + 1);
				if (cnt1 - (j + 1) >= opt->min_cnt) {
					split_reg(r, r2, as1 + j + 1 - r->as, qlen, a);
					if (zdrop_code == 2) r2->flags |= ORC_REG_SPLIT_INV;
				}
				break;
			} else r->dp_score += ez->score;
			rs = re, qs = qe;
		}
	}

	if (!dropped && qe < qe0 && re < re0) {                     /* right extension */
		qseq = &qseq0[rev][qe];
		orc_index_getseq(mi, (uint32_t)rid, (uint32_t)re, (uint32_t)re0, tseq);
		align_pair(opt, qe0 - qe, qseq, re0 - re, tseq, mat, bw, opt->end_bonus, opt->zdrop, ORC_EZ_EXTZ_ONLY, ez);
		if (ez->n_cigar > 0) {
			append_cigar(cig, ez->n_cigar, ez->cigar);
			r->flags |= ORC_REG_HAS_DP;
			r->dp_score += ez->max;
		}
		re1 = re + (ez->reach_end ? ez->mqe_t + 1 : ez->max_t + 1);
		qe1 = qe + (ez->reach_end ? qe0 - qe : ez->max_q + 1);
	}

	r->rs = rs1, r->re = re1;
	if (rev) r->qs = qlen - qe1, r->qe = qlen - qs1;
	else r->qs = qs1, r->qe = qe1;

	if (r->flags & ORC_REG_HAS_DP) {
		free(tseq);
		tseq = (uint8_t*)malloc((size_t)(re1 - rs1 > 0 ? re1 - rs1 : 1));
		orc_index_getseq(mi, (uint32_t)rid, (uint32_t)rs1, (uint32_t)re1, tseq);
		update_extra(r, cig, &qseq0[r->rev][qs1], tseq, mat, (int8_t)opt->q, (int8_t)opt->e);
	}
	free(tseq);
}

/* ------------------------------------------------------------------ the inversion between two halves */

/* mm_align1_inv: r1 / r2 are the two halves of a region split by a Z-drop whose inversion test was
 * positive (r2 carries split_inv), both aligned.  The stretch between them -- query [r1.qe, r2.qs) on the
 * halves' strand, target [r1.re, r2.rs) -- is aligned on the OTHER strand of the query: a striped
 * Smith-Waterman on the reversed sequences (ksw_ll_i16) finds where the best local alignment starts,
 * an extension from there (ksw_extd2, band (int)(bw * 1.5), end bonus -1, Z-drop `zdrop`) gives the
 * alignment.  qcat = the read's two strands back to back (forward, then reverse complement), as
 * mm_align_skeleton lays them out: with the start found in the profile's padding (orc_ksw_ll_i16) the
 * query offset is negative and the reference reads up to seven bases in front of the stretch -- inside
 * this array, except at its very beginning, where the restatement gives up (the reference reads
 * foreign memory there). */
static int align1_inv(const orc_opt_t *opt, const orc_index *mi, int qlen, const uint8_t *qcat, const orc_reg_t *r1, const orc_reg_t *r2,
                      orc_reg_t *r_inv, cig_t *cig, orc_extz_t *ez)
{
	int tl, ql, score, q_off, t_off, ret = 0;
	int64_t q_base;
	uint8_t *tseq, *trev, *qrev;
	int8_t mat[25];
	memset(r_inv, 0, sizeof(*r_inv));
	if (!(r1->flags & ORC_REG_SPLIT_L) || !(r2->flags & ORC_REG_SPLIT_R)) return 0;
	if (r1->id != r1->parent && r1->parent != ORC_PARENT_TMP_PRI) return 0;
	if (r2->id != r2->parent && r2->parent != ORC_PARENT_TMP_PRI) return 0;
	if (r1->rid != r2->rid || r1->rev != r2->rev) return 0;
	ql = r1->rev ? r1->qs - r2->qe : r2->qs - r1->qe;
	tl = r2->rs - r1->re;
	if (ql < opt->min_chain_score || ql > opt->max_gap) return 0;
	if (tl < opt->min_chain_score || tl > opt->max_gap) return 0;
	orc_gen_simple_mat(5, mat, (int8_t)opt->a, (int8_t)opt->b, (int8_t)opt->sc_ambi);
	tseq = (uint8_t*)malloc((size_t)tl * 2 + (size_t)ql);
	trev = tseq + tl, qrev = trev + tl;
	orc_index_getseq(mi, (uint32_t)r1->rid, (uint32_t)r1->re, (uint32_t)r2->rs, tseq);
	q_base = r1->rev ? (int64_t)r2->qe : (int64_t)qlen + (qlen - r2->qs);     /* the other strand's copy of the stretch */
	{
		int i;
		for (i = 0; i < ql; ++i) qrev[i] = qcat[q_base + ql - 1 - i];
		for (i = 0; i < tl; ++i) trev[i] = tseq[tl - 1 - i];
	}
	score = orc_ksw_ll_i16(ql, qrev, tl, trev, 5, mat, opt->q, opt->e, &q_off, &t_off);
	if (score < opt->min_dp_max) goto end_inv;
	q_off = ql - (q_off + 1), t_off = tl - (t_off + 1);
	if (q_base + q_off < 0) goto end_inv;                      /* in front of the read's first base: see above */
	align_pair(opt, ql - q_off, qcat + q_base + q_off, tl - t_off, tseq + t_off, mat, (int)(opt->bw * 1.5), -1, opt->zdrop, ORC_EZ_EXTZ_ONLY, ez);
	if (ez->n_cigar == 0) goto end_inv;
	append_cigar(cig, ez->n_cigar, ez->cigar);
	r_inv->flags = ORC_REG_HAS_DP | ORC_REG_INV;
	r_inv->dp_score = (int32_t)ez->max;
	r_inv->id = -1, r_inv->parent = ORC_PARENT_UNSET;
	r_inv->rev = !r1->rev, r_inv->rid = r1->rid;
	if (r_inv->rev == 0) r_inv->qs = r2->qe + q_off, r_inv->qe = r_inv->qs + ez->max_q + 1;
	else r_inv->qe = r2->qs - q_off, r_inv->qs = r_inv->qe - (ez->max_q + 1);
	r_inv->rs = r1->re + t_off, r_inv->re = r_inv->rs + ez->max_t + 1;
	update_extra(r_inv, cig, qcat + q_base + q_off, tseq + t_off, mat, (int8_t)opt->q, (int8_t)opt->e);
	ret = 1;
end_inv:
	free(tseq);
	return ret;
}

/* ------------------------------------------------------------------ all regions of a read */

static int cmp_sort_key(const void *pa, const void *pb)
{
	const orc128_t *a = (const orc128_t*)pa, *b = (const orc128_t*)pb;
	if (a->x != b->x) return a->x < b->x ? -1 : 1;
	if (a->y != b->y) return a->y < b->y ? -1 : 1;
	return 0;
}

orc_reg_t *orc_align_regs(const orc_index *mi, const orc_opt_t *opt, int qlen, const char *seq,
                          int *n_regs_, orc_reg_t *regs, orc128_t *a, uint32_t ***cigars_out)
{
	int32_t i, n_regs = *n_regs_, n_a, m_regs = *n_regs_;
	uint8_t *qseq0[2];
	orc_extz_t ez;
	cig_t *cigs;
	if (cigars_out) *cigars_out = 0;
	if (n_regs == 0) return regs;
	qseq0[0] = (uint8_t*)malloc((size_t)qlen * 2);
	qseq0[1] = qseq0[0] + qlen;
	for (i = 0; i < qlen; ++i) {
		qseq0[0][i] = orc_nt4((unsigned char)seq[i]);
		qseq0[1][qlen - 1 - i] = qseq0[0][i] < 4 ? 3 - qseq0[0][i] : 4;
	}
	n_a = orc_squeeze_a(n_regs, regs, a);
	memset(&ez, 0, sizeof(ez));
	cigs = (cig_t*)calloc((size_t)m_regs, sizeof(cig_t));
	for (i = 0; i < n_regs; ++i) {
		orc_reg_t r2;
		memset(&r2, 0, sizeof(r2));
		align1(opt, mi, qlen, qseq0, &regs[i], &r2, &cigs[i], n_a, a, &ez);
		if (r2.cnt > 0) {                                       /* the tail of a Z-dropped region: aligned next */
			if (n_regs == m_regs) {
				m_regs = m_regs ? m_regs << 1 : 4;
				regs = (orc_reg_t*)realloc(regs, (size_t)m_regs * sizeof(orc_reg_t));
				cigs = (cig_t*)realloc(cigs, (size_t)m_regs * sizeof(cig_t));
			}
			if (i + 1 != n_regs) {
				memmove(&regs[i + 2], &regs[i + 1], sizeof(orc_reg_t) * (size_t)(n_regs - i - 1));
				memmove(&cigs[i + 2], &cigs[i + 1], sizeof(cig_t) * (size_t)(n_regs - i - 1));
			}
			regs[i + 1] = r2;
			memset(&cigs[i + 1], 0, sizeof(cig_t));
			++n_regs;
		}
		if (i > 0 && (regs[i].flags & ORC_REG_SPLIT_INV)) {     /* the inversion between regs[i-1] and regs[i] */
			orc_reg_t r_inv;
			cig_t c_inv = { 0, 0, 0 };
			if (align1_inv(opt, mi, qlen, qseq0[0], &regs[i - 1], &regs[i], &r_inv, &c_inv, &ez)) {
				if (n_regs == m_regs) {
					m_regs = m_regs ? m_regs << 1 : 4;
					regs = (orc_reg_t*)realloc(regs, (size_t)m_regs * sizeof(orc_reg_t));
					cigs = (cig_t*)realloc(cigs, (size_t)m_regs * sizeof(cig_t));
				}
				if (i + 1 != n_regs) {
					memmove(&regs[i + 2], &regs[i + 1], sizeof(orc_reg_t) * (size_t)(n_regs - i - 1));
					memmove(&cigs[i + 2], &cigs[i + 1], sizeof(cig_t) * (size_t)(n_regs - i - 1));
				}
				regs[i + 1] = r_inv, cigs[i + 1] = c_inv;
				++n_regs;
				++i;                                            /* not aligned again */
			} else free(c_inv.c);
		}
	}
	free(qseq0[0]);
	free(ez.cigar);
	/* filter, then order by DP score (hash as the tie-break; TOTAL ORDER: then original position) */
	{
		int k, n_aux = 0;
		orc128_t *aux;
		orc_reg_t *t;
		cig_t *tc;
		for (i = k = 0; i < n_regs; ++i) {                      /* mm_filter_regs on (regs, cigs) */
			int one = 1;
			orc_reg_t tmp = regs[i];
			orc_filter_regs(opt, qlen, &one, &tmp);
			if (one == 0) { free(cigs[i].c); continue; }
			regs[k] = regs[i], cigs[k] = cigs[i], ++k;
		}
		n_regs = k;
		aux = (orc128_t*)malloc((size_t)(n_regs ? n_regs : 1) * sizeof(orc128_t));
		t = (orc_reg_t*)malloc((size_t)(n_regs ? n_regs : 1) * sizeof(orc_reg_t));
		tc = (cig_t*)malloc((size_t)(n_regs ? n_regs : 1) * sizeof(cig_t));
		for (i = 0; i < n_regs; ++i) {
			if ((regs[i].flags & ORC_REG_INV) || regs[i].cnt > 0) {
				int score = (regs[i].flags & ORC_REG_HAS_DP) ? regs[i].dp_max : regs[i].score;
				aux[n_aux].x = (uint64_t)(uint32_t)score << 32 | regs[i].hash;
				aux[n_aux++].y = (uint64_t)i;
			} else free(cigs[i].c);
		}
		if (n_regs > 1) {
			qsort(aux, (size_t)n_aux, sizeof(orc128_t), cmp_sort_key);
			for (i = n_aux - 1; i >= 0; --i) t[n_aux - 1 - i] = regs[aux[i].y], tc[n_aux - 1 - i] = cigs[aux[i].y];
			memcpy(regs, t, sizeof(orc_reg_t) * (size_t)n_aux);
			memcpy(cigs, tc, sizeof(cig_t) * (size_t)n_aux);
			n_regs = n_aux;
		}
		free(aux), free(t), free(tc);
	}
	/* second hierarchy pass (align_regs).  select_sub compacts regs in place; cigars follow by id */
	orc_set_parent(opt->mask_level, n_regs, regs, opt->a * 2 + opt->b);
	{
		int n0 = n_regs, k;
		int *old_of = (int*)malloc((size_t)(n0 ? n0 : 1) * sizeof(int));
		/* remember which input slot each surviving region came from: select_sub keeps order */
		for (i = 0; i < n0; ++i) regs[i].n_cigar = cigs[i].n;
		{
			/* replicate select_sub's keep / drop decisions on a copy to learn the mapping */
			orc_reg_t *cp = (orc_reg_t*)malloc((size_t)(n0 > 0 ? n0 : 1) * sizeof(orc_reg_t));
			int n1 = n0;
			if (n0 > 0) memcpy(cp, regs, sizeof(orc_reg_t) * (size_t)n0);
			for (i = 0; i < n0; ++i) cp[i].dp_score = i;           /* carry the slot through the compaction */
			orc_select_sub(opt->pri_ratio, orc_index_k(mi) * 2, opt->best_n, &n1, cp);
			for (i = 0; i < n1; ++i) old_of[i] = cp[i].dp_score;
			free(cp);
			orc_select_sub(opt->pri_ratio, orc_index_k(mi) * 2, opt->best_n, &n_regs, regs);
			for (i = k = 0; i < n0; ++i) {
				if (k < n_regs && old_of[k] == i) cigs[k++] = cigs[i];
				else free(cigs[i].c);
			}
		}
		free(old_of);
	}
	if (cigars_out) {
		*cigars_out = (uint32_t**)malloc((size_t)(n_regs ? n_regs : 1) * sizeof(uint32_t*));
		for (i = 0; i < n_regs; ++i) (*cigars_out)[i] = cigs[i].c;
	} else for (i = 0; i < n_regs; ++i) free(cigs[i].c);
	free(cigs);
	*n_regs_ = n_regs;
	return regs;
}
