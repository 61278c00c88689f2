/*
 * mm_ksw.c -- CPU ORACLE (test infrastructure, NOT product code).  See mm_oracle.h.
 *
 * Base-level alignment kernel that mappy 2.17 / minimap2 2.17 runs for every region
 * (mappy sets MM_F_CIGAR unconditionally; monica consumes hit.mapq / hit.NM / hit.mlen at
 * monica/genomes/aligner.py:194-195, 216-217): ksw2's two-piece affine-gap extension/global
 * alignment `ksw_extd2_sse`, restated here in two forms:
 *
 *   orc_ksw_extd2()        a LITERAL scalar simulation of the SSE kernel: the same flat byte
 *                          buffer (u v x y x2 y2 s | target | reversed query), int8 wrap-around
 *                          arithmetic, anti-diagonal ranges rounded to 16 lanes, in-place
 *                          updates, the 4-lane exact-max scan with its tie order, the
 *                          approximate-max walk, Z-drop, direction bytes and ksw_backtrack.
 *                          This is the oracle the pipeline uses: at band edges the SSE kernel
 *                          reads cells outside the band that only a literal simulation
 *                          reproduces.
 *   orc_dp_clean()         the same recurrence in absolute int32 scores over the full matrix
 *                          (no band): what the literal form computes whenever the band does
 *                          not clip (max(qlen, tlen) <= w).  tests/ prove the two equal on
 *                          such inputs; it documents the arithmetic without the SSE layout.
 *
 * PARITY UNPINNED: written from the published algorithm (Suzuki & Kasahara 2018 difference
 * recurrence as used in Li 2018) and knowledge of the public lh3/ksw2 sources; none of it is
 * under /root/reference and it could not be run against the real library here.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mm_oracle.h"

#define NEG_INF (-0x40000000)

void orc_extz_reset(orc_extz_t *ez)
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0;
}

void orc_gen_simple_mat(int m, int8_t *mat, int8_t a, int8_t b, int8_t sc_ambi)
{
	int i, j;
	a = a < 0 ? -a : a;
	b = b > 0 ? -b : b;
	sc_ambi = sc_ambi > 0 ? -sc_ambi : sc_ambi;
	for (i = 0; i < m - 1; ++i) {
		for (j = 0; j < m - 1; ++j) mat[i * m + j] = i == j ? a : b;
		mat[i * m + m - 1] = sc_ambi;
	}
	for (j = 0; j < m; ++j) mat[(m - 1) * m + j] = sc_ambi;
}

static void push_cigar(orc_extz_t *ez, uint32_t op, int len)
{
	if (ez->n_cigar == 0 || op != (ez->cigar[ez->n_cigar - 1] & 0xf)) {
		if (ez->n_cigar == ez->m_cigar) {
			ez->m_cigar = ez->m_cigar ? ez->m_cigar << 1 : 4;
			ez->cigar = (uint32_t*)realloc(ez->cigar, (size_t)ez->m_cigar * 4);
		}
		ez->cigar[ez->n_cigar++] = (uint32_t)len << 4 | op;
	} else ez->cigar[ez->n_cigar - 1] += (uint32_t)len << 4;
}

/* ksw_backtrack for the rotated (anti-diagonal) layout: p[r * n_col + i - off[r]] holds, for
 * cell (i = target, j = r - i), bits 0-2 = which of {0 diagonal, 1 E, 2 F, 3 E2, 4 F2} gives H,
 * bits 3-6 = "the E / F / E2 / F2 value of the NEXT cell extends this one" */
static void backtrack(orc_extz_t *ez, int is_rev, const uint8_t *p, const int *off, const int *off_end,
                      int n_col, int i0, int j0)
{
	int i = i0, j = j0, r, state = 0, k;
	uint32_t tmp;
	ez->n_cigar = 0;
	while (i >= 0 && j >= 0) {
		int force_state = -1;
		r = i + j;
		if (i < off[r]) force_state = 2;
		if (off_end && i > off_end[r]) force_state = 1;
		tmp = force_state < 0 ? p[(size_t)r * n_col + i - off[r]] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force_state >= 0) state = force_state;
		if (state == 0) push_cigar(ez, 0, 1), --i, --j;
		else if (state == 1 || state == 3) push_cigar(ez, 2, 1), --i;
		else push_cigar(ez, 1, 1), --j;
	}
	if (i >= 0) push_cigar(ez, 2, i + 1);
	if (j >= 0) push_cigar(ez, 1, j + 1);
	if (!is_rev)
		for (k = 0; k < ez->n_cigar >> 1; ++k)
			tmp = ez->cigar[k], ez->cigar[k] = ez->cigar[ez->n_cigar - 1 - k], ez->cigar[ez->n_cigar - 1 - k] = tmp;
}

static inline int apply_zdrop(orc_extz_t *ez, int32_t H, int r, int t, int zdrop, int8_t e)
{
	if (H > (int32_t)ez->max) {
		ez->max = H, ez->max_t = t, ez->max_q = r - t;
	} else if (t >= ez->max_t && r - t >= ez->max_q) {
		int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l;
		l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) {
			ez->zdropped = 1;
			return 1;
		}
	}
	return 0;
}

#define I8(v) ((int8_t)(v))

void orc_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, orc_extz_t *ez)
{
	int r, t, qe = q + e, n_col_, *off = 0, *off_end = 0, tlen_, qlen_, last_st, last_en, wl, wr, max_sc, min_sc, long_thres, long_diff;
	const int with_cigar = !(flag & ORC_EZ_SCORE_ONLY), approx_max = !!(flag & ORC_EZ_APPROX_MAX);
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	int8_t *mem, *u, *v, *x, *y, *x2, *y2, *s, sc_mch, sc_mis, sc_N, qe_, qe2_;
	uint8_t *sf, *qr, *p = 0;
	size_t T;

	orc_extz_reset(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0) return;
	if (q2 + e2 < q + e) t = q, q = q2, q2 = t, t = e, e = e2, e2 = t;   /* q+e no larger than q2+e2 */
	qe_ = q + e, qe2_ = q2 + e2;
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? -e2 : mat[m * m - 1];

	if (w < 0) w = tlen > qlen ? tlen : qlen;
	wl = wr = w;
	tlen_ = (tlen + 15) / 16;
	n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	qlen_ = (qlen + 15) / 16;
	for (t = 1, max_sc = mat[0], min_sc = mat[1]; t < m * m; ++t) {
		max_sc = max_sc > mat[t] ? max_sc : mat[t];
		min_sc = min_sc < mat[t] ? min_sc : mat[t];
	}
	if (-min_sc > 2 * (q + e)) return;

	long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	T = (size_t)tlen_ * 16;
	mem = (int8_t*)calloc(T * 8 + (size_t)(qlen_ + 2) * 16, 1);     /* one flat buffer, as the SSE kernel lays it out */
	u = mem, v = u + T, x = v + T, y = x + T, x2 = y + T, y2 = x2 + T;
	s = y2 + T, sf = (uint8_t*)(s + T), qr = sf + T;
	memset(u, -q - e, T * 4);                                       /* u, v, x, y */
	memset(x2, -q2 - e2, T * 2);                                    /* x2, y2 */
	if (!approx_max) {
		H = (int32_t*)malloc(T * 4);
		for (t = 0; t < (int)T; ++t) H[t] = NEG_INF;
	}
	if (with_cigar) {
		p = (uint8_t*)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
		off = (int*)malloc((size_t)(qlen + tlen - 1) * sizeof(int) * 2);
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, (size_t)tlen);

	for (r = 0, last_st = last_en = -1; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21, v1;
		const uint8_t *qrr = qr + (qlen - 1 - r);
		/* boundaries of the anti-diagonal inside the band */
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
		if (en > (r + wl) >> 1) en = (r + wl) >> 1;
		if (st > en) {
			ez->zdropped = 1;
			break;
		}
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;            /* whole 16-lane vectors */
		/* values left of the first lane */
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) {
				x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
			} else {
				x1 = -q - e, x21 = -q2 - e2;
				v1 = -q - e;
			}
		} else {
			x1 = -q - e, x21 = -q2 - e2;
			v1 = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		}
		if (en >= r) {
			y[r] = -q - e, y2[r] = -q2 - e2;
			u[r] = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		}
		/* scores, in 16-lane strides from st0 (may run past en0, and past the end of s[]) */
		for (t = st0; t <= en0; t += 16) {
			int l;
			int8_t tmp[16];
			for (l = 0; l < 16; ++l) {
				const uint8_t sq = sf[t + l], sq2 = qrr[t + l];
				const int mask = sq == (uint8_t)(m - 1) || sq2 == (uint8_t)(m - 1);
				tmp[l] = mask ? sc_N : sq == sq2 ? sc_mch : sc_mis;
			}
			memcpy(s + t, tmp, 16);                                 /* loads above happen before the store, as in SSE */
		}
		/* core: every lane of [st, en] reads the values of the previous anti-diagonal; going down
		 * in t keeps [t-1] old while [t] is overwritten */
		if (with_cigar) off[r] = st, off_end[r] = en;
		for (t = en; t >= st; --t) {
			int8_t z = s[t], a, b, a2, b2, xt1, x2t1, vt1, ut, tmp;
			uint8_t d;
			xt1 = t > st ? x[t - 1] : x1;
			vt1 = t > st ? v[t - 1] : v1;
			x2t1 = t > st ? x2[t - 1] : x21;
			ut = u[t];
			a = I8(xt1 + vt1), b = I8(y[t] + ut), a2 = I8(x2t1 + vt1), b2 = I8(y2[t] + ut);
			if (!(flag & ORC_EZ_RIGHT)) {                          /* gap left-alignment */
				d = a > z ? 1 : 0;  z = z > a ? z : a;
				d = b > z ? 2 : d;  z = z > b ? z : b;
				d = a2 > z ? 3 : d; z = z > a2 ? z : a2;
				d = b2 > z ? 4 : d; z = z > b2 ? z : b2;
				z = z < sc_mch ? z : sc_mch;
				u[t] = I8(z - vt1), v[t] = I8(z - ut);
				tmp = I8(z - q), a = I8(a - tmp), b = I8(b - tmp);
				tmp = I8(z - q2), a2 = I8(a2 - tmp), b2 = I8(b2 - tmp);
				x[t] = I8((a > 0 ? a : 0) - qe_);    d |= a > 0 ? 0x08 : 0;
				y[t] = I8((b > 0 ? b : 0) - qe_);    d |= b > 0 ? 0x10 : 0;
				x2[t] = I8((a2 > 0 ? a2 : 0) - qe2_); d |= a2 > 0 ? 0x20 : 0;
				y2[t] = I8((b2 > 0 ? b2 : 0) - qe2_); d |= b2 > 0 ? 0x40 : 0;
			} else {                                                /* gap right-alignment */
				d = z > a ? 0 : 1;  z = z > a ? z : a;
				d = z > b ? d : 2;  z = z > b ? z : b;
				d = z > a2 ? d : 3; z = z > a2 ? z : a2;
				d = z > b2 ? d : 4; z = z > b2 ? z : b2;
				z = z < sc_mch ? z : sc_mch;
				u[t] = I8(z - vt1), v[t] = I8(z - ut);
				tmp = I8(z - q), a = I8(a - tmp), b = I8(b - tmp);
				tmp = I8(z - q2), a2 = I8(a2 - tmp), b2 = I8(b2 - tmp);
				x[t] = I8((0 > a ? 0 : a) - qe_);    d |= 0 > a ? 0 : 0x08;
				y[t] = I8((0 > b ? 0 : b) - qe_);    d |= 0 > b ? 0 : 0x10;
				x2[t] = I8((0 > a2 ? 0 : a2) - qe2_); d |= 0 > a2 ? 0 : 0x20;
				y2[t] = I8((0 > b2 ? 0 : b2) - qe2_); d |= 0 > b2 ? 0 : 0x40;
			}
			if (with_cigar) p[((size_t)r * n_col_) * 16 + (size_t)(t - st)] = d;
		}
		if (!approx_max) {                                          /* exact max over the anti-diagonal, 32-bit H[] */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u[en0] : H[en0] + v[en0];   /* the last element first */
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4) {                    /* four lanes, each keeps its first strict maximum */
					for (i = 0; i < 4; ++i) {
						H[t + i] += (int32_t)v[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v[t];
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = v[0] - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else {                                                    /* approximate max: H of one cell per anti-diagonal */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v[last_H0_t];
					int32_t d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t];
				} else {
					++last_H0_t, H0 += u[last_H0_t];
				}
			} else H0 = v[0] - qe, last_H0_t = 0;
			if ((flag & ORC_EZ_APPROX_DROP) && apply_zdrop(ez, H0, r, last_H0_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem);
	free(H);
	if (with_cigar) {
		const int rev_cigar = !!(flag & ORC_EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag & ORC_EZ_EXTZ_ONLY)) {
			backtrack(ez, rev_cigar, p, off, off_end, n_col_ * 16, tlen - 1, qlen - 1);
		} else if (!ez->zdropped && (flag & ORC_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
			ez->reach_end = 1;
			backtrack(ez, rev_cigar, p, off, off_end, n_col_ * 16, ez->mqe_t, qlen - 1);
		} else if (ez->max_t >= 0 && ez->max_q >= 0) {
			backtrack(ez, rev_cigar, p, off, off_end, n_col_ * 16, ez->max_t, ez->max_q);
		}
		free(p), free(off);
	}
}

/* ------------------------------------------------------------------ the same recurrence, plainly
 *
 * Full matrix, absolute int32 scores.  i = target index, j = query index.
 *   H(-1,-1) = 0, H(-1,j) = -gap(j+1), H(i,-1) = -gap(i+1), gap(l) = min(q + e l, q2 + e2 l)
 *   E (i,j) = max(E (i-1,j), H(i-1,j) - q ) - e     (a run of target bases: CIGAR 'D')
 *   F (i,j) = max(F (i,j-1), H(i,j-1) - q ) - e     (a run of query bases:  CIGAR 'I')
 *   E2, F2 likewise with (q2, e2); first row / column open from the boundary H
 *   H (i,j) = max(H(i-1,j-1) + s(i,j), E, F, E2, F2), ties resolved in that order (left
 *             alignment: a later candidate must be strictly greater; right: greater or equal)
 * Extension mode tracks, per anti-diagonal, the maximum with the SSE scan's tie order and
 * applies the same Z-drop test.  Valid (and equal to orc_ksw_extd2) when the band does not
 * clip the matrix; `w` is not a parameter here. */
void orc_dp_clean(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                  int q, int e, int q2, int e2, int zdrop, int end_bonus, int flag, orc_extz_t *ez)
{
	const int right = !!(flag & ORC_EZ_RIGHT), approx = !!(flag & ORC_EZ_APPROX_MAX);
	const size_t W = (size_t)qlen;
	int32_t *H, *E, *F, *E2, *F2;
	uint8_t *D;
	int i, j, r, *off, *off_end;
	uint8_t *p;
	const int n_col = (qlen < tlen ? qlen : tlen) + 1;
	orc_extz_reset(ez);
	if (qlen <= 0 || tlen <= 0) return;
#define GAP(l) ((q + e * (l)) < (q2 + e2 * (l)) ? (q + e * (l)) : (q2 + e2 * (l)))
#define AT(A, i, j) A[(size_t)(i) * W + (size_t)(j)]
	H = (int32_t*)malloc(W * tlen * 4), E = (int32_t*)malloc(W * tlen * 4), F = (int32_t*)malloc(W * tlen * 4);
	E2 = (int32_t*)malloc(W * tlen * 4), F2 = (int32_t*)malloc(W * tlen * 4);
	D = (uint8_t*)calloc(W * tlen, 1);
	for (r = 0; r < qlen + tlen - 1; ++r) {
		const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = r < tlen - 1 ? r : tlen - 1;
		int32_t max_H = NEG_INF, max_t = -1;
		for (i = st0; i <= en0; ++i) {
			int32_t diag, ee, ff, ee2, ff2, z, sc;
			uint8_t d;
			j = r - i;
			sc = mat[target[i] * 5 + query[j]];
			if (target[i] == 4 || query[j] == 4) sc = mat[24] == 0 ? -e2 : mat[24];
			diag = i > 0 && j > 0 ? AT(H, i - 1, j - 1) : i == 0 && j == 0 ? 0 : i == 0 ? -GAP(j) : -GAP(i);
			if (i > 0) {
				const int32_t o = AT(H, i - 1, j) - q, o2 = AT(H, i - 1, j) - q2;
				const int ext = right ? AT(E, i - 1, j) >= o : AT(E, i - 1, j) > o;
				const int ext2 = right ? AT(E2, i - 1, j) >= o2 : AT(E2, i - 1, j) > o2;
				ee = (ext ? AT(E, i - 1, j) : o) - e, ee2 = (ext2 ? AT(E2, i - 1, j) : o2) - e2;
				AT(D, i - 1, j) |= (ext ? 0x08 : 0) | (ext2 ? 0x20 : 0);
			} else ee = -GAP(j + 1) - q - e, ee2 = -GAP(j + 1) - q2 - e2;
			if (j > 0) {
				const int32_t o = AT(H, i, j - 1) - q, o2 = AT(H, i, j - 1) - q2;
				const int ext = right ? AT(F, i, j - 1) >= o : AT(F, i, j - 1) > o;
				const int ext2 = right ? AT(F2, i, j - 1) >= o2 : AT(F2, i, j - 1) > o2;
				ff = (ext ? AT(F, i, j - 1) : o) - e, ff2 = (ext2 ? AT(F2, i, j - 1) : o2) - e2;
				AT(D, i, j - 1) |= (ext ? 0x10 : 0) | (ext2 ? 0x40 : 0);
			} else ff = -GAP(i + 1) - q - e, ff2 = -GAP(i + 1) - q2 - e2;
			z = diag + sc;
			if (!right) {
				d = ee > z ? 1 : 0;  z = z > ee ? z : ee;
				d = ff > z ? 2 : d;  z = z > ff ? z : ff;
				d = ee2 > z ? 3 : d; z = z > ee2 ? z : ee2;
				d = ff2 > z ? 4 : d; z = z > ff2 ? z : ff2;
			} else {
				d = z > ee ? 0 : 1;  z = z > ee ? z : ee;
				d = z > ff ? d : 2;  z = z > ff ? z : ff;
				d = z > ee2 ? d : 3; z = z > ee2 ? z : ee2;
				d = z > ff2 ? d : 4; z = z > ff2 ? z : ff2;
			}
			AT(H, i, j) = z, AT(E, i, j) = ee, AT(F, i, j) = ff, AT(E2, i, j) = ee2, AT(F2, i, j) = ff2;
			AT(D, i, j) |= d;
		}
		if (!approx) {
			/* the SSE scan's order: the last cell first, then four interleaved lanes (each keeps its
			 * first strict maximum; lanes merged in order, strictly), then the scalar tail */
			const int en1 = st0 + (en0 - st0) / 4 * 4;
			int32_t HH[4], tt[4];
			int t, l;
			max_H = AT(H, en0, r - en0), max_t = en0;
			for (l = 0; l < 4; ++l) HH[l] = max_H, tt[l] = max_t;
			for (t = st0; t < en1; t += 4)
				for (l = 0; l < 4; ++l)
					if (AT(H, t + l, r - t - l) > HH[l]) HH[l] = AT(H, t + l, r - t - l), tt[l] = t;
			for (l = 0; l < 4; ++l) if (max_H < HH[l]) max_H = HH[l], max_t = tt[l] + l;
			for (; t < en0; ++t) if (AT(H, t, r - t) > max_H) max_H = AT(H, t, r - t), max_t = t;
			if (en0 == tlen - 1 && AT(H, en0, r - en0) > ez->mte) ez->mte = AT(H, en0, r - en0), ez->mte_q = r - ((en0 + 16) / 16 * 16 - 1);
			if (r - st0 == qlen - 1 && AT(H, st0, r - st0) > ez->mqe) ez->mqe = AT(H, st0, r - st0), ez->mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, (int8_t)e2)) break;
		}
		if (r == qlen + tlen - 2) ez->score = AT(H, tlen - 1, qlen - 1);
	}
	/* direction bytes in the rotated layout the shared backtrack reads */
	off = (int*)malloc((size_t)(qlen + tlen) * 2 * sizeof(int));
	off_end = off + qlen + tlen;
	p = (uint8_t*)calloc((size_t)(qlen + tlen) * n_col, 1);
	for (r = 0; r < qlen + tlen - 1; ++r) {
		const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = r < tlen - 1 ? r : tlen - 1;
		off[r] = st0, off_end[r] = en0;
		for (i = st0; i <= en0; ++i) p[(size_t)r * n_col + i - st0] = AT(D, i, r - i);
	}
	if (!ez->zdropped && !(flag & ORC_EZ_EXTZ_ONLY)) backtrack(ez, !!(flag & ORC_EZ_REV_CIGAR), p, off, off_end, n_col, tlen - 1, qlen - 1);
	else if (!ez->zdropped && (flag & ORC_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
		ez->reach_end = 1;
		backtrack(ez, !!(flag & ORC_EZ_REV_CIGAR), p, off, off_end, n_col, ez->mqe_t, qlen - 1);
	} else if (ez->max_t >= 0 && ez->max_q >= 0) backtrack(ez, !!(flag & ORC_EZ_REV_CIGAR), p, off, off_end, n_col, ez->max_t, ez->max_q);
	free(H), free(E), free(F), free(E2), free(F2), free(D), free(off), free(p);
#undef GAP
#undef AT
}

/* ksw_ll_i16 as minimap2 uses it in mm_test_zdrop: the best local alignment score (one affine
 * gap cost q + e l), i.e. plain Smith-Waterman; only the score is consumed there */
int orc_local_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int q, int e)
{
	int32_t *H = (int32_t*)calloc((size_t)qlen + 1, 4), *E = (int32_t*)calloc((size_t)qlen + 1, 4);
	int i, j, best = 0;
	for (i = 0; i < tlen; ++i) {
		int32_t f = 0, diag = 0;
		for (j = 0; j < qlen; ++j) {
			/* E[j]: gap state entering (i, j) along the target; f: along the query */
			int32_t h = diag + mat[target[i] * 5 + query[j]], t;
			diag = H[j + 1];
			h = h > E[j + 1] ? h : E[j + 1];
			h = h > f ? h : f;
			h = h > 0 ? h : 0;
			H[j + 1] = h;
			best = best > h ? best : h;
			t = h - (q + e);
			t = t > 0 ? t : 0;
			E[j + 1] = E[j + 1] - e > t ? E[j + 1] - e : t;
			f = f - e > t ? f - e : t;
		}
	}
	free(H), free(E);
	return best > 32767 ? 32767 : best;
}


/* ksw_ll_i16 LITERALLY (ksw2_ll_sse.c, the int16 form of Farrar's striped Smith-Waterman that
 * minimap2 calls from mm_align1_inv and mm_test_zdrop): eight lanes of int16, the query striped over
 * slen = ceil(qlen / 8) vectors (position k sits in vector k % slen, lane k / slen; positions beyond
 * the query score 0), E kept per position, F carried along a column and corrected by the "lazy F"
 * loop, unsigned saturating subtractions as the floor at zero.  What mm_align1_inv consumes beyond the
 * score are the END coordinates, whose tie rules come from this layout: *te is the LAST column whose
 * maximum equals the best score (`imax >= gmax`), *qe the position of that column with the largest
 * index in the striped memory order (the loop over Hmax keeps the last match) -- which may be a
 * padding position >= qlen when the best alignment ends on the query's last base and its score is
 * carried diagonally into the padding (then the caller's offsets fall outside the sequences; the
 * restatement of mm_align1_inv gives up there).  Sequences are codes 0 .. m-1. */
static inline int16_t ll_adds(int16_t a, int16_t b) { int v = (int)a + b; return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }
static inline int16_t ll_subu(int16_t a, int16_t b) { unsigned x = (uint16_t)a, y = (uint16_t)b; return (int16_t)(uint16_t)(x > y ? x - y : 0); }
static inline int16_t ll_max(int16_t a, int16_t b) { return a > b ? a : b; }

int orc_ksw_ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                   int gapo, int gape, int *qe, int *te)
{
	const int p = 8, slen = (qlen + p - 1) / p, qlen8 = slen * p;
	int16_t *qp, *H0, *H1, *E, *Hmax, *t;
	int a, i, gmax = 0;
	const int16_t gapoe = (int16_t)(gapo + gape), ge = (int16_t)gape;
	*qe = *te = -1;
	if (qlen <= 0) return 0;
	qp = (int16_t*)malloc((size_t)m * qlen8 * 2);
	H0 = (int16_t*)calloc((size_t)qlen8, 2), H1 = (int16_t*)calloc((size_t)qlen8, 2);
	E = (int16_t*)calloc((size_t)qlen8, 2), Hmax = (int16_t*)calloc((size_t)qlen8, 2);
	for (a = 0, t = qp; a < m; ++a) {                          /* the query profile, striped */
		int k;
		const int8_t *ma = mat + a * m;
		for (i = 0; i < slen; ++i)
			for (k = i; k < qlen8; k += slen) *t++ = (int16_t)(k >= qlen ? 0 : ma[query[k]]);
	}
	for (i = 0; i < tlen; ++i) {
		int j, k, l, imax;
		int16_t h[8], f[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, mx[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, *tmp;
		const int16_t *S = qp + (size_t)target[i] * qlen8;
		for (l = 0; l < 8; ++l) h[l] = l ? H0[(slen - 1) * 8 + l - 1] : 0;      /* the last vector, shifted by one lane */
		for (j = 0; j < slen; ++j) {
			for (l = 0; l < 8; ++l) {
				int16_t v = ll_adds(h[l], S[j * 8 + l]), e = E[j * 8 + l], hu;
				v = ll_max(v, e), v = ll_max(v, f[l]);
				mx[l] = ll_max(mx[l], v);
				H1[j * 8 + l] = v;
				hu = ll_subu(v, gapoe);
				e = ll_subu(e, ge), e = ll_max(e, hu);
				E[j * 8 + l] = e;
				f[l] = ll_subu(f[l], ge), f[l] = ll_max(f[l], hu);
				h[l] = H0[j * 8 + l];
			}
		}
		for (k = 0; k < 8; ++k) {                              /* lazy F */
			int16_t g[8];
			for (l = 0; l < 8; ++l) g[l] = l ? f[l - 1] : 0;
			memcpy(f, g, sizeof(f));
			for (j = 0; j < slen; ++j) {
				int any = 0;
				for (l = 0; l < 8; ++l) {
					int16_t hh = ll_max(H1[j * 8 + l], f[l]);
					H1[j * 8 + l] = hh;
					hh = ll_subu(hh, gapoe);
					f[l] = ll_subu(f[l], ge);
					if (f[l] > hh) any = 1;
				}
				if (!any) goto end_loop;
			}
		}
end_loop:
		for (l = 0, imax = 0; l < 8; ++l) imax = imax > mx[l] ? imax : mx[l];
		if (imax >= gmax) {
			gmax = imax, *te = i;
			memcpy(Hmax, H1, (size_t)qlen8 * 2);
		}
		tmp = H1, H1 = H0, H0 = tmp;
	}
	for (i = 0; i < qlen8; ++i)
		if ((int)(uint16_t)Hmax[i] == gmax) *qe = i / 8 + i % 8 * slen;
	free(qp), free(H0), free(H1), free(E), free(Hmax);
	return gmax;
}

/* The same three numbers from the plain recurrence (a second formulation, for the tests and for what the
 * GPU computes): Smith-Waterman with one affine gap cost over the query PADDED to a multiple of eight
 * positions that score 0, the best score, the last target position whose column holds it, and in that
 * column the position k with the largest (k % slen, k / slen). */
int orc_local_end(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int q, int e, int *qe, int *te)
{
	const int slen = (qlen + 7) / 8, L = slen * 8;
	int32_t *H = (int32_t*)calloc((size_t)L + 1, 4), *E = (int32_t*)calloc((size_t)L + 1, 4), *col = (int32_t*)calloc((size_t)L + 1, 4);
	int i, j, best = 0;
	*qe = *te = -1;
	if (qlen <= 0) { free(H), free(E), free(col); return 0; }
	for (i = 0; i < tlen; ++i) {
		int32_t f = 0, diag = 0, cmax = 0;
		for (j = 0; j < L; ++j) {
			int32_t h = diag + (j < qlen ? mat[target[i] * 5 + query[j]] : 0), t;
			diag = H[j + 1];
			h = h > E[j + 1] ? h : E[j + 1];
			h = h > f ? h : f;
			h = h > 0 ? h : 0;
			H[j + 1] = h;
			cmax = cmax > h ? cmax : h;
			t = h - (q + e);
			t = t > 0 ? t : 0;
			E[j + 1] = E[j + 1] - e > t ? E[j + 1] - e : t;
			f = f - e > t ? f - e : t;
		}
		if (cmax >= best) {
			int bk = -1;
			best = cmax, *te = i;
			for (j = 0; j < L; ++j)
				if (H[j + 1] == best && (bk < 0 || j % slen > bk % slen || (j % slen == bk % slen && j / slen > bk / slen))) bk = j;
			*qe = bk;
		}
	}
	free(H), free(E), free(col);
	return best;
}
