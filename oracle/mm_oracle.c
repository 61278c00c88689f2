/*
 * mm_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See mm_oracle.h.
 *
 * PARITY UNPINNED at the mappy boundary: the arithmetic lives in
 * minimap2 v2.17 (pinned as mappy==2.17 at /root/reference/requirements.txt:3), which is
 * not vendored, not installed and not fetchable.  Every function below cites the
 * reference call site it serves and the SURVEY.md Appendix A paragraph it restates.
 *
 * Intentional strengthening (hard part 5): wherever upstream uses an
 * unstable radix sort on a partial key, this restatement defines a TOTAL order
 * (documented at each sort) so results are deterministic and a GPU can match them.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "mm_oracle.h"
#include "mm_internal.h"

#define PARENT_UNSET   (-1)
#define PARENT_TMP_PRI (-2)
#define SEED_LONG_JOIN (1ULL<<40)
#define SEED_TANDEM    (1ULL<<42)

/* ------------------------------------------------------------------ options (A.1) */

void orc_opt_init(orc_opt_t *o)
{
	memset(o, 0, sizeof(*o));
	o->seed = 11;
	o->mid_occ_frac = 2e-4f;
	o->mid_occ = 0;
	o->min_cnt = 3;
	o->min_chain_score = 40;
	o->bw = 500;
	o->max_gap = 5000;
	o->max_chain_skip = 25;
	o->max_chain_iter = 5000;
	o->mask_level = 0.5f;
	o->pri_ratio = 0.8f;
	o->best_n = 5;
	o->max_join_long = 20000;
	o->max_join_short = 2000;
	o->min_join_flank_sc = 1000;
	o->min_join_flank_ratio = 0.5f;
	o->a = 2, o->b = 4;
	o->cigar = 1;                     /* mappy ORs MM_F_CIGAR in unconditionally (A.1) */
	o->q = 4, o->e = 2, o->q2 = 24, o->e2 = 1;
	o->sc_ambi = 1;
	o->zdrop = 400, o->zdrop_inv = 200;
	o->end_bonus = -1;
	o->min_dp_max = o->min_chain_score * o->a;
	o->min_ksw_len = 200;
	o->max_clip_ratio = 1.0f;
	o->max_sw_mat = 100000000;
}

/* ------------------------------------------------------------------ sketch (A.2) */

/* base -> 0..3, everything else 4 (A.2). U/u counts as T like upstream's table. */
unsigned char orc_nt4(unsigned char c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

/* invertible integer mix under `mask` (A.2) */
uint64_t orc_hash64(uint64_t key, uint64_t mask)
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

typedef struct { orc128_t *a; int64_t n, m; } vec128;

static void vpush(vec128 *v, orc128_t e)
{
	if (v->n == v->m) {
		v->m = v->m ? v->m << 1 : 256;
		v->a = (orc128_t*)realloc(v->a, (size_t)v->m * sizeof(orc128_t));
	}
	v->a[v->n++] = e;
}

/*
 * The (w,k)-minimizer state machine of A.2, one base at a time, ring buffer of the last w
 * k-mer records.  Serves index.map(seq) at aligner.py:193,215 (query side) and
 * mappy.Aligner(fn_idx_in=...) at aligner.py:45 (reference side).
 */
static void sketch_core(const char *str, int len, int w, int k, uint32_t rid, vec128 *p)
{
	const uint64_t shift1 = 2 * (uint64_t)(k - 1), mask = (1ULL << 2 * k) - 1;
	const orc128_t none = { UINT64_MAX, UINT64_MAX };
	uint64_t kmer[2] = { 0, 0 };
	orc128_t buf[256], min = none;
	int i, j, l = 0, buf_pos = 0, min_pos = 0, kmer_span = 0;

	if (len <= 0 || w <= 0 || w >= 256 || k <= 0 || k > 28) return;
	for (j = 0; j < w; ++j) buf[j] = none;

	for (i = 0; i < len; ++i) {
		int c = orc_nt4((unsigned char)str[i]);
		orc128_t info = none;
		if (c < 4) {
			int z;
			kmer_span = l + 1 < k ? l + 1 : k;
			kmer[0] = (kmer[0] << 2 | (uint64_t)c) & mask;
			kmer[1] = (kmer[1] >> 2) | (3ULL ^ (uint64_t)c) << shift1;
			if (kmer[0] == kmer[1]) continue;      /* strand-symmetric k-mer: skipped entirely */
			z = kmer[0] < kmer[1] ? 0 : 1;
			++l;
			if (l >= k && kmer_span < 256) {
				info.x = orc_hash64(kmer[z], mask) << 8 | (uint64_t)kmer_span;
				info.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | (uint64_t)z;
			}
		} else l = 0, kmer_span = 0;
		buf[buf_pos] = info;
		if (l == w + k - 1 && min.x != UINT64_MAX) {   /* first full window: equal-hash records */
			for (j = buf_pos + 1; j < w; ++j)
				if (min.x == buf[j].x && buf[j].y != min.y) vpush(p, buf[j]);
			for (j = 0; j < buf_pos; ++j)
				if (min.x == buf[j].x && buf[j].y != min.y) vpush(p, buf[j]);
		}
		if (info.x <= min.x) {                          /* new minimum (ties go to the newest) */
			if (l >= w + k && min.x != UINT64_MAX) vpush(p, min);
			min = info, min_pos = buf_pos;
		} else if (buf_pos == min_pos) {                /* old minimum leaves the window */
			if (l >= w + k - 1 && min.x != UINT64_MAX) vpush(p, min);
			for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j)
				if (min.x >= buf[j].x) min = buf[j], min_pos = j;   /* >= : right-most wins */
			for (j = 0; j <= buf_pos; ++j)
				if (min.x >= buf[j].x) min = buf[j], min_pos = j;
			if (l >= w + k - 1 && min.x != UINT64_MAX) {
				for (j = buf_pos + 1; j < w; ++j)
					if (min.x == buf[j].x && min.y != buf[j].y) vpush(p, buf[j]);
				for (j = 0; j <= buf_pos; ++j)
					if (min.x == buf[j].x && min.y != buf[j].y) vpush(p, buf[j]);
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (min.x != UINT64_MAX) vpush(p, min);
}

int orc_sketch(const char *seq, int len, int w, int k, uint32_t rid, orc128_t *out, int cap)
{
	vec128 v = { 0, 0, 0 };
	int64_t i;
	sketch_core(seq, len, w, k, rid, &v);
	for (i = 0; i < v.n && i < cap; ++i) out[i] = v.a[i];
	free(v.a);
	return (int)v.n;
}

/* ------------------------------------------------------------------ index (A.3) */

struct orc_index {
	int k, w, n_seq;
	char **name;
	int *len;
	int64_t n_occ;          /* total minimizer occurrences */
	uint64_t *P;            /* all y words, grouped by hash, ascending y inside a group */
	/* open-addressed table over distinct hashes */
	int64_t n_keys;
	uint64_t tmask;
	uint32_t *tkey;         /* hash + 1, 0 = empty */
	uint64_t *tval;         /* offset<<32 | count  (count < 2^32, offset < 2^32) */
	uint64_t *big_off;      /* NULL unless offsets exceed 32 bits */
	uint32_t *cnts;         /* per distinct key: occurrence count (for mid_occ) */
	uint8_t **seq;          /* contig bases as codes 0..4 (minimap2 keeps them 4-bit packed; A.3) */
};

static int cmp128(const void *pa, const void *pb)
{
	const orc128_t *a = (const orc128_t*)pa, *b = (const orc128_t*)pb;
	if (a->x != b->x) return a->x < b->x ? -1 : 1;
	if (a->y != b->y) return a->y < b->y ? -1 : 1;
	return 0;
}

static int cmp64(const void *pa, const void *pb)
{
	uint64_t a = *(const uint64_t*)pa, b = *(const uint64_t*)pb;
	return a < b ? -1 : a > b;
}

static int cmp32(const void *pa, const void *pb)
{
	uint32_t a = *(const uint32_t*)pa, b = *(const uint32_t*)pb;
	return a < b ? -1 : a > b;
}

static inline uint64_t tslot(uint64_t h, uint64_t tmask)
{
	return (h * 0x9E3779B97F4A7C15ULL >> 20) & tmask;
}

/* The (hash, y) pairs are distinct, so their total order has one sorted form whatever the
 * method: slices sorted with qsort side by side, then merged pairwise (the oracle's index build
 * is test set-up time: 50 M pairs per 280 Mbp part of BASELINE config 4). */
static void sort128(orc128_t *a, int64_t n)
{
	int T = omp_get_max_threads(), t, width;
	int64_t bound[65];
	orc128_t *tmp, *src, *dst;
	if (T > 64) T = 64;
	while (T & (T - 1)) T &= T - 1;                        /* a power of two */
	if (n < 1000000 || T < 2) { qsort(a, (size_t)n, sizeof(orc128_t), cmp128); return; }
	for (t = 0; t <= T; ++t) bound[t] = n * t / T;
#pragma omp parallel for schedule(static, 1) num_threads(T)
	for (t = 0; t < T; ++t) qsort(a + bound[t], (size_t)(bound[t + 1] - bound[t]), sizeof(orc128_t), cmp128);
	tmp = (orc128_t*)malloc((size_t)n * sizeof(orc128_t));
	src = a, dst = tmp;
	for (width = 1; width < T; width *= 2) {
#pragma omp parallel for schedule(static, 1) num_threads(T)
		for (t = 0; t < T; t += 2 * width) {
			int64_t i = bound[t], ie = bound[t + width], j = ie, je = bound[t + 2 * width], o = bound[t];
			while (i < ie && j < je) dst[o++] = cmp128(&src[j], &src[i]) < 0 ? src[j++] : src[i++];
			while (i < ie) dst[o++] = src[i++];
			while (j < je) dst[o++] = src[j++];
		}
		{ orc128_t *x = src; src = dst; dst = x; }
	}
	if (src != a) memcpy(a, src, (size_t)n * sizeof(orc128_t));
	free(tmp);
}

/* all minimizers of all contigs -> (hash, y) pairs -> groups.  The position list of one
 * hash is sorted by y ascending, as A.3 says ("positions sorted ascending"). */
static orc_index *index_from_pairs(orc_index *mi, vec128 *v)
{
	int64_t i, j, nk = 0, off;
	uint64_t tsize = 16;
	for (i = 0; i < v->n; ++i) v->a[i].x >>= 8;         /* key = hash only (span dropped) */
	double t0 = omp_get_wtime();
	sort128(v->a, v->n);
	if (getenv("ORC_TIMING")) fprintf(stderr, "[oracle] sort %.2f s\n", omp_get_wtime() - t0);
	for (i = 0; i < v->n; ++i) if (i == 0 || v->a[i].x != v->a[i-1].x) ++nk;
	mi->n_occ = v->n, mi->n_keys = nk;
	while (tsize < (uint64_t)nk * 2) tsize <<= 1;
	mi->tmask = tsize - 1;
	mi->tkey = (uint32_t*)calloc(tsize, 4);
	mi->tval = (uint64_t*)calloc(tsize, 8);
	mi->P = (uint64_t*)malloc((size_t)(v->n ? v->n : 1) * 8);
	mi->cnts = (uint32_t*)malloc((size_t)(nk ? nk : 1) * 4);
	for (i = 0; i < v->n; ++i) mi->P[i] = v->a[i].y;
	for (i = 0, off = 0, nk = 0; i < v->n; i = j) {
		uint64_t h = v->a[i].x, s;
		for (j = i + 1; j < v->n && v->a[j].x == h; ++j) {}
		s = tslot(h, mi->tmask);
		while (mi->tkey[s]) s = (s + 1) & mi->tmask;
		mi->tkey[s] = (uint32_t)h + 1;
		mi->tval[s] = (uint64_t)off << 32 | (uint64_t)(j - i);
		mi->cnts[nk++] = (uint32_t)(j - i);
		off = j;
	}
	if (v->n >= (1LL << 32)) { fprintf(stderr, "[oracle] index too large for 32-bit offsets\n"); abort(); }
	if (getenv("ORC_TIMING")) fprintf(stderr, "[oracle] sort + table %.2f s\n", omp_get_wtime() - t0);
	return mi;
}

orc_index *orc_index_build_mem(int n_seq, const char *const *names, const char *const *seqs,
                               const int *lens, int k, int w)
{
	orc_index *mi = (orc_index*)calloc(1, sizeof(orc_index));
	vec128 v = { 0, 0, 0 };
	int i;
	if (2 * k > 32) { free(mi); return 0; }              /* table keys are 32-bit here */
	mi->k = k, mi->w = w, mi->n_seq = n_seq;
	mi->name = (char**)calloc((size_t)(n_seq ? n_seq : 1), sizeof(char*));
	mi->len = (int*)calloc((size_t)(n_seq ? n_seq : 1), sizeof(int));
	mi->seq = (uint8_t**)calloc((size_t)(n_seq ? n_seq : 1), sizeof(uint8_t*));
	{
		/* contigs side by side, their minimizers joined in contig order (the order does not matter to
		 * the sort below; it keeps the build the same whatever the number of threads) */
		vec128 *per = (vec128*)calloc((size_t)(n_seq ? n_seq : 1), sizeof(vec128));
		int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1)
		for (i = 0; i < n_seq; ++i) {
			int j;
			mi->name[i] = strdup(names[i]);
			mi->len[i] = lens[i];
			mi->seq[i] = (uint8_t*)malloc((size_t)(lens[i] ? lens[i] : 1));
			for (j = 0; j < lens[i]; ++j) mi->seq[i][j] = orc_nt4((unsigned char)seqs[i][j]);
			sketch_core(seqs[i], lens[i], w, k, (uint32_t)i, &per[i]);
		}
		for (i = 0; i < n_seq; ++i) total += per[i].n;
		v.a = (orc128_t*)malloc((size_t)(total ? total : 1) * sizeof(orc128_t)), v.m = total;
		for (i = 0; i < n_seq; ++i) {
			if (per[i].n) memcpy(v.a + v.n, per[i].a, (size_t)per[i].n * sizeof(orc128_t));
			v.n += per[i].n;
			free(per[i].a);
		}
		free(per);
	}
	if (getenv("ORC_TIMING")) fprintf(stderr, "[oracle] sketch done\n");
	index_from_pairs(mi, &v);
	free(v.a);
	return mi;
}

/* minimal FASTA reader (plain or gz): name = header text up to first whitespace (A.8) */
orc_index *orc_index_build_fasta(const char *path, int k, int w)
{
	gzFile fp = gzopen(path, "rb");
	char *line, **names = 0, **seqs = 0;
	int *lens = 0, n = 0, m = 0;
	size_t cap = 0, *caps = 0;
	orc_index *mi;
	int i;
	const int LINE = 1 << 16;
	if (!fp) return 0;
	line = (char*)malloc((size_t)LINE);
	(void)cap;
	while (gzgets(fp, line, LINE)) {
		size_t L = strlen(line);
		while (L && (line[L-1] == '\n' || line[L-1] == '\r')) line[--L] = 0;
		if (line[0] == '>') {
			size_t e = 1;
			if (n == m) {
				m = m ? m << 1 : 16;
				names = (char**)realloc(names, (size_t)m * sizeof(char*));
				seqs = (char**)realloc(seqs, (size_t)m * sizeof(char*));
				lens = (int*)realloc(lens, (size_t)m * sizeof(int));
				caps = (size_t*)realloc(caps, (size_t)m * sizeof(size_t));
			}
			while (line[e] && line[e] != ' ' && line[e] != '\t') ++e;
			line[e] = 0;
			names[n] = strdup(line + 1);
			seqs[n] = 0, lens[n] = 0, caps[n] = 0;
			++n;
		} else if (n > 0 && L > 0) {
			int r = n - 1;
			if ((size_t)lens[r] + L + 1 > caps[r]) {
				caps[r] = caps[r] ? caps[r] << 1 : 1 << 20;
				while ((size_t)lens[r] + L + 1 > caps[r]) caps[r] <<= 1;
				seqs[r] = (char*)realloc(seqs[r], caps[r]);
			}
			memcpy(seqs[r] + lens[r], line, L);
			lens[r] += (int)L;
		}
	}
	gzclose(fp);
	free(line);
	for (i = 0; i < n; ++i) if (!seqs[i]) seqs[i] = (char*)calloc(1, 1);
	mi = orc_index_build_mem(n, (const char *const*)names, (const char *const*)seqs, lens, k, w);
	for (i = 0; i < n; ++i) free(names[i]), free(seqs[i]);
	free(names), free(seqs), free(lens), free(caps);
	return mi;
}

void orc_index_free(orc_index *mi)
{
	int i;
	if (!mi) return;
	for (i = 0; i < mi->n_seq; ++i) free(mi->name[i]), free(mi->seq[i]);
	free(mi->seq);
	free(mi->name), free(mi->len), free(mi->P), free(mi->tkey), free(mi->tval), free(mi->cnts);
	free(mi);
}

int orc_index_getseq(const orc_index *mi, uint32_t rid, uint32_t st, uint32_t en, uint8_t *seq)
{
	if (rid >= (uint32_t)mi->n_seq || st > (uint32_t)mi->len[rid]) return -1;
	if (en > (uint32_t)mi->len[rid]) en = (uint32_t)mi->len[rid];
	memcpy(seq, mi->seq[rid] + st, en - st);
	return (int)(en - st);
}

int orc_index_k(const orc_index *mi) { return mi->k; }
int orc_index_w(const orc_index *mi) { return mi->w; }
int orc_index_n_seq(const orc_index *mi) { return mi->n_seq; }
const char *orc_index_name(const orc_index *mi, int rid) { return mi->name[rid]; }
int orc_index_len(const orc_index *mi, int rid) { return mi->len[rid]; }
int64_t orc_index_n_minimizers(const orc_index *mi) { return mi->n_occ; }
int64_t orc_index_n_keys(const orc_index *mi) { return mi->n_keys; }

const uint64_t *orc_index_get(const orc_index *mi, uint64_t hash, int *n)
{
	uint64_t s = tslot(hash, mi->tmask);
	*n = 0;
	while (mi->tkey[s]) {
		if (mi->tkey[s] == (uint32_t)hash + 1) {
			*n = (int)(uint32_t)mi->tval[s];
			return &mi->P[mi->tval[s] >> 32];
		}
		s = (s + 1) & mi->tmask;
	}
	return 0;
}

int64_t orc_index_dump(const orc_index *mi, uint64_t *hash, uint64_t *y, int64_t cap)
{
	/* walk keys in ascending hash order: rebuild from table */
	int64_t n = 0;
	uint64_t s, *keys = (uint64_t*)malloc((size_t)(mi->n_keys ? mi->n_keys : 1) * 8);
	int64_t nk = 0, i;
	for (s = 0; s <= mi->tmask; ++s)
		if (mi->tkey[s]) keys[nk++] = (uint64_t)(mi->tkey[s] - 1);
	qsort(keys, (size_t)nk, 8, cmp64);
	for (i = 0; i < nk; ++i) {
		int c, j;
		const uint64_t *p = orc_index_get(mi, keys[i], &c);
		for (j = 0; j < c; ++j, ++n)
			if (n < cap) hash[n] = keys[i], y[n] = p[j];
	}
	free(keys);
	return n;
}

/* occurrence threshold (A.3): value at rank (1-f)*n_distinct among per-key counts, plus 1 */
int orc_index_cal_mid_occ(const orc_index *mi, float f)
{
	uint32_t *a, thres;
	int64_t n = mi->n_keys;
	size_t kth;
	if (f <= 0.) return INT32_MAX;
	if (n == 0) return 1;
	a = (uint32_t*)malloc((size_t)n * 4);
	memcpy(a, mi->cnts, (size_t)n * 4);
	qsort(a, (size_t)n, 4, cmp32);
	kth = (size_t)(uint32_t)((1. - f) * n);
	if (kth >= (size_t)n) kth = (size_t)n - 1;
	thres = a[kth] + 1;
	free(a);
	return (int)thres;
}

/* ------------------------------------------------------------------ seeds (A.4) */

int64_t orc_collect_seeds(const orc_index *mi, const orc_opt_t *opt, int mid_occ,
                          const char *seq, int qlen, orc128_t **a_out, int *rep_len)
{
	vec128 mv = { 0, 0, 0 }, a = { 0, 0, 0 };
	int rep_st = 0, rep_en = 0;
	int64_t i;
	(void)opt;
	*rep_len = 0;
	sketch_core(seq, qlen, mi->w, mi->k, 0, &mv);
	for (i = 0; i < mv.n; ++i) {
		const orc128_t *p = &mv.a[i];
		uint32_t q_pos = (uint32_t)p->y, q_span = (uint32_t)(p->x & 0xff);
		int t, kk, is_tandem = 0;
		const uint64_t *r = orc_index_get(mi, p->x >> 8, &t);
		if (t >= mid_occ) {                            /* too frequent: only feeds rep_len */
			int en = (int)(q_pos >> 1) + 1, st = en - (int)q_span;
			if (st > rep_en) {
				*rep_len += rep_en - rep_st;
				rep_st = st, rep_en = en;
			} else rep_en = en;
			continue;
		}
		if (t <= 0) continue;
		if (i > 0 && p->x >> 8 == mv.a[i-1].x >> 8) is_tandem = 1;
		if (i < mv.n - 1 && p->x >> 8 == mv.a[i+1].x >> 8) is_tandem = 1;
		for (kk = 0; kk < t; ++kk) {
			orc128_t e;
			uint32_t rpos = (uint32_t)r[kk] >> 1;
			if ((r[kk] & 1) == (q_pos & 1)) {          /* same strand */
				e.x = (r[kk] & 0xffffffff00000000ULL) | rpos;
				e.y = (uint64_t)q_span << 32 | q_pos >> 1;
			} else {                                   /* opposite strand */
				e.x = 1ULL << 63 | (r[kk] & 0xffffffff00000000ULL) | rpos;
				e.y = (uint64_t)q_span << 32 | (uint32_t)(qlen - ((int)(q_pos >> 1) + 1 - (int)q_span) - 1);
			}
			if (is_tandem) e.y |= SEED_TANDEM;
			vpush(&a, e);
		}
	}
	*rep_len += rep_en - rep_st;
	free(mv.a);
	/* TOTAL ORDER (strengthening of A.4's unstable radix sort by x): (x, y) */
	qsort(a.a, (size_t)a.n, sizeof(orc128_t), cmp128);
	*a_out = a.a;
	return a.n;
}

/* ------------------------------------------------------------------ chaining (A.5) */

static inline int ilog2_32(uint32_t v)
{
	int r = 0;
	while (v >>= 1) ++r;
	return r;
}

orc128_t *orc_chain_dp(const orc_opt_t *opt, int64_t n, orc128_t *a, int *n_u_, uint64_t **u_out,
                       int32_t *f_out, int32_t *p_out, int32_t *v_out)
{
	const int max_dist_x = opt->max_gap, max_dist_y = opt->max_gap, bw = opt->bw;
	const int max_skip = opt->max_chain_skip, max_iter = opt->max_chain_iter;
	const int min_cnt = opt->min_cnt, min_sc = opt->min_chain_score;
	int32_t k, *f, *p, *t, *v, n_u, n_v;
	int64_t i, j, st = 0;
	uint64_t *u, *u2, sum_qspan = 0;
	float avg_qspan;
	orc128_t *b, *w;

	*n_u_ = 0, *u_out = 0;
	if (n == 0 || a == 0) { free(a); return 0; }
	f = (int32_t*)malloc((size_t)n * 4);
	p = (int32_t*)malloc((size_t)n * 4);
	t = (int32_t*)calloc((size_t)n, 4);
	v = (int32_t*)malloc((size_t)n * 4);

	for (i = 0; i < n; ++i) sum_qspan += a[i].y >> 32 & 0xff;
	avg_qspan = (float)sum_qspan / n;

	for (i = 0; i < n; ++i) {                           /* fill score and back-pointer arrays */
		uint64_t ri = a[i].x;
		int64_t max_j = -1;
		int32_t qi = (int32_t)a[i].y, q_span = a[i].y >> 32 & 0xff;
		int32_t max_f = q_span, n_skip = 0, min_d;
		while (st < i && ri > a[st].x + (uint64_t)max_dist_x) ++st;
		if (i - st > max_iter) st = i - max_iter;
		for (j = i - 1; j >= st; --j) {
			int64_t dr = (int64_t)(ri - a[j].x);
			int32_t dq = qi - (int32_t)a[j].y, dd, sc, log_dd, gap_cost;
			if (dr == 0 || dq <= 0) continue;
			if (dq > max_dist_y || dq > max_dist_x) continue;
			dd = dr > dq ? (int32_t)(dr - dq) : (int32_t)(dq - dr);
			if (dd > bw) continue;
			min_d = dq < dr ? dq : (int32_t)dr;
			sc = min_d > q_span ? q_span : min_d;
			log_dd = dd ? ilog2_32((uint32_t)dd) : 0;
			gap_cost = (int)(dd * .01 * avg_qspan) + (log_dd >> 1);   /* double arithmetic */
			sc -= gap_cost;
			sc += f[j];
			if (sc > max_f) {
				max_f = sc, max_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == (int32_t)i) {
				if (++n_skip > max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = (int32_t)i;
		}
		f[i] = max_f, p[i] = (int32_t)max_j;
		v[i] = max_j >= 0 && v[max_j] > max_f ? v[max_j] : max_f;   /* peak score up to i */
	}
	if (f_out) memcpy(f_out, f, (size_t)n * 4);
	if (p_out) memcpy(p_out, p, (size_t)n * 4);
	if (v_out) memcpy(v_out, v, (size_t)n * 4);

	/* chain ends: anchors that are nobody's predecessor, with peak >= min_sc */
	memset(t, 0, (size_t)n * 4);
	for (i = 0; i < n; ++i) if (p[i] >= 0) t[p[i]] = 1;
	for (i = n_u = 0; i < n; ++i) if (t[i] == 0 && v[i] >= min_sc) ++n_u;
	if (n_u == 0) { free(a), free(f), free(p), free(t), free(v); return 0; }
	u = (uint64_t*)malloc((size_t)n_u * 8);
	for (i = n_u = 0; i < n; ++i) {
		if (t[i] == 0 && v[i] >= min_sc) {
			j = i;
			while (j >= 0 && f[j] < v[j]) j = p[j];     /* walk back to the peak */
			if (j < 0) j = i;
			u[n_u++] = (uint64_t)f[j] << 32 | (uint64_t)j;
		}
	}
	/* keys (score<<32|index) are distinct, so this order is total: score desc, index desc */
	qsort(u, (size_t)n_u, 8, cmp64);
	for (i = 0; i < n_u >> 1; ++i) { uint64_t x = u[i]; u[i] = u[n_u-i-1], u[n_u-i-1] = x; }

	/* backtrack, best first; stop at anchors already used */
	memset(t, 0, (size_t)n * 4);
	for (i = n_v = k = 0; i < n_u; ++i) {
		int32_t n_v0 = n_v, k0 = k;
		j = (int32_t)u[i];
		do {
			v[n_v++] = (int32_t)j;
			t[j] = 1;
			j = p[j];
		} while (j >= 0 && t[j] == 0);
		if (j < 0) {
			if (n_v - n_v0 >= min_cnt) u[k++] = u[i] >> 32 << 32 | (uint64_t)(n_v - n_v0);
		} else if ((int32_t)(u[i] >> 32) - f[j] >= min_sc) {
			if (n_v - n_v0 >= min_cnt) u[k++] = ((u[i] >> 32) - (uint64_t)f[j]) << 32 | (uint64_t)(n_v - n_v0);
		}
		if (k0 == k) n_v = n_v0;                        /* nothing added: roll the list back */
	}
	n_u = k;
	free(f), free(p), free(t);
	if (n_u == 0) { free(a), free(v), free(u); return 0; }

	b = (orc128_t*)malloc((size_t)(n_v ? n_v : 1) * sizeof(orc128_t));
	for (i = 0, k = 0; i < n_u; ++i) {
		int32_t k0 = k, ni = (int32_t)u[i];
		for (j = 0; j < ni; ++j) b[k] = a[v[k0 + (ni - j - 1)]], ++k;
	}
	free(v);

	/* order chains by their first anchor.  TOTAL ORDER (strengthening of the unstable
	 * radix sort by x): (x of first anchor, then k<<32|i), which is unique. */
	w = (orc128_t*)malloc((size_t)n_u * sizeof(orc128_t));
	for (i = k = 0; i < n_u; ++i) {
		w[i].x = b[k].x, w[i].y = (uint64_t)k << 32 | (uint64_t)i;
		k += (int32_t)u[i];
	}
	qsort(w, (size_t)n_u, sizeof(orc128_t), cmp128);
	u2 = (uint64_t*)malloc((size_t)n_u * 8);
	for (i = k = 0; i < n_u; ++i) {
		int32_t jj = (int32_t)w[i].y, nn = (int32_t)u[jj];
		u2[i] = u[jj];
		memcpy(&a[k], &b[w[i].y >> 32], (size_t)nn * sizeof(orc128_t));
		k += nn;
	}
	free(u), free(b), free(w);
	*n_u_ = n_u, *u_out = u2;
	return a;
}

/* ------------------------------------------------------------------ regions (A.6) */

static inline uint64_t hash64_full(uint64_t key)
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

static inline uint32_t wang_hash32(uint32_t key)
{
	key += ~(key << 15);
	key ^=  (key >> 10);
	key +=  (key << 3);
	key ^=  (key >> 6);
	key += ~(key << 11);
	key ^=  (key >> 16);
	return key;
}

void orc_reg_set_coor(orc_reg_t *r, int32_t qlen, const orc128_t *a)
{
	int32_t k = r->as, q_span = (int32_t)(a[k].y >> 32 & 0xff), i;
	r->rev = (int32_t)(a[k].x >> 63);
	r->rid = (int32_t)(a[k].x << 1 >> 33);
	r->rs = (int32_t)a[k].x + 1 > q_span ? (int32_t)a[k].x + 1 - q_span : 0;
	r->re = (int32_t)a[k + r->cnt - 1].x + 1;
	if (!r->rev) {
		r->qs = (int32_t)a[k].y + 1 - q_span;
		r->qe = (int32_t)a[k + r->cnt - 1].y + 1;
	} else {
		r->qs = qlen - ((int32_t)a[k + r->cnt - 1].y + 1);
		r->qe = qlen - ((int32_t)a[k].y + 1 - q_span);
	}
	/* chain-level match / block length */
	r->mlen = r->blen = 0;
	if (r->cnt <= 0) return;
	r->mlen = r->blen = q_span;
	for (i = r->as + 1; i < r->as + r->cnt; ++i) {
		int span = (int)(a[i].y >> 32 & 0xff);
		int tl = (int32_t)a[i].x - (int32_t)a[i-1].x;
		int ql = (int32_t)a[i].y - (int32_t)a[i-1].y;
		r->blen += tl > ql ? tl : ql;
		r->mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
	}
}

static orc_reg_t *gen_regs(uint32_t hash, int qlen, int n_u, const uint64_t *u, const orc128_t *a)
{
	orc128_t *z;
	orc_reg_t *r;
	int i, k;
	if (n_u == 0) return 0;
	z = (orc128_t*)malloc((size_t)n_u * sizeof(orc128_t));
	for (i = k = 0; i < n_u; ++i) {
		uint32_t h = (uint32_t)hash64_full((hash64_full(a[k].x) + hash64_full(a[k].y)) ^ hash);
		z[i].x = u[i] ^ h;            /* score<<32 | (cnt ^ pseudo-random) */
		z[i].y = (uint64_t)k << 32 | (uint32_t)(int32_t)u[i];
		k += (int32_t)u[i];
	}
	/* TOTAL ORDER: (x, y) ascending then reversed (upstream: unstable by x only) */
	qsort(z, (size_t)n_u, sizeof(orc128_t), cmp128);
	for (i = 0; i < n_u >> 1; ++i) { orc128_t x = z[i]; z[i] = z[n_u-1-i], z[n_u-1-i] = x; }
	r = (orc_reg_t*)calloc((size_t)n_u, sizeof(orc_reg_t));
	for (i = 0; i < n_u; ++i) {
		orc_reg_t *ri = &r[i];
		ri->id = i;
		ri->parent = PARENT_UNSET;
		ri->score = ri->score0 = (int32_t)(z[i].x >> 32);
		ri->hash = (uint32_t)z[i].x;
		ri->cnt = (int32_t)z[i].y;
		ri->as = (int32_t)(z[i].y >> 32);
		orc_reg_set_coor(ri, qlen, a);
	}
	free(z);
	return r;
}

void orc_sync_regs(int n_regs, orc_reg_t *regs)
{
	int *tmp, i, max_id = -1, n_tmp;
	if (n_regs <= 0) return;
	for (i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id ? max_id : regs[i].id;
	n_tmp = max_id + 1;
	tmp = (int*)malloc((size_t)(n_tmp ? n_tmp : 1) * sizeof(int));
	for (i = 0; i < n_tmp; ++i) tmp[i] = -1;
	for (i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
	for (i = 0; i < n_regs; ++i) {
		orc_reg_t *r = &regs[i];
		r->id = i;
		if (r->parent == PARENT_TMP_PRI) r->parent = i;
		else if (r->parent >= 0 && tmp[r->parent] >= 0) r->parent = tmp[r->parent];
		else r->parent = PARENT_UNSET;
	}
	free(tmp);
}

/* parent / secondary assignment and subsc, n_sub (A.6); float32 ratio test.  Runs twice when
 * base-level alignment is on (chain_post, then align_regs): on the second pass both regions
 * carry a DP result, dp_max2 of the parent is raised and a near-equal DP score also counts as
 * a sub-optimal hit (sub_diff = a * 2 + b). */
void orc_set_parent(float mask_level, int n, orc_reg_t *r, int sub_diff)
{
	int i, j, k, *w;
	uint64_t *cov;
	if (n <= 0) return;
	for (i = 0; i < n; ++i) r[i].id = i;
	cov = (uint64_t*)malloc((size_t)n * 8);
	w = (int*)malloc((size_t)n * sizeof(int));
	w[0] = 0, r[0].parent = 0;
	for (i = 1, k = 1; i < n; ++i) {
		orc_reg_t *ri = &r[i];
		int si = ri->qs, ei = ri->qe, n_cov = 0, uncov_len = 0;
		for (j = 0; j < k; ++j) {                       /* overlapping primaries */
			orc_reg_t *rp = &r[w[j]];
			int sj = rp->qs, ej = rp->qe;
			if (ej <= si || sj >= ei) continue;
			if (sj < si) sj = si;
			if (ej > ei) ej = ei;
			cov[n_cov++] = (uint64_t)sj << 32 | (uint32_t)ej;
		}
		if (n_cov == 0) { j = k; goto set_parent_test; }
		{
			int jj, x = si;
			qsort(cov, (size_t)n_cov, 8, cmp64);
			for (jj = 0; jj < n_cov; ++jj) {
				if ((int)(cov[jj] >> 32) > x) uncov_len += (int)(cov[jj] >> 32) - x;
				x = (int32_t)cov[jj] > x ? (int32_t)cov[jj] : x;
			}
			if (ei > x) uncov_len += ei - x;
		}
		for (j = 0; j < k; ++j) {
			orc_reg_t *rp = &r[w[j]];
			int sj = rp->qs, ej = rp->qe, min, max, ol;
			if (ej <= si || sj >= ei) continue;
			min = ej - sj < ei - si ? ej - sj : ei - si;
			max = ej - sj > ei - si ? ej - sj : ei - si;
			ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj)
			             : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
			if ((float)ol / min - (float)uncov_len / max > mask_level) {
				int cnt_sub = 0;
				ri->parent = rp->parent;
				rp->subsc = rp->subsc > ri->score ? rp->subsc : ri->score;
				if (ri->cnt >= rp->cnt) cnt_sub = 1;
				if ((rp->flags & ORC_REG_HAS_DP) && (ri->flags & ORC_REG_HAS_DP) &&
				    (rp->rid != ri->rid || rp->rs != ri->rs || rp->re != ri->re || ol != min)) {   /* not the same hit twice */
					int sc = ri->dp_max;
					rp->dp_max2 = rp->dp_max2 > sc ? rp->dp_max2 : sc;
					if (rp->dp_max - ri->dp_max <= sub_diff) cnt_sub = 1;
				}
				if (cnt_sub) ++rp->n_sub;
				break;
			}
		}
set_parent_test:
		if (j == k) w[k++] = i, ri->parent = i, ri->n_sub = 0;
	}
	free(cov), free(w);
}

/* keep primaries and the best secondaries (A.6).  The in-place compaction reads r[p]
 * AFTER earlier slots may have been overwritten, exactly as upstream does. */
void orc_select_sub(float pri_ratio, int min_diff, int best_n, int *n_, orc_reg_t *r)
{
	if (pri_ratio > 0.0f && *n_ > 0) {
		int i, k, n = *n_, n_2nd = 0;
		for (i = k = 0; i < n; ++i) {
			int p = r[i].parent;
			if (p == i || (r[i].flags & ORC_REG_INV)) {         /* primary or inversion */
				r[k++] = r[i];
			} else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
				if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re))
					r[k++] = r[i], ++n_2nd;
			}
		}
		if (k != n) orc_sync_regs(k, r);
		*n_ = k;
	}
}

int orc_squeeze_a(int n_regs, orc_reg_t *regs, orc128_t *a)
{
	int i, as = 0;
	uint64_t *aux = (uint64_t*)malloc((size_t)(n_regs ? n_regs : 1) * 8);
	for (i = 0; i < n_regs; ++i) aux[i] = (uint64_t)regs[i].as << 32 | (uint32_t)i;
	qsort(aux, (size_t)n_regs, 8, cmp64);
	for (i = 0; i < n_regs; ++i) {
		orc_reg_t *r = &regs[(int32_t)aux[i]];
		if (r->as != as) {
			memmove(&a[as], &a[r->as], (size_t)r->cnt * 16);
			r->as = as;
		}
		as += r->cnt;
	}
	free(aux);
	return as;
}

void orc_filter_regs(const orc_opt_t *opt, int qlen, int *n_regs, orc_reg_t *regs)
{
	int i, k;
	for (i = k = 0; i < *n_regs; ++i) {
		const orc_reg_t *r = &regs[i];
		int flt = 0;
		if (!(r->flags & ORC_REG_INV) && r->cnt < opt->min_cnt) flt = 1;
		if (r->flags & ORC_REG_HAS_DP) {                    /* only with a base-level alignment */
			if (r->mlen < opt->min_chain_score) flt = 1;
			else if (r->dp_max < opt->min_dp_max) flt = 1;
			else if (r->qs > qlen * opt->max_clip_ratio && qlen - r->qe > qlen * opt->max_clip_ratio) flt = 1;
		}
		if (flt) continue;
		if (k < i) regs[k++] = regs[i]; else ++k;
	}
	*n_regs = k;
}

/* fuse adjacent co-linear primary chains separated by a long gap (A.6 "long-join") */
static void join_long(const orc_opt_t *opt, int qlen, int *n_regs_, orc_reg_t *regs, orc128_t *a)
{
	int i, n_aux, n_regs = *n_regs_, n_drop = 0;
	uint64_t *aux;
	if (n_regs < 2) return;
	orc_squeeze_a(n_regs, regs, a);
	aux = (uint64_t*)malloc((size_t)n_regs * 8);
	for (i = n_aux = 0; i < n_regs; ++i)
		if (regs[i].parent == i || regs[i].parent < 0)
			aux[n_aux++] = (uint64_t)regs[i].as << 32 | (uint32_t)i;
	qsort(aux, (size_t)n_aux, 8, cmp64);
	for (i = n_aux - 1; i >= 1; --i) {
		orc_reg_t *r0 = &regs[(int32_t)aux[i-1]], *r1 = &regs[(int32_t)aux[i]];
		const orc128_t *a0e, *a1s;
		int max_gap, min_gap, sc_thres, min_flank_len;
		int64_t dx;
		if (r0->as + r0->cnt != r1->as) continue;          /* not adjacent in a[] */
		if (r0->rid != r1->rid || r0->rev != r1->rev) continue;
		a0e = &a[r0->as + r0->cnt - 1];
		a1s = &a[r1->as];
		if (a1s->x <= a0e->x || (int32_t)a1s->y <= (int32_t)a0e->y) continue;   /* co-linear */
		dx = (int64_t)(a1s->x - a0e->x);
		max_gap = min_gap = (int32_t)a1s->y - (int32_t)a0e->y;
		max_gap = max_gap > dx ? max_gap : (int)dx;
		min_gap = min_gap < dx ? min_gap : (int)dx;
		if (max_gap > opt->max_join_long || min_gap > opt->max_join_short) continue;
		sc_thres = (int)((float)opt->min_join_flank_sc / opt->max_join_long * max_gap + .499);
		if (r0->score < sc_thres || r1->score < sc_thres) continue;
		min_flank_len = (int)(max_gap * opt->min_join_flank_ratio);
		if (r0->re - r0->rs < min_flank_len || r0->qe - r0->qs < min_flank_len) continue;
		if (r1->re - r1->rs < min_flank_len || r1->qe - r1->qs < min_flank_len) continue;
		a[r1->as].y |= SEED_LONG_JOIN;
		r0->cnt += r1->cnt, r0->score += r1->score;
		orc_reg_set_coor(r0, qlen, a);
		r1->cnt = 0;
		r1->parent = r0->id;
		++n_drop;
	}
	free(aux);
	if (n_drop > 0) {
		for (i = 0; i < n_regs; ++i) {                      /* re-point secondaries */
			orc_reg_t *r = &regs[i];
			if (r->parent >= 0 && r->id != r->parent)
				if (regs[r->parent].parent >= 0 && regs[r->parent].parent != r->parent)
					r->parent = regs[r->parent].parent;
		}
		orc_filter_regs(opt, qlen, n_regs_, regs);
		orc_sync_regs(*n_regs_, regs);
	}
}

/* ------------------------------------------------------------------ MAPQ (A.7) */

static void set_mapq(int n_regs, orc_reg_t *regs, int min_chain_sc, int match_sc, int rep_len)
{
	static const float q_coef = 40.0f;
	int64_t sum_sc = 0;
	float uniq_ratio;
	int i;
	if (n_regs == 0) return;
	for (i = 0; i < n_regs; ++i)
		if (regs[i].parent == regs[i].id) sum_sc += regs[i].score;
	uniq_ratio = (float)sum_sc / (sum_sc + rep_len);
	for (i = 0; i < n_regs; ++i) {
		orc_reg_t *r = &regs[i];
		if (r->flags & ORC_REG_INV) {
			r->mapq = 0;
		} else if (r->parent == r->id) {
			const int has_dp = r->flags & ORC_REG_HAS_DP;
			int mapq, subsc;
			float pen_s1 = (r->score > 100 ? 1.0f : 0.01f * r->score) * uniq_ratio;
			float pen_cm = r->cnt > 10 ? 1.0f : 0.1f * r->cnt;
			pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
			subsc = r->subsc > min_chain_sc ? r->subsc : min_chain_sc;
			if (has_dp && r->dp_max2 > 0 && r->dp_max > 0) {
				float identity = (float)r->mlen / r->blen;
				float x = (float)r->dp_max2 * subsc / r->dp_max / r->score0;
				int mapq_alt;
				mapq = (int)(identity * pen_cm * q_coef * (1.0f - x * x) * logf((float)r->dp_max / match_sc));
				mapq_alt = (int)(6.02f * identity * identity * (r->dp_max - r->dp_max2) / match_sc + .499f);
				mapq = mapq < mapq_alt ? mapq : mapq_alt;
			} else {
				float x = (float)subsc / r->score0;
				if (has_dp) {
					float identity = (float)r->mlen / r->blen;
					mapq = (int)(identity * pen_cm * q_coef * (1.0f - x) * logf((float)r->dp_max / match_sc));
				} else {
					mapq = (int)(pen_cm * q_coef * (1.0f - x) * logf(r->score));
				}
			}
			mapq -= (int)(4.343f * logf(r->n_sub + 1) + .499f);
			mapq = mapq > 0 ? mapq : 0;
			r->mapq = mapq < 60 ? mapq : 60;
			if (has_dp && r->dp_max > r->dp_max2 && r->mapq == 0) r->mapq = 1;
		} else r->mapq = 0;
	}
}

/* ------------------------------------------------------------------ one read (aligner.py:193,215) */

static int map_core(const orc_index *mi, const orc_opt_t *opt, int mid_occ, const char *seq, int qlen,
                    orc_reg_t *regs_out, int cap, uint32_t *cig_out, int cig_cap, int *cig_total);

int orc_map(const orc_index *mi, const orc_opt_t *opt, int mid_occ, const char *seq, int qlen,
            orc_reg_t *regs_out, int cap)
{
	return map_core(mi, opt, mid_occ, seq, qlen, regs_out, cap, 0, 0, 0);
}

int orc_map_cigar(const orc_index *mi, const orc_opt_t *opt, int mid_occ, const char *seq, int qlen,
                  orc_reg_t *regs_out, int cap, uint32_t *cig_out, int cig_cap, int *cig_total)
{
	return map_core(mi, opt, mid_occ, seq, qlen, regs_out, cap, cig_out, cig_cap, cig_total);
}

static int map_core(const orc_index *mi, const orc_opt_t *opt, int mid_occ, const char *seq, int qlen,
                    orc_reg_t *regs_out, int cap, uint32_t *cig_out, int cig_cap, int *cig_total)
{
	uint32_t **cigs = 0;
	orc128_t *a;
	uint64_t *u = 0;
	int rep_len = 0, n_u = 0, n_regs, i;
	int64_t n_a;
	uint32_t hash;
	orc_reg_t *regs;

	if (qlen <= 0) return 0;
	hash = 0;
	hash ^= wang_hash32((uint32_t)qlen) + wang_hash32((uint32_t)opt->seed);
	hash = wang_hash32(hash);

	n_a = orc_collect_seeds(mi, opt, mid_occ, seq, qlen, &a, &rep_len);
	a = orc_chain_dp(opt, n_a, a, &n_u, &u, 0, 0, 0);
	if (a == 0 || n_u == 0) { free(a), free(u); return 0; }
	regs = gen_regs(hash, qlen, n_u, u, a);
	n_regs = n_u;
	orc_set_parent(opt->mask_level, n_regs, regs, opt->a * 2 + opt->b);
	orc_select_sub(opt->pri_ratio, mi->k * 2, opt->best_n, &n_regs, regs);
	join_long(opt, qlen, &n_regs, regs, a);
	if (opt->cigar)                                         /* mappy: MM_F_CIGAR is always set */
		regs = orc_align_regs(mi, opt, qlen, seq, &n_regs, regs, a, cig_total ? &cigs : 0);
	set_mapq(n_regs, regs, opt->min_chain_score, opt->a, rep_len);
	for (i = 0; i < n_regs && i < cap; ++i) regs_out[i] = regs[i];
	if (cig_total) {
		int tot = 0, k;
		for (i = 0; i < n_regs; ++i) {
			for (k = 0; k < regs[i].n_cigar; ++k, ++tot)
				if (cigs && tot < cig_cap) cig_out[tot] = cigs[i][k];
			if (cigs) free(cigs[i]);
		}
		free(cigs);
		*cig_total = tot;
	}
	free(regs), free(a), free(u);
	return n_regs;
}

/* ------------------------------------------------------------------ monica layer */

/* aligner.py:328-339.  Python float semantics: float(NM)/mlen in binary64, `<=`,
 * distance = best - new at the LAST update; falsy distance (0.0) => "0" (ambiguous). */
static int best_hit_arg(const orc_hit_t *hits, int n, int *arg)
{
	double best = INFINITY, distance = 0.0;
	int best_i = -1, i;
	for (i = 0; i < n; ++i) {
		double inverse_identity = (double)hits[i].nm / (double)hits[i].mlen;
		if (inverse_identity <= best) {
			distance = best - inverse_identity;
			best = inverse_identity, best_i = i;
		}
	}
	if (arg) *arg = best_i;                  /* the last hit that set the minimum */
	if (distance == 0.0) return -1;          /* `if not distance` (NaN cannot arise: mlen > 0) */
	return best_i;
}

int orc_best_hit(const orc_hit_t *hits, int n) { return best_hit_arg(hits, n, 0); }

/* aligner.py:212-233 for a single index part: gate, then single hit | best_hit | ambiguous */
static int classify_one(const orc_index *mi, const orc_opt_t *opt, int mid_occ, const char *seq,
                        int qlen, int min_mapq, orc_hit_t *hits, int hits_cap, int *assign,
                        orc_hit_t *chosen)
{
	orc_reg_t stack_regs[64], *regs = stack_regs;
	int n_regs, i, n_h = 0, cap = 64;
	n_regs = orc_map(mi, opt, mid_occ, seq, qlen, regs, cap);
	if (n_regs > cap) {
		regs = (orc_reg_t*)malloc((size_t)n_regs * sizeof(orc_reg_t));
		cap = n_regs;
		n_regs = orc_map(mi, opt, mid_occ, seq, qlen, regs, cap);
	}
	for (i = 0; i < n_regs; ++i) {
		if (regs[i].id == regs[i].parent && regs[i].mapq >= min_mapq) {
			if (n_h < hits_cap) {
				hits[n_h].rid = regs[i].rid;
				hits[n_h].mapq = regs[i].mapq;
				hits[n_h].nm = regs[i].blen - regs[i].mlen + regs[i].n_ambi;   /* hit.NM (A.8) */
				hits[n_h].mlen = regs[i].mlen;
			}
			++n_h;
		}
	}
	if (regs != stack_regs) free(regs);
	memset(chosen, 0, sizeof(*chosen));
	if (n_h == 0) *assign = ORC_UNMAPPED;
	else if (n_h > hits_cap) *assign = ORC_AMBIGUOUS;          /* caller re-runs with more room */
	else if (n_h == 1) *assign = hits[0].rid, *chosen = hits[0];
	else {
		int arg = -1, b = best_hit_arg(hits, n_h, &arg);
		if (b < 0) *assign = ORC_AMBIGUOUS, *chosen = hits[arg];   /* the tied minimum, for cross-shard merges */
		else *assign = hits[b].rid, *chosen = hits[b];
	}
	return n_h;
}

int64_t orc_classify_batch(const orc_index *mi, const orc_opt_t *opt, int mid_occ,
                           const char *bases, const int64_t *offsets, int n_reads,
                           int min_mapq, int n_threads,
                           int32_t *out_assign, orc_hit_t *out_hit, int32_t *out_nhits,
                           orc_hit_t *hits_flat, int64_t hits_cap)
{
	int r;
	int64_t total = 0, off = 0;
	enum { HCAP = 32 };
	orc_hit_t *per = (orc_hit_t*)malloc((size_t)(n_reads ? n_reads : 1) * HCAP * sizeof(orc_hit_t));
	(void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads > 0 ? n_threads : 1)
#endif
	for (r = 0; r < n_reads; ++r) {
		int assign;
		orc_hit_t chosen;
		int n_h = classify_one(mi, opt, mid_occ, bases + offsets[r], (int)(offsets[r+1] - offsets[r]),
		                       min_mapq, per + (size_t)r * HCAP, HCAP, &assign, &chosen);
		if (n_h > HCAP) { fprintf(stderr, "[oracle] more than %d gated hits for one read\n", HCAP); abort(); }
		out_assign[r] = assign;
		if (out_hit) out_hit[r] = chosen;
		out_nhits[r] = n_h;
	}
	for (r = 0; r < n_reads; ++r) {
		int n_h = out_nhits[r], i;
		for (i = 0; i < n_h; ++i, ++off)
			if (hits_flat && off < hits_cap) hits_flat[off] = per[(size_t)r * HCAP + i];
		total += n_h;
	}
	free(per);
	return total;
}
