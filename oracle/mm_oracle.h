/*
 * mm_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Scalar C restatement of the read-classification hot path that monica's
 * aligner delegates to mappy==2.17 / minimap2 v2.17 (requirements.txt:3;
 * call sites monica/genomes/aligner.py:45,59,193,215), at the chain-level
 * contract described in SURVEY.md section 7 (hard part 1) and Appendix A:
 *
 *   sketch (k=15,w=10) -> index get + mid_occ filter -> anchor sort ->
 *   chain DP + backtrack -> regions -> parent/secondary -> select ->
 *   long-join -> chain-level MAPQ   (no ksw2 base-level DP)
 *
 * plus the Python-level semantics of monica/genomes/aligner.py:194-263
 * (gate, best_hit, taxon counts).
 *
 * PARITY UNPINNED: minimap2/mappy is an un-vendored third-party dependency
 * that is absent from /root/reference and from this image, and the
 * reference's own tests (test/test_aligner.py) hold no assertions or golden
 * vectors.  This file restates the published algorithm from SURVEY.md
 * Appendix A; it could not be checked against the real library here.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this code.  The product (monica_amd/) never links or calls it.
 */
#ifndef MM_ORACLE_H
#define MM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t x, y; } orc128_t;

/* one mapping region (minimap2's mm_reg1_t plus the fields of its mm_extra_t that the
 * pipeline reads; the base-level fields stay 0 when base-level alignment is off) */
typedef struct {
	int32_t id, parent;      /* is_primary <=> id == parent (Appendix A.8)          */
	int32_t rid;             /* contig index                                          */
	int32_t rev;             /* 0 forward, 1 reverse                                  */
	int32_t rs, re, qs, qe;  /* reference / query interval                            */
	int32_t score, score0;   /* chain score (after join) and original chain score     */
	int32_t cnt;             /* number of anchors                                     */
	int32_t as;              /* offset of first anchor in the chained anchor array    */
	int32_t mlen, blen;      /* chain-level match / block length (A.6)                */
	int32_t subsc, n_sub;    /* best secondary score, number of sub-optimal hits      */
	int32_t mapq;            /* MAPQ (A.7)                                            */
	uint32_t hash;           /* tie-break hash used in the region sort (A.6)          */
	/* ---- base-level alignment (A.6b) */
	int32_t dp_score, dp_max, dp_max2;
	int32_t n_ambi;          /* NM = blen - mlen + n_ambi (A.8)                       */
	int32_t n_cigar;
	int32_t flags;           /* ORC_REG_* bits                                        */
} orc_reg_t;
#define ORC_REG_HAS_DP     1   /* a CIGAR exists (r->p != 0)                  */
#define ORC_REG_SPLIT_L    2   /* mm_reg1_t::split & 1                         */
#define ORC_REG_SPLIT_R    4   /* mm_reg1_t::split & 2                         */
#define ORC_REG_SPLIT_INV  8   /* mm_reg1_t::split_inv                         */
#define ORC_REG_INV        16  /* mm_reg1_t::inv: the alignment of an inversion between the halves of a split region */

typedef struct orc_index orc_index;

/* ---- map options (defaults = mappy 2.17 loading an index with no preset; A.1) ---- */
typedef struct {
	int seed;                /* 11 */
	float mid_occ_frac;      /* 2e-4f */
	int mid_occ;             /* <=0: derive from the index (A.3) */
	int min_cnt;             /* 3 */
	int min_chain_score;     /* 40 */
	int bw;                  /* 500 */
	int max_gap;             /* 5000 */
	int max_chain_skip;      /* 25 */
	int max_chain_iter;      /* 5000 */
	float mask_level;        /* 0.5f */
	float pri_ratio;         /* 0.8f */
	int best_n;              /* 5 */
	int max_join_long;       /* 20000 */
	int max_join_short;      /* 2000 */
	int min_join_flank_sc;   /* 1000 */
	float min_join_flank_ratio; /* 0.5f */
	int a, b;                /* 2, 4 */
	/* ---- base-level alignment (mappy always sets MM_F_CIGAR; A.1) */
	int cigar;               /* 1: run it (what mappy does); 0: chain-level contract   */
	int q, e, q2, e2;        /* 4, 2, 24, 1 */
	int sc_ambi;             /* 1 */
	int zdrop, zdrop_inv;    /* 400, 200 */
	int end_bonus;           /* -1 */
	int min_dp_max;          /* min_chain_score * a = 80 */
	int min_ksw_len;         /* 200 */
	float max_clip_ratio;    /* 1.0f */
	int64_t max_sw_mat;      /* 100000000 */
} orc_opt_t;

void orc_opt_init(orc_opt_t *o);

/* ---- A.2 sketch ---- */
uint64_t orc_hash64(uint64_t key, uint64_t mask);
/* returns number of minimizers; writes at most cap of them */
int orc_sketch(const char *seq, int len, int w, int k, uint32_t rid, orc128_t *out, int cap);

/* ---- A.3 index ---- */
orc_index *orc_index_build_mem(int n_seq, const char *const *names, const char *const *seqs,
                               const int *lens, int k, int w);
orc_index *orc_index_build_fasta(const char *path, int k, int w);   /* plain or gzipped FASTA */
void orc_index_free(orc_index *mi);
int orc_index_k(const orc_index *mi);
int orc_index_w(const orc_index *mi);
int orc_index_n_seq(const orc_index *mi);
const char *orc_index_name(const orc_index *mi, int rid);
int orc_index_len(const orc_index *mi, int rid);
int64_t orc_index_n_minimizers(const orc_index *mi);   /* occurrences */
int64_t orc_index_n_keys(const orc_index *mi);         /* distinct hashes */
int orc_index_cal_mid_occ(const orc_index *mi, float f);
/* look one minimizer hash up: returns pointer to the ascending y-list, *n = count */
const uint64_t *orc_index_get(const orc_index *mi, uint64_t hash, int *n);
/* dump the whole index as parallel arrays sorted by (hash, y) */
int64_t orc_index_dump(const orc_index *mi, uint64_t *hash, uint64_t *y, int64_t cap);

/* ---- A.4 seeds: returns n anchors sorted by (x, y); *rep_len per A.4 ---- */
int64_t orc_collect_seeds(const orc_index *mi, const orc_opt_t *opt, int mid_occ,
                          const char *seq, int qlen, orc128_t **a_out, int *rep_len);

/* ---- A.5 chaining. a[] (n anchors, sorted) is consumed; returns the chained/reordered
 * anchor array (malloc'd) and *u_out (malloc'd, score<<32|cnt), *n_u.
 * Optional f_out/p_out/v_out (length n, caller-allocated) receive the DP arrays. ---- */
orc128_t *orc_chain_dp(const orc_opt_t *opt, int64_t n, orc128_t *a, int *n_u, uint64_t **u_out,
                       int32_t *f_out, int32_t *p_out, int32_t *v_out);

/* ---- A.6b base-level alignment: ksw2's two-piece affine kernel ---- */
#define ORC_EZ_SCORE_ONLY  0x01
#define ORC_EZ_RIGHT       0x02   /* right-align gaps */
#define ORC_EZ_GENERIC_SC  0x04
#define ORC_EZ_APPROX_MAX  0x08
#define ORC_EZ_APPROX_DROP 0x10
#define ORC_EZ_EXTZ_ONLY   0x40   /* extension: stop at the maximum or the query end */
#define ORC_EZ_REV_CIGAR   0x80
typedef struct {
	uint32_t max;
	int zdropped;
	int max_q, max_t;        /* cell of the maximum */
	int mqe, mqe_t;          /* best score reaching the end of the query */
	int mte, mte_q;          /* best score reaching the end of the target */
	int score;               /* global alignment score */
	int m_cigar, n_cigar;
	int reach_end;
	uint32_t *cigar;         /* len<<4 | op, op 0 M, 1 I, 2 D */
} orc_extz_t;
void orc_extz_reset(orc_extz_t *ez);
void orc_gen_simple_mat(int m, int8_t *mat, int8_t a, int8_t b, int8_t sc_ambi);
/* literal simulation of ksw_extd2_sse (see mm_ksw.c); sequences are codes 0..4 */
void orc_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, orc_extz_t *ez);
/* the same recurrence in absolute scores over the unbanded matrix (cross-check) */
void orc_dp_clean(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat,
                  int q, int e, int q2, int e2, int zdrop, int end_bonus, int flag, orc_extz_t *ez);
int orc_local_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int q, int e);
/* ksw_ll_i16 literally (striped int16 Smith-Waterman: score, end of the query, end of the target with its tie
 * rules) and the same from the plain recurrence; see mm_ksw.c */
int orc_ksw_ll_i16(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                   int gapo, int gape, int *qe, int *te);
int orc_local_end(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int q, int e, int *qe, int *te);

/* test hook for the seed filters of mm_align1 (mm_align.c) */
void orc_test_seed_filters(const orc_opt_t *opt, int n, orc128_t *a, int mlen, int32_t *as1, int32_t *cnt1);

/* base-level alignment of all regions of one read (mm_align_skeleton + the second hierarchy
 * pass of align_regs); regs is realloc'd when a Z-drop splits a region.  a[] are the chained
 * anchors; seq the read (ASCII). */
orc_reg_t *orc_align_regs(const orc_index *mi, const orc_opt_t *opt, int qlen, const char *seq,
                          int *n_regs, orc_reg_t *regs, orc128_t *a, uint32_t ***cigars_out);
/* contig bases as codes 0..4 (mm_idx_getseq) */
int orc_index_getseq(const orc_index *mi, uint32_t rid, uint32_t st, uint32_t en, uint8_t *seq);

/* ---- full per-read map: returns number of regions (all kept regions, primary and
 * secondary), writes at most cap ---- */
int orc_map(const orc_index *mi, const orc_opt_t *opt, int mid_occ, const char *seq, int qlen,
            orc_reg_t *regs, int cap);

/* same, also returning the CIGARs of the regions back to back (regs[i].n_cigar words each) */
int orc_map_cigar(const orc_index *mi, const orc_opt_t *opt, int mid_occ, const char *seq, int qlen,
                  orc_reg_t *regs, int cap, uint32_t *cig_out, int cig_cap, int *cig_total);

/* ---- monica layer (aligner.py:194-263, 328-339) ---- */
/* per-read decision codes */
#define ORC_UNMAPPED  (-1)
#define ORC_AMBIGUOUS (-2)

/* hits of one read that pass `is_primary and mapq >= min_mapq` */
typedef struct { int32_t rid, mapq, nm, mlen; } orc_hit_t;

/* best_hit over (nm, mlen) pairs in list order, Python float64 semantics
 * (aligner.py:328-339). returns index of best or -1 for "0"/ambiguous. */
int orc_best_hit(const orc_hit_t *hits, int n);

/* classify a batch: bases = concatenated ASCII reads, offsets[n_reads+1].
 * out_assign[r] = contig id of the chosen hit | ORC_UNMAPPED | ORC_AMBIGUOUS
 * out_hit[r]    = the hit with the smallest NM/mlen (last one when tied; zero without hits)
 * out_nhits[r]  = number of gated hits
 * hits_flat/hits_cap: optional flat list of all gated hits in read order (may be NULL)
 * returns total number of gated hits. n_threads<=1: scalar. */
int64_t orc_classify_batch(const orc_index *mi, const orc_opt_t *opt, int mid_occ,
                           const char *bases, const int64_t *offsets, int n_reads,
                           int min_mapq, int n_threads,
                           int32_t *out_assign, orc_hit_t *out_hit, int32_t *out_nhits,
                           orc_hit_t *hits_flat, int64_t hits_cap);

#ifdef __cplusplus
}
#endif
#endif
