/*
 * monica_amd.h -- C-ABI of the MI355X-native read-classification engine.
 *
 * This is the drop-in boundary for monica's aligner hot path.  In the reference the
 * boundary is the Python->C crossing into mappy (minimap2 2.17) at
 *   monica/genomes/aligner.py:45-46   mappy.Aligner(fn_idx_in=<fasta.gz>, preset='map-ont',
 *                                                   best_n=15, fn_idx_out=<index file>)
 *   monica/genomes/aligner.py:59      mappy.Aligner(fn_idx_in=<index file>)
 *   monica/genomes/aligner.py:193,215 index.map(str(seq_record.seq)) -> hits
 *                                     (.is_primary .mapq .ctg .NM .mlen, lines 194-195, 216-217)
 * and the per-read Python that consumes the hits (aligner.py:218-263, 328-339).
 * Each entry point below names the reference interface it replaces.
 *
 * Conventions: every function returns MNC_OK (0) or a negative MNC_ERR_* code and never
 * throws or aborts across the ABI.  Handles are opaque.  The library never keeps a caller
 * pointer past the call.  An mnc_index is immutable after build/load/upload and may be
 * shared by any number of engines and threads; an mnc_engine (stream + HBM workspace)
 * must be used by one thread at a time.
 *
 * There is no CPU execution path in this library: classification entry points fail with
 * MNC_ERR_NODEVICE when no gfx950 device is present.
 */
#ifndef MONICA_AMD_H
#define MONICA_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNC_OK               0
#define MNC_ERR_ARG         (-1)   /* bad argument                                        */
#define MNC_ERR_IO          (-2)   /* file cannot be opened / read / written               */
#define MNC_ERR_FORMAT      (-3)   /* damaged or empty index / FASTA                       */
#define MNC_ERR_NOMEM       (-4)   /* host or device allocation failed                     */
#define MNC_ERR_HIP         (-5)   /* HIP runtime error (see mnc_last_error())             */
#define MNC_ERR_NODEVICE    (-6)   /* no usable gfx950 device                              */
#define MNC_ERR_UNSUPPORTED (-7)   /* parameter outside what the kernels implement         */
#define MNC_ERR_RANGE       (-8)   /* caller buffer too small; required size is reported   */

/* per-read decision codes in out_assign[] (aligner.py:218-233, 264-265) */
#define MNC_UNMAPPED   (-1)        /* no gated hit        -> unmapped/<sample>             */
#define MNC_AMBIGUOUS  (-2)        /* best_hit() returned 0 -> ambiguous/<sample>          */
#define MNC_SKIPPED    (-3)        /* the read is outside what the kernels hold (see "Limits" below): not classified; no
                                      gated hits.  index.map() has no such case (aligner.py:193, 215): the host side
                                      routes the read as unmapped and says so, the rest of its batch is unaffected      */

typedef struct mnc_index  mnc_index;
typedef struct mnc_engine mnc_engine;

/* one gated hit = the (ctg, NM, mlen) tuple of aligner.py:195,217 plus its mapq */
typedef struct {
	int32_t rid;    /* contig index; ctg = mnc_index_contig_name(rid) */
	int32_t mapq;
	int32_t nm;     /* hit.NM = blen - mlen + n_ambi after base-level alignment (blen - mlen of the
	                   chain under MNC_CONTRACT_CHAIN) */
	int32_t mlen;
} mnc_hit_t;

typedef struct {
	int32_t k, w;
	int32_t n_contigs;
	int32_t n_genomes;      /* distinct contig names ("tax_unit:accession", database.py:59) */
	int32_t mid_occ;        /* occurrence cut-off derived from the index (mid_occ_frac 2e-4) */
	int32_t reserved;
	int64_t n_keys;         /* distinct minimizer hashes    */
	int64_t n_occ;          /* minimizer occurrences        */
	int64_t total_len;      /* sum of contig lengths        */
	int64_t table_slots;    /* HBM hash-table slots         */
	int64_t device_bytes;   /* bytes resident in HBM after upload (0 before) */
} mnc_index_info_t;

/* ---------------------------------------------------------------- errors */
const char *mnc_strerror(int code);
const char *mnc_last_error(void);          /* thread-local detail of the last failure */

/* ---------------------------------------------------------------- device */
int mnc_device_count(int *n);              /* counts devices without initialising them */
int mnc_device_name(int device, char *buf, size_t cap);
/* free / total HBM of a device: what the host side sizes its cache of resident index parts against
 * (the reference holds one part at a time, aligner.py:91-103; this library keeps parts resident
 * between the passes of monica's loop as long as they fit) */
int mnc_device_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes);

/* ---------------------------------------------------------------- index: build / load
 * mnc_index_build      <- mappy.Aligner(fn_idx_in=fasta(.gz), preset='map-ont', best_n=15,
 *                         fn_idx_out=out_path)                      aligner.py:45-46
 *                         out_path may be NULL (no file written).  k=15,w=10 is 'map-ont'.
 * mnc_index_load       <- mappy.Aligner(fn_idx_in=index file)       aligner.py:59
 *                         a falsy Aligner maps to MNC_ERR_FORMAT ('Damaged or empty index').
 */
int  mnc_index_build(const char *fasta_path, const char *out_path, int k, int w, mnc_index **out);
int  mnc_index_build_mem(int n_seq, const char *const *names, const char *const *seqs,
                         const int64_t *lens, int k, int w, mnc_index **out);
/* The same two with the sketch and the sort on a device (contig pieces through the batch sketch kernel, radix sort,
 * run lengths; csrc/k_idxbuild.hip): the same index, array for array, in a fraction of the time.  gz decompression and
 * the 4-bit packing of the contig bases stay on the host. */
int  mnc_index_build_device(const char *fasta_path, const char *out_path, int k, int w, int device, mnc_index **out);
int  mnc_index_build_mem_device(int n_seq, const char *const *names, const char *const *seqs,
                                const int64_t *lens, int k, int w, int device, mnc_index **out);
int  mnc_index_save(const mnc_index *idx, const char *path);
int  mnc_index_load(const char *path, mnc_index **out);
/* minimap2's own index format ("MMI\2", what mappy writes at aligner.py:45-46): mnc_index_load reads either format;
 * this writes it, so that an index built here can be handed to mappy / minimap2 as well */
int  mnc_index_save_mmi(const mnc_index *idx, const char *path);
void mnc_index_free(mnc_index *idx);
int  mnc_index_info(const mnc_index *idx, mnc_index_info_t *info);
const char *mnc_index_contig_name(const mnc_index *idx, int rid);      /* hit.ctg, aligner.py:195 */
int64_t     mnc_index_contig_len(const mnc_index *idx, int rid);
int         mnc_index_contig_genome(const mnc_index *idx, int rid);    /* contig -> genome id     */
const char *mnc_index_genome_name(const mnc_index *idx, int gid);      /* "tax_unit:accession"    */
int64_t     mnc_index_genome_len(const mnc_index *idx, int gid);       /* database.py:57-65 sum   */
/* dump (hash, y) pairs sorted by (hash, y); for tests.  *n receives the pair count. */
int  mnc_index_dump(const mnc_index *idx, uint64_t *hash, uint64_t *y, int64_t cap, int64_t *n);
/* override the occurrence cut-off (index shards of one logical index share one value) */
int  mnc_index_set_mid_occ(mnc_index *idx, int mid_occ);

/* ---------------------------------------------------------------- engine
 * One engine = one HIP stream + a growable HBM workspace on `device`.  The first engine
 * created for an index on a device uploads the index tables to that device's HBM.
 */
int  mnc_engine_create(mnc_index *idx, int device, mnc_engine **out);
/* What index.map() computes before monica's gate (aligner.py:193-195, 215-217):
 *   MNC_CONTRACT_DP     (default) minimap2's base-level alignment of every region, as mappy always
 *                       runs it (MM_F_CIGAR): mapq from dp_max / dp_max2 and identity, NM and
 *                       mlen from the CIGAR -- the values monica reads
 *   MNC_CONTRACT_CHAIN  stop after chaining: chain-level MAPQ, NM := blen - mlen of the chain
 *                       (the path BASELINE.json's north_star lists; ~10x faster, other numbers) */
#define MNC_CONTRACT_DP    0
#define MNC_CONTRACT_CHAIN 1
int  mnc_engine_set_contract(mnc_engine *eng, int contract);
void mnc_engine_destroy(mnc_engine *eng);
/* another index part behind the same engine: `index = index_loader(part)` in the reference's loop over the
 * parts of a database (aligner.py:91-103).  Same k / w / match score as the index the engine was made for. */
int  mnc_engine_set_index(mnc_engine *eng, mnc_index *idx);
void *mnc_engine_stream(mnc_engine *eng);                              /* hipStream_t */
int  mnc_index_device_bytes(const mnc_index *idx, int device, int64_t *bytes); /* HBM the index holds on that one device */
int  mnc_engine_device_bytes(mnc_engine *eng, int64_t *bytes);         /* HBM held by the engine's own batch buffers */

/* ---------------------------------------------------------------- limits, and what happens at each
 *   k = 15, w = 10 only                          mnc_index_build*: MNC_ERR_UNSUPPORTED (the one setting monica uses,
 *                                                aligner.py:45: preset 'map-ont')
 *   a read of 2^20 bases or more                 mnc_classify_batch: that read alone comes back MNC_SKIPPED (no hits); the
 *                                                other reads of the batch are classified as ever.  mnc_classify_device
 *                                                (device-resident batches; the caller states max_read_len): MNC_ERR_UNSUPPORTED
 *   one kernel call of a read's alignment whose  that read alone comes back MNC_SKIPPED.  (tlen x qlen <= 1e8 with more
 *   direction matrix exceeds 256 MB              than 256 MB of direction bytes: a few hundred query bases against > 300 000
 *                                                target bases -- mm_align1 makes no such call with max_gap = 5 000)
 *   more than 2^20 reads in one call             MNC_ERR_UNSUPPORTED before anything runs (the probe's records hold a 20-bit
 *                                                read ordinal); monica_amd.aligner passes at most 100 000 per call
 *   2^40 bases or more in one call               MNC_ERR_UNSUPPORTED before anything runs
 *   reads with > 8 192 anchors or >= 65 536      exact; slower forms (sort in HBM, sequential backtrack, a wave per region
 *   bases                                        in the plan, bases read in place by the stitch kernel)
 *   a batch that outgrows a pool (query records, exact: the batch is redone with more room (mnc_engine_get_counters [7] counts
 *   segments, CIGAR words, region slots)         the passes); MNC_ERR_NOMEM when HBM itself is exhausted
 */

/* ---------------------------------------------------------------- classify
 * mnc_classify_batch   <- the per-read loop body aligner.py:212-233 for one index part:
 *                         index.map(seq) -> gate `is_primary and mapq >= min_mapq`
 *                         -> single hit | best_hit(hits) | ambiguous.
 *   bases     host, concatenated read bases (ASCII; anything but ACGTUacgtu is ambiguous)
 *   offsets   host, n_reads+1 byte offsets into bases
 *   out_assign[n_reads]  contig index of the chosen hit | MNC_UNMAPPED | MNC_AMBIGUOUS
 *   out_best[n_reads]    the hit with the smallest NM/mlen (the last such hit when tied, i.e.
 *                        also for MNC_AMBIGUOUS; zero when there is no gated hit); may be NULL
 *   out_nhits[n_reads]   number of gated hits of the read; may be NULL
 * The gated hit lists themselves stay in HBM until the next call; fetch them with
 * mnc_engine_fetch_hits (needed for the multi-part merge, aligner.py:196-203, 218-223).
 */
int mnc_classify_batch(mnc_engine *eng, const uint8_t *bases, const int64_t *offsets,
                       uint32_t n_reads, int min_mapq,
                       int32_t *out_assign, mnc_hit_t *out_best, int32_t *out_nhits);

/* The next batch's bases on their way while this one is classified: starts the host-to-device copy of (bases,
 * offsets) into a spare device buffer on a stream of its own and returns.  The mnc_classify_batch call that
 * follows with the SAME pointers and n_reads skips its copy.  The reference has no counterpart -- its loop
 * hands mappy one read at a time (aligner.py:191-193, 212-215); here a batch crosses PCIe (5 kB of ASCII per
 * read), which would otherwise sit in front of every batch's kernels.  The caller leaves the host arrays as they
 * are until that call (page-locked arrays, e.g. mnc_fastq's, make the copy asynchronous).  *started = 0 when a
 * prefetched batch is still waiting for its call (one spare buffer): nothing was done.  Thread-safe against the
 * engine's classifying thread. */
int mnc_engine_prefetch(mnc_engine *eng, const uint8_t *bases, const int64_t *offsets, uint32_t n_reads, int *started);
/* Forget the batch announced last, if any (waits for its copy; the host arrays are the caller's again).  For a caller
 * that gives up on an announced batch -- an error in its loop (the reference's loop simply raises, aligner.py:212-215),
 * an engine handed to another sample: the announcement is matched by host address, so it must not outlive its arrays. */
int mnc_engine_prefetch_cancel(mnc_engine *eng);

/* Same, all buffers device-resident (HBM), asynchronous on the engine's stream.
 * total_bases = offsets[n_reads].  d_counts (may be NULL) is an int64[n_genomes*3] table
 * that the call ADDS to: {reads, read bases, mlen} per genome, i.e. the three counting
 * modes of aligner.py:247-263 ('basic', 'query_length', 'matching'). */
int mnc_classify_device(mnc_engine *eng, const uint8_t *d_bases, const int64_t *d_offsets,
                        uint32_t n_reads, int64_t total_bases, int max_read_len, int min_mapq,
                        int32_t *d_assign, mnc_hit_t *d_best, int32_t *d_nhits, int64_t *d_counts);
int mnc_engine_sync(mnc_engine *eng);

/* gated hits of the last batch: hit_offsets[n_reads+1] and hits[cap]; *n_hits = total.
 * Returns MNC_ERR_RANGE (with *n_hits set) if cap is too small. */
int mnc_engine_fetch_hits(mnc_engine *eng, int64_t *hit_offsets, mnc_hit_t *hits, int64_t cap,
                          int64_t *n_hits);

/* mnc_counts  <- the Counter accumulation aligner.py:247-263 on host arrays.
 * mode: 1 'basic' (+1), 2 'query_length' (+len(seq)), 3 'matching' (+mlen).
 * counts[n_genomes] is ADDED to. */
int mnc_counts(const mnc_index *idx, const int32_t *assign, const mnc_hit_t *best,
               const int64_t *offsets, uint32_t n_reads, int mode, int64_t *counts);

/* cross-part merge of gated hit lists <- aligner.py:219-233 + best_hit 328-339 in exact
 * integer arithmetic.  hits of all parts of one read are concatenated in part order;
 * rid values must already be global.  Writes the decision per read. */
int mnc_best_hit(const mnc_hit_t *hits, int n, int *best_index /* -1 = ambiguous */);

/* ---------------------------------------------------------------- profiling / introspection */
#define MNC_STAGE_PACK        0
#define MNC_STAGE_SKETCH      1
#define MNC_STAGE_PARTITION   2   /* scan + scatter of query records by table region        */
#define MNC_STAGE_PROBE       3   /* index probe: the HBM-roofline kernel                    */
#define MNC_STAGE_COLLECT     4   /* probe hits back to per-read lists                       */
#define MNC_STAGE_SORT        5   /* anchor offsets (scan) + size classes                    */
#define MNC_STAGE_SORT2       6   /* expand hits to anchors + sort                           */
#define MNC_STAGE_CHAIN       7   /* chaining DP (LDS ring per read)                         */
#define MNC_STAGE_BACKTRACK   8   /* backtrack -> chain records                              */
#define MNC_STAGE_REGIONS     9   /* regions, MAPQ, decision, counts                         */
#define MNC_STAGE_GATHER      10  /* gated hit lists -> CSR (mnc_engine_fetch_hits)          */
#define MNC_STAGE_DP_PLAN     11  /* chained anchors per region, DP windows, ksw2 segments   */
#define MNC_STAGE_DP_ALIGN    12  /* extensions, unusual gaps: ksw2's kernel in its own layout  */
#define MNC_STAGE_DP_STITCH   13  /* CIGAR merge / clean-up, mlen, blen, dp_max, Z-drop split */
#define MNC_STAGE_DP_POST     14  /* second hierarchy pass, DP MAPQ, gate, decision          */
#define MNC_STAGE_DP_FILL     15  /* gap filling between seeds: banded two-piece affine DP       */
#define MNC_STAGE_DP_FILL_T1  16  /* the next four only with debug bit 0x10000 (kernels one at a time):  */
#define MNC_STAGE_DP_FILL_T2  17  /*   the banded gap-filling kernel's 32- / 64- / 128-cell launches     */
#define MNC_STAGE_DP_FILL_T3  18
#define MNC_STAGE_DP_EXT      19  /*   all the extension kernels                                         */
#define MNC_STAGE_DP_FILL_TM  20  /*   the 42-cell launch of the banded kernel (between T1 and T2)       */
#define MNC_STAGE_DP_LFILL    21  /*   gaps of 512 .. 2047 bases on the int32 banded kernel              */
#define MNC_N_STAGES          22
int mnc_engine_set_profiling(mnc_engine *eng, int on);    /* HIP events around every stage */
int mnc_engine_set_debug(mnc_engine *eng, int mode);      /* test switches, a bit mask: 2 stress build of the
                                                             chaining ring, 4 displacement bytes read from HBM,
                                                             0x10 every region planned by mnc_dp_plan_long, 0x20 the
                                                             literal kernel's long calls on one wave (not four),
                                                             0x10000 the alignment kernels one at a time (per-kernel
                                                             timers), bits 8-15 a tuning value for the tier choice,
                                                             0x20000 / 0x40000 / 0x80000 / 0x100000 without the packed
                                                             extension / packed gap-filling / long-gap / long-extension
                                                             kernels (their calls go to the next kernel in line),
                                                             0x200000 the stitch kernel reads bases in place (its form
                                                             for regions beyond its LDS), 0x400000 without the 42-cell
                                                             tier, bits 24-30 that tier's tuning value, 0x800000 the
                                                             regions of long reads planned one lane each as all others
                                                             (not by mnc_dp_plan_long) */
/* accumulated since the last reset: ms[MNC_N_STAGES], launches[MNC_N_STAGES] */
int mnc_engine_get_timings(mnc_engine *eng, double *ms, int64_t *launches, int reset);
const char *mnc_stage_name(int stage);
const char *mnc_stage_kernel(int stage);                  /* kernel symbol, for rocprof matching */
/* counters of the last batch: [0] minimizers, [1] probe hits, [2] anchors, [3] chains,
 * [4] regions, [5] gated hits, [6] reads with ambiguous bases, [7] -; with n >= 12, of the base-level
 * alignment stage (last round): [8] kernel calls (segments), [9] gap fillings given to the banded kernel's
 * 32-cell tier, [10] those its 64-cell tier saw, [11] those handed back to the literal kernel; with
 * n >= 16: [12] / [13] / [14] anti-diagonals (steps) of the gap fillings the 32- / 64- / 128-cell tier
 * ran, [15] anti-diagonal steps x query bases of the extensions given to the packed extension kernel; with n >= 24
 * (last round): [16] / [17] calls planned for the literal kernel's large / small workspace, [18] long gaps (int32 banded
 * kernel), [19] those handed back that need the large workspace, [20] long extensions, [21] gap fillings the 128-cell tier saw */
int mnc_engine_get_counters(mnc_engine *eng, int64_t *c, int n);

/* Test hooks of the index residency: the device tables (hash-and-displace perfect hash per table region) are built
 * on the device; `on` = 1 makes the next upload of this index use the host form of the same construction instead;
 * mnc_engine_dump_tables copies what the engine's device holds ([region_bits, disp_bits] int32, salts, displacement
 * bytes, presence filter, table slots) so that the two can be compared. */
int mnc_index_set_host_tables(mnc_index *idx, int on);
/* the device tables are cut into 2^bits regions (8 .. 10; 0 = chosen from the number of keys so that a region stays at
 * 2 MiB): a speed matter only -- every value gives the same results (tests force the larger ones on small indexes) */
int mnc_index_set_region_bits(mnc_index *idx, int bits);
int mnc_engine_dump_tables(mnc_engine *eng, void *dst, int64_t cap_bytes, int64_t *n_bytes);

/* stage dumps of the last batch, for kernel-level parity tests */
#define MNC_DUMP_MINIMIZERS 1  /* u32 pairs {hash, pos<<1|strand} in read order                    */
#define MNC_DUMP_MZ_OFFSETS 2  /* int64[n_reads+1]                                              */
#define MNC_DUMP_ANCHORS    3  /* u64 x,y pairs sorted by (x,y), per read                       */
#define MNC_DUMP_AN_OFFSETS 4  /* int64[n_reads+1]                                              */
#define MNC_DUMP_CHAIN_F    5  /* int32 per anchor                                              */
#define MNC_DUMP_CHAIN_P    6  /* int32 per anchor (read-local index, -1 none)                  */
#define MNC_DUMP_CHAIN_V    7  /* int32 per anchor                                              */
#define MNC_DUMP_REGS       8  /* mnc_reg_t per region                                          */
#define MNC_DUMP_REG_OFFSETS 9 /* int64[n_reads+1]                                              */
#define MNC_DUMP_REP_LEN    10 /* int32 per read                                                */
#define MNC_DUMP_CIGARS     11 /* uint32 len<<4|op (0 M, 1 I, 2 D) of the dumped regions, back to back
                                  (regs[i].n_cigar words each); MNC_CONTRACT_DP only              */
#define MNC_DUMP_SEGS       12 /* the alignment stage's kernel calls (one per ksw_extd2 call of mm_align1): 24 int32
                                  {read, region slot, kind 0 left ext / 1 gap / 2 right ext, rid, rev, ts, tlen,
                                  qs, qlen, w, zdrop, flag, seed, kernel class, n_cigar, zdropped, zdrop_code,
                                  max, max_t, max_q, score, reach_end, mqe_t, pad} + int64 CIGAR offset; tools only */
typedef struct {
	int32_t id, parent, rid, rev, rs, re, qs, qe, score, score0, cnt, as, mlen, blen,
	        subsc, n_sub, mapq;
	uint32_t hash;
	/* base-level alignment (0 under MNC_CONTRACT_CHAIN): mm_extra_t's dp_score / dp_max / dp_max2 /
	 * n_ambi / n_cigar; flags: 1 a CIGAR exists, 2 / 4 split left / right part, 8 split_inv */
	int32_t dp_score, dp_max, dp_max2, n_ambi, n_cigar, flags;
} mnc_reg_t;
int mnc_engine_dump(mnc_engine *eng, int what, void *dst, int64_t cap_bytes, int64_t *n_bytes);

/* ---------------------------------------------------------------- FASTQ batches and read routing
 * Replaces the per-record Biopython objects of the reference's loop:
 * SeqIO.parse(sample, 'fastq') + str(seq_record.seq) (monica/genomes/aligner.py:191-193,
 * 212-215) and SeqIO.write(seq_record, handle, 'fastq') into mapped/ unmapped/ ambiguous/
 * focus/ (aligner.py:232-243, 265).  Records follow Biopython's FASTQ rules (multi-line
 * sequence / quality, '+' caption check, length check); a malformed file gives
 * MNC_ERR_FORMAT with Biopython's message in mnc_last_error().  A reader owns one batch at a
 * time; pointers returned by the accessors stay valid until the next mnc_fastq_next/close. */
typedef struct mnc_fastq mnc_fastq;
int mnc_fastq_open(const char *path, mnc_fastq **out);
void mnc_fastq_close(mnc_fastq *fq);
/* parse the next batch: stops after max_reads records or once max_bases bases are held.
 * *n_reads = 0 at the end of the file. */
int mnc_fastq_next(mnc_fastq *fq, uint32_t max_reads, uint64_t max_bases, uint32_t *n_reads);
/* the current batch as a handle of its own (the reader goes on with fresh arrays): for a host loop that parses
 * batch k + 1 while batch k is classified and batch k - 1 is written out.  Accessors, mnc_fastq_route and
 * mnc_hitmap_update take it like the reader; mnc_fastq_close frees it. */
int mnc_fastq_detach_batch(mnc_fastq *fq, mnc_fastq **out);
/* bytes of the file no batch has taken yet (-1 for a pipe): a host loop that overlaps parsing, classification and
 * output makes its first and last batches small, so that the pipeline fills and drains quickly */
int mnc_fastq_remaining(const mnc_fastq *fq, int64_t *bytes);
const uint8_t *mnc_fastq_bases(const mnc_fastq *fq);     /* concatenated sequences (page-locked when a GPU is present) */
const int64_t *mnc_fastq_offsets(const mnc_fastq *fq);   /* n_reads + 1, starts at 0 */
const uint8_t *mnc_fastq_quals(const mnc_fastq *fq);     /* quality characters, same offsets */
/* title line of read r (without '@'); *id_len = length of its first word (seq_record.id) */
int mnc_fastq_title(const mnc_fastq *fq, uint32_t r, const char **title, uint32_t *len, uint32_t *id_len);
/* Append every record of the batch to the files its dest[r] bits name:
 *   MNC_TO_UNMAPPED / MNC_TO_AMBIGUOUS / MNC_TO_FOCUS: the record as read;
 *   MNC_TO_MAPPED: the record with its id replaced by labels[label[r]]
 *   (seq_record.id = tax_unit, aligner.py:242; Biopython then writes "<id> <old title>").
 * paths[4] = {unmapped, ambiguous, mapped, focus}; a NULL path must not be addressed. */
#define MNC_TO_UNMAPPED  1
#define MNC_TO_AMBIGUOUS 2
#define MNC_TO_MAPPED    4
#define MNC_TO_FOCUS     8
int mnc_fastq_route(const mnc_fastq *fq, const uint8_t *dest, const int32_t *label,
                    const char *const *labels, int n_labels, const char *const *paths);

/* ---------------------------------------------------------------- hits carried across index parts
 * Replaces the `sample_hits` dict and its <sample>_hits.pkl (aligner.py:184-188, 196-203,
 * 218-223, 267-273): per read id, the gated hits of all index parts seen so far.  best_hit
 * (aligner.py:328-339) only asks for the smallest NM/mlen and whether it is attained once,
 * so a read's list is held as {hits, nm, mlen, contig name, tied}; extending the list and
 * reducing it commute with this summary, exactly (int64 cross-products). */
typedef struct mnc_hitmap mnc_hitmap;
int mnc_hitmap_create(mnc_hitmap **out);
int mnc_hitmap_load(const char *path, mnc_hitmap **out);
int mnc_hitmap_save(const mnc_hitmap *hm, const char *path);
void mnc_hitmap_free(mnc_hitmap *hm);
int64_t mnc_hitmap_size(const mnc_hitmap *hm);           /* ids with at least one hit */
/* For every read of the batch, in order: sample_hits[id].extend(this part's gated hits),
 * given as the engine's outputs (nhits, best = minimal hit, assign == MNC_AMBIGUOUS: tied);
 * then out[r] = the state of sample_hits[id]: {hits, nm, mlen, name id, tied}
 * (hits == 0: the id is not in the dict).  Name ids index mnc_hitmap_name(). */
int mnc_hitmap_update(mnc_hitmap *hm, const mnc_fastq *fq, const mnc_index *idx,
                      const int32_t *assign, const mnc_hit_t *best, const int32_t *nhits,
                      int32_t *out /* n_reads x 5 */);
int mnc_hitmap_n_names(const mnc_hitmap *hm);
const char *mnc_hitmap_name(const mnc_hitmap *hm, int id);

/* ---------------------------------------------------------------- collectives (RCCL over xGMI)
 * The cross-process forms of monica's two merges, on a caller-supplied ncclComm_t and HIP stream
 * (librccl.so is opened on first use; the library does not link it):
 *   mnc_allreduce_counts     <- Counter.update in alignment_update (aligner.py:286-298) when the reads
 *                               of a batch are sharded over GPUs: int64 sum of the count table
 *                               (n = n_genomes * 3 for the table mnc_classify_device fills), in place
 *   mnc_allgather_summaries  <- the hits carried between index parts (aligner.py:91-103, 196-203,
 *                               218-223) when the parts live on different GPUs: every rank's per-read
 *                               summary block {hits, nm, mlen, contig, tied} (20 B per read), in rank
 *                               order; the merge itself is mnc_best_hit's rule per read
 * mnc_comm_* make / free a communicator through this ABI alone (ncclGetUniqueId on one rank, the
 * 128 bytes handed to the others by any means, ncclCommInitRank on all). */
int mnc_comm_unique_id(void *id128);
int mnc_comm_init_rank(const void *id128, int n_ranks, int rank, void **comm);
int mnc_comm_destroy(void *comm);
int mnc_comm_count(void *comm, int *n_ranks);   /* ncclCommCount: the ranks the communicator really spans */
int mnc_allreduce_counts(int64_t *d_counts, int n, void *comm, void *stream);
int mnc_allgather_summaries(const void *d_send, void *d_recv, size_t bytes_per_rank, void *comm, void *stream);

/* ---------------------------------------------------------------- C2: the merge over index parts, on the device
 * The reference carries a read's gated hits from index part to index part (sample_hits[read_id].extend(...),
 * aligner.py:196-203, 218-223; pickled between passes, aligner.py:184-188, 267-273) and lets best_hit
 * (aligner.py:328-339) decide over the union (aligner.py:219-233).  best_hit only asks for the smallest NM/mlen and
 * whether the last update of its running minimum was a tie, so one part's list is the 20-byte record
 * {hits, nm, mlen, global contig | -1, tied} per read (the same record mnc_hitmap_update carries on the host).
 *   mnc_shard_summary     writes a part's records [n][5] from mnc_classify_device's outputs of that part
 *                         (d_assign, d_best, d_nhits); rid_offset = the part's first contig in the global numbering
 *   mnc_merge_summaries   best_hit over the parts: d_parts = [n_parts][n][5] int32 in part order (= hit order: what
 *                         mnc_allgather_summaries leaves when rank order is part order); d_assign[n] = global contig |
 *                         MNC_UNMAPPED | MNC_AMBIGUOUS; d_nm / d_mlen / d_total (the minimal hit's NM and mlen, the
 *                         number of gated hits over all parts) may be NULL.  int64 cross-products: mnc_best_hit's rule.
 * All pointers are device pointers; both calls are asynchronous on `stream` (a hipStream_t, NULL = the default stream). */
int mnc_shard_summary(const int32_t *d_assign, const mnc_hit_t *d_best, const int32_t *d_nhits, int64_t n,
                      int32_t rid_offset, int32_t *d_out, void *stream);
int mnc_merge_summaries(const int32_t *d_parts, int n_parts, int64_t n, int32_t *d_assign, int32_t *d_nm, int32_t *d_mlen,
                        int32_t *d_total, void *stream);

/* how many sample files the host works on side by side -- monica's ThreadPool runs aligner() once per sample,
 * aligner.py:89-103: every FASTQ reader's parse and routing passes then take cores / n_workers threads (1, the
 * default: all of them, at most 16).  "cores" = what the process may keep busy: the hardware threads, less what its
 * CPU affinity and its control group's quota (cpu.max / cpu.cfs_quota_us) leave of them.  The passes' helper threads
 * belong to the calling thread, sleep between passes and end with it (csrc/team.h); like an OpenMP runtime's, they do
 * not survive fork(). */
int mnc_host_set_io_workers(int n_workers);
/* page-locked host memory for batch buffers (plain malloc when no GPU is present) */
void *mnc_host_alloc(size_t bytes);
void mnc_host_free(void *p);

/* ---------------------------------------------------------------- synthetic data (bench/tests)
 * Deterministic counter-based generator (SplitMix64), SURVEY.md section 8d. */
int mnc_synth_genome(uint64_t seed, int64_t len, char *out);
int mnc_synth_diverge(const char *src, int64_t len, uint64_t seed, int rate_ppm, char *out);
/* reads: ordinal r in [first, first+n) draws from stream (seed, r).  Rates in 1e-4 units.
 * random_frac_e4 of the reads are pure random sequence (truth = -1). */
int mnc_synth_reads(int n_genomes, const char *const *genomes, const int64_t *lens,
                    uint64_t seed, int64_t first, int n_reads, int read_len,
                    int sub_e4, int ins_e4, int del_e4, int random_frac_e4,
                    char *out_bases, int32_t *out_truth);

/* the same reads made on the device (one wave per read), byte for byte: d_genomes = the contigs
 * concatenated (ASCII), d_g_off[n_genomes + 1] their starts; all pointers are device pointers, the
 * launch is asynchronous on `stream` (a hipStream_t, NULL = the default stream).  BASELINE config 3's
 * 10 M reads are 50 GB: they are generated where they are classified. */
int mnc_synth_reads_device(int n_genomes, const uint8_t *d_genomes, const int64_t *d_g_off,
                           uint64_t seed, int64_t first, int n_reads, int read_len,
                           int sub_e4, int ins_e4, int del_e4, int random_frac_e4,
                           uint8_t *d_out_bases, int32_t *d_out_truth, void *stream);

const char *mnc_version(void);

#ifdef __cplusplus
}
#endif
#endif
