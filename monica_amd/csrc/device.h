// Device-side shared declarations (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include "common.h"

namespace mnc {

constexpr int KMER = 15;             // minimap2 'map-ont' (aligner.py:45); the only setting monica uses
constexpr int WIN = 10;
constexpr uint32_t KMASK = (1u << 2 * KMER) - 1;
constexpr int GAP_LUT = 512;         // gap cost look-up (dd <= bw = 500)

struct Anchor { uint64_t x, y; };    // x = strand<<63 | rid<<32 | rpos ; y = span<<32 | qpos

// ---------------------------------------------------------------- partitioned probe
// The hash table is cut into 2^PB_BITS contiguous regions ("buckets") by the top bits of the
// hash.  Query minimizers are partitioned by bucket so that the probe of one bucket
// touches one region (2 MiB for the 20-genome index: L2-resident) instead of random HBM lines.
// How many regions follows the index (index_upload): 256 for up to ~15 M keys, 512 / 1 024 beyond, so that a region's
// slots stay at 2 MiB -- the hot set of the XCD that walks it (a 62-genome part, 35 M keys, in 256 regions of 8 MiB ran
// the probe at 0.32 of the roofline; profiles/README.md).
constexpr int PB_BITS_MIN = 8, PB_BITS_MAX = 10;
constexpr int PB_N_MAX = 1 << PB_BITS_MAX;
constexpr int PT_READS = 4;                 // reads per partition tile (= one sketch workgroup)
constexpr int PS_TILES_MIN = 64;            // tiles per super-tile (probe / collect granularity): 256 reads
constexpr int PF_BITS = 18;                 // presence filter: 2^18 bits = 32 KiB per table region (fits LDS)
constexpr int PF_WORDS = (1 << PF_BITS) / 32;
// one-word Bloom filter: a key sets two bits of the word its rest selects; a query whose two bits
// are not both set is not in the table (5.7 bits per key on the 20-genome index: 9 % false
// positives instead of 17 % with one bit, for one LDS read either way)
__host__ __device__ __forceinline__ uint32_t pf_word(uint32_t rest) { return (rest >> 5) & (uint32_t)(((1 << PF_BITS) / 32) - 1); }
__host__ __device__ __forceinline__ uint32_t pf_mask(uint32_t rest)
{
	return (1u << (rest & 31u)) | (1u << (((rest >> 17) ^ (rest >> 12) ^ (rest >> 7)) & 31u));
}
constexpr uint32_t HIT_HIGH = 0x7fffffffu;  // cnt marker: occurrences >= mid_occ (only feeds rep_len)

// 64-bit query record: [21:0] rest of the hash, [22] strand, [23] tandem,
// [43:24] query position (last base of the k-mer), [63:44] read ordinal in the batch
constexpr int PD_MAX_BITS = 14;                             // at most 16384 displacement buckets (16 KiB of LDS)
// The region comes from the LOW hash bits: minimizers are window minima, so their hash values
// crowd towards zero and the high bits are far from uniform (region 0 would hold 5x its share).
__host__ __device__ __forceinline__ uint32_t pb_bucket(uint32_t hash, int pb_bits) { return hash & ((1u << pb_bits) - 1u); }
__host__ __device__ __forceinline__ uint32_t pb_rest(uint32_t hash, int pb_bits) { return hash >> pb_bits; }
__host__ __device__ __forceinline__ uint32_t pb_hash(uint32_t rest, uint32_t bucket, int pb_bits) { return rest << pb_bits | bucket; }
// base slot of a rest inside its region (independent of the low bits that pick the displacement bucket)
__host__ __device__ __forceinline__ uint32_t pd_base(uint32_t rest, int region_bits)
{
	return (uint32_t)(((uint64_t)(rest * 0x9E3779B1u) * (uint64_t)(1u << region_bits)) >> 32);
}
// slot = base + disp * step with a key-dependent odd step (double hashing), so two keys of one
// displacement bucket that share a base still separate for some disp
__host__ __device__ __forceinline__ uint32_t pd_step(uint32_t rest) { return ((rest * 0x85EBCA6Bu) >> 7) | 1u; }
// `salt` is a per-region constant mixed into the key: two keys of one displacement bucket with
// the same (base, step) cannot be separated by any disp, so the builder then re-salts the region
__host__ __device__ __forceinline__ uint32_t pd_slot(uint32_t rest, uint32_t disp, int region_bits, uint32_t salt)
{
	const uint32_t k = rest ^ salt;
	return (pd_base(k, region_bits) + disp * pd_step(k)) & ((1u << region_bits) - 1u);
}

// start of run (bucket, tile) in the bucket-major record array.  The offsets are stored
// tile-major ([tile][bucket], so a tile's 256 starts are one contiguous 2 KiB row); the end of
// a run is the start of the next one in bucket-major order.
__device__ __forceinline__ int64_t q_start(const int64_t *q_off, uint32_t n_tiles, uint32_t bucket, uint32_t tile, uint32_t pb_n)
{
	if (tile >= n_tiles) { tile = 0; ++bucket; }
	if (bucket >= pb_n) return q_off[(size_t)n_tiles * pb_n];
	return q_off[(size_t)tile * pb_n + bucket];
}

// one probe hit = one query minimizer with 0 < cnt < mid_occ
struct HitRec {
	uint64_t val;      // cnt == 1: occurrence word ; cnt > 1: offset into positions[]
	uint32_t qinfo;    // qpos<<1 | strand   (qpos = index of the k-mer's last base)
	uint32_t cnt;
};

// one chain as produced by the backtrack stage, ordered by first anchor
struct ChainRec {
	uint64_t x0, y0;   // first anchor
	uint64_t x1, y1;   // last anchor
	int32_t score, cnt;
	int32_t mlen, blen;
	int32_t as, pad;   // as: read-local index of the chain's LAST anchor (p[] leads back from it); pad: rank
};

// ---------------------------------------------------------------- base-level alignment stage
constexpr uint64_t SEED_LONG_JOIN = 1ULL << 40, SEED_IGNORE = 1ULL << 41, SEED_TANDEM = 1ULL << 42;
// ours, on the first chained anchor of a region (mnc_dp_gather): no two consecutive anchors of the region differ by more than ten
// bases in their query and target steps -- mm_filter_bad_seeds(_alt) then find nothing to do, and the plan kernel skips their two scans
constexpr uint64_t SEED_NOGAP10 = 1ULL << 43;
constexpr int REG_HAS_DP = 1, REG_SPLIT_L = 2, REG_SPLIT_R = 4, REG_SPLIT_INV = 8, REG_INV = 16;   // REG_INV: mm_reg1_t::inv (mm_align1_inv's region)
constexpr int EZ_RIGHT = 0x02, EZ_APPROX_MAX = 0x08, EZ_EXTZ_ONLY = 0x40, EZ_REV_CIGAR = 0x80;
constexpr int SEG_NEEDS_BIG_WS = 0x10000;                 // Seg.flag, ours: the literal kernel needs its large workspace for this call
constexpr int SEG_SKIPPED = 0x20000;                      // Seg.flag, ours: the call outgrows every workspace class -- not aligned, its read is reported MNC_SKIPPED
constexpr int DP_NEG_INF = -0x40000000;
constexpr int FILL_MAX_LEN = 511;     // longest target / query of a gap filling the banded kernel (k_fill.hip) takes
constexpr int FILL_MID_CELLS = 42;    // the band between the 32- and the 64-cell tier: 21 lanes a segment, three segments a wave

// one call of the two-piece affine kernel (ksw_extd2): left extension, a gap between two seeds,
// right extension
struct Seg {
	int32_t read, reg;        // read ordinal; absolute region slot
	int32_t kind;             // 0 left extension, 1 gap filling, 2 right extension
	int32_t rid, rev;
	int32_t ts, tlen;         // target interval [ts, ts + tlen) on the contig
	int32_t qs, qlen;         // query interval on strand `rev` of the read
	int32_t w, zdrop, flag;   // band, Z-drop, EZ_* flags
	int32_t ai;               // gap filling: index (from as1) of the seed it ends at
	int32_t big;              // direction bytes do not fit a normal workspace slot
	// ---- results
	int32_t n_cigar, zdropped, zdrop_code;
	int32_t max, max_t, max_q, score, reach_end, mqe_t;
	int64_t cig_off;          // first CIGAR word in the segment pool
};

// per region: what mm_align1 keeps between its steps
struct RegDP {
	int32_t read, order;      // order: position in the skeleton's region array (orig index << 8 | split depth)
	int32_t first_seg, n_seg; // segments: [left], gap fills..., [right]
	int32_t has_left, has_right;
	int32_t as1, cnt1;        // after trimming bad ends
	int32_t rs, qs, re, qe;   // first / last DP coordinate (seed mid-points)
	int32_t rs0, qs0, re0, qe0;
	int32_t n_cigar;          // region CIGAR (after the clean-up)
	int32_t state;            // 0 unused, 1 planned this round, 2 done
	int64_t cig_off;          // in the region pool
	// what the stitch kernel would otherwise reach through three more dependent loads (written by mnc_dp_plan)
	int32_t rid, rev, qlen;
	int32_t head;             // the tail of a Z-drop split: region slot of its head + 1 (0: none) -- mm_align1_inv's r1
	int64_t coff, read_off;   // first base of the contig in seq4; of the read in the batch
	int32_t inv_after;        // slot + 1 of the inversion region inserted behind this one: it, not this region, precedes the next tail
	int32_t pad_;
};

struct RegX { uint64_t x0, y0, x1, y1; };   // first / last anchor of a region

// Everything the stage kernels need, passed by value.
struct Batch {
	// ---- inputs
	const uint8_t *bases;
	const int64_t *offsets;
	uint32_t n_reads;
	int64_t total_bases;
	int min_mapq;
	// ---- index
	const TableSlot *table;       // [pb_n][1 << region_bits]
	const uint8_t *disp;          // [pb_n][1 << disp_bits] hash-and-displace displacements
	const uint32_t *salt;         // [pb_n] per-region salt of the slot function
	const uint32_t *filter;       // [pb_n][PF_WORDS] presence bits of (region, low PF_BITS of the rest)
	int pb_bits;                  // log2 of the number of table regions (8 .. 10), from the index
	uint32_t pb_n, ps_tiles;      // 1 << pb_bits; tiles per super-tile (PS_TILES_MIN)
	int region_bits, disp_bits;
	int disp_in_lds;              // the displacement table of a region fits the probe kernel's LDS copy
	const uint64_t *positions;
	const int32_t *contig_genome;
	int mid_occ;
	int n_genomes;
	int rid_bits, rpos_bits;      // bits of a contig id / a contig position; 0: anchors do not pack into 64 bits
	// ---- parameters
	int min_cnt, min_sc, bw, max_gap, max_skip, max_iter, best_n, seed;
	int max_join_long, max_join_short, min_join_flank_sc;
	float mask_level, pri_ratio, min_join_flank_ratio;
	const int32_t *gap_lut;       // [GAP_LUT] (int)(dd*.01*avg_span) + (ilog2(dd)>>1), host-computed
	const float *logf_lut;        // [logf_n] host libm logf(i)
	const float *logf_a_lut;      // [logf_n] host libm logf((float)i / a): the DP branch of the MAPQ formula
	int logf_n;
	// ---- per base-slot arrays (capacity total_bases)
	uint32_t *packed;             // 2-bit bases, 16 per word, first base in the top bits
	uint32_t *ambig;              // per read: 1 if it holds a non-ACGTU byte
	uint32_t *skip;               // per read: 1 if a kernel call of its alignment outgrew every workspace class (-> MNC_SKIPPED)
	uint2 *mz;                    // minimizers of read r at [offsets[r], offsets[r]+mz_cnt[r])
	HitRec *hits;                 // probe hits of read r at [offsets[r], offsets[r]+hit_cnt[r])
	// ---- partitioned probe
	uint32_t n_tiles;             // ceil(n_reads / PT_READS)
	uint32_t n_super;             // ceil(n_tiles / PS_TILES)
	uint32_t *hist_tm;            // [tile][bucket] minimizer counts (tile-major, written by the sketch)
	int64_t *q_off;               // exclusive scan in bucket-major order, STORED tile-major: see q_start()
	uint64_t *qrec;               // query records in bucket-major order
	int64_t q_cap;                // capacity of qrec / bhits in records
	HitRec *bhits;                // probe hits, compacted at the start of each (bucket, super-tile) run
	uint32_t *bhit_cnt;           // [super-tile][bucket] number of hits in the run (the collect workgroup of a super-tile reads a row)
	uint32_t *overflow;           // set when the batch needs more than q_cap records
	// ---- per read
	int32_t *mz_cnt;
	int32_t *hit_cnt;
	int32_t *rep_len;
	int64_t *an_cnt;              // anchors per read (scan input)
	int64_t *an_off;              // exclusive scan, n_reads + 1
	int32_t *n_chain;             // chains per read
	int32_t *n_reg;               // kept regions per read
	// ---- per anchor (capacity an_cap)
	int64_t an_cap;
	Anchor *a;                    // sorted anchors
	int32_t *f, *p, *v, *t;
	uint64_t *u;                  // chain-end candidates
	// ---- per chain slot (capacity an_cap/3 + 1): slot base of read r = an_off[r] / 3
	ChainRec *chains_tmp;         // backtrack order
	mnc_reg_t *regs;
	int32_t *tmp_i32;             // 4 ints per chain slot of scratch
	// ---- outputs
	int32_t *assign;
	mnc_hit_t *best;
	int32_t *nhits;
	int64_t *counts;              // [n_genomes * 3] or null
	int32_t *best_mlen;           // per read: mlen of the minimal gated hit (0 without one)
	int64_t *stats;               // device counters (see mnc_engine_get_counters)
	// ---- base-level alignment stage (contract MNC_CONTRACT_DP)
	int contract;
	int debug_route;              // test switch: 1 no packed extension kernel, 2 no packed gap-filling kernel, 4 no long tiers, 16 no 42-cell tier
	int fill_pred;                // a gap filling tries the 32-cell tier when its bound is below fill_pred / 25 per base
	int fill_pred_mid;            // ... and the 42-cell tier likewise
	int fill_pred_auto;           // 1: both from the region's own anchor density instead (k_align.hip: fill_pred_of); minimap2's map-ont scores only
	const uint32_t *seq4;         // contig bases, 4 bits each
	const int64_t *seq_off;       // [n_contigs + 1]
	int sc_a, sc_b, gap_q, gap_e, gap_q2, gap_e2, sc_ambi, zdrop, zdrop_inv, end_bonus, min_dp_max, min_ksw_len;
	long long max_sw_mat;
	Anchor *ca;                   // chained anchors of the kept regions, squeezed, per read at an_off[r]
	int32_t *ca_cnt;              // per read: anchors in ca
	int32_t *chain_dst;           // per chain slot: first position in ca | LONG_JOIN bit 30, or -1
	RegDP *regdp;                 // per region slot
	Seg *segs;
	int64_t seg_cap;
	uint32_t *cig_seg, *cig_reg;  // CIGAR pools: per segment, per region
	int64_t cig_seg_cap, cig_reg_cap;
	unsigned long long *dp_ctr;   // [0] segments, [1] words in cig_seg, [2] words in cig_reg, [3] align queue,
	                              // [4] overflow, [5] new regions this round, [6] big segments, [7] big queue,
	                              // [8] first segment of the round, [9] regions this round, [10..15] banded kernel
	int32_t *work_list;           // region slots to plan / stitch this round
	int32_t *next_list;           // region slots created by Z-drop splits (next round)
	int32_t *big_list;            // segment indices for the large-workspace launch
	int32_t *fill_list1, *fill_list2, *fill_fb;   // banded gap-filling kernel: 32-lane tier, 64-lane tier, handed back
	int32_t *fill_list3;                          // ... and the 128-cell tier
	int32_t *plan_long_list;                      // region slots mnc_dp_plan leaves to mnc_dp_plan_long (reads with many chained anchors): length dp_ctr[53]
	long long plan_long_cap;
	int32_t *fill_list_mid;                       // ... and the 42-cell tier between the first two: length dp_ctr[30], queue [54], anti-diagonals [52]
	int32_t *ext_list1, *ext_list2;               // extension kernel: 32 / 64 lanes per segment
	int32_t *ext_list3, *ext_list4;               // ... 128 / 256 cells (two / four per lane)
	int32_t *gen_list;                            // the literal kernel's first pass
	int32_t *lfill_list;                          // gaps of 512 .. 2047 bases, int32 banded kernel (256 cells): length dp_ctr[31], queue [61]
	int32_t *huge_list;                           // the few calls beyond the large workspace's slots (literal kernel, pass 5): length dp_ctr[58], queue [59]
	int32_t *bigfb_list;                          // calls the banded kernel hands back that need the literal kernel's large workspace: length dp_ctr[56], queue [57]
	int32_t *lext_list;                           // extensions of 257 .. 512 bases on the shorter side (step-by-step kernel, 8 cells per lane): length dp_ctr[62], queue [63]
	int32_t *mid_list;                            // the literal kernel, segments its first pass' LDS layout cannot hold: lengths dp_ctr[28], queue [29]
	int lds0_state, lds0_p, lds0_cig;             // that layout
	int32_t *extp_list;                           // packed extension kernel: 8 lists (32 / 64 / 128 / 256 query bases x right, left), `seg_cap` apart;
	                                              // lengths dp_ctr[32 + i], queues dp_ctr[40 + i]
	int32_t *reg_cnt;             // per read: regions in the skeleton's array (kept + split tails)
	int slot_pad;                 // region slots per read beyond anchors / 3 (split tails, inversion regions): see reg_slot()
};

// size classes of the row chaining kernel (anchors per LDS tile)
constexpr int MAX_CHAIN_CLASSES = 24;
struct ChainClasses { int n; int nm[MAX_CHAIN_CLASSES]; };
// the per-class read lists laid end to end: list c holds ordinals [start[c], start[c + 1])
struct ClassSpans { uint32_t start[MAX_CHAIN_CLASSES + 2]; uint32_t stride; int n; };

// ---------------------------------------------------------------- helpers

__device__ __forceinline__ uint32_t hash30(uint32_t key)
{
	key = (~key + (key << 21)) & KMASK;
	key ^= key >> 24;
	key = (key + (key << 3) + (key << 8)) & KMASK;
	key ^= key >> 14;
	key = (key + (key << 2) + (key << 4)) & KMASK;
	key ^= key >> 28;
	return key;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// First region slot of a read (regs, regdp, regx, ... are indexed by it; chains_tmp / chain_dst keep anchors / 3).
// A chain needs three anchors, so a read's chains fit anchors / 3 slots -- but a Z-drop split peels a head of as little
// as one anchor off a region, and an inversion region has no anchors at all: every read gets slot_pad more, and a read
// that needs even more fails the batch with dp_ctr[4] = 5 (redone with a larger pad).
__device__ __forceinline__ int64_t reg_slot(const Batch &B, uint32_t rd) { return B.an_off[rd] / 3 + (int64_t)rd * B.slot_pad; }

} // namespace mnc
