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
	int32_t as, pad;   // offset of the first anchor in the read's chained-anchor order
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
	const TableSlot *table;
	uint64_t table_mask;
	const uint64_t *positions;
	const int32_t *contig_genome;
	int mid_occ;
	int n_genomes;
	// ---- parameters
	int min_cnt, min_sc, bw, max_gap, max_skip, max_iter, best_n, seed;
	int max_join_long, max_join_short, min_join_flank_sc;
	float mask_level, pri_ratio, min_join_flank_ratio;
	const int32_t *gap_lut;       // [GAP_LUT] (int)(dd*.01*avg_span) + (ilog2(dd)>>1), host-computed
	const float *logf_lut;        // [logf_n] host libm logf(i)
	int logf_n;
	// ---- per base-slot arrays (capacity total_bases)
	uint32_t *packed;             // 2-bit bases, 16 per word, first base in the top bits
	uint32_t *ambig;              // per read: 1 if it holds a non-ACGTU byte
	uint2 *mz;                    // minimizers of read r at [offsets[r], offsets[r]+mz_cnt[r])
	HitRec *hits;                 // probe hits of read r at [offsets[r], offsets[r]+hit_cnt[r])
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
	ChainRec *chains;             // final order: by first anchor
	ChainRec *chains_tmp;         // backtrack order
	mnc_reg_t *regs;
	int32_t *tmp_i32;             // 4 ints per chain slot of scratch
	// ---- outputs
	int32_t *assign;
	mnc_hit_t *best;
	int32_t *nhits;
	int64_t *counts;              // [n_genomes * 3] or null
	int64_t *stats;               // device counters (see mnc_engine_get_counters)
};

// size classes of the row chaining kernel (anchors per LDS tile)
constexpr int MAX_CHAIN_CLASSES = 24;
struct ChainClasses { int n; int nm[MAX_CHAIN_CLASSES]; };

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

__device__ __forceinline__ int ilog2_u32(uint32_t v) { return 31 - __clz((int)v); }

} // namespace mnc
