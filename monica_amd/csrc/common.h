// Internal declarations shared by the host and device translation units of
// libmonica_amd.so.  Public surface: include/monica_amd.h.
#pragma once

#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>
#include <mutex>
#include <deque>

#include "../../include/monica_amd.h"

namespace mnc {

// ---------------------------------------------------------------- errors
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

// ---------------------------------------------------------------- mapping parameters
// What mappy 2.17 uses when an index is loaded with no preset.
struct MapParams {
	int seed = 11;
	float mid_occ_frac = 2e-4f;
	int min_cnt = 3;
	int min_chain_score = 40;
	int bw = 500;
	int max_gap = 5000;
	int max_chain_skip = 25;
	int max_chain_iter = 5000;
	float mask_level = 0.5f;
	float pri_ratio = 0.8f;
	int best_n = 5;
	int max_join_long = 20000;
	int max_join_short = 2000;
	int min_join_flank_sc = 1000;
	float min_join_flank_ratio = 0.5f;
	// base-level alignment (mappy ORs MM_F_CIGAR in)
	int a = 2, b = 4, q = 4, e = 2, q2 = 24, e2 = 1, sc_ambi = 1;
	int zdrop = 400, zdrop_inv = 200, end_bonus = -1;
	int min_dp_max = 80;              // min_chain_score * a
	int min_ksw_len = 200;
	long long max_sw_mat = 100000000;
};

// ---------------------------------------------------------------- HBM table
// The 30-bit minimizer hash splits into a REGION (low PB_BITS = 8 bits) and a 22-bit REST.
// Every region owns a power-of-two block of 16-byte slots {key = hash+1, cnt, val} (load
// <= 0.5) and is perfectly hashed by hash-and-displace: the rest picks one of NB displacement
// buckets (rest & (NB-1)), a base slot and an odd step (two multiplicative mixes of the rest);
// the key sits at slot (base + disp[bucket] * step) & (R-1).  So a lookup is: one 8-bit displacement (LDS),
// ONE 16-byte gather, one compare -- no probe chain, whose longest member would otherwise
// hold up the other 63 lanes of the wave.
//   cnt  = occurrences
//   val  = cnt == 1 : the occurrence word itself (rid<<32 | pos<<1 | strand)
//          cnt  > 1 : offset of the first occurrence word in the positions array
// Queries that are not in the table at all (3 of 4) are stopped before the gather by a
// per-region presence filter, also held in LDS (k_probe.hip).
struct alignas(16) TableSlot {
	uint32_t key;
	uint32_t cnt;
	uint64_t val;
};

struct DeviceIndex {          // one per (index, device)
	int device = -1;
	int pb_bits = 8;               // log2(regions): 8 .. 10, from the number of keys (index_upload)
	TableSlot *table = nullptr;    // regions x region_slots
	int region_bits = 0;           // log2(slots per region)
	int disp_bits = 0;             // log2(displacement buckets per region)
	uint8_t *disp = nullptr;       // [PB_N][1 << disp_bits]
	uint32_t *salt = nullptr;      // [PB_N] per-region salt of the slot function (pd_slot)
	uint32_t *filter = nullptr;    // per region: 2^PF_BITS-bit presence filter (see k_probe.hip)
	uint64_t *positions = nullptr;
	int32_t *contig_genome = nullptr;
	uint32_t *seq4 = nullptr;      // contig bases, 4 bits each, 8 per word (base-level DP stage)
	int64_t *seq_off = nullptr;    // [n_contigs + 1] first base of a contig in seq4
	int64_t bytes = 0;
};

} // namespace mnc

// The opaque handle of the C-ABI.
struct mnc_index {
	int k = 15, w = 10;
	std::vector<std::string> contig_name;
	std::vector<int64_t> contig_len;
	std::vector<int32_t> contig_genome;
	std::vector<std::string> genome_name;
	std::vector<int64_t> genome_len;
	// sorted-by-hash representation (also the on-disk form)
	std::vector<uint32_t> keys;       // distinct hashes, ascending
	std::vector<uint64_t> key_off;    // n_keys + 1 offsets into pos
	std::vector<uint64_t> pos;        // occurrence words, ascending inside one key
	// contig bases for the base-level DP stage: codes 0..3 = ACGT(U), 4 = anything else, 4 bits
	// each, base i of the concatenation at bits (i & 7) * 4 of seq4[i >> 3] (as minimap2 stores
	// them next to its minimizer table)
	std::vector<uint32_t> seq4;
	std::vector<int64_t> seq_off;     // n_contigs + 1
	int32_t mid_occ = 0;
	int64_t total_len = 0;
	mnc::MapParams par;
	// device residency
	std::mutex dev_mutex;
	std::deque<mnc::DeviceIndex> dev;   // a deque: engines keep pointers to its elements
	bool host_tables = false;           // test switch: build the device tables with the host form (mnc_index_set_host_tables)
	int force_pb_bits = 0;              // test switch: this many region bits whatever the size (mnc_index_set_region_bits); 0 = by size
};

namespace mnc {
// index.cpp
int index_finalize(mnc_index *idx, std::vector<std::pair<uint64_t, uint64_t>> &pairs);
void index_genome_table(mnc_index *idx);
void pack_contigs(mnc_index *idx, const char *const *seqs, const int64_t *lens, int n_seq);
int cal_mid_occ(const mnc_index *idx, float f);
// engine side
int index_upload(mnc_index *idx, int device, DeviceIndex **out);
void index_release_device(mnc_index *idx);
} // namespace mnc
