// Internal declarations shared by the host and device translation units of
// libmonica_amd.so.  Public surface: include/monica_amd.h.
#pragma once

#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>
#include <mutex>

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
};

// ---------------------------------------------------------------- HBM table
// Open-addressed, power-of-two number of 64-byte lines, 4 keys per line, linear probing over
// LINES: a lookup reads the 16 bytes of keys of its home line (hash & line_mask) and stops
// there unless that line is full (rare at load <= 0.5), so the longest probe chain inside a
// wave stays at one or two round trips (with 16-byte slots the slowest of 64 lanes needed 5-6).
//   key  = hash + 1 (0 marks an empty slot; a line fills from slot 0 upwards)
//   cv   = {occurrences, value}: value = the occurrence word itself (rid<<32 | pos<<1 | strand)
//          when occurrences == 1, else the offset of the first occurrence in positions[]
// Queries that are not in the table at all (3 of 4) are stopped before the gather by a
// per-region presence filter held in LDS (k_probe.hip).
struct TableCV { uint32_t cnt, val_lo, val_hi; };   // 12 bytes: one dwordx3 request
struct alignas(64) TableLine {
	uint32_t key[4];
	TableCV cv[4];
};

struct DeviceIndex {          // one per (index, device)
	int device = -1;
	TableLine *table = nullptr;
	uint64_t table_mask = 0;       // number of lines - 1
	uint32_t *filter = nullptr;    // per table region: 2^18-bit presence filter (see k_probe.hip)
	uint64_t *positions = nullptr;
	int32_t *contig_genome = nullptr;
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
	int32_t mid_occ = 0;
	int64_t total_len = 0;
	mnc::MapParams par;
	// device residency
	std::mutex dev_mutex;
	std::vector<mnc::DeviceIndex> dev;
};

namespace mnc {
// index.cpp
int index_finalize(mnc_index *idx, std::vector<std::pair<uint64_t, uint64_t>> &pairs);
int cal_mid_occ(const mnc_index *idx, float f);
// engine side
int index_upload(mnc_index *idx, int device, DeviceIndex **out);
void index_release_device(mnc_index *idx);
} // namespace mnc
