// Deterministic synthetic genomes and nanopore-like reads.
// Counter-based SplitMix64: every value is a pure function of (seed, index), so any
// slice can be generated independently and identically on any machine.
#include <cstring>
#include <vector>
#include "common.h"

namespace {

constexpr uint64_t GAMMA = 0x9E3779B97F4A7C15ULL;

inline uint64_t mix(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
inline uint64_t draw(uint64_t seed, uint64_t i) { return mix(seed + (i + 1) * GAMMA); }
inline uint32_t below(uint64_t u, uint32_t n) { return (uint32_t)(((u >> 32) * (uint64_t)n) >> 32); }
inline uint32_t per(uint64_t u, uint32_t scale) { return (uint32_t)(((u >> 40) * (uint64_t)scale) >> 24); }

const char ACGT[5] = "ACGT";
inline int code(char c)
{
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
	             case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}

} // namespace

extern "C" int mnc_synth_genome(uint64_t seed, int64_t len, char *out)
{
	if (len < 0 || (len > 0 && !out)) return MNC_ERR_ARG;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < len; ++i) out[i] = ACGT[draw(seed, (uint64_t)i) >> 62];
	return MNC_OK;
}

extern "C" int mnc_synth_diverge(const char *src, int64_t len, uint64_t seed, int rate_ppm, char *out)
{
	if (len < 0 || (len > 0 && (!src || !out)) || rate_ppm < 0 || rate_ppm > 1000000) return MNC_ERR_ARG;
#pragma omp parallel for schedule(static)
	for (int64_t i = 0; i < len; ++i) {
		uint64_t u = draw(seed, (uint64_t)i);
		int c = code(src[i]);
		if (c < 4 && per(u, 1000000) < (uint32_t)rate_ppm) out[i] = ACGT[(c + 1 + (int)((u & 0xffff) % 3)) & 3];
		else out[i] = src[i];
	}
	return MNC_OK;
}

extern "C" int mnc_synth_reads(int n_genomes, const char *const *genomes, const int64_t *lens,
                               uint64_t seed, int64_t first, int n_reads, int read_len,
                               int sub_e4, int ins_e4, int del_e4, int random_frac_e4,
                               char *out_bases, int32_t *out_truth)
{
	if (n_genomes <= 0 || !genomes || !lens || n_reads < 0 || read_len <= 0 || !out_bases) return MNC_ERR_ARG;
	if (sub_e4 < 0 || ins_e4 < 0 || del_e4 < 0 || sub_e4 + ins_e4 + del_e4 > 9000) return MNC_ERR_ARG;
	const uint32_t t_del = (uint32_t)del_e4, t_ins = t_del + (uint32_t)ins_e4, t_sub = t_ins + (uint32_t)sub_e4;
#pragma omp parallel for schedule(dynamic, 64)
	for (int r = 0; r < n_reads; ++r) {
		const uint64_t rs = mix(seed + mix((uint64_t)(first + r)));
		char *o = out_bases + (int64_t)r * read_len;
		if (per(draw(rs, 0), 10000) < (uint32_t)random_frac_e4) {
			for (int j = 0; j < read_len; ++j) o[j] = ACGT[draw(rs, 8 + (uint64_t)j) >> 62];
			if (out_truth) out_truth[r] = -1;
			continue;
		}
		const int g = (int)below(draw(rs, 1), (uint32_t)n_genomes);
		const int64_t glen = lens[g];
		const int64_t span = (int64_t)read_len + read_len / 4 + 64;
		const int64_t max_start = glen > span ? glen - span : 0;
		int64_t s = (int64_t)(((__uint128_t)draw(rs, 2) * (uint64_t)(max_start + 1)) >> 64);
		const bool rev = draw(rs, 3) >> 63;
		const char *src = genomes[g];
		int n = 0;
		for (uint64_t j = 0; n < read_len; ++j) {
			uint64_t u = draw(rs, 8 + j);
			uint32_t t = per(u, 10000);
			if (s >= glen) { o[n++] = ACGT[(u >> 8) & 3]; continue; }     // ran off the contig end
			if (t < t_del) { ++s; continue; }
			if (t < t_ins) { o[n++] = ACGT[(u >> 8) & 3]; continue; }
			int c = code(src[s++]);
			if (c > 3) { o[n++] = 'N'; continue; }
			if (t < t_sub) c = (c + 1 + (int)(((u >> 8) & 0xffff) % 3)) & 3;
			o[n++] = ACGT[c];
		}
		if (rev) {
			for (int a = 0, b = read_len - 1; a <= b; ++a, --b) {
				int ca = code(o[a]), cb = code(o[b]);
				char xa = cb > 3 ? 'N' : ACGT[3 - cb], xb = ca > 3 ? 'N' : ACGT[3 - ca];
				o[a] = xa, o[b] = xb;
			}
		}
		if (out_truth) out_truth[r] = g;
	}
	return MNC_OK;
}
