// Workload generator on the device: the reads of synth.cpp (mnc_synth_reads), byte for byte, written
// straight into HBM -- gfx950.
//
// BASELINE config 3 is 10 M reads of 5 kb (50 GB of bases): the host generator makes them at ~0.3 GB/s
// and they would then cross PCIe; a read is a pure function of (seed, ordinal), so the device makes its
// own.  Measurement infrastructure (bench.py, tests), not part of the classified path: nothing in
// monica/genomes/aligner.py corresponds to it (its reads come from the sequencer's FASTQ files,
// aligner.py:191, 212).
//
// One wave per read.  The read's draws are independent by index j (counter-based SplitMix64), only the
// source position depends on the draws before it: position(j) = start + #(draws before j that consume
// a genome base) and output slot(j) = #(draws before j that emit a base) -- two prefix sums over 64
// draws at a time, a ballot and a popcount each.
#include <hip/hip_runtime.h>

#include "common.h"

namespace mnc {

namespace {

constexpr uint64_t SY_GAMMA = 0x9E3779B97F4A7C15ULL;

__device__ __forceinline__ uint64_t sy_mix(uint64_t z)
{
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
	return z ^ (z >> 31);
}
__device__ __forceinline__ uint64_t sy_draw(uint64_t seed, uint64_t i) { return sy_mix(seed + (i + 1) * SY_GAMMA); }
__device__ __forceinline__ uint32_t sy_below(uint64_t u, uint32_t n) { return (uint32_t)(((u >> 32) * (uint64_t)n) >> 32); }
__device__ __forceinline__ uint32_t sy_per(uint64_t u, uint32_t scale) { return (uint32_t)(((u >> 40) * (uint64_t)scale) >> 24); }
__device__ __forceinline__ int sy_code(uint8_t c)
{
	switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1;
	             case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
__device__ __forceinline__ uint8_t sy_base(int c) { return (uint8_t)("ACGT"[c & 3]); }

__global__ __launch_bounds__(64) void mnc_synth_reads_k(int n_genomes, const uint8_t *genomes, const int64_t *g_off,
                                                        uint64_t seed, int64_t first, int n_reads, int read_len,
                                                        uint32_t t_del, uint32_t t_ins, uint32_t t_sub, uint32_t random_frac,
                                                        uint8_t *out_bases, int32_t *out_truth)
{
	const int lane = threadIdx.x;
	for (int r = blockIdx.x; r < n_reads; r += gridDim.x) {
		const uint64_t rs = sy_mix(seed + sy_mix((uint64_t)(first + r)));
		uint8_t *o = out_bases + (int64_t)r * read_len;
		if (sy_per(sy_draw(rs, 0), 10000) < random_frac) {
			for (int j = lane; j < read_len; j += 64) o[j] = sy_base((int)(sy_draw(rs, 8 + (uint64_t)j) >> 62));
			if (out_truth && lane == 0) out_truth[r] = -1;
			continue;
		}
		const int g = (int)sy_below(sy_draw(rs, 1), (uint32_t)n_genomes);
		const int64_t glen = g_off[g + 1] - g_off[g];
		const int64_t span = (int64_t)read_len + read_len / 4 + 64;
		const int64_t max_start = glen > span ? glen - span : 0;
		const int64_t s0 = (int64_t)__umul64hi(sy_draw(rs, 2), (uint64_t)(max_start + 1));
		const bool rev = sy_draw(rs, 3) >> 63;
		const uint8_t *src = genomes + g_off[g];
		int64_t s = s0;                                            // source position before this block of draws
		int n = 0;                                                 // bases emitted before it
		for (uint64_t j0 = 0; n < read_len; j0 += 64) {
			const uint64_t u = sy_draw(rs, 8 + j0 + lane);
			const uint32_t t = sy_per(u, 10000);
			// The host loop: past the contig end every draw emits a random base and consumes nothing; otherwise a
			// deletion consumes, an insertion emits, anything else does both.  Whether a draw is "past the end"
			// depends on the position it sees, which depends on the draws before it -- but once s reaches glen it
			// stays there, and before that the position is start + consumed so far: the first lane (if any) whose
			// position reaches glen is found from the in-contig prefix sum.
			const bool is_del = t < t_del, is_ins = !is_del && t < t_ins;
			const unsigned long long cons_mask = __ballot(!is_ins);    // consumes a genome base (if in the contig)
			const int cons_before = __popcll(cons_mask & ((1ull << lane) - 1ull));
			const bool off_end = s + cons_before >= glen;              // monotone in the lane
			const bool emits = off_end || !is_del;
			const unsigned long long emit_mask = __ballot(emits);
			const int slot = n + __popcll(emit_mask & ((1ull << lane) - 1ull));
			if (emits && slot < read_len) {
				uint8_t ch;
				if (off_end || is_ins) ch = sy_base((int)((u >> 8) & 3));
				else {
					int c = sy_code(src[s + cons_before]);
					if (c > 3) ch = 'N';
					else {
						if (t < t_sub) c = (c + 1 + (int)(((u >> 8) & 0xffff) % 3)) & 3;
						ch = sy_base(c);
					}
				}
				// the reverse strand: the host reverses and complements in place at the end
				if (rev) {
					const int cc = sy_code(ch);
					o[read_len - 1 - slot] = cc > 3 ? 'N' : sy_base(3 - cc);
				} else o[slot] = ch;
			}
			// Draws after the one that filled the read are never made by the host; they change nothing here either
			// (their slots are >= read_len).  Advance by what the whole block consumed / emitted, clipped like the host.
			const unsigned long long in_contig = ~__ballot(off_end);
			s += __popcll(cons_mask & in_contig);
			n += __popcll(emit_mask);
		}
		if (out_truth && lane == 0) out_truth[r] = g;
	}
}

} // namespace

} // namespace mnc

// genomes: the contigs concatenated (ASCII), g_off[n_genomes + 1] their starts; everything on the device.
extern "C" int mnc_synth_reads_device(int n_genomes, const uint8_t *d_genomes, const int64_t *d_g_off,
                                      uint64_t seed, int64_t first, int n_reads, int read_len,
                                      int sub_e4, int ins_e4, int del_e4, int random_frac_e4,
                                      uint8_t *d_out_bases, int32_t *d_out_truth, void *stream)
{
	if (n_genomes <= 0 || !d_genomes || !d_g_off || n_reads < 0 || read_len <= 0 || !d_out_bases) return MNC_ERR_ARG;
	if (sub_e4 < 0 || ins_e4 < 0 || del_e4 < 0 || sub_e4 + ins_e4 + del_e4 > 9000) return MNC_ERR_ARG;
	if (n_reads == 0) return MNC_OK;
	const uint32_t t_del = (uint32_t)del_e4, t_ins = t_del + (uint32_t)ins_e4, t_sub = t_ins + (uint32_t)sub_e4;
	const int grid = n_reads < 65536 ? n_reads : 65536;
	hipLaunchKernelGGL(mnc::mnc_synth_reads_k, dim3(grid), dim3(64), 0, (hipStream_t)stream, n_genomes, d_genomes, d_g_off, seed, first,
	                   n_reads, read_len, t_del, t_ins, t_sub, (uint32_t)random_frac_e4, d_out_bases, d_out_truth);
	const hipError_t e = hipGetLastError();
	if (e != hipSuccess) { mnc::set_error("mnc_synth_reads_k launch failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}
