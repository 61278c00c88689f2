// Stage kernels K0 (2-bit packing) and K1 (minimizer sketch) -- gfx950.
//
// Replaces the query-side mm_sketch() that runs inside every index.map(seq) call of
// monica/genomes/aligner.py:193,215.
#include "device.h"

namespace mnc {

// ================================================================ K0: ASCII -> 2-bit
// One thread packs 16 bases (one 16-byte load) into one word, first base in the top bits.
// A byte outside ACGTUacgtu marks its read as "ambiguous"; such reads take the serial
// sketch path below, which reads the ASCII bytes directly.
__device__ __forceinline__ uint32_t pack_byte(uint32_t c, uint32_t &bad)
{
	uint32_t u = c & 0xDFu;                         // upper-case
	bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T') | (u == 'U');
	bad |= ok ? 0u : 1u;
	uint32_t x = (u >> 1) & 3u;                     // A0 C1 T2 G3
	return x ^ (x >> 1);                            // A0 C1 G2 T3
}

__device__ int64_t read_of_base(const int64_t *offsets, uint32_t n_reads, int64_t b)
{
	// largest r with offsets[r] <= b and offsets[r+1] > b
	int64_t lo = 0, hi = (int64_t)n_reads;          // answer in [lo, hi)
	while (hi - lo > 1) {
		int64_t mid = (lo + hi) >> 1;
		if (offsets[mid] <= b) lo = mid; else hi = mid;
	}
	return lo;
}

__global__ __launch_bounds__(256) void mnc_pack_bases(Batch B)
{
	const int64_t n_groups = (B.total_bases + 15) >> 4;
	const int64_t stride = (int64_t)gridDim.x * blockDim.x;
	for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += stride) {
		const int64_t b0 = g << 4;
		uint32_t word = 0, badmask = 0;
		if (b0 + 16 <= B.total_bases) {
			// four bases per 32-bit operation: upper-case, 2-bit code = x ^ (x >> 1) with
			// x = bits 1-2 of the letter (A0 C1 T2 G3 -> A0 C1 G2 T3), letters outside ACGTU
			// found with the zero-byte test, codes gathered into one byte by a multiply
			const uint4 v = *reinterpret_cast<const uint4*>(B.bases + b0);
			const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const uint32_t u = w4[k] & 0xDFDFDFDFu;
				const uint32_t x = (u >> 1) & 0x03030303u;
				const uint32_t code = x ^ ((x >> 1) & 0x01010101u);
				uint32_t okb = 0;                                  // 0x80 in every byte that is one of A C G T U
#pragma unroll
				for (int t = 0; t < 5; ++t) {
					const uint32_t z = u ^ (0x01010101u * (uint32_t)("ACGTU"[t]));
					okb |= ~(((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
				}
				const uint32_t bad4 = ~okb & 0x80808080u;          // first base of the word = lowest byte
				badmask |= ((bad4 >> 7 & 1u) | (bad4 >> 14 & 2u) | (bad4 >> 21 & 4u) | (bad4 >> 28 & 8u)) << (4 * k);
				word |= ((code * 0x40100401u) >> 24) << (24 - 8 * k);
			}
		} else {
			for (int i = 0; i < 16 && b0 + i < B.total_bases; ++i) {
				uint32_t bad = 0;
				uint32_t c = pack_byte(B.bases[b0 + i], bad);
				word |= c << (30 - 2 * i);
				badmask |= bad << i;
			}
		}
		B.packed[g] = word;
		while (badmask) {
			int i = __ffs((int)badmask) - 1;
			badmask &= badmask - 1;
			B.ambig[read_of_base(B.offsets, B.n_reads, b0 + i)] = 1u;
		}
	}
}

// ================================================================ K1: minimizers
constexpr int SK_THREADS = 256;                     // 4 waves = 4 reads per workgroup
constexpr int SK_PER = 17;                          // positions per lane: an odd stride, so no LDS bank conflicts
constexpr int SK_CHUNK = 64 * SK_PER;               // k-mer positions per LDS chunk of one wave
constexpr int SK_PAD = 10;                          // halo on each side (>= WIN - 1)
constexpr int SK_SLOTS = SK_CHUNK + 2 * SK_PAD + 4; // hash slots per wave: positions c0-PAD .. c0+CHUNK+PAD
constexpr int SK_EXTRA = SK_SLOTS - SK_CHUNK;       // slots beyond the 64 x 17 the lanes hash in their main pass
constexpr int SK_WORDS = (SK_CHUNK + 2 * SK_PAD + KMER + 32) / 16 + 4;
static_assert(SK_THREADS / 64 == PT_READS, "one sketch workgroup = one partition tile");

__device__ __forceinline__ uint32_t revcomp30(uint32_t fw)
{
	uint32_t r = __brev(~fw & KMASK);
	r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
	return r >> 2;
}

__device__ __forceinline__ void wave_lds_order()
{
	// LDS operations of one wave execute in issue order; this only pins the compiler
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	asm volatile("" ::: "memory");
}

// Serial minimizer state machine for reads that hold ambiguous bases:
// ring of the last WIN records; newest of equal hashes wins; every record equal to the
// window minimum is reported; an ambiguous base restarts the k-mer run only.
__device__ int sketch_serial(const uint8_t *s, int len, uint2 *out)
{
	const uint64_t NONE = ~0ULL;
	uint64_t ring_h[WIN], ring_y[WIN];
	uint64_t min_h = NONE, min_y = NONE;
	uint32_t fw = 0, rv = 0;
	int run = 0, at = 0, min_at = 0, n_out = 0;
	for (int j = 0; j < WIN; ++j) ring_h[j] = NONE, ring_y[j] = NONE;
#define MNC_EMIT(H, Y) (out[n_out++] = make_uint2((uint32_t)(H), (uint32_t)(Y)))
	for (int i = 0; i < len; ++i) {
		uint32_t bad = 0;
		uint32_t c = pack_byte(s[i], bad);
		uint64_t cur_h = NONE, cur_y = NONE;
		if (!bad) {
			fw = (fw << 2 | c) & KMASK;
			rv = rv >> 2 | (3u ^ c) << (2 * (KMER - 1));
			int strand = fw < rv ? 0 : 1;
			++run;
			if (run >= KMER) {
				cur_h = hash30(strand ? rv : fw);
				cur_y = (uint64_t)(uint32_t)i << 1 | (uint64_t)strand;
			}
		} else run = 0;
		ring_h[at] = cur_h, ring_y[at] = cur_y;
		if (run == WIN + KMER - 1 && min_h != NONE) {
			for (int j = at + 1; j < WIN; ++j) if (ring_h[j] == min_h && ring_y[j] != min_y) MNC_EMIT(ring_h[j], ring_y[j]);
			for (int j = 0; j < at; ++j)       if (ring_h[j] == min_h && ring_y[j] != min_y) MNC_EMIT(ring_h[j], ring_y[j]);
		}
		if (cur_h <= min_h) {
			if (run >= WIN + KMER && min_h != NONE) MNC_EMIT(min_h, min_y);
			min_h = cur_h, min_y = cur_y, min_at = at;
		} else if (at == min_at) {
			if (run >= WIN + KMER - 1 && min_h != NONE) MNC_EMIT(min_h, min_y);
			min_h = NONE;
			for (int j = at + 1; j < WIN; ++j) if (min_h >= ring_h[j]) min_h = ring_h[j], min_y = ring_y[j], min_at = j;
			for (int j = 0; j <= at; ++j)      if (min_h >= ring_h[j]) min_h = ring_h[j], min_y = ring_y[j], min_at = j;
			if (run >= WIN + KMER - 1 && min_h != NONE) {
				for (int j = at + 1; j < WIN; ++j) if (ring_h[j] == min_h && ring_y[j] != min_y) MNC_EMIT(ring_h[j], ring_y[j]);
				for (int j = 0; j <= at; ++j)      if (ring_h[j] == min_h && ring_y[j] != min_y) MNC_EMIT(ring_h[j], ring_y[j]);
			}
		}
		if (++at == WIN) at = 0;
	}
	if (min_h != NONE) MNC_EMIT(min_h, min_y);
#undef MNC_EMIT
	return n_out;
}

// One wave per read (4 reads per workgroup, no workgroup barrier).  For a read without
// ambiguous bases the emitted set has a closed form: with n k-mers
// and hashes h[0..n),
//   n <  WIN : the right-most minimum of h[0..n)
//   n >= WIN : every p whose hash is the minimum of SOME full window containing p, i.e.
//              (run of h >= h[p] to the left, clipped at 0) + (same to the right, clipped
//              at n-1) + 1 >= WIN; plus the two first-window quirks of the serial machine:
//              with m' = min h[0..WIN-2] and P' its right-most position,
//                every p <= WIN-2, p != P', h[p] == m' IS emitted, and
//                P' is NOT emitted when h[WIN-1] == m'.
// Hashes of a chunk sit in LDS with a -1 sentinel outside [0, n) (so clipping needs no
// test); a lane hashes 17 adjacent k-mers out of two 64-bit registers and decides 17 adjacent
// positions from the 35 hashes around them, held in registers (window minima, then their
// maxima over the ten windows containing a position).
// Output order is increasing position, 8 bytes per minimizer: {hash, pos<<1 | strand},
// pos = index of the k-mer's last base.
__device__ void sketch_one_read(const Batch &B, uint32_t r, int wv, int lane, int32_t *s_hash, uint32_t *s_words,
                                uint32_t *s_strand, uint32_t *s_hist);

__global__ __launch_bounds__(SK_THREADS) void mnc_sketch_minimizers(Batch B)
{
	__shared__ __align__(16) int32_t s_hash_all[SK_THREADS / 64][SK_SLOTS];
	__shared__ uint32_t s_words_all[SK_THREADS / 64][SK_WORDS];
	__shared__ uint32_t s_strand_all[SK_THREADS / 64][64 + SK_EXTRA];   // per hashing lane: strand bits of its 17 positions
	__shared__ uint32_t s_hist[PB_N_MAX];         // minimizers of this tile per table bucket

	const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t r = blockIdx.x * (SK_THREADS / 64) + wv;
	for (int k = threadIdx.x; k < (int)B.pb_n; k += SK_THREADS) s_hist[k] = 0;
	__syncthreads();
	sketch_one_read(B, r, wv, lane, s_hash_all[wv], s_words_all[wv], s_strand_all[wv], s_hist);
	__syncthreads();
	for (int k = threadIdx.x; k < (int)B.pb_n; k += SK_THREADS) B.hist_tm[(size_t)blockIdx.x * B.pb_n + k] = s_hist[k];
}

__device__ void sketch_one_read(const Batch &B, uint32_t r, int wv, int lane, int32_t *s_hash, uint32_t *s_words,
                                uint32_t *s_strand, uint32_t *s_hist)
{
	if (r >= B.n_reads) return;
	const int64_t off = B.offsets[r];
	const int len = (int)(B.offsets[r + 1] - off);
	const int n = len - (KMER - 1);
	uint2 *out = B.mz + off;
	if (n <= 0) { if (lane == 0) B.mz_cnt[r] = 0; return; }
	if (B.ambig[r]) return;                         // mnc_sketch_ambiguous handles it
	const int64_t n_packed = (B.total_bases + 15) >> 4;
	int total = 0;                                  // minimizers written so far (uniform)
	for (int c0 = 0; c0 < n; c0 += SK_CHUNK) {
		const int cend = min(c0 + SK_CHUNK, n);
		// slot q of the window holds position c0 - PAD + q; the packed words of the window start
		// at the word of its first base (before the batch's first base: zeros)
		const int64_t g0 = off + c0 - SK_PAD;                           // global base index of slot 0 (may be < 0)
		const int64_t wbase = g0 >> 4;                                  // floor
		const int bo0 = (int)(g0 - wbase * 16);                         // 0..15
		wave_lds_order();
		for (int i = lane; i < SK_WORDS; i += 64) {
			const int64_t w = wbase + i;
			s_words[i] = w >= 0 && w < n_packed ? B.packed[w] : 0u;
		}
		wave_lds_order();
		// ---- hashing: lane l takes the 17 slots q = 17 l + t.  The 31 bases they span are cut out
		// of the packed stream once (first base in the top bits), and so is their reverse
		// complement; every k-mer and its reverse complement are then 30-bit fields of those.
		{
			const int bo = bo0 + SK_PER * lane;
			const int wi = bo >> 4, sh = (bo & 15) * 2;
			const uint32_t w0 = s_words[wi], w1 = s_words[wi + 1], w2 = s_words[wi + 2];
			const uint64_t seq = (((uint64_t)w0 << 32 | w1) << sh & 0xffffffff00000000ULL) | (((uint64_t)w1 << 32 | w2) << sh) >> 32;
			uint64_t rc = __brevll(~seq);                               // complement, bit-reversed ...
			rc = ((rc & 0xAAAAAAAAAAAAAAAAULL) >> 1) | ((rc & 0x5555555555555555ULL) << 1);   // ... with the two bits of a base back in order
			uint32_t sbits = 0;
			int32_t *dst = s_hash + SK_PER * lane;
#pragma unroll
			for (int t = 0; t < SK_PER; ++t) {
				const uint32_t fw = (uint32_t)(seq >> (34 - 2 * t)) & KMASK;
				const uint32_t rv = (uint32_t)(rc >> (2 * t)) & KMASK;      // reverse complement of bases t .. t+14
				sbits |= (fw < rv ? 0u : 1u) << t;
				dst[t] = (int32_t)hash30(min(fw, rv));
			}
			s_strand[lane] = sbits;
		}
		// the few slots beyond 64 x 17, one per lane
		if (lane < SK_EXTRA) {
			const int bo = bo0 + SK_CHUNK + lane;
			const int wi = bo >> 4, sh = (bo & 15) * 2;
			const uint64_t two = (uint64_t)s_words[wi] << 32 | s_words[wi + 1];
			const uint32_t fw = (uint32_t)(two >> (34 - sh)) & KMASK;
			const uint32_t rv = revcomp30(fw);
			s_hash[SK_CHUNK + lane] = (int32_t)hash30(min(fw, rv));
			s_strand[64 + lane] = fw < rv ? 0u : 1u;
		}
		wave_lds_order();
		// positions outside [0, n) hold the -1 sentinel (smaller than every hash)
		if (c0 == 0 && lane < SK_PAD) s_hash[lane] = -1;
		for (int q = n - c0 + SK_PAD + lane; q < SK_SLOTS; q += 64) s_hash[q] = -1;
		wave_lds_order();

		// ---- selection: lane l owns the 17 positions c0 + 17 l + t.  With m(s) the minimum of
		// the window starting at s, position p is reported iff h[p] == max over the ten windows
		// that contain it (h[p] >= each of those minima, with equality iff p is a minimum of it).
		int32_t v[37];
		{
			const int32_t *src = s_hash + SK_PER * lane;
#pragma unroll
			for (int k = 1; k <= SK_PER + 18; ++k) v[k] = src[k];
		}
		uint32_t mask = 0;
		{
			int32_t m3[SK_PER + 17], m10[SK_PER + 10];
#pragma unroll
			for (int k = 1; k <= SK_PER + 16; ++k) m3[k] = min(min(v[k], v[k + 1]), v[k + 2]);
#pragma unroll
			for (int k = 1; k <= SK_PER + 9; ++k) m10[k] = min(min(min(m3[k], m3[k + 3]), m3[k + 6]), v[k + 9]);
			int32_t x3[SK_PER + 8];
#pragma unroll
			for (int k = 1; k <= SK_PER + 7; ++k) x3[k] = max(max(m10[k], m10[k + 1]), m10[k + 2]);
#pragma unroll
			for (int t = 0; t < SK_PER; ++t) {
				const int k = t + 1;
				const int32_t top = max(max(max(x3[k], x3[k + 3]), x3[k + 6]), m10[k + 9]);
				mask |= v[t + 10] == top ? 1u << t : 0u;
			}
		}
		const int first_p = c0 + SK_PER * lane;
		if (c0 == 0 && lane == 0) {                     // short reads and the first-window quirks
			if (n < WIN) {
				int32_t lowest = v[10];
				int P = 0;
#pragma unroll
				for (int k = 1; k <= WIN - 2; ++k) if (k < n && v[10 + k] <= lowest) lowest = v[10 + k], P = k;   // right-most minimum
				mask = 1u << P;
			} else {
				int32_t q_m = v[10];
				int q_P = 0;
#pragma unroll
				for (int k = 1; k <= WIN - 2; ++k) if (v[10 + k] <= q_m) q_m = v[10 + k], q_P = k;
				const bool drop = v[10 + WIN - 1] == q_m;
#pragma unroll
				for (int k = 0; k <= WIN - 2; ++k) {
					if (v[10 + k] == q_m && k != q_P) mask |= 1u << k;
					if (k == q_P && drop) mask &= ~(1u << k);
				}
			}
		}
		{
			const int nvalid = min(max(cend - first_p, 0), SK_PER);
			mask &= (1u << nvalid) - 1u;
		}
		// ---- output in position order: lane-contiguous runs
		const int cnt = __popc(mask);
		int incl = cnt;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const int o = __shfl_up(incl, d);
			if (lane >= d) incl += o;
		}
		int k_out = total + incl - cnt;
		total += __builtin_amdgcn_readlane(incl, 63);
		while (mask) {
			const int t = __ffs((int)mask) - 1;
			mask &= mask - 1;
			const int q = SK_PER * lane + SK_PAD + t;                   // its hash was made by lane q / 17 (or the extra pass)
			const int32_t h = s_hash[q];
			const int hl = t < SK_PER - SK_PAD ? lane : lane + 1, hb = t < SK_PER - SK_PAD ? t + SK_PAD : t + SK_PAD - SK_PER;
			const uint32_t strand = q < SK_CHUNK ? (s_strand[hl] >> hb) & 1u : s_strand[64 + q - SK_CHUNK];
			atomicAdd(&s_hist[pb_bucket((uint32_t)h, B.pb_bits)], 1u);
			out[k_out++] = make_uint2((uint32_t)h, (uint32_t)(first_p + t + KMER - 1) << 1 | strand);
		}
	}
	if (lane == 0) B.mz_cnt[r] = total;
}

// Reads with ambiguous bases: the serial state machine, one thread per such read.  Runs
// after mnc_sketch_minimizers, so it adds its bucket counts to the tile's histogram row.
__global__ __launch_bounds__(64) void mnc_sketch_ambiguous(Batch B)
{
	const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= B.n_reads || !B.ambig[r]) return;
	const int64_t off = B.offsets[r];
	const int len = (int)(B.offsets[r + 1] - off);
	if (len < KMER) return;                         // mz_cnt already 0
	uint2 *out = B.mz + off;
	const int n = sketch_serial(B.bases + off, len, out);
	B.mz_cnt[r] = n;
	uint32_t *row = B.hist_tm + (size_t)(r / PT_READS) * B.pb_n;
	for (int i = 0; i < n; ++i) atomicAdd(&row[pb_bucket(out[i].x, B.pb_bits)], 1u);
}

void launch_pack(const Batch &B, hipStream_t st)
{
	const int64_t n_groups = (B.total_bases + 15) >> 4;
	if (n_groups == 0) return;
	int64_t blocks = (n_groups + 255) / 256;
	if (blocks > 256 * 16) blocks = 256 * 16;
	hipLaunchKernelGGL(mnc_pack_bases, dim3((unsigned)blocks), dim3(256), 0, st, B);
}

void launch_sketch(const Batch &B, hipStream_t st)
{
	if (B.n_reads == 0) return;
	const unsigned blocks = (B.n_reads + SK_THREADS / 64 - 1) / (SK_THREADS / 64);
	hipLaunchKernelGGL(mnc_sketch_minimizers, dim3(blocks), dim3(SK_THREADS), 0, st, B);
	hipLaunchKernelGGL(mnc_sketch_ambiguous, dim3((B.n_reads + 63) / 64), dim3(64), 0, st, B);
}

} // namespace mnc
