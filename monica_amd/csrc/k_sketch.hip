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
			const uint4 v = *reinterpret_cast<const uint4*>(B.bases + b0);
			const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
			for (int i = 0; i < 16; ++i) {
				uint32_t bad = 0;
				uint32_t c = pack_byte((w4[i >> 2] >> ((i & 3) * 8)) & 0xffu, bad);
				word |= c << (30 - 2 * i);
				badmask |= bad << i;
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
constexpr int SK_THREADS = 256;
constexpr int SK_CHUNK = 8192;                      // k-mer positions per LDS chunk
constexpr int SK_HALO = WIN - 1;

__device__ __forceinline__ uint32_t revcomp30(uint32_t fw)
{
	uint32_t r = __brev(~fw & KMASK);
	r = ((r & 0xAAAAAAAAu) >> 1) | ((r & 0x55555555u) << 1);
	return r >> 2;
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

// One 256-thread workgroup per read.  For a read without ambiguous bases the emitted set
// has a closed form: with n k-mers and hashes h[0..n),
//   n <  WIN : the right-most minimum of h[0..n)
//   n >= WIN : every p whose hash is the minimum of SOME full window containing p, i.e.
//              (run of h >= h[p] to the left, clipped at 0) + (same to the right, clipped
//              at n-1) + 1 >= WIN; plus the two first-window quirks of the serial machine:
//              with m' = min h[0..WIN-2] and P' its right-most position,
//                every p <= WIN-2, p != P', h[p] == m' IS emitted, and
//                P' is NOT emitted when h[WIN-1] == m'.
// Output order is increasing position, 8 bytes per minimizer: {hash, pos<<1 | strand},
// pos = index of the k-mer's last base.
__global__ __launch_bounds__(SK_THREADS) void mnc_sketch_minimizers(Batch B)
{
	__shared__ uint32_t s_hash[SK_CHUNK + 2 * SK_HALO];
	__shared__ uint32_t s_words[(SK_CHUNK + 2 * SK_HALO + KMER) / 16 + 4];
	__shared__ int s_wcnt[SK_THREADS / 64];

	const uint32_t r = blockIdx.x;
	const int tid = threadIdx.x;
	const int64_t off = B.offsets[r];
	const int len = (int)(B.offsets[r + 1] - off);
	const int n = len - (KMER - 1);
	uint2 *out = B.mz + off;
	if (n <= 0) { if (tid == 0) B.mz_cnt[r] = 0; return; }
	if (B.ambig[r]) {
		if (tid == 0) {
			B.mz_cnt[r] = sketch_serial(B.bases + off, len, out);
			atomicAdd((unsigned long long*)&B.stats[6], 1ULL);
		}
		return;
	}

	int total = 0;                                  // minimizers written so far (uniform)
	for (int c0 = 0; c0 < n; c0 += SK_CHUNK) {
		const int cend = min(c0 + SK_CHUNK, n);
		const int lo = max(c0 - SK_HALO, 0), hi = min(cend + SK_HALO, n);
		const int64_t w_lo = (off + lo) >> 4;
		const int n_words = (int)(((off + hi - 1 + KMER - 1) >> 4) - w_lo) + 2;
		__syncthreads();
		for (int i = tid; i < n_words; i += SK_THREADS) s_words[i] = B.packed[w_lo + i];
		__syncthreads();
		for (int p = lo + tid; p < hi; p += SK_THREADS) {
			const int64_t gb = off + p;
			const int wi = (int)((gb >> 4) - w_lo), sh = (int)(gb & 15) * 2;
			const uint64_t two = (uint64_t)s_words[wi] << 32 | s_words[wi + 1];
			const uint32_t fw = (uint32_t)(two >> (34 - sh)) & KMASK;
			const uint32_t rv = revcomp30(fw);
			const uint32_t strand = fw < rv ? 0u : 1u;
			s_hash[p - lo] = hash30(strand ? rv : fw) | strand << 31;
		}
		__syncthreads();

		// first-window quirk operands (chunk 0 only)
		uint32_t q_m = 0; int q_P = -1;
		if (c0 == 0 && n >= WIN) {
			q_m = s_hash[0] & KMASK, q_P = 0;
			for (int j = 1; j <= WIN - 2; ++j) {
				uint32_t h = s_hash[j] & KMASK;
				if (h <= q_m) q_m = h, q_P = j;
			}
		}

		for (int t0 = c0; t0 < cend; t0 += SK_THREADS) {
			const int p = t0 + tid;
			bool e = false;
			uint32_t hv = 0, h = 0;
			if (p < cend) {
				hv = s_hash[p - lo], h = hv & KMASK;
				if (n < WIN) {                          // no full window: right-most minimum
					e = true;
					for (int q = 0; q < n; ++q) {
						uint32_t o = s_hash[q - lo] & KMASK;
						if (q < p && o < h) e = false;
						if (q > p && o <= h) e = false;
					}
				} else {
					int L = 0, R = 0;
					bool okL = true, okR = true;
#pragma unroll
					for (int d = 1; d < WIN; ++d) {
						const int ql = p - d, qr = p + d;
						okL = okL && ql >= 0 && (s_hash[max(ql, lo) - lo] & KMASK) >= h;
						okR = okR && qr < n && (s_hash[min(qr, hi - 1) - lo] & KMASK) >= h;
						L += okL, R += okR;
					}
					e = L + R + 1 >= WIN;
					if (p <= WIN - 2) {
						if (h == q_m && p != q_P) e = true;
						if (p == q_P && (s_hash[WIN - 1] & KMASK) == q_m) e = false;
					}
				}
			}
			const unsigned long long m = __ballot(e);
			const int wv = tid >> 6;
			if ((tid & 63) == 0) s_wcnt[wv] = __popcll(m);
			__syncthreads();
			int base = total, all = 0;
#pragma unroll
			for (int w = 0; w < SK_THREADS / 64; ++w) {
				if (w < wv) base += s_wcnt[w];
				all += s_wcnt[w];
			}
			if (e) {
				const int rank = __popcll(m & ((1ULL << (tid & 63)) - 1ULL));
				out[base + rank] = make_uint2(h, (uint32_t)(p + KMER - 1) << 1 | hv >> 31);
			}
			total += all;
			__syncthreads();
		}
	}
	if (tid == 0) B.mz_cnt[r] = total;
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
	hipLaunchKernelGGL(mnc_sketch_minimizers, dim3(B.n_reads), dim3(SK_THREADS), 0, st, B);
}

} // namespace mnc
