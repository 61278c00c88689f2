// Index build on the device: the sketch and the sort of mappy.Aligner(fn_idx_in=fasta, ...)
// (monica/genomes/aligner.py:45-46; minimap2's mm_idx_gen) -- gfx950.
//
// The host builder (index.cpp) sketches the contigs on all host threads and sorts 18 M (hash, occurrence) pairs:
// 1.5 s for the 94 Mbp index.  Here the contigs go to the device once and
//   1. are cut into pieces of 256 kb with 64 bases of context on either side (a minimizer is decided by the hashes
//      within w - 1 k-mers of it; the quirks of the scan's start only reach the first w + k bases of a sequence, inside
//      the context that is thrown away), every piece a "read" of the batch kernels: mnc_pack_bases + the sketch kernel
//      K1 (mnc_sketch_minimizers; pieces with ambiguous bases take its serial form);
//   2. a piece keeps the minimizers whose k-mer ends inside it, as occurrence words rid << 32 | pos << 1 | strand;
//   3. the pieces come in contig order and a piece's minimizers in position order, so the occurrence words are ascending
//      as they are written: ONE stable sort by the 30-bit hash gives minimap2's order -- by hash, positions ascending.
//      The sort is this file's own: least-significant-digit radix, 8 bits a pass (4 passes), per pass a histogram per
//      tile of 4 096 records, one scan, and a scatter that ranks a tile's records stably (wave w owns a contiguous
//      quarter of the tile; 64 records a round, equal digits found with eight ballots), sorts the tile in LDS and writes
//      runs of equal digits to consecutive addresses.  Run heads (hash != its left neighbour) and a scan of the head
//      flags give the distinct hashes and their offsets;
//   4. the occurrence count at rank (1 - mid_occ_frac) n of the sorted counts is mid_occ - 1 (A.3): a histogram of the
//      counts (LDS bins per workgroup), walked by the host.
// The arrays come back to the host index object (it saves them to the index file, dumps them for the tests) and the
// 4-bit contig bases are packed by the host meanwhile.  The result is the host builder's, array for array
// (tests/test_gpu_parity.py).
#include <algorithm>
#include <cstring>
#include <map>
#include <thread>

#include "device.h"
#include "scan.h"

namespace mnc {

void launch_pack(const Batch &B, hipStream_t st);
void launch_sketch(const Batch &B, hipStream_t st);
void pack_contigs(mnc_index *idx, const char *const *seqs, const int64_t *lens, int n_seq);   // index.cpp
void index_genome_table(mnc_index *idx);                                                     // index.cpp
int cal_mid_occ(const mnc_index *idx, float f);                                              // index.cpp

namespace {

constexpr int64_t IB_SEG = 1 << 18, IB_CTX = 64;

struct Piece { int32_t contig; int32_t lo, hi, from; };     // owns k-mer ends in [lo, hi); its window starts at `from`

// minimizers of piece j with their k-mer end inside [lo, hi): count (pass 0) or write at the piece's offset (pass 1)
__global__ __launch_bounds__(256) void mnc_ib_keep(const Piece *pieces, int n_pieces, const int64_t *offsets, const uint2 *mz, const int32_t *mz_cnt,
                                                   int64_t *kept, const int64_t *kept_off, uint32_t *out_h, uint64_t *out_y)
{
	// one wave per piece
	const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (j >= n_pieces) return;
	const Piece pc = pieces[j];
	const uint2 *m = mz + offsets[j];
	const int n = mz_cnt[j];
	int64_t base = out_h ? kept_off[j] : 0, total = 0;
	for (int i0 = 0; i0 < n; i0 += 64) {
		const int i = i0 + lane;
		bool keep = false;
		uint2 q = make_uint2(0, 0);
		int pos = 0;
		if (i < n) {
			q = m[i];
			pos = pc.from + (int)(q.y >> 1);                  // the k-mer's last base, in contig coordinates
			keep = pos >= pc.lo && pos < pc.hi;
		}
		const unsigned long long b = __ballot(keep);
		if (keep && out_h) {
			const int64_t at = base + total + __popcll(b & ((1ULL << lane) - 1ULL));
			out_h[at] = q.x;
			out_y[at] = (uint64_t)(uint32_t)pc.contig << 32 | (uint64_t)(uint32_t)pos << 1 | (q.y & 1u);
		}
		total += __popcll(b);
	}
	if (lane == 0 && !out_h) kept[j] = total;
}

// ---------------------------------------------------------------- stable LSD radix sort of (hash, occurrence word) by the hash
constexpr int RS_THREADS = 256, RS_WAVES = RS_THREADS / 64, RS_ROUNDS = 16, RS_TILE = RS_THREADS * RS_ROUNDS;   // 4 096 records a workgroup

// digit counts of every tile, digit-major ([256][n_tiles]: the order in which the scan must run)
__global__ __launch_bounds__(RS_THREADS) void mnc_rs_hist(const uint32_t *keys, int64_t n, int shift, uint32_t n_tiles, uint32_t *hist)
{
	__shared__ uint32_t h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	const int64_t base = (int64_t)blockIdx.x * RS_TILE;
	for (int k = 0; k < RS_ROUNDS; ++k) {
		const int64_t i = base + k * RS_THREADS + threadIdx.x;
		if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
	}
	__syncthreads();
	hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

// One tile: wave w owns records [w * 1024, (w + 1) * 1024) of it, 64 consecutive ones a round -- so "wave, round, lane" is
// the input order and a rank counted in that order is stable.  The tile is put in digit order in LDS and written from there.
__global__ __launch_bounds__(RS_THREADS) void mnc_rs_scatter(const uint32_t *keys_in, const uint64_t *vals_in, uint32_t *keys_out, uint64_t *vals_out,
                                                             int64_t n, int shift, uint32_t n_tiles, const int64_t *offs)
{
	__shared__ uint32_t cnt[RS_WAVES][256];               // records of (wave, digit) so far; then the wave's first rank for the digit
	__shared__ uint32_t lpos[256];                        // first record of the digit in the sorted tile
	__shared__ uint32_t wsum[RS_WAVES];
	__shared__ uint32_t skey[RS_TILE];
	__shared__ uint64_t sval[RS_TILE];
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	const int64_t base = (int64_t)blockIdx.x * RS_TILE;
	for (int w = 0; w < RS_WAVES; ++w) cnt[w][tid] = 0;
	__syncthreads();
	uint32_t key[RS_ROUNDS], rank[RS_ROUNDS];
	uint64_t val[RS_ROUNDS];
#pragma unroll
	for (int k = 0; k < RS_ROUNDS; ++k) {
		const int64_t i = base + (int64_t)wv * (64 * RS_ROUNDS) + k * 64 + lane;
		const bool ok = i < n;
		key[k] = ok ? keys_in[i] : 0u, val[k] = ok ? vals_in[i] : 0ull;
		const uint32_t d = (key[k] >> shift) & 255u;
		// the lanes of this round with the same digit (and a record at all)
		unsigned long long peers = __ballot(ok);
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			const unsigned long long m = __ballot((d >> b) & 1u);
			peers &= (d >> b) & 1u ? m : ~m;
		}
		const int leader = __ffsll((long long)peers) - 1;
		uint32_t before = 0;
		if (ok && lane == leader) { before = cnt[wv][d]; cnt[wv][d] = before + (uint32_t)__popcll(peers); }   // one lane per digit: no race inside the wave
		before = (uint32_t)__shfl((int)before, leader < 0 ? 0 : leader);
		rank[k] = before + (uint32_t)__popcll(peers & ((1ULL << lane) - 1ULL));
	}
	__syncthreads();
	// thread d: digit d's records per wave -> the wave's first rank; the digit's total
	uint32_t total = 0;
	for (int w = 0; w < RS_WAVES; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = total; total += c; }
	// exclusive scan of the 256 totals: first record of every digit in the sorted tile
	uint32_t inc = total;
	for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc, dd); if (lane >= dd) inc += o; }
	if (lane == 63) wsum[wv] = inc;
	__syncthreads();
	uint32_t pre = inc - total;
	for (int w = 0; w < wv; ++w) pre += wsum[w];
	lpos[tid] = pre;
	__syncthreads();
#pragma unroll
	for (int k = 0; k < RS_ROUNDS; ++k) {
		const int64_t i = base + (int64_t)wv * (64 * RS_ROUNDS) + k * 64 + lane;
		if (i < n) {
			const uint32_t d = (key[k] >> shift) & 255u, at = lpos[d] + cnt[wv][d] + rank[k];
			skey[at] = key[k], sval[at] = val[k];
		}
	}
	__syncthreads();
	const int64_t in_tile = n - base < RS_TILE ? n - base : RS_TILE;
	for (int k = 0; k < RS_ROUNDS; ++k) {
		const int at = k * RS_THREADS + tid;
		if (at < in_tile) {
			const uint32_t kk = skey[at], d = (kk >> shift) & 255u;
			const int64_t to = offs[(size_t)d * n_tiles + blockIdx.x] + (int64_t)(at - lpos[d]);
			keys_out[to] = kk, vals_out[to] = sval[at];
		}
	}
}

// ---------------------------------------------------------------- distinct hashes and their offsets
struct HeadFlag {                                           // 1 where a run of equal hashes starts
	const uint32_t *h;
	__device__ long long operator()(int64_t i) const { return i == 0 || h[i] != h[i - 1] ? 1 : 0; }
};
__global__ __launch_bounds__(256) void mnc_ib_emit(const uint32_t *h, const int64_t *head_idx, int64_t n, uint32_t *keys, uint64_t *key_off)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	if (i == 0 || h[i] != h[i - 1]) keys[head_idx[i]] = h[i], key_off[head_idx[i]] = (uint64_t)i;
	if (i == n - 1) key_off[head_idx[n]] = (uint64_t)n;     // head_idx[n]: the number of runs
}

// ---------------------------------------------------------------- mid_occ: how many hashes occur c times, c < IB_CNT_BINS - 1
constexpr int IB_CNT_BINS = 4096;
__global__ __launch_bounds__(256) void mnc_ib_count_hist(const uint64_t *key_off, int64_t n_keys, unsigned long long *bins)
{
	__shared__ uint32_t s[IB_CNT_BINS];
	for (int k = threadIdx.x; k < IB_CNT_BINS; k += 256) s[k] = 0;
	__syncthreads();
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_keys; i += (int64_t)gridDim.x * 256) {
		const uint64_t c = key_off[i + 1] - key_off[i];
		atomicAdd(&s[c < IB_CNT_BINS - 1 ? (int)c : IB_CNT_BINS - 1], 1u);
	}
	__syncthreads();
	for (int k = threadIdx.x; k < IB_CNT_BINS; k += 256) if (s[k]) atomicAdd(&bins[k], (unsigned long long)s[k]);
}

struct DBuf {
	void *p = nullptr;
	~DBuf() { if (p) (void)hipFree(p); }
	hipError_t get(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8); }
	template <class T> T *as() { return static_cast<T*>(p); }
};

#define IB_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); \
	return e_ == hipErrorOutOfMemory ? MNC_ERR_NOMEM : MNC_ERR_HIP; } } while (0)

} // namespace

int index_build_on_device(mnc_index *idx, int n_seq, const char *const *seqs, const int64_t *lens, int device)
{
	IB_TRY(hipSetDevice(device));
	// ---- pieces and their windows
	std::vector<Piece> pieces;
	std::vector<int64_t> offsets(1, 0);
	for (int i = 0; i < n_seq; ++i)
		for (int64_t lo = 0; lo == 0 || lo < lens[i]; lo += IB_SEG) {
			const int64_t hi = std::min(lens[i], lo + IB_SEG), from = std::max<int64_t>(0, lo - IB_CTX), to = std::min(lens[i], hi + IB_CTX);
			pieces.push_back({ i, (int32_t)lo, (int32_t)hi, (int32_t)from });
			offsets.push_back(offsets.back() + ((to - from + 15) & ~(int64_t)15));     // windows start on 16-byte boundaries (the pack kernel reads 16 bases at a time)
		}
	const size_t np = pieces.size();
	const int64_t nb = offsets.back();
	// (nothing to sketch: still a whole index object -- the contig / genome tables and mid_occ as index_finalize leaves them)
	auto empty_index = [&]() {
		idx->keys.clear(), idx->key_off.assign(1, 0), idx->pos.clear();
		pack_contigs(idx, seqs, lens, n_seq);
		index_genome_table(idx);
		idx->mid_occ = cal_mid_occ(idx, idx->par.mid_occ_frac);
		return MNC_OK;
	};
	if (np == 0 || nb == 0) return empty_index();
	// the windows' real lengths differ from the padded slots: the sketch takes a read's length from the offsets, so the
	// slots are laid out back to back with their true lengths and the padding is avoided instead -- true lengths, 16-aligned
	// by giving every window its own start rounded up
	std::vector<int64_t> start(np), true_off(np + 1, 0);
	for (size_t j = 0; j < np; ++j) {
		const Piece &pc = pieces[j];
		const int64_t to = std::min<int64_t>(lens[pc.contig], (int64_t)pc.hi + IB_CTX);
		start[j] = true_off[j];
		true_off[j + 1] = true_off[j] + (to - pc.from);
	}
	const int64_t total = true_off[np];
	DBuf d_bases, d_off, d_pieces, d_packed, d_ambig, d_mz, d_cnt, d_hist, d_kept, d_kept_off;
	IB_TRY(d_bases.get((size_t)total + 64));
	IB_TRY(d_off.get((np + 1) * 8));
	IB_TRY(d_pieces.get(np * sizeof(Piece)));
	IB_TRY(d_packed.get(((size_t)total / 16 + 4) * 4));
	IB_TRY(d_ambig.get((np + 1) * 4));
	IB_TRY(d_mz.get(((size_t)total + 1) * sizeof(uint2)));
	IB_TRY(d_cnt.get((np + 1) * 4));
	const size_t n_tiles = (np + PT_READS - 1) / PT_READS;
	IB_TRY(d_hist.get((n_tiles + 1) * ((size_t)1 << PB_BITS_MIN) * 4));
	IB_TRY(d_kept.get((np + 1) * 8));
	IB_TRY(d_kept_off.get((np + 1) * 8));
	hipStream_t st = nullptr;                                 // the default stream: everything here is one sequence
	for (size_t j = 0; j < np; ++j) {
		const Piece &pc = pieces[j];
		IB_TRY(hipMemcpyAsync(d_bases.as<uint8_t>() + start[j], seqs[pc.contig] + pc.from, (size_t)(true_off[j + 1] - true_off[j]), hipMemcpyHostToDevice, st));
	}
	IB_TRY(hipMemcpyAsync(d_off.p, true_off.data(), (np + 1) * 8, hipMemcpyHostToDevice, st));
	IB_TRY(hipMemcpyAsync(d_pieces.p, pieces.data(), np * sizeof(Piece), hipMemcpyHostToDevice, st));
	IB_TRY(hipMemsetAsync(d_ambig.p, 0, (np + 1) * 4, st));
	IB_TRY(hipMemsetAsync(d_cnt.p, 0, (np + 1) * 4, st));
	IB_TRY(hipMemsetAsync(d_packed.as<uint32_t>() + total / 16, 0, 16, st));
	// the host packs the contigs' 4-bit bases meanwhile (what the alignment stage reads; minimap2 keeps them the same way)
	std::thread packer([&]() { pack_contigs(idx, seqs, lens, n_seq); });
	struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{ packer };
	Batch B;
	memset(&B, 0, sizeof(B));
	B.bases = d_bases.as<uint8_t>(), B.offsets = d_off.as<int64_t>(), B.n_reads = (uint32_t)np, B.total_bases = total;
	B.packed = d_packed.as<uint32_t>(), B.ambig = d_ambig.as<uint32_t>(), B.mz = d_mz.as<uint2>(), B.mz_cnt = d_cnt.as<int32_t>();
	B.hist_tm = d_hist.as<uint32_t>(), B.n_tiles = (uint32_t)n_tiles;
	B.pb_bits = PB_BITS_MIN, B.pb_n = 1u << PB_BITS_MIN, B.ps_tiles = PS_TILES_MIN;   // (the sketch kernel fills a histogram row per tile: unused here)
	launch_pack(B, st);
	launch_sketch(B, st);
	// ---- what every piece keeps
	hipLaunchKernelGGL(mnc_ib_keep, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, st, d_pieces.as<Piece>(), (int)np, d_off.as<int64_t>(), d_mz.as<uint2>(),
	                   d_cnt.as<int32_t>(), d_kept.as<int64_t>(), nullptr, nullptr, nullptr);
	std::vector<int64_t> kept(np), kept_off(np + 1, 0);
	IB_TRY(hipMemcpyAsync(kept.data(), d_kept.p, np * 8, hipMemcpyDeviceToHost, st));
	IB_TRY(hipStreamSynchronize(st));
	for (size_t j = 0; j < np; ++j) kept_off[j + 1] = kept_off[j] + kept[j];
	const int64_t n_occ = kept_off[np];
	if (n_occ >= (1LL << 31)) { set_error("more than 2^31 minimizer occurrences in one index part"); return MNC_ERR_UNSUPPORTED; }
	if (n_occ == 0) {                                        // every contig shorter than k, or all ambiguous
		packer.join();
		idx->keys.clear(), idx->key_off.assign(1, 0), idx->pos.clear();
		index_genome_table(idx);
		idx->mid_occ = cal_mid_occ(idx, idx->par.mid_occ_frac);
		return MNC_OK;
	}
	DBuf d_h, d_y, d_h2, d_y2, d_rhist, d_offs, d_sums, d_keys, d_koff, d_bins;
	IB_TRY(d_h.get((size_t)n_occ * 4)); IB_TRY(d_h2.get((size_t)n_occ * 4));
	IB_TRY(d_y.get((size_t)n_occ * 8)); IB_TRY(d_y2.get((size_t)n_occ * 8));
	IB_TRY(hipMemcpyAsync(d_kept_off.p, kept_off.data(), (np + 1) * 8, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(mnc_ib_keep, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, st, d_pieces.as<Piece>(), (int)np, d_off.as<int64_t>(), d_mz.as<uint2>(),
	                   d_cnt.as<int32_t>(), d_kept.as<int64_t>(), d_kept_off.as<int64_t>(), d_h.as<uint32_t>(), d_y.as<uint64_t>());
	// ---- stable sort by the hash (the occurrence words are ascending as written: pieces in contig order, positions
	// ascending inside a piece): four passes of 8 bits over the 30-bit hash
	const uint32_t rs_tiles = (uint32_t)((n_occ + RS_TILE - 1) / RS_TILE);
	const int64_t n_hist = (int64_t)rs_tiles * 256;
	IB_TRY(d_rhist.get((size_t)n_hist * 4)); IB_TRY(d_offs.get(((size_t)n_hist + 1) * 8));
	// (scratch of the scans below: one partial sum per 1 024 inputs of the longest, the head flags)
	IB_TRY(d_sums.get(((size_t)std::max<int64_t>(n_hist, n_occ) / SC_TILE + 4) * 8));
	{
		uint32_t *kin = d_h.as<uint32_t>(), *kout = d_h2.as<uint32_t>();
		uint64_t *vin = d_y.as<uint64_t>(), *vout = d_y2.as<uint64_t>();
		for (int shift = 0; shift < 2 * KMER; shift += 8) {
			hipLaunchKernelGGL(mnc_rs_hist, dim3(rs_tiles), dim3(RS_THREADS), 0, st, kin, n_occ, shift, rs_tiles, d_rhist.as<uint32_t>());
			exclusive_scan(ScanInPlain<uint32_t>{d_rhist.as<uint32_t>()}, n_hist, d_offs.as<int64_t>(), d_sums.as<int64_t>(), st);
			hipLaunchKernelGGL(mnc_rs_scatter, dim3(rs_tiles), dim3(RS_THREADS), 0, st, kin, vin, kout, vout, n_occ, shift, rs_tiles, d_offs.as<int64_t>());
			std::swap(kin, kout), std::swap(vin, vout);
		}
		static_assert((2 * KMER + 7) / 8 % 2 == 0, "an even number of passes: the sorted records are back in d_h / d_y");
	}
	// now d_h = hashes ascending, d_y = their occurrence words (ascending inside one hash)
	// ---- run heads -> distinct hashes and their offsets
	int n_keys = 0;
	{
		DBuf d_idx;                                             // per record: the run heads before it; [n_occ]: the number of runs
		IB_TRY(d_idx.get(((size_t)n_occ + 1) * 8));
		exclusive_scan(HeadFlag{d_h.as<uint32_t>()}, n_occ, d_idx.as<int64_t>(), d_sums.as<int64_t>(), st);
		int64_t n_keys64 = 0;
		IB_TRY(hipMemcpyAsync(&n_keys64, d_idx.as<int64_t>() + n_occ, 8, hipMemcpyDeviceToHost, st));
		IB_TRY(hipStreamSynchronize(st));
		n_keys = (int)n_keys64;
		IB_TRY(d_keys.get((size_t)n_keys * 4)); IB_TRY(d_koff.get(((size_t)n_keys + 1) * 8));
		hipLaunchKernelGGL(mnc_ib_emit, dim3((unsigned)((n_occ + 255) / 256)), dim3(256), 0, st, d_h.as<uint32_t>(), d_idx.as<int64_t>(), n_occ,
		                   d_keys.as<uint32_t>(), d_koff.as<uint64_t>());
		IB_TRY(hipStreamSynchronize(st));                       // (d_idx is freed here)
	}
	// ---- mid_occ: the occurrence count at rank (1 - f) n, plus one (index.cpp: cal_mid_occ)
	int mid_occ = 1;
	bool mid_occ_on_host = false;
	{
		const float f = idx->par.mid_occ_frac;
		if (f <= 0.) mid_occ = INT32_MAX;
		else {
			IB_TRY(d_bins.get((size_t)IB_CNT_BINS * 8));
			IB_TRY(hipMemsetAsync(d_bins.p, 0, (size_t)IB_CNT_BINS * 8, st));
			hipLaunchKernelGGL(mnc_ib_count_hist, dim3((unsigned)std::min<int64_t>(1024, (n_keys + 255) / 256)), dim3(256), 0, st, d_koff.as<uint64_t>(), (int64_t)n_keys,
			                   d_bins.as<unsigned long long>());
			std::vector<unsigned long long> bins(IB_CNT_BINS);
			IB_TRY(hipMemcpyAsync(bins.data(), d_bins.p, (size_t)IB_CNT_BINS * 8, hipMemcpyDeviceToHost, st));
			IB_TRY(hipStreamSynchronize(st));
			size_t kth = (size_t)(uint32_t)((1. - f) * n_keys);
			if (kth >= (size_t)n_keys) kth = (size_t)n_keys - 1;
			unsigned long long below = 0;
			int c = 0;
			while (c < IB_CNT_BINS && below + bins[c] <= kth) below += bins[c++];     // the count whose bin holds rank kth
			if (c >= IB_CNT_BINS - 1) mid_occ_on_host = true;                         // a count beyond the bins: the host's selection
			else mid_occ = c + 1;
		}
	}
	// ---- back to the host index object
	try {
		idx->keys.resize((size_t)n_keys), idx->key_off.resize((size_t)n_keys + 1), idx->pos.resize((size_t)n_occ);
	} catch (const std::bad_alloc &) { return MNC_ERR_NOMEM; }
	IB_TRY(hipMemcpyAsync(idx->keys.data(), d_keys.p, (size_t)n_keys * 4, hipMemcpyDeviceToHost, st));
	IB_TRY(hipMemcpyAsync(idx->key_off.data(), d_koff.p, ((size_t)n_keys + 1) * 8, hipMemcpyDeviceToHost, st));
	IB_TRY(hipMemcpyAsync(idx->pos.data(), d_y.p, (size_t)n_occ * 8, hipMemcpyDeviceToHost, st));
	IB_TRY(hipStreamSynchronize(st));
	IB_TRY(hipGetLastError());
	packer.join();
	index_genome_table(idx);
	idx->mid_occ = mid_occ_on_host ? cal_mid_occ(idx, idx->par.mid_occ_frac) : mid_occ;
	return MNC_OK;
}

} // namespace mnc

using namespace mnc;

// mnc_index_build_mem with the sketch and the sort on a device: the same index, array for array.
extern "C" int mnc_index_build_mem_device(int n_seq, const char *const *names, const char *const *seqs, const int64_t *lens,
                                          int k, int w, int device, mnc_index **out)
{
	if (!out || n_seq < 0 || (n_seq > 0 && (!names || !seqs || !lens))) return MNC_ERR_ARG;
	*out = nullptr;
	if (k != 15 || w != 10) { set_error("only k=15, w=10 (minimap2 'map-ont', the setting monica uses) is implemented; got k=%d w=%d", k, w); return MNC_ERR_UNSUPPORTED; }
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) { (void)hipGetLastError(); set_error("no HIP device %d", device); return MNC_ERR_NODEVICE; }
	mnc_index *idx = new (std::nothrow) mnc_index;
	if (!idx) return MNC_ERR_NOMEM;
	idx->k = k, idx->w = w;
	try {
		for (int i = 0; i < n_seq; ++i) {
			if (lens[i] < 0 || lens[i] > 0x7fffffffLL) { delete idx; set_error("contig %d too long", i); return MNC_ERR_UNSUPPORTED; }
			idx->contig_name.emplace_back(names[i]);
			idx->contig_len.push_back(lens[i]);
		}
		idx->seq_off.assign((size_t)n_seq + 1, 0);
		for (int i = 0; i < n_seq; ++i) idx->seq_off[i + 1] = idx->seq_off[i] + lens[i];
		idx->seq4.assign((size_t)(idx->seq_off[n_seq] + 7) / 8 + 1, 0u);
	} catch (const std::bad_alloc &) { delete idx; return MNC_ERR_NOMEM; }
	const int rc = index_build_on_device(idx, n_seq, seqs, lens, device);
	if (rc) { delete idx; return rc; }
	*out = idx;
	return MNC_OK;
}
