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
//   3. two stable radix sorts (by occurrence word, then by the 30-bit hash: rocPRIM through hipCUB) give minimap2's
//      order -- by hash, positions ascending -- and a run-length pass the distinct hashes and their offsets;
//   4. the occurrence count at rank (1 - mid_occ_frac) n of the sorted counts is mid_occ - 1 (A.3).
// The arrays come back to the host index object (it saves them to the index file, dumps them for the tests) and the
// 4-bit contig bases are packed by the host meanwhile.  The result is the host builder's, array for array
// (tests/test_gpu_parity.py).
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <map>
#include <thread>

#include "device.h"

namespace mnc {

void launch_pack(const Batch &B, hipStream_t st);
void launch_sketch(const Batch &B, hipStream_t st);
void pack_contigs(mnc_index *idx, const char *const *seqs, const int64_t *lens, int n_seq);   // index.cpp
void index_genome_table(mnc_index *idx);                                                     // index.cpp

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

__global__ __launch_bounds__(256) void mnc_ib_counts(const uint64_t *key_off, int64_t n_keys, uint32_t *cnt)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n_keys) cnt[i] = (uint32_t)(key_off[i + 1] - key_off[i]);
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
	if (np == 0 || nb == 0) { idx->keys.clear(), idx->key_off.assign(1, 0), idx->pos.clear(); return MNC_OK; }
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
	if (n_occ == 0) { idx->keys.clear(), idx->key_off.assign(1, 0), idx->pos.clear(); packer.join(); return MNC_OK; }
	DBuf d_h, d_y, d_h2, d_y2, d_tmp, d_keys, d_runs, d_nruns, d_koff;
	IB_TRY(d_h.get((size_t)n_occ * 4)); IB_TRY(d_h2.get((size_t)n_occ * 4));
	IB_TRY(d_y.get((size_t)n_occ * 8)); IB_TRY(d_y2.get((size_t)n_occ * 8));
	IB_TRY(hipMemcpyAsync(d_kept_off.p, kept_off.data(), (np + 1) * 8, hipMemcpyHostToDevice, st));
	hipLaunchKernelGGL(mnc_ib_keep, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, st, d_pieces.as<Piece>(), (int)np, d_off.as<int64_t>(), d_mz.as<uint2>(),
	                   d_cnt.as<int32_t>(), d_kept.as<int64_t>(), d_kept_off.as<int64_t>(), d_h.as<uint32_t>(), d_y.as<uint64_t>());
	// ---- sort by (hash, occurrence word): stable, the minor key first
	size_t tb1 = 0, tb2 = 0, tb3 = 0, tb4 = 0;
	IB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, d_y.as<uint64_t>(), d_y2.as<uint64_t>(), d_h.as<uint32_t>(), d_h2.as<uint32_t>(), (int)n_occ, 0, 64, st));
	IB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tb2, d_h2.as<uint32_t>(), d_h.as<uint32_t>(), d_y2.as<uint64_t>(), d_y.as<uint64_t>(), (int)n_occ, 0, 2 * KMER, st));
	IB_TRY(d_keys.get((size_t)n_occ * 4)); IB_TRY(d_runs.get((size_t)n_occ * 8)); IB_TRY(d_nruns.get(8));
	IB_TRY(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb3, d_h.as<uint32_t>(), d_keys.as<uint32_t>(), d_runs.as<uint64_t>(), d_nruns.as<int>(), (int)n_occ, st));
	IB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb4, d_runs.as<uint64_t>(), d_runs.as<uint64_t>(), (int)n_occ + 1, st));
	IB_TRY(d_tmp.get(std::max(std::max(tb1, tb2), std::max(tb3, tb4)) + 16));
	size_t tb = tb1;
	IB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tb, d_y.as<uint64_t>(), d_y2.as<uint64_t>(), d_h.as<uint32_t>(), d_h2.as<uint32_t>(), (int)n_occ, 0, 64, st));
	tb = tb2;
	IB_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp.p, tb, d_h2.as<uint32_t>(), d_h.as<uint32_t>(), d_y2.as<uint64_t>(), d_y.as<uint64_t>(), (int)n_occ, 0, 2 * KMER, st));
	// now d_h = hashes ascending, d_y = their occurrence words (ascending inside one hash)
	tb = tb3;
	IB_TRY(hipcub::DeviceRunLengthEncode::Encode(d_tmp.p, tb, d_h.as<uint32_t>(), d_keys.as<uint32_t>(), d_runs.as<uint64_t>(), d_nruns.as<int>(), (int)n_occ, st));
	int n_keys = 0;
	IB_TRY(hipMemcpyAsync(&n_keys, d_nruns.p, 4, hipMemcpyDeviceToHost, st));
	IB_TRY(hipStreamSynchronize(st));
	// run lengths -> offsets (n_keys + 1 of them: one zero behind the last run gives the total)
	IB_TRY(hipMemsetAsync(d_runs.as<uint64_t>() + n_keys, 0, 8, st));
	IB_TRY(d_koff.get(((size_t)n_keys + 1) * 8));
	tb = tb4;
	IB_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp.p, tb, d_runs.as<uint64_t>(), d_koff.as<uint64_t>(), n_keys + 1, st));
	// ---- mid_occ: the occurrence count at rank (1 - f) n, plus one (index.cpp: cal_mid_occ)
	int mid_occ = 1;
	{
		const float f = idx->par.mid_occ_frac;
		if (f <= 0.) mid_occ = INT32_MAX;
		else {
			DBuf d_c, d_c2, d_t2;
			IB_TRY(d_c.get((size_t)n_keys * 4)); IB_TRY(d_c2.get((size_t)n_keys * 4));
			hipLaunchKernelGGL(mnc_ib_counts, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, st, d_koff.as<uint64_t>(), (int64_t)n_keys, d_c.as<uint32_t>());
			size_t tbc = 0;
			IB_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, tbc, d_c.as<uint32_t>(), d_c2.as<uint32_t>(), n_keys, 0, 32, st));
			IB_TRY(d_t2.get(tbc + 16));
			IB_TRY(hipcub::DeviceRadixSort::SortKeys(d_t2.p, tbc, d_c.as<uint32_t>(), d_c2.as<uint32_t>(), n_keys, 0, 32, st));
			size_t kth = (size_t)(uint32_t)((1. - f) * n_keys);
			if (kth >= (size_t)n_keys) kth = (size_t)n_keys - 1;
			uint32_t c = 0;
			IB_TRY(hipMemcpyAsync(&c, d_c2.as<uint32_t>() + kth, 4, hipMemcpyDeviceToHost, st));
			IB_TRY(hipStreamSynchronize(st));
			mid_occ = (int)(c + 1);
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
	idx->mid_occ = mid_occ;
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
