// Host orchestration: HBM residency of the index, per-engine workspace, stage launches,
// HIP-event timing, stage dumps.  Public surface: include/monica_amd.h.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

#include "device.h"
#include <atomic>
#include <thread>

#include "scan.h"

namespace mnc {

// launches implemented in the k_*.hip units
void launch_pack(const Batch &B, hipStream_t st);
void launch_sketch(const Batch &B, hipStream_t st);
void launch_partition(const Batch &B, hipStream_t st);
int partition_prepare();
void launch_probe(const Batch &B, hipStream_t st);
void launch_collect(const Batch &B, hipStream_t st);
void launch_expand_sort(const Batch &B, const uint32_t *lists, const ClassSpans &spans, int NM, hipStream_t st);
int expand_sort_prepare(int max_nm);
void launch_backtrack(const Batch &B, const uint32_t *read_list, uint32_t n_list, hipStream_t st);
void launch_bin_reads(const Batch &B, const ChainClasses &C, uint32_t *cls_count, uint32_t *cls_list, hipStream_t st);
void launch_chain_dp_ring(const Batch &B, const uint32_t *lists, const ClassSpans &spans, int stress, hipStream_t st);
void launch_chain_tail(const Batch &B, const uint32_t *lists, const ClassSpans &spans, int NM, hipStream_t st);
size_t chain_tail_lds_bytes(int NM);
int chain_tail_prepare(size_t max_lds);

// LDS tile sizes (anchors per read) of the sort and backtrack kernels; reads above the last
// one, or with >= 65 536 bases, are sorted in HBM and take the sequential backtrack
// anchors per read: ~390 for a 5 kb read; the classes reach reads of ~100 kb (8192 anchors: 128 KB of LDS for the sort's tile,
// 135 KB for two reads of the backtrack), longer ones take the forms that work in HBM
static const ChainClasses CHAIN_CLASSES = { 22, { 64, 128, 192, 256, 320, 384, 448, 512, 576, 640, 768, 896, 1024, 1280, 1536, 1792, 2048, 2560,
                                                  3072, 4096, 6144, 8192 } };
void launch_regions(const Batch &B, void *regx, uint64_t *k64a, uint64_t *k64b, mnc_hit_t *gated, hipStream_t st);
void launch_regions_post(const Batch &B, mnc_reg_t *work, void *regx, uint64_t *k64a, int32_t *tmp, mnc_hit_t *gated, hipStream_t st);
// base-level alignment stage (k_align.hip)
void launch_dp_gather(const Batch &B, hipStream_t st);
int dp_gather_long_prepare(int lds_anchors);
void launch_dp_gather_long(const Batch &B, const uint32_t *lists, const ClassSpans &spans, int lds_anchors, hipStream_t st);
void launch_dp_round(const Batch &B, int first, hipStream_t st);
void launch_dp_round_end(const Batch &B, hipStream_t st);
int dp_plan_prepare();
void launch_dp_plan(const Batch &B, const int32_t *work_list, unsigned max_work, int wave_form, int state_max, long long p_max, int cig_max,
                    long long big_state, long long big_p, long long big_cig, long long huge_state, long long huge_p, long long huge_cig, hipStream_t st);
size_t dp_align_ws_bytes(long long state_max, long long p_max, long long cig_max);
int dp_align_prepare(int lds_bytes);
int dp_stitch_prepare();
size_t dp_inv_ws_words(int max_gap);
void launch_dp_inv(const Batch &B, const int32_t *work_list, int32_t *next_list, int32_t *ws, int n_wg, hipStream_t st);
size_t dp_stitch_pool_slack(int n_wg);
void launch_dp_align(const Batch &B, uint8_t *ws, int n_wg, long long state_max, long long p_max, long long cig_max,
                     int lds_state, int lds_p, int lds_cig, int big_pass, hipStream_t st, int forms = 3);
void launch_dp_stitch(const Batch &B, const int32_t *work_list, int32_t *next_list, int max_read_len, int n_wg, hipStream_t st);
bool dp_align_long_packed(const Batch &B);
void launch_dp_align_long(const Batch &B, uint8_t *ws_huge, int n_huge, long long st_huge, long long p_huge, long long cig_huge,
                          uint8_t *ws_big, int n_big, long long st_big, long long p_big, long long cig_big,
                          uint8_t *ws_small, int n_small, long long st_small, long long p_small, long long cig_small, hipStream_t st);
size_t dp_fill_p_slot();
size_t dp_fillp_slot();
size_t dp_fillp_cig_slot();
size_t dp_lfill_p_slot();
size_t dp_lext_p_slot();
void launch_dp_lext(const Batch &B, const int32_t *list, int ctr_n, int ctr_q, int32_t *fb_list, int ctr_fb, uint8_t *p_all, int n_wg, hipStream_t st);
void launch_dp_lfill(const Batch &B, const int32_t *list, int ctr_n, int ctr_q, int32_t *fb_list, int ctr_fb, uint8_t *p_all, int n_wg, hipStream_t st);
size_t dp_extp_slot();
size_t dp_extp_cig_slot();
void launch_dp_extp(const Batch &B, int cells, int rgt, const int32_t *list, int ctr_n, int ctr_q, uint8_t *p_all, uint32_t *cig_all, int n_wg, hipStream_t st);
void launch_dp_fill(const Batch &B, int lanes, const int32_t *list, int ctr_n, int ctr_q, int32_t *next_list, int ctr_next,
                    int32_t *fb_list, int ctr_fb, uint8_t *p_all, uint32_t *cig_all, int n_wg, hipStream_t st);
void launch_dp_ext(const Batch &B, int lanes, const int32_t *list, int ctr_n, int ctr_q, int32_t *fb_list, int ctr_fb, uint8_t *p_all, int n_wg, hipStream_t st);
#ifndef MNC_DP_WG_FILL
#define MNC_DP_WG_FILL (256 * 16)
#endif
#ifndef MNC_DP_WG_EXT
#define MNC_DP_WG_EXT (256 * 8)
#endif
constexpr int DP_WG_FILL = MNC_DP_WG_FILL, DP_WG_EXT = MNC_DP_WG_EXT, DP_WG_STITCH = 4096, DP_WG_INV = 256;
// workspace classes of the alignment kernel: a normal slot per workgroup, a few large ones
constexpr int DP_LDS_BYTES = 16 * 1024;
constexpr int DP_LDS0_STATE = 4 * 1024, DP_LDS0_P = 10 * 1024, DP_LDS0_CIG = 256;   // pass 0: 15 KB per workgroup
constexpr int DP_WG_SMALL = 2048, DP_WG_BIG = 768, DP_WG_HUGE = 8, DP_WG_MID = 1536, DP_WG_LFILL = 4096, DP_WG_LEXT = 2048, DP_WG_BIGFB = 512;   // (BIGFB: large slots of pass 4's round inside the window, beside pass 1's)
constexpr long long DP_STATE_SMALL = 96 * 1024, DP_P_SMALL = 1 << 20, DP_CIG_SMALL = 4096;
constexpr long long DP_STATE_BIG = 13 * 32768, DP_P_BIG = 8LL << 20, DP_CIG_BIG = 65536;       // 768 slots of 9 MB: any extension (max_gap 5000 on both sides: 7.7 MB of direction bytes)
constexpr long long DP_STATE_HUGE = 13 * 32768, DP_P_HUGE = 256LL << 20, DP_CIG_HUGE = 65536;   // and 8 of 257 MB for anything up to max_sw_mat
void launch_gather_hits(const Batch &B, const mnc_hit_t *gated, const int64_t *hit_off, mnc_hit_t *out, hipStream_t st);

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	return e_ == hipErrorOutOfMemory ? MNC_ERR_NOMEM : MNC_ERR_HIP; } } while (0)

// ---------------------------------------------------------------- run offsets of the partition
// The sketch writes one histogram row per tile ([tile][bucket]); the query records are laid out
// bucket-major, so the offset of run (bucket b, tile t) is the exclusive sum over that order.
// Rows are read and written whole (coalesced): chunks of HC_TILES tiles are summed per bucket,
// the [bucket][chunk] sums are scanned as one flat array (which IS the bucket-major order), and
// each chunk then walks its rows again with the running offsets.
constexpr int HC_TILES = 64;

__global__ __launch_bounds__(PB_N_MAX) void mnc_hist_chunk_sums(const uint32_t *hist_tm, uint32_t n_tiles, uint32_t n_chunks, uint32_t *sums_bm)
{
	const uint32_t c = blockIdx.x, b = threadIdx.x, PB_N = blockDim.x;        // one thread per table region
	const uint32_t t0 = c * HC_TILES, t1 = min(n_tiles, t0 + HC_TILES);
	uint32_t s = 0;
	for (uint32_t t = t0; t < t1; ++t) s += hist_tm[(size_t)t * PB_N + b];
	sums_bm[(size_t)b * n_chunks + c] = s;
}

__global__ __launch_bounds__(PB_N_MAX) void mnc_hist_offsets(const uint32_t *hist_tm, uint32_t n_tiles, uint32_t n_chunks,
                                                             const int64_t *chunk_off_bm, int64_t *q_off)
{
	const uint32_t c = blockIdx.x, b = threadIdx.x, PB_N = blockDim.x;
	const uint32_t t0 = c * HC_TILES, t1 = min(n_tiles, t0 + HC_TILES);
	int64_t run = chunk_off_bm[(size_t)b * n_chunks + c];
	for (uint32_t t = t0; t < t1; ++t) {
		q_off[(size_t)t * PB_N + b] = run;
		run += hist_tm[(size_t)t * PB_N + b];
	}
	if (c == 0 && b == 0) q_off[(size_t)n_tiles * PB_N] = chunk_off_bm[(size_t)PB_N * n_chunks];   // the total
}

// ================================================================ small read-backs
// The few numbers the host needs in the middle of a batch (anchor total, size-class counts, round counters) are written
// by a one-wave kernel straight into page-locked host memory instead of being fetched with hipMemcpyAsync: a copy
// request queues behind every copy submitted before it on the same engine, and with the next batch's bases on their
// way (mnc_engine_prefetch: half a gigabyte) a four-byte read-back waited 7 ms for them -- measured, profiles/r03*.
struct MailItem { const void *src; uint32_t dst_word, n_words; };
constexpr int MAIL_MAX = 8;
struct MailList { MailItem it[MAIL_MAX]; int n; };
__global__ __launch_bounds__(64) void mnc_mail(MailList L, uint32_t *host_box)
{
	for (int k = 0; k < L.n; ++k) {
		const uint32_t *s = static_cast<const uint32_t*>(L.it[k].src);
		for (uint32_t i = threadIdx.x; i < L.it[k].n_words; i += 64) host_box[L.it[k].dst_word + i] = s[i];
	}
	__threadfence_system();
}

// ================================================================ device buffer
struct Buf {
	void *p = nullptr;
	size_t cap = 0;
	int ensure(size_t bytes)
	{
		if (bytes <= cap) return MNC_OK;
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		size_t want = bytes + bytes / 8 + 256;
		hipError_t e = hipMalloc(&p, want);
		if (e != hipSuccess) { p = nullptr; set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return MNC_ERR_NOMEM; }
		cap = want;
		return MNC_OK;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
	template <class T> T *as() const { return reinterpret_cast<T*>(p); }
};

// The alignment kernels' scratch -- direction bytes of the persistent workgroups, CIGAR scratch, the
// literal kernel's workspaces: ~30 GB whatever the batch -- exists once per device and is lent to one
// engine at a time for the alignment stage of its batch (the kernels fill the chip: two engines
// would not run them faster side by side).  monica's thread pool makes an engine per thread.
struct SharedWs {
	std::mutex mu;
	int refs = 0;
	Buf fill_p, fill_cig, extp_p, extp_cig, ext_p, lfill_p, lext_p, dp_ws, dp_ws_mid, dp_ws_big, dp_ws_bigfb, dp_ws_huge;
	void release() { for (Buf *b : { &fill_p, &fill_cig, &extp_p, &extp_cig, &ext_p, &lfill_p, &lext_p, &dp_ws, &dp_ws_mid, &dp_ws_big, &dp_ws_bigfb, &dp_ws_huge }) b->release(); }
};
static std::mutex g_ws_mu;
static SharedWs *g_ws[64];
static SharedWs *shared_ws(int device, int add_ref)
{
	std::lock_guard<std::mutex> g(g_ws_mu);
	if (device < 0 || device >= 64) return nullptr;
	if (!g_ws[device]) g_ws[device] = new SharedWs();
	SharedWs *w = g_ws[device];
	w->refs += add_ref;
	if (w->refs == 0 && add_ref < 0) {                       // the last engine of the device is gone
		std::lock_guard<std::mutex> h(w->mu);
		w->release();
	}
	return w;
}

// ================================================================ index residency
static int check_device(int device)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device visible"); return MNC_ERR_NODEVICE; }
	if (device < 0 || device >= n) { set_error("device %d out of range (have %d)", device, n); return MNC_ERR_NODEVICE; }
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		set_error("device %d is %s; this library holds gfx950 code only", device, prop.gcnArchName);
		return MNC_ERR_NODEVICE;
	}
	return MNC_OK;
}

int index_tables_on_device(const std::vector<uint32_t> &keys, const std::vector<uint64_t> &key_off, const std::vector<uint32_t> &reg_count,
                           int pb_bits, int region_bits, int disp_bits, TableSlot *d_table, uint8_t *d_disp, uint32_t *d_salt, uint32_t *d_filter,
                           const uint64_t *d_positions);

// the rest of the index (occurrence words, genome of a contig, contig bases) next to tables that are on the device already
static int index_upload_rest(mnc_index *idx, DeviceIndex &d)
{
	auto upload = [&](void **dst, const void *src, size_t bytes, size_t spare = 0) -> int {
		HIP_TRY(hipMalloc(dst, bytes + spare ? bytes + spare : 8));
		if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
		return MNC_OK;
	};
	int urc = MNC_OK;
	if (!d.positions) urc = upload((void**)&d.positions, idx->pos.data(), idx->pos.size() * 8, 8);   // one spare word behind the positions
	if (!urc) urc = upload((void**)&d.contig_genome, idx->contig_genome.data(), idx->contig_genome.size() * 4);
	if (!urc) urc = upload((void**)&d.seq4, idx->seq4.data(), idx->seq4.size() * 4, 16);
	if (!urc) urc = upload((void**)&d.seq_off, idx->seq_off.data(), idx->seq_off.size() * 8);
	return urc;
}

int index_upload(mnc_index *idx, int device, DeviceIndex **out)
{
	std::lock_guard<std::mutex> lk(idx->dev_mutex);
	for (auto &d : idx->dev) if (d.device == device) { *out = &d; return MNC_OK; }
	if (idx->keys.empty()) { set_error("empty index"); return MNC_ERR_FORMAT; }
	HIP_TRY(hipSetDevice(device));
	// ---- regions: 2^pb_bits blocks of R slots (load <= 0.5), perfectly hashed by hash-and-displace.  As few regions
	// as keep a region at 2^17 slots (2 MiB: what an XCD's L2 holds of the two or three regions its probe workgroups
	// walk at a time), at most 2^PB_BITS_MAX
	int pb_bits = idx->force_pb_bits ? idx->force_pb_bits : PB_BITS_MIN, region_bits = 6;
	std::vector<uint32_t> reg_count;
	for (;; ++pb_bits) {
		reg_count.assign((size_t)1 << pb_bits, 0);
		for (size_t i = 0; i < idx->keys.size(); ++i) ++reg_count[pb_bucket(idx->keys[i], pb_bits)];
		size_t biggest = 0;
		for (uint32_t c : reg_count) biggest = std::max(biggest, (size_t)c);
		region_bits = 6;
		while ((1ULL << region_bits) < biggest * 2) ++region_bits;
		if (region_bits <= 17 || pb_bits == PB_BITS_MAX || idx->force_pb_bits) break;
	}
	const int PB_N = 1 << pb_bits;
	// ---- the construction on the device (k_index.hip): the same tables in a few milliseconds; the host form below
	// takes over when a region does not fit that kernel (or needs larger regions)
	if (!idx->host_tables && region_bits <= 28) {
		DeviceIndex d;
		d.device = device, d.pb_bits = pb_bits, d.region_bits = region_bits, d.disp_bits = std::max(0, region_bits - 3);
		const size_t Rd = (size_t)1 << d.region_bits, NBd = (size_t)1 << d.disp_bits;
		bool ok = hipMalloc((void**)&d.filter, (size_t)PB_N * PF_WORDS * 4) == hipSuccess && hipMalloc((void**)&d.disp, (size_t)PB_N * NBd) == hipSuccess &&
		          hipMalloc((void**)&d.salt, PB_N * 4) == hipSuccess && hipMalloc((void**)&d.table, (size_t)PB_N * Rd * sizeof(TableSlot)) == hipSuccess &&
		          hipMalloc((void**)&d.positions, idx->pos.size() * 8 + 8) == hipSuccess &&
		          hipMemcpy(d.positions, idx->pos.data(), idx->pos.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
		int trc = ok ? index_tables_on_device(idx->keys, idx->key_off, reg_count, pb_bits, d.region_bits, d.disp_bits, d.table, d.disp, d.salt, d.filter, d.positions) : 1;
		if (trc == 0) trc = index_upload_rest(idx, d) == MNC_OK ? 0 : -1;
		if (trc == 0) {
			d.bytes = (int64_t)((size_t)PB_N * Rd * sizeof(TableSlot) + (size_t)PB_N * PF_WORDS * 4 + (size_t)PB_N * NBd + (idx->pos.size() + 1) * 8 +
			                    idx->contig_genome.size() * 4 + idx->seq4.size() * 4 + idx->seq_off.size() * 8);
			idx->dev.push_back(d);
			*out = &idx->dev.back();
			return MNC_OK;
		}
		(void)hipGetLastError();
		void *parts[] = { d.filter, d.disp, d.salt, d.table, d.positions, d.contig_genome, d.seq4, d.seq_off };
		for (void *q : parts) if (q) (void)hipFree(q);
		if (trc < 0) { set_error("index upload failed"); return MNC_ERR_NOMEM; }
	}
	std::vector<std::vector<uint32_t>> reg(PB_N);       // key indices per region
	for (size_t i = 0; i < idx->keys.size(); ++i) reg[pb_bucket(idx->keys[i], pb_bits)].push_back((uint32_t)i);
	int disp_bits = 0;
	size_t R = 0, NB = 0;
	std::vector<TableSlot> tab;
	std::vector<uint8_t> disp;
	std::vector<uint32_t> filt((size_t)PB_N * PF_WORDS, 0u), salt(PB_N, 0u);
	constexpr int MAX_SALTS = 32, MAX_GROW = 3;
	for (int grow = 0;; ++grow, ++region_bits) {
		if (region_bits > 28) { set_error("index too large for one device table"); return MNC_ERR_UNSUPPORTED; }
		disp_bits = std::max(0, region_bits - 3);            // about four keys per displacement bucket at most
		R = (size_t)1 << region_bits, NB = (size_t)1 << disp_bits;
		disp.assign((size_t)PB_N * NB, 0);
		std::fill(filt.begin(), filt.end(), 0u);
		try { tab.assign((size_t)PB_N * R, TableSlot{0, 0, 0}); } catch (const std::bad_alloc &) { return MNC_ERR_NOMEM; }
		// regions are independent: build them on a few host threads
		std::atomic<int> next_region{0}, failed{-1};
		auto work = [&]() {
			std::vector<std::vector<uint32_t>> bk(NB);
			std::vector<uint32_t> order(NB);
			for (;;) {
				const int b = next_region.fetch_add(1);
				if (b >= PB_N || failed.load() >= 0) return;
				TableSlot *T = tab.data() + (size_t)b * R;
				for (auto &v : bk) v.clear();
				for (uint32_t ki : reg[b]) {
					const uint32_t rest = pb_rest(idx->keys[ki], pb_bits);
					bk[rest & (NB - 1)].push_back(ki);
					filt[(size_t)b * PF_WORDS + pf_word(rest)] |= pf_mask(rest);
				}
				for (size_t i = 0; i < NB; ++i) order[i] = (uint32_t)i;
				std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return bk[x].size() != bk[y].size() ? bk[x].size() > bk[y].size() : x < y; });
				bool placed = false;
				for (int sx = 0; sx < MAX_SALTS && !placed; ++sx) {
					// salt 0 first; a region is re-salted only when two keys of one displacement
					// bucket share base and step (no displacement can separate those)
					const uint32_t sv = (uint32_t)sx * 0x9E3779B9u;
					if (sx > 0) {
						std::fill(T, T + R, TableSlot{0, 0, 0});
						std::fill(disp.begin() + (size_t)b * NB, disp.begin() + (size_t)(b + 1) * NB, (uint8_t)0);
					}
					placed = true;
					for (uint32_t o : order) {               // largest displacement buckets first
						const auto &keys = bk[o];
						if (keys.empty()) break;
						int d = 0;
						for (; d < 256; ++d) {
							bool ok = true;
							for (size_t a = 0; a < keys.size() && ok; ++a) {
								const uint32_t sa = pd_slot(pb_rest(idx->keys[keys[a]], pb_bits), (uint32_t)d, region_bits, sv);
								if (T[sa].key) ok = false;
								for (size_t c = 0; c < a && ok; ++c)
									if (sa == pd_slot(pb_rest(idx->keys[keys[c]], pb_bits), (uint32_t)d, region_bits, sv)) ok = false;
							}
							if (ok) break;
						}
						if (d == 256) { placed = false; break; }
						disp[(size_t)b * NB + o] = (uint8_t)d;
						for (uint32_t ki : keys) {
							const uint32_t h = idx->keys[ki];
							const uint64_t off = idx->key_off[ki], c = idx->key_off[ki + 1] - off;
							TableSlot &sl = T[pd_slot(pb_rest(h, pb_bits), (uint32_t)d, region_bits, sv)];
							sl.key = h + 1, sl.cnt = (uint32_t)c, sl.val = c == 1 ? idx->pos[off] : off;
						}
					}
					if (placed) salt[b] = sv;
				}
				if (!placed) { failed.store(b); return; }
			}
		};
		unsigned nt = std::thread::hardware_concurrency();
		nt = nt == 0 ? 4 : nt > 16 ? 16 : nt;
		std::vector<std::thread> pool;
		for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
		work();
		for (auto &t : pool) t.join();
		if (failed.load() < 0) break;
		if (grow == MAX_GROW) {                                   // more room did not help either
			const int b = failed.load();
			set_error("perfect hashing of table region %d failed (region of %zu keys, R %zu, NB %zu)", b, reg[b].size(), R, NB);
			return MNC_ERR_UNSUPPORTED;
		}
	}
	DeviceIndex d;
	d.device = device, d.pb_bits = pb_bits, d.region_bits = region_bits, d.disp_bits = disp_bits;
	auto upload = [&](void **dst, const void *src, size_t bytes, size_t spare = 0) -> int {
		HIP_TRY(hipMalloc(dst, bytes + spare ? bytes + spare : 8));
		if (bytes) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
		return MNC_OK;
	};
	int urc = upload((void**)&d.filter, filt.data(), filt.size() * 4);
	if (!urc) urc = upload((void**)&d.disp, disp.data(), disp.size());
	if (!urc) urc = upload((void**)&d.salt, salt.data(), salt.size() * 4);
	if (!urc) urc = upload((void**)&d.table, tab.data(), tab.size() * sizeof(TableSlot));
	if (!urc) urc = upload((void**)&d.positions, idx->pos.data(), idx->pos.size() * 8, 8);   // one spare word behind the positions
	if (!urc) urc = upload((void**)&d.contig_genome, idx->contig_genome.data(), idx->contig_genome.size() * 4);
	if (!urc) urc = upload((void**)&d.seq4, idx->seq4.data(), idx->seq4.size() * 4, 16);
	if (!urc) urc = upload((void**)&d.seq_off, idx->seq_off.data(), idx->seq_off.size() * 8);
	if (urc) {                                               // do not leak the part that made it to the device
		void *parts[] = { d.filter, d.disp, d.salt, d.table, d.positions, d.contig_genome, d.seq4, d.seq_off };
		for (void *q : parts) if (q) (void)hipFree(q);
		return urc;
	}
	d.bytes = (int64_t)(tab.size() * sizeof(TableSlot) + filt.size() * 4 + disp.size() + (idx->pos.size() + 1) * 8 + idx->contig_genome.size() * 4 +
	                    idx->seq4.size() * 4 + idx->seq_off.size() * 8);
	idx->dev.push_back(d);
	*out = &idx->dev.back();
	return MNC_OK;
}

void index_release_device(mnc_index *idx)
{
	std::lock_guard<std::mutex> lk(idx->dev_mutex);
	for (auto &d : idx->dev) {
		if (hipSetDevice(d.device) != hipSuccess) continue;
		if (d.table) (void)hipFree(d.table);
		if (d.filter) (void)hipFree(d.filter);
		if (d.disp) (void)hipFree(d.disp);
		if (d.salt) (void)hipFree(d.salt);
		if (d.positions) (void)hipFree(d.positions);
		if (d.contig_genome) (void)hipFree(d.contig_genome);
		if (d.seq4) (void)hipFree(d.seq4);
		if (d.seq_off) (void)hipFree(d.seq_off);
	}
	idx->dev.clear();
}

} // namespace mnc

using namespace mnc;

// ================================================================ engine
struct mnc_engine {
	mnc_index *idx = nullptr;
	DeviceIndex *didx = nullptr;
	int device = 0;
	hipStream_t stream = nullptr;
	// side streams: the per-size-class launches of one stage are independent and run side by side
	static constexpr int N_SIDE = 4;
	hipStream_t side[N_SIDE]{};
	hipEvent_t ev_fork = nullptr, ev_join[N_SIDE]{}, ev_lfill = nullptr;
	// constant tables
	Buf gap_lut, logf_lut, logf_a_lut;
	int logf_n = 0;
	// inputs / outputs for the host-buffer entry point
	Buf in_bases, in_offsets, out_assign, out_best, out_nhits;
	// the next batch's bases on their way to the device while this one is classified (mnc_engine_prefetch)
	Buf pf_bases_buf, pf_offsets_buf;
	hipStream_t copy_stream = nullptr;
	hipEvent_t ev_prefetch = nullptr;
	std::mutex pf_mu;
	bool pf_valid = false;
	const uint8_t *pf_bases = nullptr;
	const int64_t *pf_offsets = nullptr;
	uint32_t pf_n = 0;
	int64_t pf_total = 0;
	// per base slot
	Buf packed, mz, hits, hist_tm, q_off, qrec, bhits, bhit_cnt;
	size_t q_cap_override = 0;              // grown after an overflowing batch
	// per read
	Buf ambig, skip, mz_cnt, hit_cnt, rep_len, an_cnt, an_off, n_chain, n_reg, scan_sums, hit_off, best_mlen, hist_sums, hist_offs;
	// per anchor
	Buf a, f, p, v, t, u;
	// per chain slot
	Buf chains_tmp, regs, regx, k64a, k64b, tmp_i32, gated, hits_csr;
	Buf stats, cls_count, cls_list;
	// base-level alignment stage
	int contract = MNC_CONTRACT_DP;
	Buf ca, ca_cnt, chain_dst, regdp, segs, cig_seg, cig_reg, dp_ctr, work_a, work_b, big_list, huge_list, reg_cnt, regs2;
	SharedWs *ws = nullptr;                  // the device's alignment scratch (held during the alignment stage only)
	Buf fill1, fill2, fill3, fill_mid, fill_fb, plan_long, extp, mid_list, lfill, lext, bigfb, ext1, ext2, ext3, ext4, gen_list;
	size_t seg_cap_override = 0, cig_cap_override = 0;
	uint32_t *mailbox = nullptr;             // 1 KiB of page-locked host memory the device writes its small read-backs to (mnc_mail)
	int last_redos = 0;                      // how often the last batch was redone with larger budgets (query records; segment / CIGAR pools; region slots)
	long long prev_wide_calls = 0;           // calls the last large batch planned for the literal kernel's large-workspace passes (1 and 5)
	int slot_pad = 2;                        // region slots per read beyond anchors / 3 (device.h: reg_slot); grown when a batch runs out
	Buf inv_ws;                              // mnc_dp_inv: two columns per workgroup
	int cur_max_read_len = 0;                // of the batch being classified (sizes the stitch kernel's LDS)
	int debug = 0;                           // bit mask (tests): 2 stress build of the chaining ring, 4 displacement bytes read from HBM, 0x10000 alignment kernels one at a time (with stage timers), 0x200000 the stitch kernel reads bases in place (its form for regions beyond its LDS)
	// last batch
	Batch B{};
	bool have_batch = false;
	int64_t last_total_anchors = 0, last_total_hits = -1;
	std::vector<int64_t> h_offsets;          // host copy of the last offsets (for dumps)
	// profiling
	bool profiling = false;
	hipEvent_t ev[MNC_N_STAGES][2]{};
	bool ev_used[MNC_N_STAGES]{};
	double ms[MNC_N_STAGES]{};
	int64_t launches[MNC_N_STAGES]{};
};

static const char *STAGE_NAME[MNC_N_STAGES] = { "pack", "sketch", "partition", "probe", "collect", "offsets", "sort", "chain", "backtrack", "regions", "gather",
                                                "dp_plan", "dp_align", "dp_stitch", "dp_post", "dp_fill", "dp_fill_t1", "dp_fill_t2", "dp_fill_t3", "dp_ext", "dp_fill_tm", "dp_lfill" };
static const char *STAGE_KERNEL[MNC_N_STAGES] = {
	"mnc_pack_bases", "mnc_sketch_minimizers", "mnc_partition_queries", "mnc_probe_buckets", "mnc_collect_hits",
	"mnc_bin_reads", "mnc_expand_sort", "mnc_chain_dp_ring", "mnc_chain_tail", "mnc_regions_decide", "mnc_gather_hits",
	"mnc_dp_plan", "mnc_dp_align", "mnc_dp_stitch", "mnc_regions_post", "mnc_dp_fillp", "mnc_dp_fillp<16>", "mnc_dp_fillp<32>", "mnc_dp_fillp<64>", "mnc_dp_extp", "mnc_dp_fillp<21>", "mnc_dp_fill<64, 4, 2047, 1024>" };

extern "C" const char *mnc_stage_name(int s) { return s >= 0 && s < MNC_N_STAGES ? STAGE_NAME[s] : nullptr; }
extern "C" const char *mnc_stage_kernel(int s) { return s >= 0 && s < MNC_N_STAGES ? STAGE_KERNEL[s] : nullptr; }

// page-locked host buffers for the batch readers (hostio.cpp); plain malloc without a GPU.  Locking a
// gigabyte of pages takes ~0.1 s, as long as classifying the reads in it: freed buffers are kept
// (up to PINNED_KEEP bytes) and handed out again.
static std::mutex g_pinned_mu;
struct PinnedBuf { void *p; size_t bytes; bool in_use; };
static std::vector<PinnedBuf> g_pinned;
constexpr size_t PINNED_KEEP = 6ull << 30;

extern "C" void *mnc_host_alloc(size_t bytes)
{
	if (bytes == 0) bytes = 1;
	{
		std::lock_guard<std::mutex> g(g_pinned_mu);
		size_t best = g_pinned.size();
		for (size_t i = 0; i < g_pinned.size(); ++i)             // the smallest kept buffer that is large enough (and not wasteful)
			if (!g_pinned[i].in_use && g_pinned[i].bytes >= bytes && g_pinned[i].bytes <= 2 * bytes + (1u << 20) &&
			    (best == g_pinned.size() || g_pinned[i].bytes < g_pinned[best].bytes)) best = i;
		if (best != g_pinned.size()) { g_pinned[best].in_use = true; return g_pinned[best].p; }
	}
	int c = 0;
	void *p = nullptr;
	if (hipGetDeviceCount(&c) == hipSuccess && c > 0 && hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) {
		std::lock_guard<std::mutex> g(g_pinned_mu);
		g_pinned.push_back({ p, bytes, true });
		return p;
	}
	(void)hipGetLastError();
	return malloc(bytes);
}

extern "C" void mnc_host_free(void *p)
{
	if (!p) return;
	{
		std::lock_guard<std::mutex> g(g_pinned_mu);
		for (size_t i = 0; i < g_pinned.size(); ++i)
			if (g_pinned[i].p == p) {
				g_pinned[i].in_use = false;
				size_t kept = 0;
				for (const PinnedBuf &b : g_pinned) if (!b.in_use) kept += b.bytes;
				if (kept > PINNED_KEEP) {                            // over the budget: this one goes back to the system
					g_pinned[i] = g_pinned.back();
					g_pinned.pop_back();
					(void)hipHostFree(p);
				}
				return;
			}
	}
	free(p);
}

extern "C" int mnc_device_count(int *n)
{
	if (!n) return MNC_ERR_ARG;
	int c = 0;
	if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
	*n = c;
	return MNC_OK;
}

extern "C" int mnc_device_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes)
{
	if (int rc = check_device(device)) return rc;
	HIP_TRY(hipSetDevice(device));
	size_t f = 0, t = 0;
	HIP_TRY(hipMemGetInfo(&f, &t));
	if (free_bytes) *free_bytes = (int64_t)f;
	if (total_bytes) *total_bytes = (int64_t)t;
	return MNC_OK;
}

extern "C" int mnc_device_name(int device, char *buf, size_t cap)
{
	if (!buf || cap == 0) return MNC_ERR_ARG;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return MNC_ERR_NODEVICE;
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	snprintf(buf, cap, "%s (%s)", prop.name, prop.gcnArchName);
	return MNC_OK;
}

// every device buffer an engine owns (release, accounting)
template <class F> static void engine_bufs(mnc_engine *e, F f)
{
	Buf *all[] = { &e->gap_lut, &e->logf_lut, &e->logf_a_lut, &e->ca, &e->ca_cnt, &e->chain_dst, &e->regdp, &e->segs, &e->cig_seg, &e->cig_reg, &e->dp_ctr,
	               &e->work_a, &e->work_b, &e->big_list, &e->huge_list, &e->reg_cnt, &e->regs2, &e->inv_ws, &e->fill1, &e->fill2, &e->fill3, &e->fill_mid, &e->fill_fb, &e->plan_long, &e->extp, &e->mid_list, &e->lfill, &e->lext, &e->bigfb, &e->ext1, &e->ext2, &e->ext3, &e->ext4, &e->gen_list, &e->in_bases, &e->in_offsets, &e->pf_bases_buf, &e->pf_offsets_buf, &e->out_assign, &e->out_best, &e->out_nhits,
	               &e->packed, &e->mz, &e->hits, &e->hist_tm, &e->q_off, &e->qrec, &e->bhits, &e->bhit_cnt, &e->ambig, &e->skip, &e->mz_cnt, &e->hit_cnt, &e->rep_len, &e->an_cnt, &e->an_off,
	               &e->n_chain, &e->n_reg, &e->best_mlen, &e->hist_sums, &e->hist_offs, &e->scan_sums, &e->hit_off, &e->a, &e->f, &e->p, &e->v, &e->t, &e->u,
	               &e->chains_tmp, &e->regs, &e->regx, &e->k64a, &e->k64b, &e->tmp_i32, &e->gated,
	               &e->hits_csr, &e->stats, &e->cls_count, &e->cls_list };
	for (Buf *b : all) f(b);
}

extern "C" void mnc_engine_destroy(mnc_engine *e)
{
	if (!e) return;
	(void)hipSetDevice(e->device);
	if (e->stream) (void)hipStreamSynchronize(e->stream);
	if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
	engine_bufs(e, [](Buf *b) { b->release(); });
	if (e->ws) { (void)shared_ws(e->device, -1); e->ws = nullptr; }
	for (int s = 0; s < MNC_N_STAGES; ++s) for (int k = 0; k < 2; ++k) if (e->ev[s][k]) (void)hipEventDestroy(e->ev[s][k]);
	for (int k = 0; k < mnc_engine::N_SIDE; ++k) {
		if (e->side[k]) (void)hipStreamDestroy(e->side[k]);
		if (e->ev_join[k]) (void)hipEventDestroy(e->ev_join[k]);
	}
	if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
	if (e->ev_lfill) (void)hipEventDestroy(e->ev_lfill);
	if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
	if (e->ev_prefetch) (void)hipEventDestroy(e->ev_prefetch);
	if (e->mailbox) (void)hipHostFree(e->mailbox);
	if (e->stream) (void)hipStreamDestroy(e->stream);
	delete e;
}

extern "C" int mnc_engine_create(mnc_index *idx, int device, mnc_engine **out)
{
	if (!idx || !out) return MNC_ERR_ARG;
	*out = nullptr;
	if (int rc = check_device(device)) return rc;
	if (idx->par.bw >= GAP_LUT) { set_error("bw too large for the gap look-up"); return MNC_ERR_UNSUPPORTED; }
	mnc_engine *e = new (std::nothrow) mnc_engine;
	if (!e) return MNC_ERR_NOMEM;
	e->idx = idx, e->device = device;
	e->ws = shared_ws(device, +1);
	int rc = index_upload(idx, device, &e->didx);
	if (rc) { mnc_engine_destroy(e); return rc; }      // gives the workspace reference back, too
	hipError_t he = hipSetDevice(device);
	int least = 0, greatest = 0;
	if (he == hipSuccess) {
		// the batch's own stream carries the latency-bound kernels (a few long calls on single waves) beside the
		// chip-filling ones of the side streams: its workgroups go first when wave slots come free
		if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
		he = hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, greatest);
	}
	for (int k = 0; k < mnc_engine::N_SIDE && he == hipSuccess; ++k) {
		// (three streams of the default class get a hardware queue each; a fourth shares one with the first -- measured:
		// the long extensions behind the literal kernel's long passes, profiles/r05j_timeline.txt -- so the fourth is
		// made in the batch stream's class, whose queues only that stream uses)
		he = k < 3 ? hipStreamCreateWithFlags(&e->side[k], hipStreamNonBlocking) : hipStreamCreateWithPriority(&e->side[k], hipStreamNonBlocking, greatest);
		if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_join[k], hipEventDisableTiming);
	}
	if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
	if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_lfill, hipEventDisableTiming);
	// The copy stream gets the LOWEST priority -- not for the priority: the runtime keeps one set of hardware queues per
	// priority class and deals a class's streams round-robin over its queues, so a fifth stream of the side streams' class
	// shares a queue with one of them, and the barrier packet behind a half-gigabyte copy (the event record of
	// mnc_engine_prefetch) then holds back that side stream's kernels until the copy is over: the sort launches of a
	// batch took 7.9 ms instead of 0.8 whenever the next batch's copy had been issued before them
	// (profiles/r03k_e2e_delay.txt).  Only copies and markers ever go to this stream.
	if (he == hipSuccess) he = hipStreamCreateWithPriority(&e->copy_stream, hipStreamNonBlocking, least);
	if (he == hipSuccess) he = hipEventCreateWithFlags(&e->ev_prefetch, hipEventDisableTiming);
	if (he == hipSuccess) he = hipHostMalloc((void**)&e->mailbox, 1024, hipHostMallocDefault);
	if (he != hipSuccess) { set_error("stream creation failed: %s", hipGetErrorString(he)); mnc_engine_destroy(e); return MNC_ERR_HIP; }
	for (int s = 0; s < MNC_N_STAGES; ++s) for (int k = 0; k < 2; ++k) (void)hipEventCreate(&e->ev[s][k]);
	// gap cost: (int)(dd * .01 * avg_span) + (ilog2(dd) >> 1), evaluated in double exactly as
	// minimap2 does; avg_span is the float mean of the spans == k
	std::vector<int32_t> gap(GAP_LUT);
	const float avg_span = (float)idx->k;
	for (int dd = 0; dd < GAP_LUT; ++dd) {
		int lg = 0;
		for (uint32_t v = (uint32_t)dd; v >>= 1;) ++lg;
		gap[dd] = (int)(dd * .01 * avg_span) + (dd ? lg >> 1 : 0);
	}
	e->logf_n = 1 << 21;
	std::vector<float> lg((size_t)e->logf_n);
	for (int i = 0; i < e->logf_n; ++i) lg[i] = logf((float)i);
	// logf((float)dp_max / a) of the DP branch of the MAPQ formula (A.7), indexed by dp_max
	std::vector<float> lga((size_t)e->logf_n);
	for (int i = 0; i < e->logf_n; ++i) lga[i] = logf((float)i / idx->par.a);
	rc = e->gap_lut.ensure(GAP_LUT * 4);
	if (!rc) rc = e->logf_lut.ensure((size_t)e->logf_n * 4);
	if (!rc) rc = e->logf_a_lut.ensure((size_t)e->logf_n * 4);
	if (!rc) rc = e->dp_ctr.ensure(64 * 8);
	if (!rc) rc = dp_align_prepare(DP_LDS_BYTES);
	if (!rc) rc = e->stats.ensure(16 * 8);
	if (!rc) rc = e->cls_count.ensure((MAX_CHAIN_CLASSES + 1) * 4 + 64);
	if (!rc) rc = chain_tail_prepare(chain_tail_lds_bytes(CHAIN_CLASSES.nm[CHAIN_CLASSES.n - 1]));
	if (!rc) rc = expand_sort_prepare(CHAIN_CLASSES.nm[CHAIN_CLASSES.n - 1]);
	if (!rc) rc = dp_stitch_prepare();
	if (!rc) rc = partition_prepare();
	if (!rc) rc = dp_gather_long_prepare(CHAIN_CLASSES.nm[CHAIN_CLASSES.n - 1]);
	if (!rc) rc = dp_plan_prepare();
	if (!rc) {
		he = hipMemcpy(e->gap_lut.p, gap.data(), GAP_LUT * 4, hipMemcpyHostToDevice);
		if (he == hipSuccess) he = hipMemcpy(e->logf_lut.p, lg.data(), (size_t)e->logf_n * 4, hipMemcpyHostToDevice);
		if (he == hipSuccess) he = hipMemcpy(e->logf_a_lut.p, lga.data(), (size_t)e->logf_n * 4, hipMemcpyHostToDevice);
		if (he == hipSuccess) he = hipMemset(e->stats.p, 0, 16 * 8);
		if (he != hipSuccess) { set_error("table upload failed: %s", hipGetErrorString(he)); rc = MNC_ERR_HIP; }
	}
	if (rc) { mnc_engine_destroy(e); return rc; }
	*out = e;
	return MNC_OK;
}

// HBM held by this engine's own buffers (the per-device alignment scratch it shares is not counted)
extern "C" int mnc_engine_device_bytes(mnc_engine *e, int64_t *bytes)
{
	if (!e || !bytes) return MNC_ERR_ARG;
	size_t n = 0;
	engine_bufs(e, [&](Buf *b) { n += b->cap; });
	*bytes = (int64_t)n;
	return MNC_OK;
}

// Another index part behind the same engine (stream, batch buffers): the reference's loop over the parts of a
// database rebinds `index` and keeps its thread (aligner.py:91-103).  The part is made resident if it is not yet;
// the tables an engine holds of its own (gap costs by k, logf(dp_max / a)) must fit the new part.
extern "C" int mnc_engine_set_index(mnc_engine *e, mnc_index *idx)
{
	if (!e || !idx) return MNC_ERR_ARG;
	if (idx == e->idx) return MNC_OK;
	if (idx->k != e->idx->k || idx->w != e->idx->w || idx->par.a != e->idx->par.a) {
		set_error("the engine was made for k = %d, w = %d, a = %d; the index has k = %d, w = %d, a = %d", e->idx->k, e->idx->w, e->idx->par.a, idx->k, idx->w, idx->par.a);
		return MNC_ERR_UNSUPPORTED;
	}
	if (idx->par.bw >= GAP_LUT) { set_error("bw too large for the gap look-up"); return MNC_ERR_UNSUPPORTED; }
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (int rc = mnc_engine_prefetch_cancel(e)) return rc;       // a batch announced for the previous part's run is not this run's
	DeviceIndex *d = nullptr;
	if (int rc = index_upload(idx, e->device, &d)) return rc;
	e->idx = idx, e->didx = d;
	e->have_batch = false;
	return MNC_OK;
}

extern "C" void *mnc_engine_stream(mnc_engine *e) { return e ? (void*)e->stream : nullptr; }

// test / tuning switch: this many region bits (8 .. 10) for the device tables made from now on; 0 = by the index's size
extern "C" int mnc_index_set_region_bits(mnc_index *idx, int bits)
{
	if (!idx || (bits != 0 && (bits < PB_BITS_MIN || bits > PB_BITS_MAX))) return MNC_ERR_ARG;
	std::lock_guard<std::mutex> lk(idx->dev_mutex);
	idx->force_pb_bits = bits;
	return MNC_OK;
}

extern "C" int mnc_index_set_host_tables(mnc_index *idx, int on)
{
	if (!idx) return MNC_ERR_ARG;
	std::lock_guard<std::mutex> lk(idx->dev_mutex);
	idx->host_tables = on != 0;
	return MNC_OK;
}

// the device tables of an engine's index, for tests: [region_bits, disp_bits] int32, salt[PB_N], displacement bytes,
// presence filter words, table slots
extern "C" int mnc_engine_dump_tables(mnc_engine *e, void *dst, int64_t cap_bytes, int64_t *n_bytes)
{
	if (!e || !n_bytes || !e->didx) return MNC_ERR_ARG;
	HIP_TRY(hipSetDevice(e->device));
	const DeviceIndex &d = *e->didx;
	const size_t R = (size_t)1 << d.region_bits, NB = (size_t)1 << d.disp_bits, PB_N = (size_t)1 << d.pb_bits;
	const size_t sz[] = { 8, PB_N * 4, (size_t)PB_N * NB, (size_t)PB_N * PF_WORDS * 4, (size_t)PB_N * R * sizeof(TableSlot) };
	*n_bytes = (int64_t)(sz[0] + sz[1] + sz[2] + sz[3] + sz[4]);
	if (!dst || cap_bytes < *n_bytes) return MNC_ERR_RANGE;
	uint8_t *o = (uint8_t*)dst;
	const int32_t hdr[2] = { d.region_bits | d.pb_bits << 16, d.disp_bits };
	memcpy(o, hdr, 8), o += 8;
	HIP_TRY(hipMemcpy(o, d.salt, sz[1], hipMemcpyDeviceToHost));
	o += sz[1];
	HIP_TRY(hipMemcpy(o, d.disp, sz[2], hipMemcpyDeviceToHost));
	o += sz[2];
	HIP_TRY(hipMemcpy(o, d.filter, sz[3], hipMemcpyDeviceToHost));
	o += sz[3];
	HIP_TRY(hipMemcpy(o, d.table, sz[4], hipMemcpyDeviceToHost));
	return MNC_OK;
}

extern "C" int mnc_engine_sync(mnc_engine *e)
{
	if (!e) return MNC_ERR_ARG;
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (e->profiling) {
		for (int s = 0; s < MNC_N_STAGES; ++s) {
			if (!e->ev_used[s]) continue;
			float t = 0;
			if (hipEventElapsedTime(&t, e->ev[s][0], e->ev[s][1]) == hipSuccess) e->ms[s] += t, e->launches[s] += 1;
			e->ev_used[s] = false;
		}
	}
	return MNC_OK;
}

extern "C" int mnc_engine_set_profiling(mnc_engine *e, int on)
{
	if (!e) return MNC_ERR_ARG;
	e->profiling = on != 0;
	return MNC_OK;
}

extern "C" int mnc_engine_set_contract(mnc_engine *e, int contract)
{
	if (!e || (contract != MNC_CONTRACT_DP && contract != MNC_CONTRACT_CHAIN)) return MNC_ERR_ARG;
	e->contract = contract;
	return MNC_OK;
}

extern "C" int mnc_engine_set_debug(mnc_engine *e, int on)
{
	if (!e) return MNC_ERR_ARG;
	e->debug = on;
	return MNC_OK;
}

extern "C" int mnc_engine_get_timings(mnc_engine *e, double *ms, int64_t *launches, int reset)
{
	if (!e) return MNC_ERR_ARG;
	for (int s = 0; s < MNC_N_STAGES; ++s) {
		if (ms) ms[s] = e->ms[s];
		if (launches) launches[s] = e->launches[s];
		if (reset) e->ms[s] = 0, e->launches[s] = 0;
	}
	return MNC_OK;
}

namespace {
struct StageTimer {
	mnc_engine *e; int s;
	StageTimer(mnc_engine *e_, int s_) : e(e_), s(s_) { if (e->profiling) (void)hipEventRecord(e->ev[s][0], e->stream); }
	~StageTimer() { if (e->profiling) { (void)hipEventRecord(e->ev[s][1], e->stream); e->ev_used[s] = true; } }
};
}

// The kernel calls of a round run on four streams side by side -- they touch different segments --
// and meet before the literal kernel's last passes, which take what the others handed back:
//   s0  gap fillings on the packed banded kernel: 32 cells per segment, then 64, then 128 for those whose band
//       could not be proven wide enough
//   s1  the extensions: the packed kernel (by query length and side), then the step-by-step kernel for what it
//       handed back
//   s2  (the batch's own stream) the long gaps first, the long extensions, the literal kernel's small calls
//   s3  (the side stream with a hardware queue of its own) the literal kernel's few long calls
// The chip-filling kernels (s0, s1) are bound by vector issue; the others are serial work on single waves that
// fills the gaps.  With debug bit 0x10000 all four are the same stream (one kernel at a time, per-kernel timers).
static void align_round(const Batch &B, mnc_engine *e, hipStream_t s0, hipStream_t s1, hipStream_t s2, hipStream_t s3, hipStream_t s4)
{
	// per-kernel timers: only when everything runs on one stream (debug bit 0x10000)
	const bool timed = e->profiling && s0 == s1;
	const bool serial = s0 == s1;
	auto mark = [&](int stage, int which) { if (timed) { (void)hipEventRecord(e->ev[stage][which], s0); if (which) e->ev_used[stage] = true; } };
	// Five streams side by side (round 5: the long extensions have one of their own, and what the long kernels hand back
	// that needs the large workspace is aligned INSIDE the window):
	//   s0  gap fillings on the packed banded kernel, tier by tier          (chip-filling, bound by vector issue)
	//   s1  the extensions: the packed kernel by class, then the step-by-step kernel (chip-filling)
	//   s2  (the batch's own stream) the long gaps, then the literal kernel's small calls
	//   s3  the literal kernel's long passes: one launch of one-wave workgroups that stay until their queues are empty
	//   s4  the long extensions, and behind them (and behind the long gaps) the first round of pass 4
	// The single-wave kernels (s2 .. s4) are launched FIRST: their waves are on the chip before the persistent
	// workgroups of the tiers take every wave slot, and each is a chain of long calls whose length is a latency, not
	// work -- behind one another on one stream they were the window of a batch of divergent reads
	// (profiles/r05h_timeline.txt: long gaps 22 ms, then long extensions 23 ms at 16 % errors).
	const bool small_batch = B.n_reads < 4096;
	auto long_passes = [&](hipStream_t sl) {
		if (dp_align_long_packed(B) && small_batch) {
			// a micro-batch has a handful of long calls and waits for the longest: each pass in both packed forms, the
			// pass's call count picks one on the device -- four waves on a call for up to 64 calls, one wave a call for more
			launch_dp_align(B, e->ws->dp_ws_big.as<uint8_t>(), DP_WG_BIG, DP_STATE_BIG, DP_P_BIG, DP_CIG_BIG, DP_LDS_BYTES, 0, 0, 1, sl);
			launch_dp_align(B, e->ws->dp_ws_mid.as<uint8_t>(), DP_WG_MID, DP_STATE_SMALL, DP_P_SMALL, DP_CIG_SMALL, DP_LDS_BYTES, 0, 0, 3, sl);
			launch_dp_align(B, e->ws->dp_ws_huge.as<uint8_t>(), DP_WG_HUGE, DP_STATE_HUGE, DP_P_HUGE, DP_CIG_HUGE, DP_LDS_BYTES, 0, 0, 5, sl);
			return;
		}
		if (dp_align_long_packed(B)) {                          // round 5: the three passes as one launch of one-wave workgroups (packed pairs in registers)
			// (a micro-batch has a handful of such calls: a launch of thousands of workgroups that find nothing costs it more
			// than the calls themselves: workgroups by the batch's reads -- a read rarely has more than two such calls)
			const int n_big = std::min<int>(DP_WG_BIG, std::max<int>(64, (int)B.n_reads / 2)), n_mid = std::min<int>(DP_WG_MID, std::max<int>(128, (int)B.n_reads));
			launch_dp_align_long(B, e->ws->dp_ws_huge.as<uint8_t>(), DP_WG_HUGE, DP_STATE_HUGE, DP_P_HUGE, DP_CIG_HUGE,
			                     e->ws->dp_ws_big.as<uint8_t>(), n_big, DP_STATE_BIG, DP_P_BIG, DP_CIG_BIG,
			                     e->ws->dp_ws_mid.as<uint8_t>(), n_mid, DP_STATE_SMALL, DP_P_SMALL, DP_CIG_SMALL, sl);
			return;
		}
		const int forms_long = small_batch || e->prev_wide_calls > 0 ? 3 : 1, forms_mid = small_batch ? 3 : 1;
		launch_dp_align(B, e->ws->dp_ws_big.as<uint8_t>(), DP_WG_BIG, DP_STATE_BIG, DP_P_BIG, DP_CIG_BIG, DP_LDS_BYTES, 0, 0, 1, sl, forms_long | (forms_long == 1 ? 4 : 0));
		launch_dp_align(B, e->ws->dp_ws_mid.as<uint8_t>(), DP_WG_MID, DP_STATE_SMALL, DP_P_SMALL, DP_CIG_SMALL, DP_LDS_BYTES, 0, 0, 3, sl, forms_mid | (forms_mid == 1 ? 4 : 0));
		launch_dp_align(B, e->ws->dp_ws_huge.as<uint8_t>(), DP_WG_HUGE, DP_STATE_HUGE, DP_P_HUGE, DP_CIG_HUGE, DP_LDS_BYTES, 0, 0, 5, sl, forms_long | (forms_long == 1 ? 4 : 0));
	};
	const bool packed_long = dp_align_long_packed(B);
	if (!serial) {
		if (packed_long || !small_batch) long_passes(s3);
		launch_dp_lext(B, B.lext_list, 62, 63, B.fill_fb, 12, e->ws->lext_p.as<uint8_t>(), std::min<int>(DP_WG_LEXT, std::max<int>(256, (int)B.n_reads)), s4);
		launch_dp_lfill(B, B.lfill_list, 31, 61, B.fill_fb, 12, e->ws->lfill_p.as<uint8_t>(), std::min<int>(DP_WG_LFILL, std::max<int>(512, (int)B.n_reads * 2)), s2);
		// what the two have handed back for the large workspace (a long extension whose Z-drop may fire is one literal
		// call of thousands of anti-diagonals): aligned now, beside the tiers, not behind the window's join
		(void)hipEventRecord(e->ev_lfill, s2);
		(void)hipStreamWaitEvent(s4, e->ev_lfill, 0);
		launch_dp_align(B, e->ws->dp_ws_bigfb.as<uint8_t>(), std::min<int>(DP_WG_BIGFB, std::max<int>(32, (int)B.n_reads / 8)), DP_STATE_BIG, DP_P_BIG, DP_CIG_BIG, DP_LDS_BYTES, 0, 0, 4, s4);
	}
	mark(MNC_STAGE_DP_FILL_T1, 0);
	launch_dp_fill(B, 32, B.fill_list1, 10, 13, B.fill_list_mid, 30, B.fill_fb, 12, e->ws->fill_p.as<uint8_t>(), e->ws->fill_cig.as<uint32_t>(), DP_WG_FILL, s0);
	mark(MNC_STAGE_DP_FILL_T1, 1), mark(MNC_STAGE_DP_FILL_TM, 0);
	launch_dp_fill(B, FILL_MID_CELLS, B.fill_list_mid, 30, 54, B.fill_list2, 11, B.fill_fb, 12, e->ws->fill_p.as<uint8_t>(), e->ws->fill_cig.as<uint32_t>(), DP_WG_FILL, s0);
	mark(MNC_STAGE_DP_FILL_TM, 1), mark(MNC_STAGE_DP_FILL_T2, 0);
	launch_dp_fill(B, 64, B.fill_list2, 11, 14, B.fill_list3, 22, B.fill_fb, 12, e->ws->fill_p.as<uint8_t>(), e->ws->fill_cig.as<uint32_t>(), DP_WG_FILL, s0);
	mark(MNC_STAGE_DP_FILL_T2, 1), mark(MNC_STAGE_DP_FILL_T3, 0);
	launch_dp_fill(B, 128, B.fill_list3, 22, 23, B.fill_fb, 12, B.fill_fb, 12, e->ws->fill_p.as<uint8_t>(), e->ws->fill_cig.as<uint32_t>(), DP_WG_FILL, s0);
	if (!serial && !packed_long && small_batch) long_passes(s0);   // (the older forms in a micro-batch: behind the tiers)
	mark(MNC_STAGE_DP_FILL_T3, 1), mark(MNC_STAGE_DP_LFILL, 0);
	if (serial) launch_dp_lfill(B, B.lfill_list, 31, 61, B.fill_fb, 12, e->ws->lfill_p.as<uint8_t>(), DP_WG_LFILL, s0);   // one kernel at a time (profiling)
	mark(MNC_STAGE_DP_LFILL, 1), mark(MNC_STAGE_DP_EXT, 0);
	// the classes with the longest queries first: few calls, each long -- at the end of the stream they would be a tail
	// of a few busy waves; the short ones (most of the calls) drain evenly
	for (int i = 7; i >= 0; --i)
		launch_dp_extp(B, 32 << (i >> 1), i & 1, B.extp_list + (int64_t)i * B.seg_cap, 32 + i, 40 + i, e->ws->extp_p.as<uint8_t>(), e->ws->extp_cig.as<uint32_t>(), DP_WG_EXT, s1);
	launch_dp_ext(B, 32, B.ext_list1, 16, 18, B.fill_fb, 12, e->ws->ext_p.as<uint8_t>(), DP_WG_EXT, s1);
	launch_dp_ext(B, 64, B.ext_list2, 17, 19, B.fill_fb, 12, e->ws->ext_p.as<uint8_t>(), DP_WG_EXT, s1);
	launch_dp_ext(B, 128, B.ext_list3, 24, 26, B.fill_fb, 12, e->ws->ext_p.as<uint8_t>(), DP_WG_EXT / 2, s1);
	launch_dp_ext(B, 256, B.ext_list4, 25, 27, B.fill_fb, 12, e->ws->ext_p.as<uint8_t>(), DP_WG_EXT / 4, s1);
	mark(MNC_STAGE_DP_EXT, 1);
	if (serial) {                                              // one kernel at a time (profiling)
		launch_dp_lext(B, B.lext_list, 62, 63, B.fill_fb, 12, e->ws->lext_p.as<uint8_t>(), DP_WG_LEXT, s0);
		long_passes(s0);
	}
	launch_dp_align(B, e->ws->dp_ws.as<uint8_t>(), 2 * DP_WG_SMALL, DP_STATE_SMALL, DP_P_SMALL, DP_CIG_SMALL, DP_LDS0_STATE, DP_LDS0_P, DP_LDS0_CIG, 0, s2);
}
// what the other kernels handed back, on the literal kernel
static void align_rest(const Batch &B, mnc_engine *e, hipStream_t st)
{
	launch_dp_align(B, e->ws->dp_ws.as<uint8_t>(), DP_WG_SMALL, DP_STATE_SMALL, DP_P_SMALL, DP_CIG_SMALL, DP_LDS_BYTES, 0, 0, 2, st);
	// (pass 4 has run inside the window on what the long kernels handed back -- all of it, as a rule: this second round
	// goes on where that one stopped, with a launch sized for leftovers unless the older forms are asked for)
	launch_dp_align(B, e->ws->dp_ws_big.as<uint8_t>(), (dp_align_long_packed(B) && !(e->debug & 0x10000)) ? 64 : DP_WG_BIG, DP_STATE_BIG, DP_P_BIG, DP_CIG_BIG, DP_LDS_BYTES, 0, 0, 4, st);
}

// ---------------------------------------------------------------- one batch, device-resident
static int classify_once(mnc_engine *e, const uint8_t *d_bases, const int64_t *d_offsets,
                         uint32_t n_reads, int64_t total_bases, int min_mapq,
                         int32_t *d_assign, mnc_hit_t *d_best, int32_t *d_nhits, int64_t *d_counts, int *overflowed)
{
	hipStream_t st = e->stream;
	const mnc_index *idx = e->idx;
	*overflowed = 0;                                  // 1: query records, 2: segments / CIGAR pools of the alignment stage
	const size_t nr = (size_t)n_reads, nb = (size_t)total_bases;
	const int pb_bits = e->didx->pb_bits;
	const size_t PB_N = (size_t)1 << pb_bits, PS_TILES = (size_t)PS_TILES_MIN;
	const size_t n_tiles = (nr + PT_READS - 1) / PT_READS, n_super = (n_tiles + PS_TILES - 1) / PS_TILES;
	// query records: a (w,k)-minimizer sketch keeps ~2/(w+1) of the k-mers; room for a third
	// of the bases, and the whole batch is redone with room for all of them if that overflows
	size_t q_cap = nb / 3 + 4096;
	if (e->q_cap_override > q_cap) q_cap = e->q_cap_override < nb + 4096 ? e->q_cap_override : nb + 4096;

	int rc = MNC_OK;
#define ENS(buf, bytes) do { if (!rc) rc = e->buf.ensure(bytes); } while (0)
	ENS(packed, (nb / 16 + 4) * 4);
	ENS(mz, (nb + 1) * sizeof(uint2));
	ENS(hits, (nb + 1) * sizeof(HitRec));
	ENS(ambig, (nr + 1) * 4);
	ENS(skip, (nr + 1) * 4);
	ENS(mz_cnt, (nr + 1) * 4);
	ENS(hit_cnt, (nr + 1) * 4);
	ENS(rep_len, (nr + 1) * 4);
	ENS(an_cnt, (nr + 1) * 8);
	ENS(an_off, (nr + 2) * 8);
	ENS(n_chain, (nr + 1) * 4);
	ENS(n_reg, (nr + 1) * 4);
	ENS(best_mlen, (nr + 1) * 4);
	ENS(hit_off, (nr + 2) * 8);
	ENS(hist_tm, (n_tiles + 1) * PB_N * 4);
	ENS(q_off, (n_tiles * PB_N + 2) * 8);
	ENS(qrec, (q_cap + 1) * 8);
	ENS(bhits, (q_cap + 1) * sizeof(HitRec));
	ENS(bhit_cnt, (n_super + 1) * PB_N * 4);
	ENS(scan_sums, ((n_tiles * PB_N + nr) / SC_TILE + 4) * 8);
	ENS(hist_sums, ((n_tiles / HC_TILES + 2) * PB_N) * 4);
	ENS(hist_offs, ((n_tiles / HC_TILES + 2) * PB_N + 2) * 8);
	ENS(cls_list, (size_t)(CHAIN_CLASSES.n + 1) * (nr + 1) * 4);
	if (!d_nhits) ENS(out_nhits, (nr + 1) * 4);
	if (rc) return rc;

	Batch &B = e->B;
	memset(&B, 0, sizeof(B));
	B.bases = d_bases, B.offsets = d_offsets, B.n_reads = n_reads, B.total_bases = total_bases, B.min_mapq = min_mapq;
	B.table = e->didx->table, B.filter = e->didx->filter, B.disp = e->didx->disp, B.salt = e->didx->salt, B.disp_in_lds = (e->didx->disp_bits <= PD_MAX_BITS && !(e->debug & 4)) ? 1 : 0, B.positions = e->didx->positions;
	B.region_bits = e->didx->region_bits, B.disp_bits = e->didx->disp_bits;
	B.pb_bits = pb_bits, B.pb_n = (uint32_t)PB_N, B.ps_tiles = (uint32_t)PS_TILES;
	B.contig_genome = e->didx->contig_genome, B.mid_occ = idx->mid_occ, B.n_genomes = (int)idx->genome_name.size();
	{
		// an anchor is (strand, contig, position, tandem flag, query position < 2^20): when that
		// fits 64 bits the sort works on packed words (k_sort.hip)
		int64_t longest = 1;
		for (int64_t l : idx->contig_len) longest = std::max(longest, l);
		int rb = 1, pb = 1;
		while ((1LL << rb) < (int64_t)idx->contig_name.size()) ++rb;
		while ((1LL << pb) < longest) ++pb;
		const bool fits = 1 + rb + pb + 21 <= 64;
		B.rid_bits = fits ? rb : 0, B.rpos_bits = fits ? pb : 0;
	}
	const MapParams &P = idx->par;
	B.min_cnt = P.min_cnt, B.min_sc = P.min_chain_score, B.bw = P.bw, B.max_gap = P.max_gap, B.max_skip = P.max_chain_skip;
	B.max_iter = P.max_chain_iter, B.best_n = P.best_n, B.seed = P.seed, B.max_join_long = P.max_join_long;
	B.max_join_short = P.max_join_short, B.min_join_flank_sc = P.min_join_flank_sc, B.mask_level = P.mask_level;
	B.pri_ratio = P.pri_ratio, B.min_join_flank_ratio = P.min_join_flank_ratio;
	B.gap_lut = e->gap_lut.as<int32_t>(), B.logf_lut = e->logf_lut.as<float>(), B.logf_a_lut = e->logf_a_lut.as<float>(), B.logf_n = e->logf_n;
	B.debug_route = (e->debug >> 17 & 15) | ((e->debug >> 22 & 1) << 4) | ((e->debug >> 5 & 1) << 5)    // 0x20: the literal kernel's long calls on one wave each
	              | ((e->debug >> 6 & 1) << 6) | ((e->debug >> 7 & 1) << 7)                                 // 0x40: several waves, but cells in the workspace; 0x80: round 3's four-wave form
	              | ((e->debug >> 3 & 1) << 8) | ((e->debug & 1) << 9)                                      // 0x8: always the wide form; 0x1: always the four-wave form
	              | (((unsigned)e->debug >> 31 & 1) << 10);                                                  // 0x80000000: the packed gap-filling kernels without the drifting frame
	// tuning knobs: debug bits 8-15 and 24-30.  Trying a tier pays when the chance that its band can be proven outweighs
	// the cost of running the next tier after it as well: 32 cells (1 unit) before 42 (4/3): above three in four;
	// 42 before 64 (2 units): above two in three -- a read with 10 % errors scores 1.36 +- 0.13 per base
	B.fill_pred = (e->debug >> 8 & 0xff) ? (e->debug >> 8 & 0xff) : 32;
	B.fill_pred_mid = (e->debug >> 24 & 0x7f) ? (e->debug >> 24 & 0x7f) : 34;   // (tools/sweep_pred.sh: flat around 32 / 34)
	// (neither given: by the region's anchor density -- the law it uses was fitted with the scores mappy maps with)
	B.fill_pred_auto = !(e->debug >> 8 & 0xff) && !(e->debug >> 24 & 0x7f) && P.a == 2 && P.b == 4 && P.q == 4 && P.e == 2 &&
	                   P.q2 == 24 && P.e2 == 1 && !getenv("MNC_FILL_PRED_FIXED");
	B.contract = e->contract, B.seq4 = e->didx->seq4, B.seq_off = e->didx->seq_off;
	B.sc_a = P.a, B.sc_b = P.b, B.gap_q = P.q, B.gap_e = P.e, B.gap_q2 = P.q2, B.gap_e2 = P.e2, B.sc_ambi = P.sc_ambi;
	B.zdrop = P.zdrop, B.zdrop_inv = P.zdrop_inv, B.end_bonus = P.end_bonus, B.min_dp_max = P.min_dp_max, B.min_ksw_len = P.min_ksw_len;
	B.max_sw_mat = P.max_sw_mat;
	B.packed = e->packed.as<uint32_t>(), B.ambig = e->ambig.as<uint32_t>(), B.skip = e->skip.as<uint32_t>(), B.mz = e->mz.as<uint2>(), B.hits = e->hits.as<HitRec>();
	B.mz_cnt = e->mz_cnt.as<int32_t>(), B.hit_cnt = e->hit_cnt.as<int32_t>(), B.rep_len = e->rep_len.as<int32_t>();
	B.an_cnt = e->an_cnt.as<int64_t>(), B.an_off = e->an_off.as<int64_t>(), B.n_chain = e->n_chain.as<int32_t>(), B.n_reg = e->n_reg.as<int32_t>(), B.best_mlen = e->best_mlen.as<int32_t>();
	B.n_tiles = (uint32_t)n_tiles, B.n_super = (uint32_t)n_super;
	B.hist_tm = e->hist_tm.as<uint32_t>(), B.q_off = e->q_off.as<int64_t>(), B.qrec = e->qrec.as<uint64_t>();
	B.q_cap = (int64_t)q_cap, B.bhits = e->bhits.as<HitRec>(), B.bhit_cnt = e->bhit_cnt.as<uint32_t>();
	B.overflow = reinterpret_cast<uint32_t*>(e->stats.as<int64_t>() + 8);
	B.assign = d_assign, B.best = d_best, B.nhits = d_nhits ? d_nhits : e->out_nhits.as<int32_t>(), B.counts = d_counts;
	B.stats = e->stats.as<int64_t>();

	HIP_TRY(hipMemsetAsync(B.ambig, 0, (nr + 1) * 4, st));
	HIP_TRY(hipMemsetAsync(B.skip, 0, (nr + 1) * 4, st));
	HIP_TRY(hipMemsetAsync(B.packed + nb / 16, 0, 16, st));
	HIP_TRY(hipMemsetAsync(B.overflow, 0, 8, st));

	{ StageTimer t(e, MNC_STAGE_PACK);   launch_pack(B, st); }
	{ StageTimer t(e, MNC_STAGE_SKETCH); launch_sketch(B, st); }
	{
		StageTimer t(e, MNC_STAGE_PARTITION);
		{
			const uint32_t n_chunks = (uint32_t)((n_tiles + HC_TILES - 1) / HC_TILES);
			uint32_t *sums_bm = e->hist_sums.as<uint32_t>();
			int64_t *off_bm = e->hist_offs.as<int64_t>();
			hipLaunchKernelGGL(mnc_hist_chunk_sums, dim3(n_chunks), dim3((unsigned)PB_N), 0, st, B.hist_tm, (uint32_t)n_tiles, n_chunks, sums_bm);
			exclusive_scan(ScanInPlain<uint32_t>{sums_bm}, (int64_t)n_chunks * PB_N, off_bm, e->scan_sums.as<int64_t>(), st);
			hipLaunchKernelGGL(mnc_hist_offsets, dim3(n_chunks), dim3((unsigned)PB_N), 0, st, B.hist_tm, (uint32_t)n_tiles, n_chunks, off_bm, B.q_off);
		}
		launch_partition(B, st);
	}
	{ StageTimer t(e, MNC_STAGE_PROBE);   launch_probe(B, st); }
	{ StageTimer t(e, MNC_STAGE_COLLECT); launch_collect(B, st); }
	HIP_TRY(hipGetLastError());
	int64_t total_anchors = 0, total_q = 0;
	uint32_t cls_count[MAX_CHAIN_CLASSES + 1] = {0}, overflow = 0;
	{
		StageTimer t(e, MNC_STAGE_SORT);
		exclusive_scan(ScanInPlain<int64_t>{B.an_cnt}, (int64_t)n_reads, B.an_off, e->scan_sums.as<int64_t>(), st);
		HIP_TRY(hipMemsetAsync(e->cls_count.p, 0, (MAX_CHAIN_CLASSES + 1) * 4, st));
		launch_bin_reads(B, CHAIN_CLASSES, e->cls_count.as<uint32_t>(), e->cls_list.as<uint32_t>(), st);
		// the anchor total sizes every later buffer: one small read-back per batch
		MailList ml;
		ml.n = 4;
		ml.it[0] = { B.an_off + n_reads, 0, 2 }, ml.it[1] = { B.q_off + n_tiles * PB_N, 2, 2 };
		ml.it[2] = { e->cls_count.p, 4, (uint32_t)(MAX_CHAIN_CLASSES + 1) }, ml.it[3] = { B.overflow, 4 + MAX_CHAIN_CLASSES + 1, 1 };
		hipLaunchKernelGGL(mnc_mail, dim3(1), dim3(64), 0, st, ml, e->mailbox);
		HIP_TRY(hipStreamSynchronize(st));
		memcpy(&total_anchors, e->mailbox, 8), memcpy(&total_q, e->mailbox + 2, 8);
		memcpy(cls_count, e->mailbox + 4, (MAX_CHAIN_CLASSES + 1) * 4);
		overflow = e->mailbox[4 + MAX_CHAIN_CLASSES + 1];
	}
	if (overflow || total_q > (int64_t)q_cap) {
		e->q_cap_override = (size_t)total_q + (size_t)total_q / 8 + 4096;
		*overflowed = 1;
		return MNC_OK;
	}
	if (total_anchors < 0 || total_anchors >= (1LL << 31) * 16) { set_error("anchor count %lld out of range", (long long)total_anchors); return MNC_ERR_UNSUPPORTED; }
	e->last_total_anchors = total_anchors;
	// ns chain slots (a chain has three anchors or more); nsr region slots: slot_pad more per read (device.h: reg_slot)
	const size_t na = (size_t)total_anchors + 4, ns = (size_t)total_anchors / 3 + 4, nsr = ns + nr * (size_t)e->slot_pad;
	ENS(a, na * sizeof(Anchor));
	ENS(f, na * 4); ENS(p, na * 4); ENS(v, na * 4); ENS(t, na * 4);
	ENS(u, na * 8);
	ENS(chains_tmp, ns * sizeof(ChainRec));
	ENS(regs, nsr * sizeof(mnc_reg_t)); ENS(regx, nsr * 32);
	ENS(k64a, nsr * 8); ENS(k64b, nsr * 8); ENS(tmp_i32, nsr * 16); ENS(gated, nsr * sizeof(mnc_hit_t));
	if (rc) return rc;
#undef ENS
	B.an_cap = (int64_t)na;
	B.slot_pad = e->slot_pad;
	B.a = e->a.as<Anchor>(), B.chains_tmp = e->chains_tmp.as<ChainRec>();
	B.f = e->f.as<int32_t>(), B.p = e->p.as<int32_t>(), B.v = e->v.as<int32_t>(), B.t = e->t.as<int32_t>(), B.u = e->u.as<uint64_t>();
	B.regs = e->regs.as<mnc_reg_t>(), B.tmp_i32 = e->tmp_i32.as<int32_t>();

	// The launches of one stage differ only in their LDS tile and touch different reads: they go
	// to side streams between a fork and a join on the engine stream, so that one class fills
	// the CUs another class's last waves leave idle.
	auto fork = [&]() -> int {
		HIP_TRY(hipEventRecord(e->ev_fork, st));
		for (int k = 0; k < mnc_engine::N_SIDE; ++k) HIP_TRY(hipStreamWaitEvent(e->side[k], e->ev_fork, 0));
		return MNC_OK;
	};
	auto join = [&]() -> int {
		for (int k = 0; k < mnc_engine::N_SIDE; ++k) {
			HIP_TRY(hipEventRecord(e->ev_join[k], e->side[k]));
			HIP_TRY(hipStreamWaitEvent(st, e->ev_join[k], 0));
		}
		return MNC_OK;
	};
	{
		StageTimer t(e, MNC_STAGE_SORT2);
		const uint32_t *lists = e->cls_list.as<uint32_t>();
		if (int rc = fork()) return rc;
		// the classes up to SORT_MERGE anchors share one launch (their tiles are small either way:
		// no launch tails between them); the larger ones keep a launch each, on the side streams
		constexpr int SORT_MERGE = 512;
		int first_big = 0;
		while (first_big < CHAIN_CLASSES.n && CHAIN_CLASSES.nm[first_big] <= SORT_MERGE) ++first_big;
		int k = 0;
		if (first_big > 0) {
			ClassSpans sp;
			sp.n = first_big, sp.stride = (uint32_t)n_reads, sp.start[0] = 0;
			for (int c = 0; c < first_big; ++c) sp.start[c + 1] = sp.start[c] + cls_count[c];
			if (sp.start[sp.n]) launch_expand_sort(B, lists, sp, CHAIN_CLASSES.nm[first_big - 1], e->side[k++ % mnc_engine::N_SIDE]);
		}
		for (int c = first_big; c <= CHAIN_CLASSES.n; ++c) {    // the last list: reads too large for LDS (NM = 0)
			if (cls_count[c] == 0) continue;
			ClassSpans sp;
			sp.n = 1, sp.stride = 0, sp.start[0] = 0, sp.start[1] = cls_count[c];
			launch_expand_sort(B, lists + (size_t)c * n_reads, sp, c < CHAIN_CLASSES.n ? CHAIN_CLASSES.nm[c] : 0,
			                   e->side[k++ % mnc_engine::N_SIDE]);
		}
		if (int rc = join()) return rc;
	}
	{
		StageTimer t(e, MNC_STAGE_CHAIN);           // DP: every read, whatever its size
		const uint32_t *lists = e->cls_list.as<uint32_t>();
		ClassSpans spans;
		spans.n = CHAIN_CLASSES.n + 1, spans.stride = (uint32_t)n_reads, spans.start[0] = 0;
		for (int c = 0; c <= CHAIN_CLASSES.n; ++c) spans.start[c + 1] = spans.start[c] + cls_count[c];
		launch_chain_dp_ring(B, lists, spans, (e->debug & 2) != 0, st);
	}
	{
		StageTimer t(e, MNC_STAGE_BACKTRACK);       // LDS form per size class; sequential form beyond
		const uint32_t *lists = e->cls_list.as<uint32_t>();
		if (int rc = fork()) return rc;
		int k = 0;
		constexpr int merge_to = 384;                  // smaller classes share a launch (their LDS tiles differ little)
		int first_big = 0;
		while (first_big < CHAIN_CLASSES.n && CHAIN_CLASSES.nm[first_big] <= merge_to) ++first_big;
		if (first_big > 0) {
			ClassSpans sp;
			sp.n = first_big, sp.stride = (uint32_t)n_reads, sp.start[0] = 0;
			for (int c = 0; c < first_big; ++c) sp.start[c + 1] = sp.start[c] + cls_count[c];
			if (sp.start[sp.n]) launch_chain_tail(B, lists, sp, CHAIN_CLASSES.nm[first_big - 1], e->side[k++ % mnc_engine::N_SIDE]);
		}
		for (int c = first_big; c < CHAIN_CLASSES.n; ++c) {
			if (cls_count[c] == 0) continue;
			ClassSpans sp;
			sp.n = 1, sp.stride = 0, sp.start[0] = 0, sp.start[1] = cls_count[c];
			launch_chain_tail(B, lists + (size_t)c * n_reads, sp, CHAIN_CLASSES.nm[c], e->side[k++ % mnc_engine::N_SIDE]);
		}
		launch_backtrack(B, lists + (size_t)CHAIN_CLASSES.n * n_reads, cls_count[CHAIN_CLASSES.n], e->side[k % mnc_engine::N_SIDE]);
		if (int rc = join()) return rc;
	}
	if (e->contract == MNC_CONTRACT_DP) {
		// ---- base-level alignment stage: buffers
		// room for the segments: ~23 per 5 kb read with 10 % errors is one per 17 anchors; one per 6, and
		// the batch is redone with the worst case if that overflows (`seg_cap_override`)
		size_t seg_cap = na / 6 + 4096, cig_cap = nb / 2 + (1u << 20);
		if (e->seg_cap_override > seg_cap) seg_cap = e->seg_cap_override;
		if (e->cig_cap_override > cig_cap) cig_cap = e->cig_cap_override;
		int rc2 = MNC_OK;
#define ENS2(buf, bytes) do { if (!rc2) rc2 = e->buf.ensure(bytes); } while (0)
#define ENSW(buf, bytes) do { if (!rc2) rc2 = e->ws->buf.ensure(bytes); } while (0)
		// the device's alignment scratch is this engine's from here to the end of the stage
		std::unique_lock<std::mutex> ws_hold(e->ws->mu);
		ENS2(ca, na * sizeof(Anchor)); ENS2(ca_cnt, (nr + 1) * 4); ENS2(chain_dst, ns * 4); ENS2(regdp, nsr * sizeof(RegDP));
		ENS2(inv_ws, dp_inv_ws_words(B.max_gap) * 4 * DP_WG_INV);
		// the stitch kernel's waves reserve the region pool a chunk at a time (k_align.hip): what its two launches
		// can leave unused is room on top of the CIGARs' own
		const size_t cig_reg_cap = cig_cap + dp_stitch_pool_slack(DP_WG_STITCH);
		ENS2(segs, seg_cap * sizeof(Seg)); ENS2(cig_seg, cig_cap * 4); ENS2(cig_reg, cig_reg_cap * 4);
		ENS2(work_a, nsr * 4); ENS2(work_b, nsr * 4); ENS2(big_list, seg_cap * 4); ENS2(reg_cnt, (nr + 1) * 4); ENS2(regs2, nsr * sizeof(mnc_reg_t));
		const size_t ws_small = dp_align_ws_bytes(DP_STATE_SMALL, DP_P_SMALL, DP_CIG_SMALL), ws_big = dp_align_ws_bytes(DP_STATE_BIG, DP_P_BIG, DP_CIG_BIG), ws_huge = dp_align_ws_bytes(DP_STATE_HUGE, DP_P_HUGE, DP_CIG_HUGE);
		ENSW(dp_ws, ws_small * DP_WG_SMALL * 2); ENSW(dp_ws_big, ws_big * DP_WG_BIG); ENSW(dp_ws_bigfb, ws_big * DP_WG_BIGFB); ENSW(dp_ws_huge, ws_huge * DP_WG_HUGE); ENS2(huge_list, seg_cap * 4); ENSW(dp_ws_mid, ws_small * DP_WG_MID); ENS2(mid_list, seg_cap * 4); ENS2(lfill, seg_cap * 4); ENSW(lfill_p, dp_lfill_p_slot() * DP_WG_LFILL); ENS2(lext, seg_cap * 4); ENS2(bigfb, seg_cap * 4); ENSW(lext_p, dp_lext_p_slot() * DP_WG_LEXT);
		ENS2(plan_long, (nr * 4 + 1024) * 4);
		ENS2(fill1, seg_cap * 4); ENS2(fill2, seg_cap * 4); ENS2(fill3, seg_cap * 4); ENS2(fill_mid, seg_cap * 4); ENS2(fill_fb, seg_cap * 4); ENSW(fill_p, dp_fillp_slot() * DP_WG_FILL); ENSW(fill_cig, dp_fillp_cig_slot() * DP_WG_FILL); ENS2(extp, seg_cap * 4 * 8); ENSW(extp_p, dp_extp_slot() * DP_WG_EXT); ENSW(extp_cig, dp_extp_cig_slot() * DP_WG_EXT); ENS2(ext1, seg_cap * 4); ENS2(ext2, seg_cap * 4); ENS2(ext3, seg_cap * 4); ENS2(ext4, seg_cap * 4); ENS2(gen_list, seg_cap * 4); ENSW(ext_p, dp_fill_p_slot() * DP_WG_EXT);
#undef ENS2
		if (rc2) return rc2;
		B.ca = e->ca.as<Anchor>(), B.ca_cnt = e->ca_cnt.as<int32_t>(), B.chain_dst = e->chain_dst.as<int32_t>(), B.regdp = e->regdp.as<RegDP>();
		B.segs = e->segs.as<Seg>(), B.seg_cap = (int64_t)seg_cap, B.cig_seg = e->cig_seg.as<uint32_t>(), B.cig_reg = e->cig_reg.as<uint32_t>();
		B.cig_seg_cap = (int64_t)cig_cap, B.cig_reg_cap = (int64_t)cig_reg_cap, B.dp_ctr = e->dp_ctr.as<unsigned long long>();
		B.big_list = e->big_list.as<int32_t>(), B.huge_list = e->huge_list.as<int32_t>(), B.reg_cnt = e->reg_cnt.as<int32_t>();
		B.plan_long_list = e->plan_long.as<int32_t>(), B.plan_long_cap = (long long)(nr * 4 + 1024);
		B.fill_list1 = e->fill1.as<int32_t>(), B.fill_list2 = e->fill2.as<int32_t>(), B.fill_list3 = e->fill3.as<int32_t>(), B.fill_list_mid = e->fill_mid.as<int32_t>(), B.fill_fb = e->fill_fb.as<int32_t>();
		B.ext_list1 = e->ext1.as<int32_t>(), B.ext_list2 = e->ext2.as<int32_t>(), B.ext_list3 = e->ext3.as<int32_t>(), B.ext_list4 = e->ext4.as<int32_t>(), B.gen_list = e->gen_list.as<int32_t>(), B.extp_list = e->extp.as<int32_t>(), B.mid_list = e->mid_list.as<int32_t>(), B.lfill_list = e->lfill.as<int32_t>(), B.lext_list = e->lext.as<int32_t>(), B.bigfb_list = e->bigfb.as<int32_t>();
		B.lds0_state = DP_LDS0_STATE, B.lds0_p = DP_LDS0_P, B.lds0_cig = DP_LDS0_CIG;
		int32_t *lists[2] = { e->work_a.as<int32_t>(), e->work_b.as<int32_t>() };
		B.next_list = lists[0];                          // the regions kernel files every kept region here
		HIP_TRY(hipMemsetAsync(e->dp_ctr.p, 0, 64 * 8, st));
		{ StageTimer t(e, MNC_STAGE_REGIONS);   launch_regions(B, e->regx.p, e->k64a.as<uint64_t>(), e->k64b.as<uint64_t>(), e->gated.as<mnc_hit_t>(), st); }
		{
			StageTimer t(e, MNC_STAGE_DP_PLAN);
			launch_dp_gather(B, st);
			// reads of the size classes above 1024 anchors (and beyond the classes): a wave each
			int c0 = 0;
			while (c0 < CHAIN_CLASSES.n && CHAIN_CLASSES.nm[c0] <= 1024) ++c0;
			ClassSpans sp;
			sp.n = CHAIN_CLASSES.n + 1 - c0, sp.stride = (uint32_t)n_reads, sp.start[0] = 0;
			for (int c = c0; c <= CHAIN_CLASSES.n; ++c) sp.start[c - c0 + 1] = sp.start[c - c0] + cls_count[c];
			launch_dp_gather_long(B, e->cls_list.as<uint32_t>() + (size_t)c0 * n_reads, sp, CHAIN_CLASSES.nm[CHAIN_CLASSES.n - 1], st);
		}
		unsigned max_work = (unsigned)nsr;
		// reads long enough for 512 chained anchors (a 5 kb read has ~200 a chain): their regions are planned a wave each;
		// debug bit 0x800000: everything on the lane form (tests)
		// 0x10: every region on the wave form (tests)
		// A micro-batch: every region on the wave form -- a lane's plan is 0.5 ms of dependent loads however few regions there
		// are, a wave's 45 us, and a few hundred regions are one round of waves (p50 of a 400-read batch 3.9 -> 3.5 ms)
		const int long_reads = (e->debug & 0x10) ? 2 : (e->debug & 0x800000) ? 0 : n_reads <= 4096 ? 2
		                     : (e->cur_max_read_len <= 0 || e->cur_max_read_len > 6144) ? 1 : 0;
		for (int round = 0;; ++round) {
			const int32_t *work = lists[round & 1];
			int32_t *next = lists[(round + 1) & 1];
			B.next_list = next;
			launch_dp_round(B, round == 0, st);
			if (round == 0) {
				{ StageTimer t(e, MNC_STAGE_DP_PLAN);   launch_dp_plan(B, work, max_work, long_reads, (int)DP_STATE_SMALL, DP_P_SMALL, (int)DP_CIG_SMALL, DP_STATE_BIG, DP_P_BIG, DP_CIG_BIG, DP_STATE_HUGE, DP_P_HUGE, DP_CIG_HUGE, st); }
				{
					StageTimer t(e, MNC_STAGE_DP_FILL);               // the four streams, fork to join
					if (int rcf = fork()) return rcf;
					if (e->debug & 0x10000) align_round(B, e, e->side[0], e->side[0], e->side[0], e->side[0], e->side[0]);   // profiling: one kernel at a time
					else align_round(B, e, e->side[0], e->side[1], st, e->side[2], e->side[3]);
					if (int rcj = join()) return rcj;
				}
				{
					StageTimer t(e, MNC_STAGE_DP_ALIGN);
					align_rest(B, e, st);
				}
				{ StageTimer t(e, MNC_STAGE_DP_STITCH); launch_dp_stitch(B, work, next, (e->debug & 0x200000) ? 0 : e->cur_max_read_len > 0 ? e->cur_max_read_len : 8192, DP_WG_STITCH, st); }
			} else {
				launch_dp_plan(B, work, max_work, long_reads, (int)DP_STATE_SMALL, DP_P_SMALL, (int)DP_CIG_SMALL, DP_STATE_BIG, DP_P_BIG, DP_CIG_BIG, DP_STATE_HUGE, DP_P_HUGE, DP_CIG_HUGE, st);
				if (int rcf = fork()) return rcf;
				if (e->debug & 0x10000) align_round(B, e, e->side[0], e->side[0], e->side[0], e->side[0], e->side[0]);
				else align_round(B, e, e->side[0], e->side[1], st, e->side[2], e->side[3]);
				if (int rcj = join()) return rcj;
				align_rest(B, e, st);
				launch_dp_stitch(B, work, next, (e->debug & 0x200000) ? 0 : e->cur_max_read_len > 0 ? e->cur_max_read_len : 8192, DP_WG_STITCH, st);
				// the tails of Z-drop splits are aligned now: the inversion between a head and its tail (mm_align1_inv), whose
				// region -- one extension -- is aligned in the next round
				launch_dp_inv(B, work, next, e->inv_ws.as<int32_t>(), (int)std::min<unsigned>(DP_WG_INV, max_work), st);
			}
			launch_dp_round_end(B, st);
			// Z-drop splits make new regions for the next round (rare); one small read-back per round
			unsigned long long ctr[6] = {0};
			{
				MailList ml;
				ml.n = 3;
				ml.it[0] = { e->dp_ctr.p, 0, 12 };
				ml.it[1] = { e->dp_ctr.as<unsigned long long>() + 6, 12, 2 }, ml.it[2] = { e->dp_ctr.as<unsigned long long>() + 58, 14, 2 };   // calls of the passes 1 and 5
				hipLaunchKernelGGL(mnc_mail, dim3(1), dim3(64), 0, st, ml, e->mailbox);
				HIP_TRY(hipStreamSynchronize(st));
				memcpy(ctr, e->mailbox, sizeof(ctr));
				if (round == 0 && n_reads >= 4096) {
					unsigned long long big = 0, huge = 0;
					memcpy(&big, e->mailbox + 12, 8), memcpy(&huge, e->mailbox + 14, 8);
					e->prev_wide_calls = (long long)(big + huge);
				}
			}
			if (ctr[4] != 0) {
				if (ctr[4] >= 9) { set_error("a gap between two seeds is too large for the alignment workspace"); return MNC_ERR_UNSUPPORTED; }
				if (ctr[4] == 5) {                            // a read ran out of region slots (split tails, inversion regions)
					e->slot_pad = e->slot_pad * 4;
					*overflowed = 2;
					return MNC_OK;
				}
				// segments: the worst case; CIGAR words: four times the room, at most one word per base of every
				// region's query and target span (regions of a read may overlap: bounded by the retry count instead)
				e->seg_cap_override = na + 2 * ns + 1024;
				e->cig_cap_override = cig_cap * 4;
				*overflowed = 2;
				return MNC_OK;
			}
			if (ctr[5] == 0) break;
			if (round > 256) { set_error("base-level alignment does not settle"); return MNC_ERR_HIP; }
			max_work = (unsigned)ctr[5];
		}
		{ StageTimer t(e, MNC_STAGE_DP_POST); launch_regions_post(B, e->regs2.as<mnc_reg_t>(), e->regx.p, e->k64a.as<uint64_t>(), e->tmp_i32.as<int32_t>(), e->gated.as<mnc_hit_t>(), st); }
	} else {
		StageTimer t(e, MNC_STAGE_REGIONS);
		launch_regions(B, e->regx.p, e->k64a.as<uint64_t>(), e->k64b.as<uint64_t>(), e->gated.as<mnc_hit_t>(), st);
	}
	HIP_TRY(hipGetLastError());
	e->have_batch = true;
	return MNC_OK;
}

extern "C" int mnc_classify_device(mnc_engine *e, const uint8_t *d_bases, const int64_t *d_offsets,
                                   uint32_t n_reads, int64_t total_bases, int max_read_len, int min_mapq,
                                   int32_t *d_assign, mnc_hit_t *d_best, int32_t *d_nhits, int64_t *d_counts)
{
	if (!e || !d_offsets || !d_assign || (total_bases > 0 && !d_bases) || total_bases < 0) return MNC_ERR_ARG;
	if (((uintptr_t)d_bases & 15u) != 0) { set_error("d_bases must be 16-byte aligned"); return MNC_ERR_ARG; }
	if (max_read_len < 0 || max_read_len >= (1 << 20)) { set_error("reads of 2^20 bases or more are not supported"); return MNC_ERR_UNSUPPORTED; }
	if (n_reads > (1u << 20)) { set_error("at most 2^20 reads per batch (query records hold a 20-bit read ordinal)"); return MNC_ERR_UNSUPPORTED; }
	if (total_bases >= (1LL << 40)) return MNC_ERR_UNSUPPORTED;
	HIP_TRY(hipSetDevice(e->device));
	e->have_batch = false, e->last_total_hits = -1;
	if (n_reads == 0) {
		memset(&e->B, 0, sizeof(e->B));
		e->have_batch = true, e->last_total_anchors = 0;
		return MNC_OK;
	}
	int overflowed = 0, q_redone = 0, dp_redone = 0;
	e->cur_max_read_len = max_read_len, e->last_redos = 0;
	int rc = MNC_OK;
	// A batch that outgrows a budget is redone with more room: the query records once (the second try has exact
	// room), the alignment stage's pools four times as large per try (4^6 times the first budget is beyond any CIGAR
	// the batch's bases can make) -- and either may happen in the same batch.
	for (;;) {
		rc = classify_once(e, d_bases, d_offsets, n_reads, total_bases, min_mapq, d_assign, d_best, d_nhits, d_counts, &overflowed);
		if (rc || !overflowed) break;
		++e->last_redos;
		if (overflowed == 1 && ++q_redone > 1) { set_error("query record budget exceeded twice"); rc = MNC_ERR_NOMEM; break; }
		if (overflowed == 2 && ++dp_redone > 6) { set_error("segment / CIGAR pools of the alignment stage exceeded after %d enlargements", dp_redone - 1); rc = MNC_ERR_NOMEM; break; }
	}
	return rc;
}

// ---------------------------------------------------------------- the next batch's bases, ahead of its call
// Starts the copy of a batch's bases and offsets to a spare device buffer on a stream of its own and returns; the
// mnc_classify_batch call that follows with the SAME pointers and read count finds them there and skips its copy --
// so one batch's transfer runs behind the previous batch's kernels.  The caller leaves the host arrays alone in
// between (page-locked ones, e.g. a FASTQ reader's, make the copy asynchronous).  *started = 0: a prefetched batch
// is still waiting for its call (one spare buffer); nothing was done, the batch will be copied by its own call.
// May be called from another host thread than the one classifying.
extern "C" int mnc_engine_prefetch(mnc_engine *e, const uint8_t *bases, const int64_t *offsets, uint32_t n_reads, int *started)
{
	if (!e || !offsets || !started) return MNC_ERR_ARG;
	*started = 0;
	const int64_t total = offsets[n_reads] - offsets[0];
	if (offsets[0] != 0 || total < 0 || (total > 0 && !bases)) { set_error("offsets must start at 0 and be non-decreasing"); return MNC_ERR_ARG; }
	std::lock_guard<std::mutex> lk(e->pf_mu);
	if (e->pf_valid) return MNC_OK;
	HIP_TRY(hipSetDevice(e->device));
	// the spare buffers may still be the source of the running batch's kernels only after a swap -- and a swap
	// happens when a prefetched batch is taken, which clears pf_valid: they are free here
	int rc = e->pf_bases_buf.ensure((size_t)total + 32);
	if (!rc) rc = e->pf_offsets_buf.ensure(((size_t)n_reads + 1) * 8);
	if (rc) return rc;
	// In pieces: the copy engines take their requests in turn, and the running batch's small read-backs (anchor totals,
	// round counters: a few bytes each, the host waits for them) would otherwise queue behind half a gigabyte --
	// measured: a batch that starts together with a 500 MB copy takes 48.4 ms instead of 41.0 (profiles/r03i_e2e_probe.txt)
	constexpr size_t PF_PIECE = 4u << 20;
	for (size_t o = 0; o < (size_t)total; o += PF_PIECE) {
		const size_t len = std::min(PF_PIECE, (size_t)total - o);
		HIP_TRY(hipMemcpyAsync((uint8_t*)e->pf_bases_buf.p + o, bases + o, len, hipMemcpyHostToDevice, e->copy_stream));
	}
	HIP_TRY(hipMemcpyAsync(e->pf_offsets_buf.p, offsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, e->copy_stream));
	HIP_TRY(hipEventRecord(e->ev_prefetch, e->copy_stream));
	e->pf_valid = true, e->pf_bases = bases, e->pf_offsets = offsets, e->pf_n = n_reads, e->pf_total = total;
	*started = 1;
	return MNC_OK;
}

// Forget the batch announced last, if any: waits for its copy (the host arrays are the caller's again afterwards)
// and frees the spare buffer for the next announcement.  For a caller that gives up on a batch it has announced --
// an error between announcement and call, an engine handed to another sample.  Matching is by the host pointers, so
// an abandoned announcement whose page-locked arrays are recycled for another batch must be cancelled, not left.
extern "C" int mnc_engine_prefetch_cancel(mnc_engine *e)
{
	if (!e) return MNC_ERR_ARG;
	std::lock_guard<std::mutex> lk(e->pf_mu);
	// whether or not an announcement stands: mnc_engine_prefetch sets pf_valid only after its last copy is queued, so a
	// call that failed half-way has left copies from the caller's arrays in flight with pf_valid still false
	e->pf_valid = false, e->pf_bases = nullptr, e->pf_offsets = nullptr;
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipStreamSynchronize(e->copy_stream));
	return MNC_OK;
}

// ---------------------------------------------------------------- one batch, host buffers
extern "C" int mnc_classify_batch(mnc_engine *e, const uint8_t *bases, const int64_t *offsets, uint32_t n_reads,
                                  int min_mapq, int32_t *out_assign, mnc_hit_t *out_best, int32_t *out_nhits)
{
	if (!e || !offsets || !out_assign) return MNC_ERR_ARG;
	const int64_t total = offsets[n_reads] - offsets[0];
	if (offsets[0] != 0 || total < 0 || (total > 0 && !bases)) { set_error("offsets must start at 0 and be non-decreasing"); return MNC_ERR_ARG; }
	int max_len = 0;
	size_t n_over = 0;
	for (uint32_t r = 0; r < n_reads; ++r) {
		const int64_t l = offsets[r + 1] - offsets[r];
		if (l < 0) { set_error("offsets must be non-decreasing"); return MNC_ERR_ARG; }
		if (l >= (1 << 20)) ++n_over;
		else if (l > max_len) max_len = (int)l;
	}
	if (n_over) {
		// Reads of 2^20 bases or more are outside what the kernels hold (a query position is a 20-bit field of the
		// probe's records and of the packed anchors).  index.map() takes any length (aligner.py:193, 215), so such a read
		// must not cost the batch: it is taken out (an empty read in its place), the others are classified as ever, and
		// it comes back as MNC_SKIPPED -- the one limit of this entry point that shows in its results.
		std::vector<int64_t> foff((size_t)n_reads + 1);
		std::vector<uint8_t> fb;
		try { fb.reserve((size_t)total); } catch (const std::bad_alloc &) { return MNC_ERR_NOMEM; }
		foff[0] = 0;
		for (uint32_t r = 0; r < n_reads; ++r) {
			const int64_t l = offsets[r + 1] - offsets[r];
			if (l < (1 << 20)) fb.insert(fb.end(), bases + offsets[r], bases + offsets[r + 1]);
			foff[r + 1] = (int64_t)fb.size();
		}
		const int rc = mnc_classify_batch(e, fb.empty() ? bases : fb.data(), foff.data(), n_reads, min_mapq, out_assign, out_best, out_nhits);
		if (rc) return rc;
		for (uint32_t r = 0; r < n_reads; ++r)
			if (offsets[r + 1] - offsets[r] >= (1 << 20)) {
				out_assign[r] = MNC_SKIPPED;
				if (out_best) memset(&out_best[r], 0, sizeof(mnc_hit_t));
				if (out_nhits) out_nhits[r] = 0;
			}
		return MNC_OK;
	}
	HIP_TRY(hipSetDevice(e->device));
	const size_t nr = n_reads;
	int rc = e->out_assign.ensure((nr + 1) * 4);
	if (!rc) rc = e->out_best.ensure((nr + 1) * sizeof(mnc_hit_t));
	if (!rc) rc = e->out_nhits.ensure((nr + 1) * 4);
	if (rc) return rc;
	hipStream_t st = e->stream;
	bool prefetched = false;
	{
		std::lock_guard<std::mutex> lk(e->pf_mu);
		if (e->pf_valid) {
			const bool mine = e->pf_bases == bases && e->pf_offsets == offsets && e->pf_n == n_reads && e->pf_total == total;
			e->pf_valid = false, e->pf_bases = nullptr, e->pf_offsets = nullptr;     // whatever happens below, the announcement is spent
			if (mine) {
				// this batch is on the device already (or on its way): its buffers become the input, the old input the spare
				HIP_TRY(hipStreamWaitEvent(st, e->ev_prefetch, 0));
				std::swap(e->in_bases, e->pf_bases_buf), std::swap(e->in_offsets, e->pf_offsets_buf);
				prefetched = true;
			} else HIP_TRY(hipStreamSynchronize(e->copy_stream));      // another batch was announced: let its copy finish, then forget it
		}
	}
	if (!prefetched) {
		rc = e->in_bases.ensure((size_t)total + 32);
		if (!rc) rc = e->in_offsets.ensure((nr + 1) * 8);
		if (rc) return rc;
		if (total > 0) HIP_TRY(hipMemcpyAsync(e->in_bases.p, bases, (size_t)total, hipMemcpyHostToDevice, st));
		HIP_TRY(hipMemcpyAsync(e->in_offsets.p, offsets, (nr + 1) * 8, hipMemcpyHostToDevice, st));
	}
	rc = mnc_classify_device(e, e->in_bases.as<uint8_t>(), e->in_offsets.as<int64_t>(), n_reads, total, max_len, min_mapq,
	                         e->out_assign.as<int32_t>(), e->out_best.as<mnc_hit_t>(), e->out_nhits.as<int32_t>(), nullptr);
	if (rc) return rc;
	if (nr) {
		HIP_TRY(hipMemcpyAsync(out_assign, e->out_assign.p, nr * 4, hipMemcpyDeviceToHost, st));
		if (out_best) HIP_TRY(hipMemcpyAsync(out_best, e->out_best.p, nr * sizeof(mnc_hit_t), hipMemcpyDeviceToHost, st));
		if (out_nhits) HIP_TRY(hipMemcpyAsync(out_nhits, e->out_nhits.p, nr * 4, hipMemcpyDeviceToHost, st));
	}
	rc = mnc_engine_sync(e);
	if (rc) return rc;
	HIP_TRY(hipGetLastError());
	e->h_offsets.assign(offsets, offsets + nr + 1);
	return MNC_OK;
}

// ---------------------------------------------------------------- gated hits of the last batch
extern "C" int mnc_engine_fetch_hits(mnc_engine *e, int64_t *hit_offsets, mnc_hit_t *hits, int64_t cap, int64_t *n_hits)
{
	if (!e || !n_hits) return MNC_ERR_ARG;
	if (!e->have_batch) { set_error("no batch has been classified"); return MNC_ERR_ARG; }
	HIP_TRY(hipSetDevice(e->device));
	hipStream_t st = e->stream;
	Batch &B = e->B;
	const size_t nr = B.n_reads;
	if (nr == 0) { *n_hits = 0; if (hit_offsets) hit_offsets[0] = 0; return MNC_OK; }
	int64_t *d_off = e->hit_off.as<int64_t>();
	if (e->last_total_hits < 0) {
		StageTimer t(e, MNC_STAGE_GATHER);
		exclusive_scan(ScanInPlain<int32_t>{B.nhits}, (int64_t)nr, d_off, e->scan_sums.as<int64_t>(), st);
		int64_t total = 0;
		HIP_TRY(hipMemcpyAsync(&total, d_off + nr, 8, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		int rc = e->hits_csr.ensure(((size_t)total + 1) * sizeof(mnc_hit_t));
		if (rc) return rc;
		launch_gather_hits(B, e->gated.as<mnc_hit_t>(), d_off, e->hits_csr.as<mnc_hit_t>(), st);
		e->last_total_hits = total;
	}
	*n_hits = e->last_total_hits;
	if (cap < e->last_total_hits || !hits || !hit_offsets) { (void)mnc_engine_sync(e); return MNC_ERR_RANGE; }
	HIP_TRY(hipMemcpyAsync(hit_offsets, d_off, (nr + 1) * 8, hipMemcpyDeviceToHost, st));
	if (e->last_total_hits > 0) HIP_TRY(hipMemcpyAsync(hits, e->hits_csr.p, (size_t)e->last_total_hits * sizeof(mnc_hit_t), hipMemcpyDeviceToHost, st));
	return mnc_engine_sync(e);
}

// ---------------------------------------------------------------- counters / dumps
extern "C" int mnc_engine_get_counters(mnc_engine *e, int64_t *c, int n)
{
	if (!e || !c || n < 8) return MNC_ERR_ARG;
	if (!e->have_batch) { set_error("no batch has been classified"); return MNC_ERR_ARG; }
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	const Batch &B = e->B;
	const size_t nr = B.n_reads;
	std::vector<int32_t> a(nr), b(nr), d(nr), g(nr), h(nr);
	std::vector<uint32_t> amb(nr);
	if (nr) {
		HIP_TRY(hipMemcpy(a.data(), B.mz_cnt, nr * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(b.data(), B.hit_cnt, nr * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(d.data(), B.n_chain, nr * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(g.data(), B.n_reg, nr * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(h.data(), B.nhits, nr * 4, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(amb.data(), B.ambig, nr * 4, hipMemcpyDeviceToHost));
	}
	for (int i = 0; i < n; ++i) c[i] = 0;
	for (size_t r = 0; r < nr; ++r) c[0] += a[r], c[1] += b[r], c[3] += d[r], c[4] += g[r], c[5] += h[r], c[6] += amb[r] ? 1 : 0;
	c[2] = e->last_total_anchors;
	c[7] = e->last_redos;                                   // passes over the batch beyond the first (a budget was outgrown: the batch ran again with more room)
	if (n >= 12 && B.contract == MNC_CONTRACT_DP && e->dp_ctr.p) {
		unsigned long long d[64];
		HIP_TRY(hipMemcpy(d, e->dp_ctr.p, sizeof(d), hipMemcpyDeviceToHost));
		c[8] = (int64_t)d[0], c[9] = (int64_t)d[10], c[10] = (int64_t)d[11], c[11] = (int64_t)d[12];
		if (n >= 16) c[12] = (int64_t)d[48], c[13] = (int64_t)d[49], c[14] = (int64_t)d[50], c[15] = (int64_t)d[51];
		if (n >= 24) c[16] = (int64_t)d[6], c[17] = (int64_t)d[28], c[18] = (int64_t)d[31], c[19] = (int64_t)d[56], c[20] = (int64_t)d[62], c[21] = (int64_t)d[22];
		if (n >= 24) c[22] = (int64_t)d[30], c[23] = (int64_t)d[52];       // the 42-cell tier: segments, anti-diagonals


	}
	return MNC_OK;
}

namespace {
// copy per-read segments [base[r], base[r]+cnt[r]) of a device array into one host run
template <class T>
int dump_segments(const T *d_arr, const std::vector<int64_t> &base, const std::vector<int64_t> &cnt,
                  void *dst, int64_t cap_bytes, int64_t *n_bytes)
{
	int64_t total = 0;
	for (size_t r = 0; r < cnt.size(); ++r) total += cnt[r];
	*n_bytes = total * (int64_t)sizeof(T);
	if (!dst || cap_bytes < *n_bytes) return MNC_ERR_RANGE;
	T *o = reinterpret_cast<T*>(dst);
	for (size_t r = 0; r < cnt.size(); ++r) {
		if (cnt[r] > 0) HIP_TRY(hipMemcpy(o, d_arr + base[r], (size_t)cnt[r] * sizeof(T), hipMemcpyDeviceToHost));
		o += cnt[r];
	}
	return MNC_OK;
}
template <class T>
int dump_offsets(const std::vector<T> &cnt, void *dst, int64_t cap_bytes, int64_t *n_bytes)
{
	*n_bytes = (int64_t)(cnt.size() + 1) * 8;
	if (!dst || cap_bytes < *n_bytes) return MNC_ERR_RANGE;
	int64_t *o = reinterpret_cast<int64_t*>(dst), s = 0;
	for (size_t r = 0; r < cnt.size(); ++r) { o[r] = s; s += (int64_t)cnt[r]; }
	o[cnt.size()] = s;
	return MNC_OK;
}
}

extern "C" int mnc_engine_dump(mnc_engine *e, int what, void *dst, int64_t cap_bytes, int64_t *n_bytes)
{
	if (!e || !n_bytes) return MNC_ERR_ARG;
	if (!e->have_batch) { set_error("no batch has been classified"); return MNC_ERR_ARG; }
	HIP_TRY(hipSetDevice(e->device));
	HIP_TRY(hipStreamSynchronize(e->stream));
	const Batch &B = e->B;
	const size_t nr = B.n_reads;
	std::vector<int64_t> off(nr + 1), an_off(nr + 1);
	std::vector<int32_t> c32(nr);
	if (nr) {
		HIP_TRY(hipMemcpy(off.data(), B.offsets, (nr + 1) * 8, hipMemcpyDeviceToHost));
		HIP_TRY(hipMemcpy(an_off.data(), B.an_off, (nr + 1) * 8, hipMemcpyDeviceToHost));
	}
	std::vector<int64_t> base(nr), cnt(nr);
	switch (what) {
	case MNC_DUMP_MINIMIZERS:
	case MNC_DUMP_MZ_OFFSETS: {
		if (nr) HIP_TRY(hipMemcpy(c32.data(), B.mz_cnt, nr * 4, hipMemcpyDeviceToHost));
		if (what == MNC_DUMP_MZ_OFFSETS) return dump_offsets(c32, dst, cap_bytes, n_bytes);
		for (size_t r = 0; r < nr; ++r) base[r] = off[r], cnt[r] = c32[r];
		return dump_segments(B.mz, base, cnt, dst, cap_bytes, n_bytes);
	}
	case MNC_DUMP_AN_OFFSETS:
		*n_bytes = (int64_t)(nr + 1) * 8;
		if (!dst || cap_bytes < *n_bytes) return MNC_ERR_RANGE;
		memcpy(dst, an_off.data(), (size_t)*n_bytes);
		return MNC_OK;
	case MNC_DUMP_ANCHORS: case MNC_DUMP_CHAIN_F: case MNC_DUMP_CHAIN_P: case MNC_DUMP_CHAIN_V: {
		const int64_t n = nr ? an_off[nr] : 0;
		const size_t es = what == MNC_DUMP_ANCHORS ? sizeof(Anchor) : 4;
		*n_bytes = n * (int64_t)es;
		if (!dst || cap_bytes < *n_bytes) return MNC_ERR_RANGE;
		const void *src = what == MNC_DUMP_ANCHORS ? (const void*)B.a : what == MNC_DUMP_CHAIN_F ? (const void*)B.f
		                : what == MNC_DUMP_CHAIN_P ? (const void*)B.p : (const void*)B.v;
		if (n) HIP_TRY(hipMemcpy(dst, src, (size_t)*n_bytes, hipMemcpyDeviceToHost));
		return MNC_OK;
	}
	case MNC_DUMP_REGS:
	case MNC_DUMP_REG_OFFSETS: {
		if (nr) HIP_TRY(hipMemcpy(c32.data(), B.n_reg, nr * 4, hipMemcpyDeviceToHost));
		if (what == MNC_DUMP_REG_OFFSETS) return dump_offsets(c32, dst, cap_bytes, n_bytes);
		for (size_t r = 0; r < nr; ++r) base[r] = an_off[r] / 3 + (int64_t)r * B.slot_pad, cnt[r] = c32[r];
		return dump_segments(B.regs, base, cnt, dst, cap_bytes, n_bytes);
	}
	case MNC_DUMP_CIGARS: {
		if (B.contract != MNC_CONTRACT_DP) { set_error("CIGARs exist under MNC_CONTRACT_DP only"); return MNC_ERR_ARG; }
		if (nr) HIP_TRY(hipMemcpy(c32.data(), B.n_reg, nr * 4, hipMemcpyDeviceToHost));
		std::vector<uint32_t> out;
		std::vector<RegDP> rd;
		for (size_t r = 0; r < nr; ++r) {
			if (c32[r] <= 0) continue;
			rd.resize((size_t)c32[r]);
			HIP_TRY(hipMemcpy(rd.data(), B.regdp + an_off[r] / 3 + (int64_t)r * B.slot_pad, (size_t)c32[r] * sizeof(RegDP), hipMemcpyDeviceToHost));
			for (int i = 0; i < c32[r]; ++i) {
				const size_t o = out.size();
				out.resize(o + (size_t)rd[i].n_cigar);
				if (rd[i].n_cigar > 0) HIP_TRY(hipMemcpy(out.data() + o, B.cig_reg + rd[i].cig_off, (size_t)rd[i].n_cigar * 4, hipMemcpyDeviceToHost));
			}
		}
		*n_bytes = (int64_t)out.size() * 4;
		if (!dst || cap_bytes < *n_bytes) return MNC_ERR_RANGE;
		if (!out.empty()) memcpy(dst, out.data(), out.size() * 4);
		return MNC_OK;
	}
	case MNC_DUMP_SEGS: {
		// the kernel calls of the alignment stage (every round's), as the plan kernel filed them and the kernels answered
		if (B.contract != MNC_CONTRACT_DP || !e->dp_ctr.p) { set_error("segments exist under MNC_CONTRACT_DP only"); return MNC_ERR_ARG; }
		unsigned long long n_seg = 0;
		HIP_TRY(hipMemcpy(&n_seg, e->dp_ctr.p, 8, hipMemcpyDeviceToHost));
		*n_bytes = (int64_t)(n_seg * sizeof(Seg));
		if (!dst || cap_bytes < *n_bytes) return MNC_ERR_RANGE;
		if (n_seg) HIP_TRY(hipMemcpy(dst, B.segs, (size_t)*n_bytes, hipMemcpyDeviceToHost));
		return MNC_OK;
	}
	case MNC_DUMP_REP_LEN:
		*n_bytes = (int64_t)nr * 4;
		if (!dst || cap_bytes < *n_bytes) return MNC_ERR_RANGE;
		if (nr) HIP_TRY(hipMemcpy(dst, B.rep_len, nr * 4, hipMemcpyDeviceToHost));
		return MNC_OK;
	default:
		return MNC_ERR_ARG;
	}
}
