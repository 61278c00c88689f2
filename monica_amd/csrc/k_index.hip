// Index residency: the device tables (per-region hash-and-displace perfect hash, displacement
// bytes, presence filter) built ON the device -- gfx950.
//
// The host form of this construction (engine.hip: index_upload, kept as the fallback) took 0.6 s for
// the 94 Mbp index on 16 host threads, every time an index part is first used on a device
// (monica/genomes/aligner.py:59 loads a part per pass of its loop).  The regions are independent:
// one wave per region, the region's occupancy as a bit map in LDS, and the same greedy order as the
// host form -- displacement buckets by size (largest first, then by index), for each the smallest
// displacement under which all its keys fall on free, distinct slots -- so the tables are the same
// bit for bit (tests/test_gpu_parity.py compares them).  64 candidate displacements are tried at once,
// one per lane.
#include "device.h"

namespace mnc {

constexpr int IDX_MAX_BUCKET = 32;          // keys of one displacement bucket handled here (more: host form)
constexpr int IDX_MAX_SALTS = 32;

__global__ __launch_bounds__(256) void mnc_idx_scatter(const uint32_t *keys, int64_t n_keys, const uint32_t *reg_off, uint32_t *cursor, uint32_t *reg_list, int pb_bits)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_keys) return;
	const uint32_t b = pb_bucket(keys[i], pb_bits);
	reg_list[reg_off[b] + atomicAdd(&cursor[b], 1u)] = (uint32_t)i;
}

__device__ __forceinline__ void idx_order()
{
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	asm volatile("" ::: "memory");
}

// one wave per region.  Scratch per region: cnt / start / fillc [NB] and the region's keys grouped by bucket.
__global__ __launch_bounds__(64) void mnc_idx_build(const uint32_t *keys, const uint64_t *key_off, const uint64_t *positions,
                                                    const uint32_t *reg_off, const uint32_t *reg_list, TableSlot *table, uint8_t *disp,
                                                    uint32_t *salt, uint32_t *filter, uint32_t *cnt_all, uint32_t *start_all, uint32_t *fill_all,
                                                    uint32_t *rk_all, int32_t *fail, int pb_bits, int region_bits, int disp_bits)
{
	extern __shared__ uint32_t s_mem[];
	const int lane = threadIdx.x, b = blockIdx.x;
	const uint32_t R = 1u << region_bits, NB = 1u << disp_bits;
	uint32_t *s_bits = s_mem;                               // R bits: slot taken
	uint32_t *s_slot = s_mem + R / 32;                      // [64][IDX_MAX_BUCKET] slots of a candidate displacement
	const uint32_t nk = reg_off[b + 1] - reg_off[b];
	const uint32_t *list = reg_list + reg_off[b];
	uint32_t *cnt = cnt_all + (size_t)b * NB, *start = start_all + (size_t)b * NB, *fillc = fill_all + (size_t)b * NB, *rk = rk_all + reg_off[b];
	TableSlot *T = table + (size_t)b * R;
	uint8_t *D = disp + (size_t)b * NB;
	for (uint32_t i = lane; i < nk; i += 64) {
		const uint32_t rest = pb_rest(keys[list[i]], pb_bits);
		atomicOr(&filter[(size_t)b * PF_WORDS + pf_word(rest)], pf_mask(rest));
		atomicAdd(&cnt[rest & (NB - 1)], 1u);
	}
	idx_order();
	uint32_t running = 0, maxs = 0;
	for (uint32_t base = 0; base < NB; base += 64) {
		const uint32_t c = base + lane < NB ? cnt[base + lane] : 0u;
		uint32_t incl = c;
#pragma unroll
		for (int sft = 1; sft < 64; sft <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, sft); if (lane >= sft) incl += o; }
		if (base + lane < NB) start[base + lane] = running + incl - c;
		running += (uint32_t)__shfl((int)incl, 63);
		uint32_t m = c;
#pragma unroll
		for (int sft = 32; sft > 0; sft >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, sft); m = m > o ? m : o; }
		maxs = maxs > m ? maxs : m;
	}
	idx_order();
	for (uint32_t i = lane; i < nk; i += 64) {
		const uint32_t ki = list[i], bkt = pb_rest(keys[ki], pb_bits) & (NB - 1);
		rk[start[bkt] + atomicAdd(&fillc[bkt], 1u)] = ki;
	}
	idx_order();
	if (maxs > (uint32_t)IDX_MAX_BUCKET) { if (lane == 0) fail[b] = 2; return; }
	bool placed = false;
	for (int sx = 0; sx < IDX_MAX_SALTS && !placed; ++sx) {
		// salt 0 first; a region is re-salted only when two keys of one displacement bucket share base and
		// step (no displacement can separate those)
		const uint32_t sv = (uint32_t)sx * 0x9E3779B9u;
		if (sx > 0) {
			for (uint32_t i = lane; i < R; i += 64) T[i] = TableSlot{0, 0, 0};
			for (uint32_t i = lane; i < NB; i += 64) D[i] = 0;
		}
		for (uint32_t i = lane; i < R / 32; i += 64) s_bits[i] = 0;
		idx_order();
		placed = true;
		for (uint32_t s = maxs; s >= 1 && placed; --s) {            // largest displacement buckets first, then by index
			for (uint32_t base = 0; base < NB && placed; base += 64) {
				unsigned long long m = __ballot(base + lane < NB && cnt[base + lane] == s);
				while (m && placed) {
					const uint32_t bkt = base + (uint32_t)__ffsll((long long)m) - 1u;
					m &= m - 1;
					const uint32_t st = start[bkt];
					const uint32_t my_ki = (uint32_t)lane < s ? rk[st + lane] : 0u;
					const uint32_t my_rest = (uint32_t)lane < s ? pb_rest(keys[my_ki], pb_bits) : 0u;
					int found = -1;
					for (uint32_t d0 = 0; d0 < 256 && found < 0; d0 += 64) {
						const uint32_t dd = d0 + lane;
						bool ok = true;
						for (uint32_t a = 0; a < s; ++a) {
							const uint32_t ra = (uint32_t)__shfl((int)my_rest, (int)a);
							const uint32_t sa = pd_slot(ra, dd, region_bits, sv);
							if (s_bits[sa >> 5] >> (sa & 31) & 1u) ok = false;
							for (uint32_t c = 0; c < a; ++c) if (s_slot[lane * IDX_MAX_BUCKET + c] == sa) ok = false;
							s_slot[lane * IDX_MAX_BUCKET + a] = sa;
						}
						const unsigned long long good = __ballot(ok);
						if (good) found = (int)d0 + __ffsll((long long)good) - 1;
					}
					if (found < 0) { placed = false; break; }
					if ((uint32_t)lane < s) {
						const uint32_t h = keys[my_ki];
						const uint64_t off = key_off[my_ki], c = key_off[my_ki + 1] - off;
						const uint32_t sl = pd_slot(my_rest, (uint32_t)found, region_bits, sv);
						atomicOr(&s_bits[sl >> 5], 1u << (sl & 31));
						TableSlot t;
						t.key = h + 1, t.cnt = (uint32_t)c, t.val = c == 1 ? positions[off] : off;
						T[sl] = t;
					}
					if (lane == 0) D[bkt] = (uint8_t)found;
					idx_order();
				}
			}
		}
		if (placed && lane == 0) salt[b] = sv;
	}
	if (!placed && lane == 0) fail[b] = 1;
}

// 0 = built; 1 = not applicable here (the host form takes over); < 0 = error
int index_tables_on_device(const std::vector<uint32_t> &keys, const std::vector<uint64_t> &key_off, const std::vector<uint32_t> &reg_count,
                           int pb_bits, int region_bits, int disp_bits, TableSlot *d_table, uint8_t *d_disp, uint32_t *d_salt, uint32_t *d_filter,
                           const uint64_t *d_positions)
{
	const size_t R = (size_t)1 << region_bits, NB = (size_t)1 << disp_bits;
	const int PB_N = 1 << pb_bits;
	const size_t lds = R / 8 + (size_t)64 * IDX_MAX_BUCKET * 4;
	if (lds > 150 * 1024 || keys.empty() || NB < 64) return 1;   // (a tiny index: fewer displacement buckets than lanes -- the host form)
	if (hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_idx_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return 1; }
	std::vector<uint32_t> reg_off(PB_N + 1, 0);
	for (int b = 0; b < PB_N; ++b) reg_off[b + 1] = reg_off[b] + reg_count[b];
	const size_t n = keys.size();
	uint32_t *d_keys = nullptr, *d_reg_off = nullptr, *d_cursor = nullptr, *d_list = nullptr, *d_cnt = nullptr, *d_rk = nullptr;
	uint64_t *d_koff = nullptr;
	int32_t *d_fail = nullptr;
	auto cleanup = [&]() { for (void *p : { (void*)d_keys, (void*)d_reg_off, (void*)d_cursor, (void*)d_list, (void*)d_cnt, (void*)d_rk, (void*)d_koff, (void*)d_fail }) if (p) (void)hipFree(p); };
#define IDX_TRY(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); cleanup(); return 1; } } while (0)
	IDX_TRY(hipMalloc(&d_keys, n * 4));
	IDX_TRY(hipMalloc(&d_koff, (n + 1) * 8));
	IDX_TRY(hipMalloc(&d_reg_off, (PB_N + 1) * 4));
	IDX_TRY(hipMalloc(&d_cursor, PB_N * 4));
	IDX_TRY(hipMalloc(&d_list, n * 4));
	IDX_TRY(hipMalloc(&d_cnt, (size_t)PB_N * NB * 4 * 3));
	IDX_TRY(hipMalloc(&d_rk, n * 4));
	IDX_TRY(hipMalloc(&d_fail, PB_N * 4));
	IDX_TRY(hipMemcpy(d_keys, keys.data(), n * 4, hipMemcpyHostToDevice));
	IDX_TRY(hipMemcpy(d_koff, key_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
	IDX_TRY(hipMemcpy(d_reg_off, reg_off.data(), (PB_N + 1) * 4, hipMemcpyHostToDevice));
	IDX_TRY(hipMemset(d_cursor, 0, PB_N * 4));
	IDX_TRY(hipMemset(d_cnt, 0, (size_t)PB_N * NB * 4 * 3));
	IDX_TRY(hipMemset(d_fail, 0, PB_N * 4));
	IDX_TRY(hipMemset(d_table, 0, (size_t)PB_N * R * sizeof(TableSlot)));
	IDX_TRY(hipMemset(d_disp, 0, (size_t)PB_N * NB));
	IDX_TRY(hipMemset(d_salt, 0, PB_N * 4));
	IDX_TRY(hipMemset(d_filter, 0, (size_t)PB_N * PF_WORDS * 4));
	hipLaunchKernelGGL(mnc_idx_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_keys, (int64_t)n, d_reg_off, d_cursor, d_list, pb_bits);
	hipLaunchKernelGGL(mnc_idx_build, dim3(PB_N), dim3(64), lds, 0, d_keys, d_koff, d_positions, d_reg_off, d_list, d_table, d_disp, d_salt, d_filter,
	                   d_cnt, d_cnt + (size_t)PB_N * NB, d_cnt + (size_t)PB_N * NB * 2, d_rk, d_fail, pb_bits, region_bits, disp_bits);
	std::vector<int32_t> fail(PB_N);
	IDX_TRY(hipMemcpy(fail.data(), d_fail, PB_N * 4, hipMemcpyDeviceToHost));
	IDX_TRY(hipGetLastError());
#undef IDX_TRY
	cleanup();
	for (int f : fail) if (f) return 1;                      // a bucket too large, or a region no salt could place: the host form decides (and may grow the regions)
	return 0;
}

} // namespace mnc
