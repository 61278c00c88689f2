// Issue rate of the vector instructions the alignment kernels are made of, on gfx950: cycles per wave64
// instruction and SIMD with 1, 2, 4, 8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
// Round 5: the shader clock is READ, not assumed -- wave 0 of every launch takes s_memtime (shader cycles) and
// s_memrealtime (the constant 100 MHz counter) at both ends; the rows are in measured cycles, "clock_mhz" is what the
// launches ran at, and "at_2p4_ghz" repeats the figures of the earlier rounds' convention (time x 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(64) void k(uint32_t *out, int iters, uint32_t seed, unsigned long long *clk)
{
	const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
	const uint32_t b = seed * 2654435761u | 0x00010001u;
	for (int i = 0; i < iters; ++i) {
#define ONE(INS) asm volatile(INS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
		if (OP == 0) { REP16(ONE("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8")) }
		if (OP == 1) { REP16(ONE("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8")) }
		if (OP == 2) { REP16(ONE("v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_add_i16 %3, %3, %8 clamp\n v_pk_add_i16 %4, %4, %8 clamp\n v_pk_add_i16 %5, %5, %8 clamp\n v_pk_add_i16 %6, %6, %8 clamp\n v_pk_add_i16 %7, %7, %8 clamp")) }
		if (OP == 3) { REP16(ONE("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8")) }
		if (OP == 4) { REP16(ONE("v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8")) }
		if (OP == 5) { REP16(ONE("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf")) }
		if (OP == 6) { REP16(ONE("v_and_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_alignbit_b32 %4, %4, %8, 16\n v_alignbit_b32 %5, %5, %8, 16\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8")) }
		if (OP == 7) { REP16(ONE("v_pk_fma_f16 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n v_pk_fma_f16 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_pk_fma_f16 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8")) }
		if (OP == 8) { REP16(ONE("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8")) }
		if (OP == 9) { REP16(ONE("v_pk_mad_u16 %0, %0, %8, %8\n v_pk_mad_u16 %1, %1, %8, %8\n v_pk_mad_u16 %2, %2, %8, %8\n v_pk_mad_u16 %3, %3, %8, %8\n v_pk_mad_u16 %4, %4, %8, %8\n v_pk_mad_u16 %5, %5, %8, %8\n v_pk_mad_u16 %6, %6, %8, %8\n v_pk_mad_u16 %7, %7, %8, %8")) }
		if (OP == 10) { REP16(ONE("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8")) }
		if (OP == 11) { REP16(ONE("v_pk_max_f16 %0, %0, %8\n v_pk_max_f16 %1, %1, %8\n v_pk_max_f16 %2, %2, %8\n v_pk_max_f16 %3, %3, %8\n v_pk_max_f16 %4, %4, %8\n v_pk_max_f16 %5, %5, %8\n v_pk_max_f16 %6, %6, %8\n v_pk_max_f16 %7, %7, %8")) }
		if (OP == 12) { REP16(ONE("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8")) }
		if (OP == 13) { REP16(ONE("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8")) }
		if (OP == 14) { REP16(ONE("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8")) }
		if (OP == 15) { REP16(ONE("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8")) }
		if (OP == 16) { REP16(ONE("v_lshlrev_b32 %0, 4, %0\n v_lshlrev_b32 %1, 4, %1\n v_lshlrev_b32 %2, 4, %2\n v_lshlrev_b32 %3, 4, %3\n v_lshlrev_b32 %4, 4, %4\n v_lshlrev_b32 %5, 4, %5\n v_lshlrev_b32 %6, 4, %6\n v_lshlrev_b32 %7, 4, %7")) }
		if (OP == 17) { REP16(ONE("v_alignbit_b32 %0, %0, %8, 16\n v_alignbit_b32 %1, %1, %8, 16\n v_alignbit_b32 %2, %2, %8, 16\n v_alignbit_b32 %3, %3, %8, 16\n v_alignbit_b32 %4, %4, %8, 16\n v_alignbit_b32 %5, %5, %8, 16\n v_alignbit_b32 %6, %6, %8, 16\n v_alignbit_b32 %7, %7, %8, 16")) }
		if (OP == 18) { REP16(ONE("v_bfi_b32 %0, %8, %0, %8\n v_bfi_b32 %1, %8, %1, %8\n v_bfi_b32 %2, %8, %2, %8\n v_bfi_b32 %3, %8, %3, %8\n v_bfi_b32 %4, %8, %4, %8\n v_bfi_b32 %5, %8, %5, %8\n v_bfi_b32 %6, %8, %6, %8\n v_bfi_b32 %7, %8, %7, %8")) }
		if (OP == 19) { REP16(ONE("v_bitop3_b32 %0, %0, %8, %8 bitop3:0x96\n v_bitop3_b32 %1, %1, %8, %8 bitop3:0x96\n v_bitop3_b32 %2, %2, %8, %8 bitop3:0x96\n v_bitop3_b32 %3, %3, %8, %8 bitop3:0x96\n v_bitop3_b32 %4, %4, %8, %8 bitop3:0x96\n v_bitop3_b32 %5, %5, %8, %8 bitop3:0x96\n v_bitop3_b32 %6, %6, %8, %8 bitop3:0x96\n v_bitop3_b32 %7, %7, %8, %8 bitop3:0x96")) }
		if (OP == 20) { REP16(ONE("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8")) }
		if (OP == 21) { REP16(ONE("v_pk_sub_u16 %0, %0, %8\n v_pk_sub_u16 %1, %1, %8\n v_pk_sub_u16 %2, %2, %8\n v_pk_sub_u16 %3, %3, %8\n v_pk_sub_u16 %4, %4, %8\n v_pk_sub_u16 %5, %5, %8\n v_pk_sub_u16 %6, %6, %8\n v_pk_sub_u16 %7, %7, %8")) }
		if (OP == 22) { REP16(ONE("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8")) }
		if (OP == 23) { REP16(ONE("v_pk_max_u16 %0, %0, %8\n v_pk_max_u16 %1, %1, %8\n v_pk_max_u16 %2, %2, %8\n v_pk_max_u16 %3, %3, %8\n v_pk_max_u16 %4, %4, %8\n v_pk_max_u16 %5, %5, %8\n v_pk_max_u16 %6, %6, %8\n v_pk_max_u16 %7, %7, %8")) }
		if (OP == 24) { REP16(ONE("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8")) }
		if (OP == 25) { REP16(ONE("v_add3_u32 %0, %0, %8, %8\n v_add3_u32 %1, %1, %8, %8\n v_add3_u32 %2, %2, %8, %8\n v_add3_u32 %3, %3, %8, %8\n v_add3_u32 %4, %4, %8, %8\n v_add3_u32 %5, %5, %8, %8\n v_add3_u32 %6, %6, %8, %8\n v_add3_u32 %7, %7, %8, %8")) }
		if (OP == 26) { REP16(ONE("v_lshl_add_u32 %0, %0, 4, %8\n v_lshl_add_u32 %1, %1, 4, %8\n v_lshl_add_u32 %2, %2, 4, %8\n v_lshl_add_u32 %3, %3, 4, %8\n v_lshl_add_u32 %4, %4, 4, %8\n v_lshl_add_u32 %5, %5, 4, %8\n v_lshl_add_u32 %6, %6, 4, %8\n v_lshl_add_u32 %7, %7, 4, %8")) }
		if (OP == 27) { REP16(ONE("v_and_or_b32 %0, %0, %8, %8\n v_and_or_b32 %1, %1, %8, %8\n v_and_or_b32 %2, %2, %8, %8\n v_and_or_b32 %3, %3, %8, %8\n v_and_or_b32 %4, %4, %8, %8\n v_and_or_b32 %5, %5, %8, %8\n v_and_or_b32 %6, %6, %8, %8\n v_and_or_b32 %7, %7, %8, %8")) }
		if (OP == 28) { REP16(ONE("v_lshl_or_b32 %0, %0, 4, %8\n v_lshl_or_b32 %1, %1, 4, %8\n v_lshl_or_b32 %2, %2, 4, %8\n v_lshl_or_b32 %3, %3, 4, %8\n v_lshl_or_b32 %4, %4, 4, %8\n v_lshl_or_b32 %5, %5, 4, %8\n v_lshl_or_b32 %6, %6, 4, %8\n v_lshl_or_b32 %7, %7, 4, %8")) }
		if (OP == 29) { REP16(ONE("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf")) }
		if (OP == 30) { REP16(ONE("v_max_i16 %0, %0, %8\n v_max_i16 %1, %1, %8\n v_max_i16 %2, %2, %8\n v_max_i16 %3, %3, %8\n v_max_i16 %4, %4, %8\n v_max_i16 %5, %5, %8\n v_max_i16 %6, %6, %8\n v_max_i16 %7, %7, %8")) }
		if (OP == 31) { REP16(ONE("v_add_u16 %0, %0, %8\n v_add_u16 %1, %1, %8\n v_add_u16 %2, %2, %8\n v_add_u16 %3, %3, %8\n v_add_u16 %4, %4, %8\n v_add_u16 %5, %5, %8\n v_add_u16 %6, %6, %8\n v_add_u16 %7, %7, %8")) }
	}
	out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
	if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = __builtin_readcyclecounter() - c0, clk[1] = wall_clock64() - w0;
}

static double g_mhz_sum = 0, g_cyc2p4[40][4];
static int g_mhz_n = 0;
template <int OP> double run(uint32_t *d, int waves_per_simd, int iters)
{
	static unsigned long long *clk = nullptr;
	if (!clk) hipHostMalloc((void**)&clk, 16, hipHostMallocDefault);
	const int n_wg = 256 * 4 * waves_per_simd;
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipLaunchKernelGGL(k<OP>, dim3(n_wg), dim3(64), 0, 0, d, 10, 1u, clk);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k<OP>, dim3(n_wg), dim3(64), 0, 0, d, iters, 1u, clk);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double instr_per_simd = (double)waves_per_simd * iters * 16 * 8;
	hipDeviceSynchronize();
	const double mhz = clk[1] ? (double)clk[0] / (double)clk[1] * 100.0 : 0;       // shader cycles per tick of the 100 MHz counter
	g_mhz_sum += mhz, ++g_mhz_n;
	const int wi = waves_per_simd == 1 ? 0 : waves_per_simd == 2 ? 1 : waves_per_simd == 4 ? 2 : 3;
	g_cyc2p4[OP][wi] = ms * 1e-3 * 2.4e9 / instr_per_simd;
	return ms * 1e-3 * mhz * 1e6 / instr_per_simd;        // cycles per wave64 instruction and SIMD, at the clock the launch ran at
}

int main()
{
	uint32_t *d;
	hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
	const char *names[] = { "v_add_u32", "v_pk_max_i16", "v_pk_add_i16 clamp", "v_fma_f32", "v_perm_b32", "v_mov_b32_dpp row_shr:1", "and/or/alignbit/xor mix", "v_pk_fma_f16", "v_max_i32", "v_pk_mad_u16", "v_max_f32", "v_pk_max_f16", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_alignbit_b32", "v_bfi_b32", "v_bitop3_b32", "v_min_u32", "v_pk_sub_u16", "v_pk_add_u16", "v_pk_max_u16", "v_mov_b32", "v_add3_u32", "v_lshl_add_u32", "v_and_or_b32", "v_lshl_or_b32", "v_mov_b32_dpp wave_shr:1", "v_max_i16", "v_add_u16" };
	printf("{\n");
#define ROW(OP) printf(" \"%s\": {\"1\": %.2f, \"2\": %.2f, \"4\": %.2f, \"8\": %.2f}%s\n", names[OP], run<OP>(d, 1, 4000), run<OP>(d, 2, 4000), run<OP>(d, 4, 2000), run<OP>(d, 8, 1000), OP == 31 ? "" : ",");
	ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16) ROW(17) ROW(18) ROW(19) ROW(20) ROW(21) ROW(22) ROW(23) ROW(24) ROW(25) ROW(26) ROW(27) ROW(28) ROW(29) ROW(30) ROW(31)
	printf("}\n");
	fprintf(stderr, "{\"clock_mhz\": %.1f, \"at_2p4_ghz\": {\"v_pk_max_i16\": {\"1\": %.2f, \"8\": %.2f}, \"v_perm_b32\": {\"1\": %.2f, \"8\": %.2f}, \"v_add_u32\": {\"1\": %.2f, \"8\": %.2f}}}\n",
	        g_mhz_sum / (g_mhz_n ? g_mhz_n : 1), g_cyc2p4[1][0], g_cyc2p4[1][3], g_cyc2p4[4][0], g_cyc2p4[4][3], g_cyc2p4[0][0], g_cyc2p4[0][3]);
	return 0;
}
