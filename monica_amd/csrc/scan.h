// Exclusive scan (any integer input -> int64 prefix sums) on the device: reduce per tile, one wave over the tile sums,
// apply.  Shared by the engine (anchor / query / hit offsets of a batch) and the index builder (k_idxbuild.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace mnc {

// ================================================================ exclusive scan (int -> int64)
constexpr int SC_THREADS = 256, SC_ITEMS = 4, SC_TILE = SC_THREADS * SC_ITEMS;

template <class T> struct ScanInPlain {
	const T *p;
	__device__ long long operator()(int64_t i) const { return (long long)p[i]; }
};
template <class In>
__global__ __launch_bounds__(SC_THREADS) void mnc_scan_reduce(In in, int64_t n, int64_t *sums)
{
	__shared__ long long s[SC_THREADS / 64];
	const int64_t base = (int64_t)blockIdx.x * SC_TILE;
	long long x = 0;
	for (int k = 0; k < SC_ITEMS; ++k) {
		const int64_t i = base + (int64_t)threadIdx.x * SC_ITEMS + k;
		if (i < n) x += in(i);
	}
	for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
	if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = x;
	__syncthreads();
	if (threadIdx.x == 0) sums[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

static __global__ __launch_bounds__(64) void mnc_scan_sums(int64_t *sums, int64_t n_blocks)
{
	// one wave: serial over 64-wide strips; n_blocks is small (n / 1024)
	long long carry = 0;
	for (int64_t i0 = 0; i0 < n_blocks; i0 += 64) {
		const int64_t i = i0 + threadIdx.x;
		long long x = i < n_blocks ? sums[i] : 0, inc = x;
		for (int d = 1; d < 64; d <<= 1) {
			long long o = __shfl_up(inc, d);
			if ((int)threadIdx.x >= d) inc += o;
		}
		if (i < n_blocks) sums[i] = carry + inc - x;
		carry += __shfl(inc, 63);
	}
}

template <class In>
__global__ __launch_bounds__(SC_THREADS) void mnc_scan_apply(In in, int64_t n, const int64_t *sums, int64_t *out, int64_t out_tiles, int64_t out_stride)
{
	__shared__ long long s[SC_THREADS / 64];
	const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
	long long v[SC_ITEMS], x = 0;
	for (int k = 0; k < SC_ITEMS; ++k) { v[k] = base + k < n ? in(base + k) : 0; x += v[k]; }
	long long inc = x;
	for (int d = 1; d < 64; d <<= 1) {
		long long o = __shfl_up(inc, d);
		if ((int)(threadIdx.x & 63) >= d) inc += o;
	}
	if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = inc;
	__syncthreads();
	long long pre = sums[blockIdx.x] + inc - x;
	for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pre += s[w];
	for (int k = 0; k < SC_ITEMS; ++k) {
		if (base + k < n) {
			const int64_t i = base + k;
			out[out_tiles ? (i % out_tiles) * out_stride + i / out_tiles : i] = pre;
		}
		pre += v[k];
		if (base + k == n - 1) out[n] = pre;
	}
}

template <class In>
inline void exclusive_scan(In in, int64_t n, int64_t *out, int64_t *sums, hipStream_t st, int64_t out_tiles = 0, int64_t out_stride = 0)
{
	if (n <= 0) { (void)hipMemsetAsync(out, 0, 8, st); return; }
	const int64_t nb = (n + SC_TILE - 1) / SC_TILE;
	hipLaunchKernelGGL(mnc_scan_reduce<In>, dim3((unsigned)nb), dim3(SC_THREADS), 0, st, in, n, sums);
	hipLaunchKernelGGL(mnc_scan_sums, dim3(1), dim3(64), 0, st, sums, nb);
	hipLaunchKernelGGL(mnc_scan_apply<In>, dim3((unsigned)nb), dim3(SC_THREADS), 0, st, in, n, sums, out, out_tiles, out_stride);
}


} // namespace mnc
