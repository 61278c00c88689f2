// C2: the cross-part form of monica's hit merge, on the device.
//
// In the reference a read's gated hits are carried from index part to index part -- `sample_hits[read_id]`
// extended per part (monica/genomes/aligner.py:196-203, 218-223), pickled between passes (aligner.py:184-188,
// 267-273) -- and `best_hit` (aligner.py:328-339) decides over the union at the end.  `best_hit` only asks for the
// smallest NM/mlen and whether the LAST update of its running minimum was a tie, so a part is summarised per read by
// five integers {hits, nm, mlen, global contig, tied} (20 bytes; csrc/hostio.cpp's mnc_hitmap_update carries the same
// record between parts on the host).  When the parts live on different GPUs (BASELINE config 4) every rank writes its
// parts' records with mnc_shard_summary, all-gathers them (mnc_allgather_summaries) and reduces them with
// mnc_merge_summaries -- identically on every rank, part order = hit order.
//
// Both kernels are one thread per read over 20-byte records: HBM-bound, 20 B x parts read + 16 B written per read.
#include <hip/hip_runtime.h>

#include "common.h"

namespace mnc {

// {hits, nm, mlen, global contig | -1, tied} of one part from mnc_classify_device's outputs.  `best` is the part's
// minimal gated hit (also for MNC_AMBIGUOUS reads: the last of the tied ones).
__global__ __launch_bounds__(256) void mnc_k_shard_summary(const int32_t *assign, const mnc_hit_t *best, const int32_t *nhits, int64_t n,
                                                           int32_t rid_offset, int32_t *out)
{
	const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (r >= n) return;
	const int32_t h = nhits[r];
	const mnc_hit_t b = best[r];
	int32_t *o = out + r * 5;
	o[0] = h, o[1] = b.nm, o[2] = b.mlen, o[3] = h > 0 ? b.rid + rid_offset : -1, o[4] = assign[r] == MNC_AMBIGUOUS ? 1 : 0;
}

// best_hit over the union of the parts' lists, from their summaries.  The running minimum of aligner.py:331-337 walks
// the hits in list order with `<=`: a part whose minimum is smaller takes over with its own tie state (its tie was the
// last update inside that part); an equal one is an update by a tie; a larger one changes nothing.  NM/mlen compared
// as int64 cross-products (mlen > 0 for every gated hit), the rule of mnc_best_hit.
__global__ __launch_bounds__(256) void mnc_k_merge_summaries(const int32_t *parts, int P, int64_t n, int32_t *assign, int32_t *nm_out, int32_t *mlen_out, int32_t *total_out)
{
	const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (r >= n) return;
	bool has = false, tied = false;
	int64_t nm = 0, ml = 1, total = 0;
	int32_t rid = -1;
	for (int p = 0; p < P; ++p) {
		const int32_t *s = parts + ((int64_t)p * n + r) * 5;
		const int32_t c_hits = s[0];
		if (c_hits <= 0) continue;
		const int64_t c_nm = s[1], c_ml = s[2] > 0 ? s[2] : 1;
		const int64_t lhs = c_nm * ml, rhs = nm * c_ml;       // c_nm / c_ml ? nm / ml
		if (!has || lhs < rhs) tied = s[4] > 0, nm = c_nm, ml = c_ml, rid = s[3];
		else if (lhs == rhs) tied = true, nm = c_nm, ml = c_ml, rid = s[3];
		has = true;
		total += c_hits;
	}
	assign[r] = !has ? MNC_UNMAPPED : tied ? MNC_AMBIGUOUS : rid;
	if (nm_out) nm_out[r] = (int32_t)nm;
	if (mlen_out) mlen_out[r] = (int32_t)ml;
	if (total_out) total_out[r] = (int32_t)total;
}

static int need_device()
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); set_error("no HIP device visible"); return MNC_ERR_NODEVICE; }
	return MNC_OK;
}

static int launched(const char *what)
{
	const hipError_t e = hipGetLastError();
	if (e == hipSuccess) return MNC_OK;
	set_error("%s failed: %s", what, hipGetErrorString(e));
	return MNC_ERR_HIP;
}

} // namespace mnc

using namespace mnc;

extern "C" int mnc_shard_summary(const int32_t *d_assign, const mnc_hit_t *d_best, const int32_t *d_nhits, int64_t n,
                                 int32_t rid_offset, int32_t *d_out, void *stream)
{
	if (n < 0 || (n > 0 && (!d_assign || !d_best || !d_nhits || !d_out))) return MNC_ERR_ARG;
	if (int rc = need_device()) return rc;
	if (n == 0) return MNC_OK;
	hipLaunchKernelGGL(mnc_k_shard_summary, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_assign, d_best, d_nhits, n, rid_offset, d_out);
	return launched("mnc_shard_summary");
}

extern "C" int mnc_merge_summaries(const int32_t *d_parts, int n_parts, int64_t n, int32_t *d_assign, int32_t *d_nm, int32_t *d_mlen,
                                   int32_t *d_total, void *stream)
{
	if (n < 0 || n_parts < 1 || (n > 0 && (!d_parts || !d_assign))) return MNC_ERR_ARG;
	if (int rc = need_device()) return rc;
	if (n == 0) return MNC_OK;
	hipLaunchKernelGGL(mnc_k_merge_summaries, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_parts, n_parts, n, d_assign, d_nm, d_mlen, d_total);
	return launched("mnc_merge_summaries");
}
