// Stage kernels K8: base-level alignment of every region -- gfx950.
//
// Replaces mm_align_skeleton() / mm_align1() / ksw_extd2_sse() inside index.map(seq)
// (monica/genomes/aligner.py:193, 215: mappy always sets MM_F_CIGAR), whose
// results monica reads as hit.mapq / hit.NM / hit.mlen (aligner.py:194-195, 216-217).
//
//   mnc_dp_gather   wave / read: the anchors of the kept chains, squeezed together in `as` order
//                   (mm_squeeze_a), LONG_JOIN flag on the first anchor of a fused chain
//   mnc_dp_plan     thread / region: trim bad chain ends, flag seeds around long indels, DP window
//                   from neighbouring seeds, the list of kernel calls ("segments": left extension,
//                   one gap filling per >= min_ksw_len of seeds, right extension)
//   mnc_dp_plan_long  wave / region, for the regions of long reads: the same, the passes over all
//                   anchors as wave-wide steps on a copy of the anchors in LDS
//   mnc_dp_inv      wave / region tail after an inversion-like Z-drop: mm_align1_inv (local alignment
//                   of the reverse-complemented query stretch, extension back from its end)
//   mnc_dp_align    wave / segment from a work queue: ksw2's two-piece affine kernel in its
//                   anti-diagonal difference form (u, v, x, y, x2, y2 as int8 in LDS, 64 cells per
//                   step), the exact / approximate maximum, Z-drop, direction bytes to HBM, the
//                   backtrack; gap fillings run minimap2's two passes (approximate first, exact
//                   only when a walk over the CIGAR shows a large drop)
//   mnc_dp_stitch   wave / region: CIGARs of the segments joined (a Z-drop ends the region and
//                   its tail becomes a new region for the next round), indel left-alignment and
//                   I/D merging (mm_fix_cigar), one walk giving mlen, blen, n_ambi, dp_max
//                   (mm_update_extra)
//
// The kernel follows the SSE kernel's layout literally -- flat byte buffer, 16-lane rounding of
// every anti-diagonal, in-place update, int8 wrap-around -- because at band edges ksw2 reads
// cells outside the band whose contents only that layout defines; tests/ compare it with the
// CPU oracle's simulation of the same layout, CIGAR for CIGAR.
#include <cstring>
#include <cstdlib>
#include "device.h"
#include "ksw_pk.h"

namespace mnc {

__device__ __forceinline__ void mem_order()
{
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	asm volatile("" ::: "memory");
}

__device__ __forceinline__ int nt4_code(uint8_t c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

// base p of strand `rev` of a read (qseq0[rev][p] of mm_align_skeleton)
__device__ __forceinline__ int qcode(const uint8_t *read, int qlen, int rev, int p)
{
	const int c = nt4_code(read[rev ? qlen - 1 - p : p]);
	return rev ? (c < 4 ? 3 - c : 4) : c;
}

// base `pos` of contig `rid` (mm_idx_getseq)
__device__ __forceinline__ int tcode(const Batch &B, int64_t contig_off, int pos)
{
	const int64_t o = contig_off + pos;
	return (int)(B.seq4[o >> 3] >> ((o & 7) * 4) & 15u);
}

// ================================================================ gather: chained anchors, squeezed
constexpr int GATHER_LONG = 1024;       // reads with more anchors: mnc_dp_gather_long

// One wave per read: p[] of the read in LDS; lane c walks chain c there (its anchors' indices go to the chain's place in
// the squeezed order), then all lanes copy the anchors, coalesced.  (One LANE per read walking p[] through memory --
// the first form -- was 0.7 ms of dependent round trips per batch.)
__global__ __launch_bounds__(64) void mnc_dp_gather(Batch B)
{
	__shared__ int32_t s_p[GATHER_LONG], s_j[GATHER_LONG];
	__shared__ unsigned long long s_gapb[GATHER_LONG / 64 + 1], s_startb[GATHER_LONG / 64 + 1];   // per position: a step of more than ten bases; a region's first anchor (or a hole)
	const uint32_t rd = blockIdx.x;
	const int lane = threadIdx.x;
	const int n = B.n_chain[rd];
	if (n <= 0) return;
	const int64_t a_off = B.an_off[rd], slot = a_off / 3;
	const int n_an = (int)(B.an_off[rd + 1] - a_off);
	if (n_an > GATHER_LONG) return;                          // mnc_dp_gather_long takes it
	const ChainRec *ch = B.chains_tmp + slot;
	const Anchor *a = B.a + a_off;
	const int32_t *p = B.p + a_off;
	Anchor *ca = B.ca + a_off;
	for (int i = lane; i < n_an; i += 64) s_p[i] = p[i], s_j[i] = INT32_MAX;
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	int total = 0;
	for (int c = lane; c < n; c += 64) {
		const int d = B.chain_dst[slot + c];
		if (d < 0) continue;
		const int dst = d & ((1 << 30) - 1), cnt = ch[c].cnt;
		int j = ch[c].as;                                   // the chain's last anchor; p[] leads back
		for (int k = cnt - 1; k >= 0; --k) {
			s_j[dst + k] = k != 0 ? j : (d >> 30 & 1) ? j | INT32_MIN : j | 1 << 30;   // the sign: first anchor of a long-joined chain; bit 30: of a region
			j = s_p[j];
		}
		total = total > dst + cnt ? total : dst + cnt;
	}
#pragma unroll
	for (int sft = 32; sft > 0; sft >>= 1) { const int o = __shfl_xor(total, sft); total = total > o ? total : o; }
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	// the copy, and on the way: where do consecutive anchors of a region step more than ten bases apart in query and target?
	// (what mm_filter_bad_seeds and mm_filter_bad_seeds_alt look for first; a region without such a step gets SEED_NOGAP10)
	int cx = 0, cy = 0, ce = 1;                             // anchor i0 - 1: its coordinates; nothing there
	for (int i0 = 0; i0 < total; i0 += 64) {
		const int i = i0 + lane;
		const int jj = i < total ? s_j[i] : INT32_MAX;
		const bool empty = jj == INT32_MAX;                 // (no chain put an anchor here)
		Anchor x;
		x.x = x.y = 0;
		if (!empty) x = a[jj & ((1 << 30) - 1)];
		const bool rstart = !empty && jj >= 0 && (jj >> 30 & 1);
		if (jj < 0) x.y |= SEED_LONG_JOIN;
		int px = __shfl_up((int)(uint32_t)x.x, 1), py = __shfl_up((int)(uint32_t)x.y, 1), pe = __shfl_up((int)empty, 1);
		if (lane == 0) px = cx, py = cy, pe = ce;
		cx = __shfl((int)(uint32_t)x.x, 63), cy = __shfl((int)(uint32_t)x.y, 63), ce = __shfl((int)empty, 63);
		const int gap = ((int32_t)(uint32_t)x.y - py) - ((int32_t)(uint32_t)x.x - px);
		const bool big = !empty && !rstart && (pe || gap < -10 || gap > 10);
		const unsigned long long bg = __ballot(big), bs = __ballot(rstart || empty);
		if (lane == 0) s_gapb[i0 >> 6] = bg, s_startb[i0 >> 6] = bs;
		if (!empty && !rstart) ca[i] = x;
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	for (int i = lane; i < total; i += 64) {
		const int jj = s_j[i];
		if (jj == INT32_MAX || jj < 0 || !(jj >> 30 & 1)) continue;
		// a region's first anchor: any long step between here and the next first anchor (or the end)?
		bool any = false;
		for (int w = i >> 6, first = 1; w <= (total - 1) >> 6; ++w, first = 0) {
			const unsigned long long after = first ? ((i & 63) == 63 ? 0ULL : ~0ULL << ((i & 63) + 1)) : ~0ULL;
			const unsigned long long st = s_startb[w] & after;
			const unsigned long long upto = st ? (st & (0 - st)) - 1ULL : ~0ULL;            // the positions before the next first anchor
			if (s_gapb[w] & after & upto) { any = true; break; }
			if (st) break;
		}
		Anchor x = a[jj & ((1 << 30) - 1)];
		if (!any) x.y |= SEED_NOGAP10;
		ca[i] = x;
	}
}

// The same for a long read (a 60 kb read has ~4 700 chained anchors: one lane following p[] through HBM is 4 ms of
// round trips): one wave per read, p[] in LDS, lane 0 walks a chain there and leaves the indices, all lanes copy.
// `lists` / `spans`: the reads of the chain stage's size classes above GATHER_LONG anchors (and of the class beyond
// the largest, whose p[] does not fit LDS: those walk in HBM as before).
__global__ __launch_bounds__(64) void mnc_dp_gather_long(Batch B, const uint32_t *lists, ClassSpans spans, int lds_anchors)
{
	extern __shared__ int32_t gl_smem[];
	if (blockIdx.x >= spans.start[spans.n]) return;
	int cls = 0;
	while (blockIdx.x >= spans.start[cls + 1]) ++cls;
	const uint32_t rd = lists[(size_t)cls * spans.stride + (blockIdx.x - spans.start[cls])];
	const int lane = threadIdx.x;
	const int n = B.n_chain[rd];
	if (n <= 0) return;
	const int64_t a_off = B.an_off[rd], slot = a_off / 3;
	const int n_an = (int)(B.an_off[rd + 1] - a_off);
	if (n_an <= GATHER_LONG) return;                         // mnc_dp_gather took it
	const ChainRec *ch = B.chains_tmp + slot;
	const Anchor *a = B.a + a_off;
	const int32_t *p = B.p + a_off;
	Anchor *ca = B.ca + a_off;
	const bool in_lds = n_an <= lds_anchors;
	int32_t *s_p = gl_smem, *s_j = gl_smem + lds_anchors;
	if (in_lds) for (int i = lane; i < n_an; i += 64) s_p[i] = p[i];
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	for (int c = 0; c < n; ++c) {
		const int d = B.chain_dst[slot + c];
		if (d < 0) continue;
		const int dst = d & ((1 << 30) - 1), cnt = ch[c].cnt;
		if (in_lds) {
			if (lane == 0) {
				int j = ch[c].as;
				for (int k = cnt - 1; k >= 0; --k) s_j[k] = j, j = s_p[j];
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
			for (int k = lane; k < cnt; k += 64) {
				Anchor x = a[s_j[k]];
				if (k == 0 && (d >> 30 & 1)) x.y |= SEED_LONG_JOIN;
				ca[dst + k] = x;
			}
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		} else if (lane == 0) {
			int j = ch[c].as;
			for (int k = cnt - 1; k >= 0; --k) {
				Anchor x = a[j];
				if (k == 0 && (d >> 30 & 1)) x.y |= SEED_LONG_JOIN;
				ca[dst + k] = x;
				j = p[j];
			}
		}
	}
}

// ================================================================ plan: one region (mm_align1 up to the kernel calls)
__device__ __forceinline__ int seed_gap(const Anchor *a, int i)
{
	return ((int32_t)a[i].y - (int32_t)a[i - 1].y) - ((int32_t)a[i].x - (int32_t)a[i - 1].x);
}

// (eight anchors per turn, loaded before any is used: one lane walks a region's anchors alone, and a turn's time is its
// loads' round trip)
__device__ int collect_long_gaps(const Anchor *a, int cnt1, int min_gap, int32_t *K)
{
	int n = 0;
	int32_t px = (int32_t)a[0].x, py = (int32_t)a[0].y;
	for (int i = 1; i < cnt1; i += 8) {
		int32_t x[8], y[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) { const int ii = i + u < cnt1 ? i + u : cnt1 - 1; x[u] = (int32_t)a[ii].x, y[u] = (int32_t)a[ii].y; }
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			if (i + u < cnt1) {
				const int gap = (y[u] - py) - (x[u] - px);
				if (gap < -min_gap || gap > min_gap) K[n++] = i + u;
				px = x[u], py = y[u];
			}
		}
	}
	return n <= 1 ? 0 : n;
}

// The kernel a segment goes to (Seg.big: 0..3 the literal kernel's workspace classes, 4 + t the work list t of a banded /
// extension kernel) and its bookkeeping fields; `widx` / `wamt`: what it adds to the work counters dp_ctr[48 + widx].
struct PlanLimits { int slot_state_max; long long slot_p_max; int slot_cig_max; long long big_state_max, big_p_max, big_cig_max, huge_state_max, huge_p_max, huge_cig_max; };
// The score per base (x 100) a gap filling of a region has to be expected to reach for a tier to be worth trying: Batch.fill_pred
// / fill_pred_mid / 32, all x 4 -- fitted on reads with 10 % errors -- or, round 5, from the region itself.  The density of
// its chained anchors, rho = cnt (w + 1) / (2 x query span), is (1 - eps)^k in expectation, and it is that to half a per cent
// of eps read by read (3 / 7 / 10 / 13 / 16 % errors: 3.1 / 7.2 / 10.3 / 13.3 / 16.3 % from rho, tools/pred_fit.py); with
// minimap2's map-ont scores a gap filling reaches 2 - 6.0 eps per base of its shorter side (1.81 / 1.57 / 1.40 / 1.23 / 1.07).
// The constants sit 0.115 (32- and wider tiers) and 0.035 (42 cells) under that at 10 %: the same distances here.  A read at
// 13 % used to send a quarter of its 42-cell tries on to the next tier; a read at 3 % took wide tiers it did not need.
struct FillPred { int t1, mid, wide, last; };
__device__ __forceinline__ FillPred fill_pred_of(const Batch &B, int cnt, int q_span)
{
	FillPred p = { B.fill_pred * 4, B.fill_pred_mid * 4, 32 * 4, 32 * 4 };
	if (B.fill_pred_auto && cnt >= 20 && q_span >= 500) {
		float rho = (float)cnt * 11.0f / (2.0f * (float)q_span);            // (k = 15, w = 10: the only sketch there is, check_kw)
		rho = rho < 0.02f ? 0.02f : rho > 1.0f ? 1.0f : rho;
		float eps = 1.0f - __expf(__logf(rho) * (1.0f / 15.0f)) - 0.003f;
		eps = eps < 0.0f ? 0.0f : eps > 0.25f ? 0.25f : eps;
		const float mu = 2.0f - 6.0f * eps;
		// What stands behind a tier decides how sure it has to be: trying a tier that costs a before one that costs b pays
		// when the chance of its proof is above a / b.  A segment's step costs 1 : 1.35 : 3.1 : 8.6 in the four tiers as they
		// run (not 1 : 4/3 : 2 : 4 as their cells: the wide tiers fill fewer lanes and waves), so 32 cells want three chances
		// in four, 42 cells and 64 cells less than even odds -- quantiles of the score per base, whose spread from segment
		// to segment is 0.39 sqrt(eps (1 - eps)) (0.12 at 10 %).  The 128-cell tier stands before the literal kernel, a
		// call of milliseconds on a wave of its own: tried on any chance at all, never below the fixed threshold.
		const float sd = 0.39f * __fsqrt_rn(eps * (1.0f - eps) + 1e-4f);
		p.t1 = (int)(100.0f * (mu - 0.64f * sd)), p.mid = (int)(100.0f * (mu + 0.15f * sd)), p.wide = (int)(100.0f * (mu + 0.36f * sd));
		const int opt = (int)(100.0f * (mu + 2.0f * sd));
		p.last = opt > p.last ? opt : p.last;
	}
	return p;
}
__device__ void plan_seg_class(const Batch &B, const PlanLimits &lim, const FillPred &pred, int bw, uint32_t rd, int64_t rslot, int rid, int rev, int32_t seg_index,
                               Seg &g, int &widx, unsigned long long &wamt)
{
	widx = -1, wamt = 0;
	g.read = (int32_t)rd, g.reg = (int32_t)rslot, g.rid = rid, g.rev = rev;
	g.n_cigar = 0, g.zdropped = 0, g.zdrop_code = 0, g.max = 0, g.max_t = g.max_q = -1, g.score = DP_NEG_INF, g.reach_end = 0, g.mqe_t = -1, g.cig_off = 0;
	// workspace class
	const long long T = (g.tlen + 15) / 16 * 16, Q = (g.qlen + 15) / 16 * 16 + 32;
	long long nc = g.qlen < g.tlen ? g.qlen : g.tlen;
	const int wb = g.w < 0 ? (g.tlen > g.qlen ? g.tlen : g.qlen) : g.w;
	nc = ((nc < wb + 1 ? nc : wb + 1) + 15) / 16 + 1;
	const long long p_bytes = ((long long)(g.qlen + g.tlen - 1) * nc + 1) * 16;
	g.big = (12 * T + Q > lim.slot_state_max || p_bytes > lim.slot_p_max || g.qlen + g.tlen + 8 > lim.slot_cig_max) ? 1 : 0;
	if ((long long)g.tlen * g.qlen > B.max_sw_mat) g.big = 0;      // not aligned at all (ksw_reset_extz + zdropped)
	else if (12 * T + Q > lim.big_state_max || p_bytes > lim.big_p_max || g.qlen + g.tlen + 8 > lim.big_cig_max) {
		g.big = 3;                                              // the few largest: a class of their own, a handful of very large slots
		if (12 * T + Q > lim.huge_state_max || p_bytes > lim.huge_p_max || g.qlen + g.tlen + 8 > lim.huge_cig_max) {
			// beyond even those (a direction matrix of more than 256 MB with tlen x qlen <= max_sw_mat: a few hundred
			// query bases against hundreds of thousands of target bases -- nothing mm_align1 makes with max_gap 5 000):
			// the call is not run, the read is reported MNC_SKIPPED and the rest of the batch is classified
			g.big = 0, g.flag |= SEG_SKIPPED;
			B.skip[rd] = 1;
		}
	}
	bool lfill = false;
	if (!(B.debug_route & 4) && g.big <= 1 && g.kind == 1 && g.w == bw && g.tlen >= 1 && g.qlen >= 1 && g.tlen <= 2047 && g.qlen <= 2047 &&
	    (g.tlen > FILL_MAX_LEN || g.qlen > FILL_MAX_LEN) && (long long)g.tlen * g.qlen <= B.max_sw_mat) {
		// a longer gap between two seeds (512 .. 2047 bases): the banded kernel's int32 form, a band of 256
		// cells (one launch, early, beside the packed tiers: a narrower first try would put its failures
		// behind them)
		const int ad = g.tlen > g.qlen ? g.tlen - g.qlen : g.qlen - g.tlen;
		const int tier = (510 - ad) / 2 >= 8 ? 18 : 0;
		if (tier) {
			if (g.big == 1) g.flag |= SEG_NEEDS_BIG_WS;           // should the banded kernel hand it back
			g.big = 4 + tier, lfill = true;
		}
	}
	if (lfill) {
	} else if (!(B.debug_route & 8) && g.big == 1 && g.kind != 1 && g.tlen >= 1 && g.qlen >= 1 && g.tlen <= g.w && g.qlen <= g.w &&
	           g.tlen + g.qlen - 1 <= 2 * 1535 && (g.tlen < g.qlen ? g.tlen : g.qlen) <= 512) {
		g.flag |= SEG_NEEDS_BIG_WS;                             // a longer extension (below) that would need the large workspace
		g.big = 4 + 19;
	} else if (g.big == 1) {
		const unsigned long long bi = atomicAdd(&B.dp_ctr[6], 1ULL);
		B.big_list[bi] = seg_index;
	} else if (g.big == 3) {
		const unsigned long long bi = atomicAdd(&B.dp_ctr[58], 1ULL);
		B.huge_list[bi] = seg_index;
	} else if (!(B.debug_route & 2) && g.big == 0 && g.kind == 1 && g.w == bw && g.tlen >= 1 && g.qlen >= 1 && g.tlen <= FILL_MAX_LEN && g.qlen <= FILL_MAX_LEN) {
		// a gap between two seeds whose matrix the band never clips: the banded kernel of
		// k_fill.hip, 32 lanes per segment when |tlen - qlen| leaves a band worth trying, else 64
		const int ad = g.tlen > g.qlen ? g.tlen - g.qlen : g.qlen - g.tlen;
		// the 32-lane tier only when its proof has a chance: the bound a band of that width leaves
		// against what a read with ~10 % errors scores (~1.28 per base)
		int tier = (62 - ad) / 2 >= 12 ? 1 : (2 * FILL_MID_CELLS - 2 - ad) / 2 >= 12 ? 18 : (126 - ad) / 2 >= 8 ? 2 : 6;
		if (tier == 1) {
			const int bb = (62 - ad) / 2, mn = g.tlen < g.qlen ? g.tlen : g.qlen;
			const int g1 = B.gap_q + B.gap_e * (bb + 1), g2 = B.gap_q2 + B.gap_e2 * (bb + 1);
			const int U = B.sc_a * (mn - bb - 1) - 2 * (g1 < g2 ? g1 : g2);
			if (U * 100 > mn * pred.t1) tier = 18;         // trying costs one unit, failing more than that again: worth it below even odds
		}
		if (tier == 18) {                                     // the 42-cell tier (three segments a wave: 4/3 units) against the 64-cell one (2 units)
			const int bb = (2 * FILL_MID_CELLS - 2 - ad) / 2, mn = g.tlen < g.qlen ? g.tlen : g.qlen;
			const int g1 = B.gap_q + B.gap_e * (bb + 1), g2 = B.gap_q2 + B.gap_e2 * (bb + 1);
			const int U = B.sc_a * (mn - bb - 1) - 2 * (g1 < g2 ? g1 : g2);
			if ((B.debug_route & 16) || bb < 8 || U * 100 > mn * pred.mid) tier = 2;
		}
		if (tier == 2) {                                      // and the 64-lane tier likewise, against the literal kernel
			const int bb = (126 - ad) / 2, mn = g.tlen < g.qlen ? g.tlen : g.qlen;
			const int g1 = B.gap_q + B.gap_e * (bb + 1), g2 = B.gap_q2 + B.gap_e2 * (bb + 1);
			const int U = B.sc_a * (mn - bb - 1) - 2 * (g1 < g2 ? g1 : g2);
			if (U * 100 > mn * pred.wide) tier = 6;                   // two cells per lane: a band of 128
		}
		if (tier == 6) {
			const int bb = (254 - ad) / 2, mn = g.tlen < g.qlen ? g.tlen : g.qlen;
			const int g1 = B.gap_q + B.gap_e * (bb + 1), g2 = B.gap_q2 + B.gap_e2 * (bb + 1);
			const int U = B.sc_a * (mn - bb - 1) - 2 * (g1 < g2 ? g1 : g2);
			if (bb < 8 || U * 100 > mn * pred.last) tier = 0;
		}
		if (tier) g.big = 3 + tier;          // not for the literal kernel's first pass
		if (tier) widx = tier == 1 ? 0 : tier == 2 ? 1 : tier == 18 ? 4 : 2, wamt = (unsigned long long)(g.tlen + g.qlen - 1);
	} else if (!(B.debug_route & 1) && g.big == 0 && g.kind != 1 && g.tlen >= 1 && g.qlen >= 1 && g.tlen <= g.w && g.qlen <= g.w &&
	           g.tlen + g.qlen - 1 <= 2 * FILL_MAX_LEN && g.qlen <= 256) {
		// an extension whose matrix the band never clips, one cell per query base: the packed
		// extension kernel of k_fill.hip (tiers 8..15: by query length, right / left)
		const int tier = 8 + 2 * (g.qlen <= 32 ? 0 : g.qlen <= 64 ? 1 : g.qlen <= 128 ? 2 : 3) + ((g.flag & EZ_RIGHT) ? 1 : 0);
		g.big = 4 + tier;
		widx = 3, wamt = (unsigned long long)(g.tlen + g.qlen - 1) * g.qlen;
	} else if (g.big == 0 && g.kind != 1 && g.tlen >= 1 && g.qlen >= 1 && g.tlen <= g.w && g.qlen <= g.w &&
	           g.tlen + g.qlen - 1 <= 2 * FILL_MAX_LEN && (g.tlen < g.qlen ? g.tlen : g.qlen) <= 256) {
		// ... or with anti-diagonals that fit a wave (up to four cells per lane): the step-by-step one
		const int mn = g.tlen < g.qlen ? g.tlen : g.qlen;
		const int tier = mn <= 32 ? 3 : mn <= 64 ? 4 : mn <= 128 ? 7 : 8;
		g.big = 3 + tier;
	}
	if (!(B.debug_route & 8) && g.big == 0 && g.kind != 1 && g.tlen >= 1 && g.qlen >= 1 && g.tlen <= g.w && g.qlen <= g.w &&
	    g.tlen + g.qlen - 1 <= 2 * 1535 && (g.tlen < g.qlen ? g.tlen : g.qlen) <= 512) {
		g.big = 4 + 19;                           // a longer extension: the step-by-step kernel, eight cells per lane
	}
	if (g.big == 0) {
		// the literal kernel: its first pass with everything in LDS, or from the start on its own list
		const bool all_lds = 12 * T + Q <= B.lds0_state && p_bytes <= B.lds0_p && g.qlen + g.tlen + 2 <= B.lds0_cig;
		if (all_lds) g.big = 8;
		else g.big = 20;
	}
}

// work list `tier` (Seg.big - 4) of the alignment kernels: its counter in dp_ctr, its array
__device__ __forceinline__ int plan_list_ctr(int tier)
{
	return tier == 19 ? 62 : tier >= 17 ? 13 + tier : tier == 16 ? 28 : tier >= 8 ? 24 + tier : tier < 2 ? 10 + tier : tier < 4 ? 14 + tier : tier == 4 ? 20 : tier == 5 ? 22 : 18 + tier;   // 6 -> 24, 7 -> 25; 8.. -> 32..
}
__device__ __forceinline__ int32_t *plan_list(const Batch &B, int tier)
{
	return tier == 19 ? B.lext_list : tier == 18 ? B.lfill_list : tier == 17 ? B.fill_list_mid : tier == 16 ? B.mid_list : tier >= 8 ? B.extp_list + (int64_t)(tier - 8) * B.seg_cap
	     : tier == 0 ? B.fill_list1 : tier == 1 ? B.fill_list2 : tier == 2 ? B.ext_list1 : tier == 3 ? B.ext_list2 : tier == 4 ? B.gen_list
	     : tier == 5 ? B.fill_list3 : tier == 6 ? B.ext_list3 : B.ext_list4;
}

__global__ __launch_bounds__(64) void mnc_dp_plan(Batch B, const int32_t *work_list, int long_from, int slot_state_max, long long slot_p_max, int slot_cig_max,
                                                  long long big_state_max, long long big_p_max, long long big_cig_max,
                                                  long long huge_state_max, long long huge_p_max, long long huge_cig_max)
{
	const unsigned long long n_work = B.dp_ctr[9];
	const unsigned long long wi = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (wi >= n_work) return;
	const int64_t rslot = work_list[wi];
	mnc_reg_t r = B.regs[rslot];
	RegDP d = B.regdp[rslot];
	const uint32_t rd = (uint32_t)d.read;
	const int64_t a_off = B.an_off[rd];
	const int qlen = (int)(B.offsets[rd + 1] - B.offsets[rd]);
	Anchor *a = B.ca + a_off;
	const int n_a = B.ca_cnt[rd];
	if (long_from > 0 && n_a >= long_from) {                // a long read's region: mnc_dp_plan_long, unless its list is full
		const unsigned long long li = atomicAdd(&B.dp_ctr[53], 1ULL);
		if (li < (unsigned long long)B.plan_long_cap) { B.plan_long_list[li] = (int32_t)rslot; return; }
	}
	int32_t *K = B.t + a_off + r.as;                       // scratch: one int per anchor of the region
	d.n_seg = 0, d.first_seg = 0, d.has_left = d.has_right = 0;
	// an inversion region (mnc_dp_inv): no seeds -- one extension from the start the local alignment found, on the strand
	// and with the window that kernel left in the region's record
	const bool inv = (r.flags & REG_INV) != 0;
	if (r.cnt == 0 && !inv) { d.state = 2; B.regdp[rslot] = d; return; }
	const int k2 = KMER >> 1;
	const int rid = inv ? d.rid : (int32_t)(a[r.as].x << 1 >> 33), rev = inv ? d.rev : (int32_t)(a[r.as].x >> 63);
	d.rid = rid, d.rev = rev, d.qlen = qlen, d.pad_ = 0, d.coff = B.seq_off[rid], d.read_off = B.offsets[rd];
	const int ref_len = (int)(B.seq_off[rid + 1] - B.seq_off[rid]);
	const int bw = inv ? (int)(B.bw * 1.5) : (int)(B.bw * 1.5 + 1.);   // mm_align1_inv passes (int)(bw * 1.5), mm_align1 one more
	int as1 = r.as, cnt1 = r.cnt;
	int rs = 0, qs = 0, re = 0, qe = 0, rs0 = 0, qs0 = 0, re0 = 0, qe0 = 0;
	Anchor *b = a;
	if (inv) {
		rs = re = d.rs, qs = qe = d.qs, rs0 = rs, qs0 = qs, re0 = d.re0, qe0 = d.qe0;
		as1 = 0, cnt1 = 0;
	} else {
	// ---- mm_fix_bad_ends
	if (r.cnt >= 3) {
		const int min_match = B.min_sc * 2;
		// (eight anchors a turn, loaded before any is looked at: anchor by anchor each of the ~40 steps from either end was a
		// round trip to memory of its own -- more of them than all the other passes of this kernel together)
		int m, l;
		{
			uint64_t px = a[r.as].x, py = a[r.as].y;            // the anchor before
			m = l = (int)(py >> 32 & 0xff);
			const int i_end = r.as + r.cnt - 1;                   // (the last anchor is not visited)
			bool go = true;
			for (int i0 = r.as + 1; i0 < i_end && go; i0 += 8) {
				uint64_t bx[8], by[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) { const int ii = i0 + u < i_end ? i0 + u : i_end - 1; bx[u] = a[ii].x, by[u] = a[ii].y; }
#pragma unroll
				for (int u = 0; u < 8; ++u) {
					const int i = i0 + u;
					if (go && i < i_end) {
						const int q_span = (int)(by[u] >> 32 & 0xff);
						if (by[u] & SEED_LONG_JOIN) go = false;
						else {
							const int lr = (int32_t)bx[u] - (int32_t)px, lq = (int32_t)by[u] - (int32_t)py;
							const int mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
							if (mx - mn > l >> 1) as1 = i;
							l += mn;
							m += mn < q_span ? mn : q_span;
							px = bx[u], py = by[u];
							if (l >= B.bw << 1 || (m >= min_match && m >= B.bw) || m >= r.mlen >> 1) go = false;
						}
					}
				}
			}
		}
		cnt1 = r.as + r.cnt - as1;
		{
			uint64_t nx = a[r.as + r.cnt - 1].x, ny = a[r.as + r.cnt - 1].y;   // the anchor after
			m = l = (int)(ny >> 32 & 0xff);
			bool go = true;
			for (int i0 = r.as + r.cnt - 2; i0 > as1 && go; i0 -= 8) {
				uint64_t bx[8], by[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) { const int ii = i0 - u > as1 ? i0 - u : as1 + 1; bx[u] = a[ii].x, by[u] = a[ii].y; }
#pragma unroll
				for (int u = 0; u < 8; ++u) {
					const int i = i0 - u;
					if (go && i > as1) {
						const int q_span = (int)(ny >> 32 & 0xff);
						if (ny & SEED_LONG_JOIN) go = false;
						else {
							const int lr = (int32_t)nx - (int32_t)bx[u], lq = (int32_t)ny - (int32_t)by[u];
							const int mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
							if (mx - mn > l >> 1) cnt1 = i + 1 - as1;
							l += mn;
							m += mn < q_span ? mn : q_span;
							nx = bx[u], ny = by[u];
							if (l >= B.bw << 1 || (m >= min_match && m >= B.bw) || m >= r.mlen >> 1) go = false;
						}
					}
				}
			}
		}
	}
	b = a + as1;
	// (a region whose consecutive anchors never step more than ten bases apart -- mnc_dp_gather saw it while it copied them --
	// has nothing for either filter: both begin by collecting such steps)
	const bool nogap10 = (a[r.as].y & SEED_NOGAP10) != 0;
	// ---- mm_filter_bad_seeds(as1, cnt1, a, 10, 40, max_gap >> 1, 10)
	if (!nogap10) {
		const int n = collect_long_gaps(b, cnt1, 10, K);
		if (n > 0) {
			const int diff_thres = 40, max_ext_len = B.max_gap >> 1, max_ext_cnt = 10;
			int mx = 0, max_st = -1, max_en = -1;
			for (int k = 0;; ++k) {
				int gap, l, n_ins = 0, n_del = 0, max_diff = 0, max_diff_l = -1;
				if (k == n || k >= max_en) {
					if (max_en > 0)
						for (int i = K[max_st]; i < K[max_en]; ++i) b[i].y |= SEED_IGNORE;
					mx = 0, max_st = max_en = -1;
					if (k == n) break;
				}
				const int i = K[k];
				gap = seed_gap(b, i);
				if (gap > 0) n_ins += gap; else n_del += -gap;
				const int qs_ = (int32_t)b[i - 1].y, rs_ = (int32_t)b[i - 1].x;
				for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
					const int j = K[l];
					if ((int32_t)b[j].y - qs_ > max_ext_len || (int32_t)b[j].x - rs_ > max_ext_len) break;
					gap = seed_gap(b, j);
					if (gap > 0) n_ins += gap; else n_del += -gap;
					const int diff = n_ins + n_del - abs(n_ins - n_del);
					if (max_diff < diff) max_diff = diff, max_diff_l = l;
				}
				if (max_diff > diff_thres && max_diff > mx) mx = max_diff, max_st = k, max_en = max_diff_l;
			}
		}
	}
	// ---- mm_filter_bad_seeds_alt(as1, cnt1, a, 30, max_gap >> 1)
	if (!nogap10) {
		const int n = collect_long_gaps(b, cnt1, 30, K);
		const int max_ext = B.max_gap >> 1;
		for (int k = 0; k < n;) {
			const int i = K[k];
			int l, gap1 = seed_gap(b, i), re1 = (int32_t)b[i].x, qe1 = (int32_t)b[i].y;
			gap1 = gap1 > 0 ? gap1 : -gap1;
			for (l = k + 1; l < n; ++l) {
				const int j = K[l];
				if ((int32_t)b[j].y - qe1 > max_ext || (int32_t)b[j].x - re1 > max_ext) break;
				int gap2 = seed_gap(b, j);
				const int q_span_pre = (int)(b[j - 1].y >> 32 & 0xff);
				const int rs2 = (int32_t)b[j - 1].x + q_span_pre, qs2 = (int32_t)b[j - 1].y + q_span_pre;
				const int m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
				gap2 = gap2 > 0 ? gap2 : -gap2;
				if (m > gap1 + gap2) break;
				re1 = (int32_t)b[j].x, qe1 = (int32_t)b[j].y, gap1 = gap2;
			}
			if (l > k + 1) {
				const int end = K[l - 1];
				for (int j = K[k]; j < end; ++j) b[j].y |= SEED_IGNORE;
				b[end].y |= SEED_LONG_JOIN;
			}
			k = l;
		}
	}
	// ---- DP window
	rs = (int32_t)b[0].x - k2, qs = (int32_t)b[0].y - k2;
	re = (int32_t)b[cnt1 - 1].x - k2, qe = (int32_t)b[cnt1 - 1].y - k2;
	rs0 = (int32_t)a[r.as].x + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
	qs0 = (int32_t)a[r.as].y + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
	if (rs0 < 0) rs0 = 0;
	int rs1 = 0, qs1 = 0, l;
	for (int i = r.as - 1, c = 0; i >= 0 && a[i].x >> 32 == a[r.as].x >> 32; --i) {
		const int x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		const int y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		if (x < rs0 && y < qs0) {
			if (++c > B.min_cnt) {
				l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
				rs1 = rs0 - l, qs1 = qs0 - l;
				if (rs1 < 0) rs1 = 0;
				break;
			}
		}
	}
	if (qs > 0 && rs > 0) {
		l = qs < B.max_gap ? qs : B.max_gap;
		qs1 = qs1 > qs - l ? qs1 : qs - l;
		qs0 = qs0 < qs1 ? qs0 : qs1;
		l += l * B.sc_a > B.gap_q ? (l * B.sc_a - B.gap_q) / B.gap_e : 0;
		l = l < B.max_gap ? l : B.max_gap;
		l = l < rs ? l : rs;
		rs1 = rs1 > rs - l ? rs1 : rs - l;
		rs0 = rs0 < rs1 ? rs0 : rs1;
		rs0 = rs0 < rs ? rs0 : rs;
	} else rs0 = rs, qs0 = qs;
	re0 = (int32_t)a[r.as + r.cnt - 1].x + 1, qe0 = (int32_t)a[r.as + r.cnt - 1].y + 1;
	int re1 = ref_len, qe1 = qlen;
	for (int i = r.as + r.cnt, c = 0; i < n_a && a[i].x >> 32 == a[r.as].x >> 32; ++i) {
		const int x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		const int y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
		if (x > re0 && y > qe0) {
			if (++c > B.min_cnt) {
				l = x - re0 > y - qe0 ? x - re0 : y - qe0;
				re1 = re0 + l, qe1 = qe0 + l;
				break;
			}
		}
	}
	if (qe < qlen && re < ref_len) {
		l = qlen - qe < B.max_gap ? qlen - qe : B.max_gap;
		qe1 = qe1 < qe + l ? qe1 : qe + l;
		qe0 = qe0 > qe1 ? qe0 : qe1;
		l += l * B.sc_a > B.gap_q ? (l * B.sc_a - B.gap_q) / B.gap_e : 0;
		l = l < B.max_gap ? l : B.max_gap;
		l = l < ref_len - re ? l : ref_len - re;
		re1 = re1 < re + l ? re1 : re + l;
		re0 = re0 > re1 ? re0 : re1;
	} else re0 = re, qe0 = qe;
	}   // !inv
	d.as1 = as1, d.cnt1 = cnt1, d.rs = rs, d.qs = qs, d.re = re, d.qe = qe;
	d.rs0 = rs0, d.qs0 = qs0, d.re0 = re0, d.qe0 = qe0;

	// ---- segments: count, allocate, fill
	const bool left = !inv && qs > 0 && rs > 0;
	int n_fill = 0;
	// the seeds the gap fillings end at: counted here (the segments' places are reserved before they are written), and kept from
	// the END of the region's scratch downwards, so that the pass that writes the records loads those twenty seeds instead of
	// walking all anchors again (the front of the scratch takes the records' tier codes below)
	int32_t *KI = K + (r.cnt > 0 ? r.cnt - 1 : 0);
	{
		int prs = rs, pqs = qs;
		for (int i0 = 1; i0 < cnt1; i0 += 8) {
			uint64_t bx[8], by[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) { const int ii = i0 + u < cnt1 ? i0 + u : cnt1 - 1; bx[u] = b[ii].x, by[u] = b[ii].y; }
#pragma unroll
			for (int u = 0; u < 8; ++u) {
				const int i = i0 + u;
				if (i >= cnt1) break;
				if ((by[u] & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
				const int cre = (int32_t)bx[u] - k2, cqe = (int32_t)by[u] - k2;
				if (i == cnt1 - 1 || (by[u] & SEED_LONG_JOIN) || (cqe - pqs >= B.min_ksw_len && cre - prs >= B.min_ksw_len)) {
					if (n_fill < r.cnt) KI[-n_fill] = i;
					++n_fill, prs = cre, pqs = cqe;
				}
			}
		}
	}
	// the last gap filling ends at the last seed: the right extension starts from (re, qe) above
	const bool right = qe < qe0 && re < re0;
	const int n_seg = (left ? 1 : 0) + n_fill + (right ? 1 : 0);
	d.n_seg = n_seg, d.has_left = left, d.has_right = right, d.state = 1;
	if (n_seg > 0) {
		const unsigned long long s0 = atomicAdd(&B.dp_ctr[0], (unsigned long long)n_seg);
		d.first_seg = (int32_t)s0;
		if ((long long)(s0 + n_seg) > B.seg_cap) {
			atomicMax(&B.dp_ctr[4], 1ULL);                          // overflow: the batch is redone with more room
			d.n_seg = 0, d.has_left = d.has_right = 0;
		} else {
			Seg *sg = B.segs + s0;
			const bool k_ok = n_seg <= r.cnt;                        // the region's scratch: one int per anchor
			int n_tier[20] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
			unsigned long long work[5] = { 0, 0, 0, 0, 0 };          // anti-diagonals given to the banded tiers (32 / 64 / 128 cells; [4]: 42 cells); [3]: steps x cells of the packed extensions
			const PlanLimits lim = { slot_state_max, slot_p_max, slot_cig_max, big_state_max, big_p_max, big_cig_max, huge_state_max, huge_p_max, huge_cig_max };
			const FillPred pred = fill_pred_of(B, cnt1, qe - qs);
			auto emit = [&](Seg g) {
				int widx;
				unsigned long long wamt;
				plan_seg_class(B, lim, pred, bw, rd, rslot, rid, rev, (int32_t)(sg - B.segs), g, widx, wamt);
				if (g.big >= 4) ++n_tier[g.big - 4];
				if (widx >= 0) work[widx] += wamt;
				if (k_ok) K[sg - (B.segs + s0)] = g.big;                // for the lists below (the scratch ints are free again)
				*sg++ = g;
			};
			if (left) {
				Seg g;
				g.kind = 0, g.ts = rs0, g.tlen = rs - rs0, g.qs = qs0, g.qlen = qs - qs0, g.w = bw;
				g.zdrop = (r.flags & REG_SPLIT_INV) ? B.zdrop_inv : B.zdrop;
				g.flag = EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR, g.ai = 0;
				emit(g);
			}
			int prs = rs, pqs = qs;
			if (n_seg + n_fill <= r.cnt) {
				// the end seeds from the scratch's end (the tier codes at its front stay clear of them), eight at a time
				for (int k0 = 0; k0 < n_fill; k0 += 8) {
					int32_t ix[8];
					uint64_t bx[8], by[8];
#pragma unroll
					for (int u = 0; u < 8; ++u) ix[u] = KI[-(k0 + u < n_fill ? k0 + u : n_fill - 1)];
#pragma unroll
					for (int u = 0; u < 8; ++u) bx[u] = b[ix[u]].x, by[u] = b[ix[u]].y;
					for (int u = 0; u < 8 && k0 + u < n_fill; ++u) {
						int i = ix[0];
						uint64_t vx = bx[0], vy = by[0];
#pragma unroll
						for (int w8 = 1; w8 < 8; ++w8) if (u == w8) i = ix[w8], vx = bx[w8], vy = by[w8];
						const int cre = (int32_t)vx - k2, cqe = (int32_t)vy - k2;
						Seg g;
						g.kind = 1, g.ts = prs, g.tlen = cre - prs, g.qs = pqs, g.qlen = cqe - pqs;
						g.w = (vy & SEED_LONG_JOIN) ? (cqe - pqs > cre - prs ? cqe - pqs : cre - prs) : bw;
						g.zdrop = B.zdrop, g.flag = EZ_APPROX_MAX, g.ai = i;
						emit(g);
						prs = cre, pqs = cqe;
					}
				}
			} else
			for (int i0 = 1; i0 < cnt1; i0 += 8) {
				uint64_t bx[8], by[8];
#pragma unroll
				for (int u = 0; u < 8; ++u) { const int ii = i0 + u < cnt1 ? i0 + u : cnt1 - 1; bx[u] = b[ii].x, by[u] = b[ii].y; }
				for (int u = 0; u < 8 && i0 + u < cnt1; ++u) {
					const int i = i0 + u;
					uint64_t vx = bx[0], vy = by[0];
#pragma unroll
					for (int w8 = 1; w8 < 8; ++w8) if (u == w8) vx = bx[w8], vy = by[w8];
					if ((vy & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
					const int cre = (int32_t)vx - k2, cqe = (int32_t)vy - k2;
					if (i == cnt1 - 1 || (vy & SEED_LONG_JOIN) || (cqe - pqs >= B.min_ksw_len && cre - prs >= B.min_ksw_len)) {
						Seg g;
						g.kind = 1, g.ts = prs, g.tlen = cre - prs, g.qs = pqs, g.qlen = cqe - pqs;
						g.w = (vy & SEED_LONG_JOIN) ? (cqe - pqs > cre - prs ? cqe - pqs : cre - prs) : bw;
						g.zdrop = B.zdrop, g.flag = EZ_APPROX_MAX, g.ai = i;
						emit(g);
						prs = cre, pqs = cqe;
					}
				}
			}
			if (right) {
				Seg g;
				g.kind = 2, g.ts = re, g.tlen = re0 - re, g.qs = qe, g.qlen = qe0 - qe, g.w = bw;
				g.zdrop = B.zdrop, g.flag = EZ_EXTZ_ONLY, g.ai = cnt1 - 1;
				emit(g);
			}
			for (int k = 0; k < 5; ++k) if (work[k]) atomicAdd(&B.dp_ctr[48 + k], work[k]);
			// the banded kernel's lists: one reservation per region and tier
			for (int tier = 0; tier < 20; ++tier) {
				if (n_tier[tier] == 0) continue;
				unsigned long long fi = atomicAdd(&B.dp_ctr[plan_list_ctr(tier)], (unsigned long long)n_tier[tier]);
				int32_t *lst = plan_list(B, tier);
				for (int k = 0; k < n_seg; ++k)
					if ((k_ok ? K[k] : B.segs[s0 + k].big) == 4 + tier) lst[fi++] = (int32_t)(s0 + k);
			}
		}
	}
	B.regdp[rslot] = d;
}

// ================================================================ plan, regions of long reads
// The lane that plans a region reads its anchors eight at a time from memory, four passes: a 60 kb read's region has
// 4 700 anchors -- 2 400 round trips, 4 ms that a batch with such a read waits for.  Regions of reads with `long_from`
// chained anchors or more are therefore planned by a WAVE each: the read's squeezed anchors in LDS; the passes over all
// anchors as wave-wide steps (the two collections of long gaps as ballot compactions; the choice of the gap fillings'
// end seeds 64 candidates at a time; the segments' records one per lane); the short sequential pieces -- mm_fix_bad_ends,
// the walks over the collected long gaps, the DP window -- on lane 0, out of LDS.  Same results as mnc_dp_plan, which
// tests/test_gpu_dp.py::test_long_reads_and_mixed_lengths and debug bit 0x800000 (everything on the lane form) hold it to.
__device__ int plan_long_gaps_wave(const Anchor *b, int cnt1, int min_gap, int32_t *K, int lane)
{
	int n = 0;
	for (int i0 = 1; i0 < cnt1; i0 += 64) {
		const int i = i0 + lane;
		bool hit = false;
		if (i < cnt1) { const int gap = seed_gap(b, i); hit = gap < -min_gap || gap > min_gap; }
		const unsigned long long m = __ballot(hit);
		if (hit) K[n + __popcll(m & ((1ULL << lane) - 1ULL))] = i;
		n += __popcll(m);
	}
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	__builtin_amdgcn_wave_barrier();
	return n <= 1 ? 0 : n;
}

__global__ __launch_bounds__(64) void mnc_dp_plan_long(Batch B, int lds_anchors, int n_a_from, int n_a_below, PlanLimits lim)
{
	extern __shared__ __align__(16) uint8_t pl_smem[];       // [anchors | one int per anchor]
	const int lane = threadIdx.x;
	const unsigned long long lt = (1ULL << lane) - 1ULL;
	auto sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); };
	auto bc = [&](int v) { return __builtin_amdgcn_readfirstlane(v); };
	// the regions mnc_dp_plan left in the list (it ran before this kernel on the same stream), dealt out one at a time:
	// neighbours in the list are regions of neighbouring reads, about as long as each other
	{
		const unsigned long long listed = B.dp_ctr[53];
		const unsigned long long n_long = listed < (unsigned long long)B.plan_long_cap ? listed : (unsigned long long)B.plan_long_cap;
		for (unsigned long long wi = blockIdx.x; wi < n_long; wi += gridDim.x) {
			const int64_t rslot = B.plan_long_list[wi];
			mnc_reg_t r = B.regs[rslot];
			RegDP d = B.regdp[rslot];
			const uint32_t rd = (uint32_t)d.read;
			const int64_t a_off = B.an_off[rd];
			const int qlen = (int)(B.offsets[rd + 1] - B.offsets[rd]);
			const int n_a = B.ca_cnt[rd];
			if (n_a < n_a_from || n_a >= n_a_below) continue;      // the launch with the other LDS size takes it
			d.n_seg = 0, d.first_seg = 0, d.has_left = d.has_right = 0;
			const bool inv = (r.flags & REG_INV) != 0;
			if (r.cnt == 0 && !inv) { d.state = 2; if (lane == 0) B.regdp[rslot] = d; continue; }
			const bool staged = r.cnt > 0 && n_a <= lds_anchors;
			Anchor *a = B.ca + a_off;
			int32_t *K = B.t + a_off + r.as;                   // scratch: one int per anchor of the region
			if (staged) {
				Anchor *s_a = reinterpret_cast<Anchor*>(pl_smem);
				for (int i = lane; i < n_a; i += 64) s_a[i] = a[i];
				a = s_a, K = reinterpret_cast<int32_t*>(s_a + lds_anchors);
			}
			sync();
			const int k2 = KMER >> 1;
			const int rid = inv ? d.rid : (int32_t)(a[r.as].x << 1 >> 33), rev = inv ? d.rev : (int32_t)(a[r.as].x >> 63);
			d.rid = rid, d.rev = rev, d.qlen = qlen, d.pad_ = 0, d.coff = B.seq_off[rid], d.read_off = B.offsets[rd];
			const int ref_len = (int)(B.seq_off[rid + 1] - B.seq_off[rid]);
			const int bw = inv ? (int)(B.bw * 1.5) : (int)(B.bw * 1.5 + 1.);
			int as1 = r.as, cnt1 = r.cnt;
			int rs = 0, qs = 0, re = 0, qe = 0, rs0 = 0, qs0 = 0, re0 = 0, qe0 = 0;
			Anchor *b = a;
			if (inv) {
				rs = re = d.rs, qs = qe = d.qs, rs0 = rs, qs0 = qs, re0 = d.re0, qe0 = d.qe0;
				as1 = 0, cnt1 = 0;
			} else {
				// ---- mm_fix_bad_ends (lane 0: it stops after 2 bw bases from either end)
				if (lane == 0 && r.cnt >= 3) {
					const int min_match = B.min_sc * 2;
					int m, l;
					m = l = (int)(a[r.as].y >> 32 & 0xff);
					for (int i = r.as + 1; i < r.as + r.cnt - 1; ++i) {
						const int q_span = (int)(a[i].y >> 32 & 0xff);
						if (a[i].y & SEED_LONG_JOIN) break;
						const int lr = (int32_t)a[i].x - (int32_t)a[i - 1].x, lq = (int32_t)a[i].y - (int32_t)a[i - 1].y;
						const int mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
						if (mx - mn > l >> 1) as1 = i;
						l += mn;
						m += mn < q_span ? mn : q_span;
						if (l >= B.bw << 1 || (m >= min_match && m >= B.bw) || m >= r.mlen >> 1) break;
					}
					cnt1 = r.as + r.cnt - as1;
					m = l = (int)(a[r.as + r.cnt - 1].y >> 32 & 0xff);
					for (int i = r.as + r.cnt - 2; i > as1; --i) {
						const int q_span = (int)(a[i + 1].y >> 32 & 0xff);
						if (a[i + 1].y & SEED_LONG_JOIN) break;
						const int lr = (int32_t)a[i + 1].x - (int32_t)a[i].x, lq = (int32_t)a[i + 1].y - (int32_t)a[i].y;
						const int mn = lr < lq ? lr : lq, mx = lr > lq ? lr : lq;
						if (mx - mn > l >> 1) cnt1 = i + 1 - as1;
						l += mn;
						m += mn < q_span ? mn : q_span;
						if (l >= B.bw << 1 || (m >= min_match && m >= B.bw) || m >= r.mlen >> 1) break;
					}
				}
				as1 = bc(as1), cnt1 = bc(cnt1);
				b = a + as1;
				// ---- mm_filter_bad_seeds(as1, cnt1, a, 10, 40, max_gap >> 1, 10): the long gaps by all lanes, the walk over them on lane 0
				{
					const int n = plan_long_gaps_wave(b, cnt1, 10, K, lane);
					if (lane == 0 && n > 0) {
						const int diff_thres = 40, max_ext_len = B.max_gap >> 1, max_ext_cnt = 10;
						int mx = 0, max_st = -1, max_en = -1;
						for (int k = 0;; ++k) {
							int gap, l, n_ins = 0, n_del = 0, max_diff = 0, max_diff_l = -1;
							if (k == n || k >= max_en) {
								if (max_en > 0)
									for (int i = K[max_st]; i < K[max_en]; ++i) b[i].y |= SEED_IGNORE;
								mx = 0, max_st = max_en = -1;
								if (k == n) break;
							}
							const int i = K[k];
							gap = seed_gap(b, i);
							if (gap > 0) n_ins += gap; else n_del += -gap;
							const int qs_ = (int32_t)b[i - 1].y, rs_ = (int32_t)b[i - 1].x;
							for (l = k + 1; l < n && l <= k + max_ext_cnt; ++l) {
								const int j = K[l];
								if ((int32_t)b[j].y - qs_ > max_ext_len || (int32_t)b[j].x - rs_ > max_ext_len) break;
								gap = seed_gap(b, j);
								if (gap > 0) n_ins += gap; else n_del += -gap;
								const int diff = n_ins + n_del - abs(n_ins - n_del);
								if (max_diff < diff) max_diff = diff, max_diff_l = l;
							}
							if (max_diff > diff_thres && max_diff > mx) mx = max_diff, max_st = k, max_en = max_diff_l;
						}
					}
					sync();
				}
				// ---- mm_filter_bad_seeds_alt(as1, cnt1, a, 30, max_gap >> 1)
				{
					const int n = plan_long_gaps_wave(b, cnt1, 30, K, lane);
					if (lane == 0) {
						const int max_ext = B.max_gap >> 1;
						for (int k = 0; k < n;) {
							const int i = K[k];
							int l, gap1 = seed_gap(b, i), re1 = (int32_t)b[i].x, qe1 = (int32_t)b[i].y;
							gap1 = gap1 > 0 ? gap1 : -gap1;
							for (l = k + 1; l < n; ++l) {
								const int j = K[l];
								if ((int32_t)b[j].y - qe1 > max_ext || (int32_t)b[j].x - re1 > max_ext) break;
								int gap2 = seed_gap(b, j);
								const int q_span_pre = (int)(b[j - 1].y >> 32 & 0xff);
								const int rs2 = (int32_t)b[j - 1].x + q_span_pre, qs2 = (int32_t)b[j - 1].y + q_span_pre;
								const int m = rs2 - re1 < qs2 - qe1 ? rs2 - re1 : qs2 - qe1;
								gap2 = gap2 > 0 ? gap2 : -gap2;
								if (m > gap1 + gap2) break;
								re1 = (int32_t)b[j].x, qe1 = (int32_t)b[j].y, gap1 = gap2;
							}
							if (l > k + 1) {
								const int end = K[l - 1];
								for (int j = K[k]; j < end; ++j) b[j].y |= SEED_IGNORE;
								b[end].y |= SEED_LONG_JOIN;
							}
							k = l;
						}
					}
					sync();
				}
				// ---- DP window (lane 0: it looks at a few neighbours of the region)
				if (lane == 0) {
					rs = (int32_t)b[0].x - k2, qs = (int32_t)b[0].y - k2;
					re = (int32_t)b[cnt1 - 1].x - k2, qe = (int32_t)b[cnt1 - 1].y - k2;
					rs0 = (int32_t)a[r.as].x + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
					qs0 = (int32_t)a[r.as].y + 1 - (int32_t)(a[r.as].y >> 32 & 0xff);
					if (rs0 < 0) rs0 = 0;
					int rs1 = 0, qs1 = 0, l;
					for (int i = r.as - 1, c = 0; i >= 0 && a[i].x >> 32 == a[r.as].x >> 32; --i) {
						const int x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
						const int y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
						if (x < rs0 && y < qs0) {
							if (++c > B.min_cnt) {
								l = rs0 - x > qs0 - y ? rs0 - x : qs0 - y;
								rs1 = rs0 - l, qs1 = qs0 - l;
								if (rs1 < 0) rs1 = 0;
								break;
							}
						}
					}
					if (qs > 0 && rs > 0) {
						l = qs < B.max_gap ? qs : B.max_gap;
						qs1 = qs1 > qs - l ? qs1 : qs - l;
						qs0 = qs0 < qs1 ? qs0 : qs1;
						l += l * B.sc_a > B.gap_q ? (l * B.sc_a - B.gap_q) / B.gap_e : 0;
						l = l < B.max_gap ? l : B.max_gap;
						l = l < rs ? l : rs;
						rs1 = rs1 > rs - l ? rs1 : rs - l;
						rs0 = rs0 < rs1 ? rs0 : rs1;
						rs0 = rs0 < rs ? rs0 : rs;
					} else rs0 = rs, qs0 = qs;
					re0 = (int32_t)a[r.as + r.cnt - 1].x + 1, qe0 = (int32_t)a[r.as + r.cnt - 1].y + 1;
					int re1 = ref_len, qe1 = qlen;
					for (int i = r.as + r.cnt, c = 0; i < n_a && a[i].x >> 32 == a[r.as].x >> 32; ++i) {
						const int x = (int32_t)a[i].x + 1 - (int32_t)(a[i].y >> 32 & 0xff);
						const int y = (int32_t)a[i].y + 1 - (int32_t)(a[i].y >> 32 & 0xff);
						if (x > re0 && y > qe0) {
							if (++c > B.min_cnt) {
								l = x - re0 > y - qe0 ? x - re0 : y - qe0;
								re1 = re0 + l, qe1 = qe0 + l;
								break;
							}
						}
					}
					if (qe < qlen && re < ref_len) {
						l = qlen - qe < B.max_gap ? qlen - qe : B.max_gap;
						qe1 = qe1 < qe + l ? qe1 : qe + l;
						qe0 = qe0 > qe1 ? qe0 : qe1;
						l += l * B.sc_a > B.gap_q ? (l * B.sc_a - B.gap_q) / B.gap_e : 0;
						l = l < B.max_gap ? l : B.max_gap;
						l = l < ref_len - re ? l : ref_len - re;
						re1 = re1 < re + l ? re1 : re + l;
						re0 = re0 > re1 ? re0 : re1;
					} else re0 = re, qe0 = qe;
				}
				rs = bc(rs), qs = bc(qs), re = bc(re), qe = bc(qe), rs0 = bc(rs0), qs0 = bc(qs0), re0 = bc(re0), qe0 = bc(qe0);
			}   // !inv
			d.as1 = as1, d.cnt1 = cnt1, d.rs = rs, d.qs = qs, d.re = re, d.qe = qe;
			d.rs0 = rs0, d.qs0 = qs0, d.re0 = re0, d.qe0 = qe0;

			// ---- the seeds the gap fillings end at: the next seed (not ignored, not tandem) at least min_ksw_len past the
			// last one on both sequences, or a long-join seed, or the last seed -- 64 candidates at a time
			const bool left = !inv && qs > 0 && rs > 0;
			int n_fill = 0;
			{
				int pos = 0, prs = rs, pqs = qs;
				while (pos < cnt1 - 1) {
					const int i = pos + 1 + lane;
					bool elig = false;
					int cre = 0, cqe = 0;
					if (i < cnt1) {
						const uint64_t vx = b[i].x, vy = b[i].y;
						cre = (int32_t)vx - k2, cqe = (int32_t)vy - k2;
						const bool skip = (vy & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1;
						elig = !skip && (i == cnt1 - 1 || (vy & SEED_LONG_JOIN) || (cqe - pqs >= B.min_ksw_len && cre - prs >= B.min_ksw_len));
					}
					const unsigned long long m = __ballot(elig);
					if (!m) { pos += 64; continue; }
					const int l = __ffsll((long long)m) - 1;
					pos += 1 + l, prs = __shfl(cre, l), pqs = __shfl(cqe, l);
					if (lane == 0) K[n_fill] = pos;                    // (the long gaps' list is done with)
					++n_fill;
				}
				sync();
			}
			const bool right = qe < qe0 && re < re0;
			const int n_seg = (left ? 1 : 0) + n_fill + (right ? 1 : 0);
			d.n_seg = n_seg, d.has_left = left, d.has_right = right, d.state = 1;
			if (n_seg > 0) {
				unsigned long long s0 = 0;
				if (lane == 0) s0 = atomicAdd(&B.dp_ctr[0], (unsigned long long)n_seg);
				s0 = (unsigned long long)__shfl((long long)s0, 0);
				d.first_seg = (int32_t)s0;
				if ((long long)(s0 + n_seg) > B.seg_cap) {
					if (lane == 0) atomicMax(&B.dp_ctr[4], 1ULL);       // overflow: the batch is redone with more room
					d.n_seg = 0, d.has_left = d.has_right = 0;
				} else {
					// one segment per lane, 64 at a time
					const FillPred pred = fill_pred_of(B, cnt1, qe - qs);
					for (int k0 = 0; k0 < n_seg; k0 += 64) {
						const int k = k0 + lane;
						const bool in = k < n_seg;
						Seg g;
						int widx = -1;
						unsigned long long wamt = 0;
						if (in) {
							if (left && k == 0) {
								g.kind = 0, g.ts = rs0, g.tlen = rs - rs0, g.qs = qs0, g.qlen = qs - qs0, g.w = bw;
								g.zdrop = (r.flags & REG_SPLIT_INV) ? B.zdrop_inv : B.zdrop;
								g.flag = EZ_EXTZ_ONLY | EZ_RIGHT | EZ_REV_CIGAR, g.ai = 0;
							} else if (right && k == n_seg - 1) {
								g.kind = 2, g.ts = re, g.tlen = re0 - re, g.qs = qe, g.qlen = qe0 - qe, g.w = bw;
								g.zdrop = B.zdrop, g.flag = EZ_EXTZ_ONLY, g.ai = cnt1 - 1;
							} else {
								const int f = k - (left ? 1 : 0), i = K[f];
								const uint64_t vx = b[i].x, vy = b[i].y;
								int prs = rs, pqs = qs;
								if (f > 0) { const int ip = K[f - 1]; prs = (int32_t)b[ip].x - k2, pqs = (int32_t)b[ip].y - k2; }
								const int cre = (int32_t)vx - k2, cqe = (int32_t)vy - k2;
								g.kind = 1, g.ts = prs, g.tlen = cre - prs, g.qs = pqs, g.qlen = cqe - pqs;
								g.w = (vy & SEED_LONG_JOIN) ? (cqe - pqs > cre - prs ? cqe - pqs : cre - prs) : bw;
								g.zdrop = B.zdrop, g.flag = EZ_APPROX_MAX, g.ai = i;
							}
							plan_seg_class(B, lim, pred, bw, rd, rslot, rid, rev, (int32_t)(s0 + k), g, widx, wamt);
							B.segs[s0 + k] = g;
						}
						// the work counters and the kernels' lists: one reservation per tier and 64 segments
						for (int w5 = 0; w5 < 5; ++w5) {
							unsigned long long v = in && widx == w5 ? wamt : 0ULL;
#pragma unroll
							for (int sft = 32; sft > 0; sft >>= 1) v += (unsigned long long)__shfl_xor((long long)v, sft);
							if (lane == 0 && v) atomicAdd(&B.dp_ctr[48 + w5], v);
						}
						const int tier = in && g.big >= 4 ? g.big - 4 : -1;
						unsigned long long left_t = __ballot(tier >= 0);
						while (left_t) {
							const int t = __shfl(tier, __ffsll((long long)left_t) - 1);
							const unsigned long long mt = __ballot(tier == t);
							left_t &= ~mt;
							unsigned long long fi = 0;
							if (lane == 0) fi = atomicAdd(&B.dp_ctr[plan_list_ctr(t)], (unsigned long long)__popcll(mt));
							fi = (unsigned long long)__shfl((long long)fi, 0);
							if (tier == t) plan_list(B, t)[fi + __popcll(mt & lt)] = (int32_t)(s0 + k);
						}
					}
				}
			}
			if (lane == 0) B.regdp[rslot] = d;
			sync();
			// the seed filters' flags (SEED_IGNORE, SEED_LONG_JOIN) live in the anchors' y words: the stitch kernel reads them
			if (staged) for (int i = lane; i < r.cnt; i += 64) B.ca[a_off + r.as + i].y = a[r.as + i].y;
			sync();
		}
	}
}

// ================================================================ align: ksw_extd2 on one wave
struct Ez {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, score, reach_end, n_cigar;
};

__device__ __forceinline__ bool apply_zdrop(Ez &ez, int H, int r, int t, int zdrop, int e)
{
	if (H > ez.max) {
		ez.max = H, ez.max_t = t, ez.max_q = r - t;
	} else if (t >= ez.max_t && r - t >= ez.max_q) {
		const int tl = t - ez.max_t, ql = (r - t) - ez.max_q;
		const int l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && ez.max - H > zdrop + l * e) { ez.zdropped = 1; return true; }
	}
	return false;
}

__device__ __forceinline__ int wave_max_i32(int v)
{
	for (int d = 32; d > 0; d >>= 1) { const int o = __shfl_xor(v, d); v = v > o ? v : o; }
	return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
	for (int d = 32; d > 0; d >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)v, d); v = v < o ? v : o; }
	return v;
}

#define I8(v) ((int)(int8_t)(v))

// LDS pointers keep their address space in the type, so that the compiler emits ds_* instructions
// (a generic pointer that may be LDS or HBM costs a flat access: several times the latency)
typedef __attribute__((address_space(3))) int8_t *lds_i8p;
typedef __attribute__((address_space(3))) uint8_t *lds_u8p;
typedef __attribute__((address_space(3))) int32_t *lds_i32p;
typedef __attribute__((address_space(3))) uint32_t *lds_u32p;
template <class P> struct PtrTraits;
template <> struct PtrTraits<int8_t*> { typedef uint8_t *u8; static constexpr bool lds = false; };
template <> struct PtrTraits<lds_i8p> { typedef lds_u8p u8; static constexpr bool lds = true; };

// One call of the kernel on sequences already laid out in `mem` (sf = target codes, qr = the
// query reversed), exactly as ksw_extd2_sse works on its buffer.  `cig` receives the CIGAR the
// way ksw_backtrack pushes it (reversed unless EZ_REV_CIGAR asks for that order).
template <bool LDS, int NW = 1>
__device__ __forceinline__ void st_order()
{
	if (NW > 1) { __syncthreads(); return; }                  // several waves on one call: they meet (state in the workspace)
	// state in LDS: its operations execute in issue order within the wave, only the compiler has to
	// be pinned (the direction bytes stream to HBM without being waited for); state in HBM: wait
	if (LDS) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); asm volatile("" ::: "memory"); }
	else mem_order();
}

// reductions and a broadcast over the NW waves of a call (NW == 1: the wave's own)
template <int NW> __device__ __forceinline__ int wg_max_i32(int v)
{
	v = wave_max_i32(v);
	if (NW > 1) {
		__shared__ int s_red[16];
		__syncthreads();
		if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
		__syncthreads();
		v = s_red[0];
#pragma unroll
		for (int k = 1; k < NW; ++k) v = v > s_red[k] ? v : s_red[k];
	}
	return v;
}
template <int NW> __device__ __forceinline__ unsigned wg_min_u32(unsigned v)
{
	v = wave_min_u32(v);
	if (NW > 1) {
		__shared__ unsigned s_redu[16];
		__syncthreads();
		if ((threadIdx.x & 63) == 0) s_redu[threadIdx.x >> 6] = v;
		__syncthreads();
		v = s_redu[0];
#pragma unroll
		for (int k = 1; k < NW; ++k) v = v < s_redu[k] ? v : s_redu[k];
	}
	return v;
}
template <int NW> __device__ __forceinline__ unsigned long long wg_bcast0_u64(unsigned long long v)
{
	if (NW == 1) return (unsigned long long)__shfl((long long)v, 0);
	__shared__ unsigned long long s_bc64;
	__syncthreads();
	if (threadIdx.x == 0) s_bc64 = v;
	__syncthreads();
	return s_bc64;
}
template <int NW> __device__ __forceinline__ int wg_bcast0(int v)
{
	if (NW == 1) return __shfl(v, 0);
	__shared__ int s_bc;
	__syncthreads();
	if (threadIdx.x == 0) s_bc = v;
	__syncthreads();
	return s_bc;
}

template <int NW, class S8, class S32, class PP, class CP>
__device__ void ksw_wave(int qlen, int tlen, S8 mem, S32 H, PP p, CP cig,
                         int q, int e, int q2, int e2, int sc_mch, int sc_mis, int sc_N,
                         int w, int zdrop, int end_bonus, int flag, Ez &ez)
{
	constexpr bool LDS = PtrTraits<S8>::lds;
	typedef typename PtrTraits<S8>::u8 SU8;
	constexpr int KW = PtrTraits<S8>::lds ? 2 : 4;            // chunks of 64 cells in flight per turn of the per-step loops (state in LDS / in the workspace)
	// NW waves share the call (NW > 1: state in the workspace only; `lane` is then the index in the workgroup, a "chunk" 64 NW cells)
	constexpr int CH = 64 * NW;
	const int lane = threadIdx.x;
	static_assert(NW == 1 || !LDS, "several waves per call: the workspace layout only");
	const bool approx_max = (flag & EZ_APPROX_MAX) != 0, right = (flag & EZ_RIGHT) != 0;
	ez.max = 0, ez.zdropped = 0, ez.max_q = ez.max_t = ez.mqe_t = -1, ez.mqe = ez.score = DP_NEG_INF, ez.reach_end = 0, ez.n_cigar = 0;
	const int qe = q + e;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int wl = w, wr = w;
	const int tlen_ = (tlen + 15) / 16, qlen_ = (qlen + 15) / 16;
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int ncol = n_col_ * 16;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const int T = tlen_ * 16;
	S8 u = mem, v = u + T, x = v + T, y = x + T, x2 = y + T, y2 = x2 + T, s = y2 + T;
	SU8 sf = (SU8)(s + T), qr = sf + T;
	(void)qlen_;
	// u, v, x, y = -q - e ; x2, y2 = -q2 - e2 ; s = 0 ; H = -inf  (sf / qr were filled by the caller)
	for (int i = lane; i < 4 * T; i += CH) u[i] = (int8_t)(-q - e);
	for (int i = lane; i < 2 * T; i += CH) x2[i] = (int8_t)(-q2 - e2);
	for (int i = lane; i < T; i += CH) s[i] = 0;
	if (!approx_max) for (int i = lane; i < T; i += CH) H[i] = DP_NEG_INF;
	st_order<LDS, NW>();

	int last_st = -1, last_en = -1, H0 = 0, last_H0_t = 0;
	const int n_r = qlen + tlen - 1;
	for (int r = 0; r < n_r; ++r) {
		int st = 0, en = tlen - 1;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
		if (en > (r + wl) >> 1) en = (r + wl) >> 1;
		if (st > en) { ez.zdropped = 1; break; }
		const int st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		int x1, x21, v1;
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = x[st - 1], x21 = x2[st - 1], v1 = v[st - 1];
			else x1 = -q - e, x21 = -q2 - e2, v1 = -q - e;
		} else {
			x1 = -q - e, x21 = -q2 - e2;
			v1 = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		}
		st_order<LDS, NW>();
		if (en >= r && lane == 0) {
			y[r] = (int8_t)(-q - e), y2[r] = (int8_t)(-q2 - e2);
			u[r] = (int8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
		}
		// scores: 16-lane strides from st0, all loads before all stores
		{
			SU8 qrr = qr + (qlen - 1 - r);
			const int n16 = (en0 - st0) / 16 + 1;
			// (KW chunks of 64 cells per turn: a long call is one wave's serial work, and a turn is a round trip to LDS)
			for (int c0 = 0; c0 < n16 * 16; c0 += CH * KW) {
				int sc[KW];
#pragma unroll
				for (int k = 0; k < KW; ++k) {
					const int i = c0 + CH * k + lane;
					sc[k] = 0;
					if (i < n16 * 16) {
						const int sq = sf[st0 + i], sq2 = qrr[st0 + i];
						sc[k] = (sq == 4 || sq2 == 4) ? sc_N : sq == sq2 ? sc_mch : sc_mis;
					}
				}
				st_order<LDS, NW>();
#pragma unroll
				for (int k = 0; k < KW; ++k) {
					const int i = c0 + CH * k + lane;
					if (i < n16 * 16) s[st0 + i] = (int8_t)sc[k];
				}
			}
		}
		st_order<LDS, NW>();
		// core: chunks of 64 lanes from the top; a chunk reads [t-1] of the chunk below before that is updated
		PP pr = p + (size_t)r * ncol;
		// (every cell of the anti-diagonal reads old values only -- its own and [t-1] of the cell below -- so KW chunks may
		// load before any of them stores: the same values as one chunk at a time from the top)
		for (int c0 = (en - st) / CH * CH; c0 >= 0; c0 -= CH * KW) {
			int z[KW], xt1[KW], vt1[KW], x2t1[KW], ut[KW], yt[KW], y2t[KW];
#pragma unroll
			for (int k = 0; k < KW; ++k) {
				const int t = st + c0 - CH * k + lane;
				const bool act = c0 - CH * k >= 0 && t <= en;
				z[k] = xt1[k] = vt1[k] = x2t1[k] = ut[k] = yt[k] = y2t[k] = 0;
				if (act) {
					z[k] = s[t];
					xt1[k] = t > st ? (int)x[t - 1] : x1;
					vt1[k] = t > st ? (int)v[t - 1] : v1;
					x2t1[k] = t > st ? (int)x2[t - 1] : x21;
					ut[k] = u[t], yt[k] = y[t], y2t[k] = y2[t];
				}
			}
			st_order<LDS, NW>();
#pragma unroll
			for (int k = 0; k < KW; ++k) {
				const int t = st + c0 - CH * k + lane;
				const bool act = c0 - CH * k >= 0 && t <= en;
				if (act) {
					int zz = z[k];
					int a = I8(xt1[k] + vt1[k]), b = I8(yt[k] + ut[k]), a2 = I8(x2t1[k] + vt1[k]), b2 = I8(y2t[k] + ut[k]), d, tmp;
					if (!right) {
						d = a > zz ? 1 : 0;  zz = zz > a ? zz : a;
						d = b > zz ? 2 : d;  zz = zz > b ? zz : b;
						d = a2 > zz ? 3 : d; zz = zz > a2 ? zz : a2;
						d = b2 > zz ? 4 : d; zz = zz > b2 ? zz : b2;
					} else {
						d = zz > a ? 0 : 1;  zz = zz > a ? zz : a;
						d = zz > b ? d : 2;  zz = zz > b ? zz : b;
						d = zz > a2 ? d : 3; zz = zz > a2 ? zz : a2;
						d = zz > b2 ? d : 4; zz = zz > b2 ? zz : b2;
					}
					zz = zz < sc_mch ? zz : sc_mch;
					u[t] = (int8_t)(zz - vt1[k]), v[t] = (int8_t)(zz - ut[k]);
					tmp = I8(zz - q), a = I8(a - tmp), b = I8(b - tmp);
					tmp = I8(zz - q2), a2 = I8(a2 - tmp), b2 = I8(b2 - tmp);
					if (!right) {
						x[t] = (int8_t)((a > 0 ? a : 0) - qe);          d |= a > 0 ? 0x08 : 0;
						y[t] = (int8_t)((b > 0 ? b : 0) - qe);          d |= b > 0 ? 0x10 : 0;
						x2[t] = (int8_t)((a2 > 0 ? a2 : 0) - (q2 + e2)); d |= a2 > 0 ? 0x20 : 0;
						y2[t] = (int8_t)((b2 > 0 ? b2 : 0) - (q2 + e2)); d |= b2 > 0 ? 0x40 : 0;
					} else {
						x[t] = (int8_t)((0 > a ? 0 : a) - qe);          d |= 0 > a ? 0 : 0x08;
						y[t] = (int8_t)((0 > b ? 0 : b) - qe);          d |= 0 > b ? 0 : 0x10;
						x2[t] = (int8_t)((0 > a2 ? 0 : a2) - (q2 + e2)); d |= 0 > a2 ? 0 : 0x20;
						y2[t] = (int8_t)((0 > b2 ? 0 : b2) - (q2 + e2)); d |= 0 > b2 ? 0 : 0x40;
					}
					pr[t - st] = (uint8_t)d;
				}
			}
			st_order<LDS, NW>();
		}
		if (!approx_max) {
			int max_H, max_t;
			if (r > 0) {
				// H[en0] first (from the old H[en0-1]); then H[t] += v[t] for t in [st0, en0); the maximum in
				// the SSE scan's tie order: en0, four interleaved lanes over [st0, en1), the tail [en1, en0)
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				const int h_en0 = en0 > 0 ? H[en0 - 1] + (int)u[en0] : H[en0] + (int)v[en0];
				st_order<LDS, NW>();
				int best_h = DP_NEG_INF - 1;
				unsigned best_rank = 0xffffffffu;
				for (int c0 = 0; c0 < en0 - st0; c0 += CH * KW) {
					int hh[KW];
#pragma unroll
					for (int k = 0; k < KW; ++k) {
						const int t = st0 + c0 + CH * k + lane;
						hh[k] = t < en0 ? H[t] + (int)v[t] : 0;
					}
#pragma unroll
					for (int k = 0; k < KW; ++k) {
						const int t = st0 + c0 + CH * k + lane;
						if (t < en0) {
							const int h = hh[k];
							H[t] = h;
							const unsigned rank = t < en1 ? 1u + ((unsigned)(t - st0) & 3u) * 0x1000000u + ((unsigned)(t - st0) >> 2)
							                              : 1u + 4u * 0x1000000u + (unsigned)(t - en1);
							if (h > best_h || (h == best_h && rank < best_rank)) best_h = h, best_rank = rank;
						}
					}
				}
				if (lane == 0) {
					H[en0] = h_en0;
					if (h_en0 > best_h || (h_en0 == best_h)) best_h = h_en0, best_rank = 0;   // en0 comes first in the scan
				}
				const int mh = wg_max_i32<NW>(best_h);
				const unsigned mr = wg_min_u32<NW>(best_h == mh ? best_rank : 0xffffffffu);
				max_H = mh;
				if (mr == 0) max_t = en0;
				else if (mr < 1u + 4u * 0x1000000u) { const unsigned k = mr - 1u; max_t = st0 + (int)((k & 0xffffffu) * 4u + (k >> 24)); }
				else max_t = en1 + (int)(mr - 1u - 4u * 0x1000000u);
				st_order<LDS, NW>();
			} else {
				if (lane == 0) H[0] = (int)v[0] - qe;
				st_order<LDS, NW>();
				max_H = H[0], max_t = 0;
			}
			if (r - st0 == qlen - 1 && H[st0] > ez.mqe) ez.mqe = H[st0], ez.mqe_t = st0;
			if (apply_zdrop(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H[tlen - 1];
		} else {
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					const int d0 = v[last_H0_t], d1 = u[last_H0_t + 1];
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v[last_H0_t];
				} else {
					++last_H0_t, H0 += u[last_H0_t];
				}
			} else H0 = (int)v[0] - qe, last_H0_t = 0;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H0;
		}
		last_st = st, last_en = en;
	}
	st_order<false, NW>();                                  // the direction bytes have to be in memory before the walk reads them
	// ---- backtrack (ksw_backtrack, rotated layout): one lane walks, all lanes prefetch nothing yet
	int i0 = -1, j0 = -1;
	if (!ez.zdropped && !(flag & EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
	else if (!ez.zdropped && (flag & EZ_EXTZ_ONLY) && ez.mqe + end_bonus > ez.max) ez.reach_end = 1, i0 = ez.mqe_t, j0 = qlen - 1;
	else if (ez.max_t >= 0 && ez.max_q >= 0) i0 = ez.max_t, j0 = ez.max_q;
	int n_cigar = 0;
	if (i0 >= 0 && j0 >= 0 && lane == 0) {
		int i = i0, j = j0, state = 0;
		uint32_t cur = 0;                                     // op being grown: len << 4 | op, 0 = none
		auto push = [&](uint32_t op, int len) {
			if (cur != 0 && (cur & 0xf) == op) cur += (uint32_t)len << 4;
			else { if (cur != 0) cig[n_cigar++] = cur; cur = (uint32_t)len << 4 | op; }
		};
		while (i >= 0 && j >= 0) {
			const int r = i + j;
			int st = 0, en = tlen - 1;
			if (st < r - qlen + 1) st = r - qlen + 1;
			if (en > r) en = r;
			if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
			if (en > (r + wl) >> 1) en = (r + wl) >> 1;
			st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;    // off[r], off_end[r]
			int force_state = -1;
			if (i < st) force_state = 2;
			if (i > en) force_state = 1;
			const uint32_t tmp = force_state < 0 ? p[(size_t)r * ncol + (size_t)(i - st)] : 0;
			if (state == 0) state = tmp & 7;
			else if (!(tmp >> (state + 2) & 1)) state = 0;
			if (state == 0) state = tmp & 7;
			if (force_state >= 0) state = force_state;
			if (state == 0) push(0, 1), --i, --j;
			else if (state == 1 || state == 3) push(2, 1), --i;
			else push(1, 1), --j;
		}
		if (i >= 0) push(2, i + 1);
		if (j >= 0) push(1, j + 1);
		if (cur != 0) cig[n_cigar++] = cur;
		if (!(flag & EZ_REV_CIGAR))
			for (int k = 0; k < n_cigar >> 1; ++k) { const uint32_t t2 = cig[k]; cig[k] = cig[n_cigar - 1 - k], cig[n_cigar - 1 - k] = t2; }
	}
	ez.n_cigar = wg_bcast0<NW>(n_cigar);
	st_order<false, NW>();
}

// ================================================================ ksw_extd2 on a whole workgroup, state in registers
// The same call as ksw_wave, for the long ones (an extension of 1 500 query bases is 4 500 anti-diagonals of up to 751 + 31
// cells): NW waves, C consecutive cells per thread, one barrier per anti-diagonal.
//
// ksw2 keeps u v x y x2 y2 s (int8) and H (int32) in arrays indexed by the target position t and updates them in place;
// an anti-diagonal touches t in [st, en] (both rounded to 16, which is why cells outside the band hold values that only
// this layout defines) and reads x, v, x2 of t - 1.  Both ends only move up, and en - st stays below the workgroup's
// CW = 64 NW C cells: so the arrays live in REGISTERS, cell t in thread (t mod CW) / C -- the window of cells the workgroup
// holds is [st - 16, st - 16 + CW); a thread whose cells have dropped out below takes the next C above the window, which no
// step has touched yet (the arrays' initial values: the window's top stays more than a rounding step above en).  st and
// en + 1 are multiples of 16, so a thread's C cells (C divides 16) are inside the anti-diagonal together or not at all.
// What a step reads of ANOTHER thread's cells -- x, v, x2 (and H, below en0) of the cell below a thread's first -- comes
// from the neighbouring lane by a DPP shift, from the neighbouring wave through one LDS word per wave, written before
// the step's barrier (two buffers in turn: one barrier a step).  What ksw2's bookkeeping reads of single cells (the
// exact maximum's H[st0], H[tlen - 1]; the approximate maximum's u, v) their owners put into four LDS words likewise.
// The exact-maximum scan is a DPP reduction per wave and NW words in LDS; its result, the Z-drop test and `mqe` are
// evaluated one step late, after the next step's barrier -- a step more is computed than ksw2 would (its direction bytes
// are never read: the walk starts at or below the maximum's anti-diagonal).
// The sequences lie in LDS, one byte per base; the walk afterwards reads the direction bytes through LDS tiles of
// 128 anti-diagonals x 128 target positions that all threads load (a walk step is an LDS round trip instead of one to L2).
// <8, 2> (first <16, 1>): one call as fast as it goes (a micro-batch, a block with a handful of long calls waits for the longest);
// <4, 4>: four times the calls side by side at ~1.5 times the time each (a batch of divergent reads has thousands).
#define MNC_DPPW(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xf, false)
typedef __attribute__((address_space(3))) unsigned long long *lds_u64p;
#ifdef MNC_WG_TIMING
// profiling build only: cycles of thread 0 per phase of a call, summed over the calls -- [0] sequences into LDS, [1] the
// anti-diagonals, [2] the walk, [3] calls, [4] anti-diagonals, [5] mm_test_zdrop on lane 0, [6] the kernel's own set-up of a call
__device__ unsigned long long g_wg_cycles[8];
#define WG_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define WG_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_wg_cycles[i], (unsigned long long)(v)); } while (0)
#else
#define WG_T(var) do {} while (0)
#define WG_ADD(i, v) do {} while (0)
#endif
constexpr int WG_TILE = 128;               // the walk's LDS tile: anti-diagonals x target positions
// LDS of a workgroup: [2][NW] boundary records | [2][NW] maximum keys | [2][4] single cells | 16 words of the walk | the
// reversed query's codes | the target's (`seq` bytes each; at least the walk's tile together)
template <int NW> constexpr int wg_lds_head() { return 2 * NW * 8 * 2 + 2 * 4 * 8 + 64; }
template <int NW> constexpr int wg_lds_bytes(int seq) { return wg_lds_head<NW>() + 2 * seq; }

__device__ __forceinline__ int wave_max_dpp(int v)           // the maximum over the wave, uniform
{
	const int lo = INT32_MIN;
	int o;
	o = MNC_DPPW(lo, v, 0x111, 0xf); v = v > o ? v : o;  o = MNC_DPPW(lo, v, 0x112, 0xf); v = v > o ? v : o;
	o = MNC_DPPW(lo, v, 0x114, 0xf); v = v > o ? v : o;  o = MNC_DPPW(lo, v, 0x118, 0xf); v = v > o ? v : o;
	o = MNC_DPPW(lo, v, 0x142, 0xa); v = v > o ? v : o;  o = MNC_DPPW(lo, v, 0x143, 0xc); v = v > o ? v : o;
	return __builtin_amdgcn_readlane(v, 63);
}

// can this call run on ksw_wg<NW, C>?  (the widest anti-diagonal, its rounding at both ends, the 16 cells below st, one
// rounding step of head room; the sequences in `seq` bytes of LDS each)
template <int NW, int C> __device__ __forceinline__ bool wg_fits(int qlen, int tlen, int w, int seq)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	int width = qlen < tlen ? qlen : tlen;
	if (w + 1 < width) width = w + 1;
	return width + 30 + 16 + 17 <= 64 * NW * C && (tlen + 15) / 16 * 16 <= seq && (qlen + 15) / 16 * 16 + 32 <= seq;
}

// DEC: 0 the direction bytes are ksw2's; 1 / 2 the packed kernel's codes (left- / right-aligned gaps: kpk::decode)
template <int NW, class PP, class CP, int DEC = 0>
__device__ int walk_wg(int qlen, int tlen, int wl, int wr, int ncol, PP p, CP cig, int i0, int j0, int flag, lds_u8p tile, lds_i32p bc)
{
	constexpr int CW = 64 * NW, TR = WG_TILE;
	const int lane = threadIdx.x;
	int i = i0, j = j0, state = 0, n_cigar = 0;
	uint32_t cur = 0;                                         // op being grown: len << 4 | op, 0 = none
	auto push = [&](uint32_t op, int len) {
		if (cur != 0 && (cur & 0xf) == op) cur += (uint32_t)len << 4;
		else { if (cur != 0) cig[n_cigar++] = cur; cur = (uint32_t)len << 4 | op; }
	};
	auto row_range = [&](int r, int &st, int &en) {           // off[r], off_end[r]
		st = 0, en = tlen - 1;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
		if (en > (r + wl) >> 1) en = (r + wl) >> 1;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
	};
	for (;;) {
		if (lane == 0) bc[0] = i, bc[1] = j;
		__syncthreads();
		const int ti = bc[0], tj = bc[1];
		if (ti < 0 || tj < 0) break;                          // uniform: the walk has left the matrix
		const int r_top = ti + tj;
		// the tile: anti-diagonals r_top - k, target positions ti - c (a walk step goes down one or two anti-diagonals
		// and at most one target position)
		for (int idx = lane; idx < TR * TR; idx += CW) {
			const int k = idx / TR, c = idx % TR, rr = r_top - k, ii = ti - c;
			if (rr >= 0 && ii >= 0) {
				int st, en;
				row_range(rr, st, en);
				if (ii >= st && ii <= en) tile[idx] = p[(size_t)rr * ncol + (size_t)(ii - st)];
			}
		}
		__syncthreads();
		if (lane == 0) {
			while (i >= 0 && j >= 0) {
				const int r = i + j, k = r_top - r, c = ti - i;
				if (k >= TR || c >= TR) break;                    // the next tile
				int st, en;
				row_range(r, st, en);
				int force_state = -1;
				if (i < st) force_state = 2;
				if (i > en) force_state = 1;
				uint32_t tmp = force_state < 0 ? tile[k * TR + c] : 0;
				if (DEC == 1 && force_state < 0) tmp = kpk::decode<false>(tmp);
				if (DEC == 2 && force_state < 0) tmp = kpk::decode<true>(tmp);
				if (state == 0) state = tmp & 7;
				else if (!(tmp >> (state + 2) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (force_state >= 0) state = force_state;
				if (state == 0) push(0, 1), --i, --j;
				else if (state == 1 || state == 3) push(2, 1), --i;
				else push(1, 1), --j;
			}
		}
	}
	if (lane == 0) {
		if (i >= 0) push(2, i + 1);
		if (j >= 0) push(1, j + 1);
		if (cur != 0) cig[n_cigar++] = cur;
		if (!(flag & EZ_REV_CIGAR))
			for (int k = 0; k < n_cigar >> 1; ++k) { const uint32_t t2 = cig[k]; cig[k] = cig[n_cigar - 1 - k], cig[n_cigar - 1 - k] = t2; }
	}
	return n_cigar;                                           // thread 0's
}

// an LDS word every lane reads from the same address: as scalars
__device__ __forceinline__ int uni_hi(unsigned long long v) { return __builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)); }
__device__ __forceinline__ unsigned uni_lo(unsigned long long v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v); }

template <int NW, int C, class PP, class CP>
__device__ __attribute__((noinline)) void ksw_wg(int qlen, int tlen, const uint8_t *sf_g, const uint8_t *qr_g, lds_u8p lds, int seq_lds, PP p, CP cig,
                       int q, int e, int q2, int e2, int sc_mch, int sc_mis, int sc_N,
                       int w, int zdrop, int end_bonus, int flag, Ez &ez_out)
{
	constexpr int CW = 64 * NW * C, M = CW - 1;
	static_assert((NW & (NW - 1)) == 0 && NW <= 16 && (C == 1 || C == 2 || C == 4), "waves: a power of two, a row of 16 lanes at most; cells per thread divide 16");
	// a function of its own (not inlined into the kernel's other forms): its arguments arrive in vector registers, and
	// every one of them is the same in all lanes -- back into scalars, or the bookkeeping of every step runs on the vector ALU
	{
		auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
		auto unip = [&](auto ptr) { const unsigned long long v = (unsigned long long)ptr; return (decltype(ptr))((unsigned long long)(unsigned)uni((int)(unsigned)v) | (unsigned long long)(unsigned)uni((int)(unsigned)(v >> 32)) << 32); };
		qlen = uni(qlen), tlen = uni(tlen), seq_lds = uni(seq_lds), q = uni(q), e = uni(e), q2 = uni(q2), e2 = uni(e2), sc_mch = uni(sc_mch), sc_mis = uni(sc_mis), sc_N = uni(sc_N);
		w = uni(w), zdrop = uni(zdrop), end_bonus = uni(end_bonus), flag = uni(flag);
		sf_g = unip(sf_g), qr_g = unip(qr_g), p = unip(p), cig = unip(cig);
		lds = (lds_u8p)(unsigned)uni((int)(unsigned)(unsigned long long)lds);
	}
	const int lane = threadIdx.x, wv = lane >> 6;
	lds_u64p bnd = (lds_u64p)lds;                             // [2][NW] {x, v, x2 | H} of the last cell of a wave's last lane
	lds_u64p sl = bnd + 2 * NW;                               // [2][NW] the waves' (H, rank) of the exact-maximum scan
	lds_u64p one = sl + 2 * NW;                               // [2][4] single cells: H[st0], H[tlen - 1], v[last_H0_t], u[last_H0_t + 1]
	lds_i32p bc = (lds_i32p)(one + 2 * 4);                    // 16 words for the walk
	lds_u8p sq = (lds_u8p)(bc + 16);                          // the reversed query's codes
	lds_u8p stg = sq + seq_lds;                               // the target's
	const bool approx_max = (flag & EZ_APPROX_MAX) != 0, right = (flag & EZ_RIGHT) != 0;
	// ksw2's `ez`, field by field in scalars of this function (the caller's struct lives in memory; every field is wave-uniform)
	int z_max = 0, z_zdropped = 0, z_max_q = -1, z_max_t = -1, z_mqe_t = -1, z_mqe = DP_NEG_INF, z_score = DP_NEG_INF, z_reach_end = 0;
	const int qe = q + e;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int wl = w, wr = w;
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int ncol = n_col_ * 16;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const int T = (tlen + 15) / 16 * 16, Q = (qlen + 15) / 16 * 16 + 32;
	WG_T(tm0);
	__syncthreads();                                          // (the LDS may still hold the previous call's walk)
	for (int i = lane; i < T; i += 64 * NW) stg[i] = sf_g[i];
	for (int i = lane; i < Q; i += 64 * NW) sq[i] = qr_g[i];
	if (lane < 2 * NW) sl[lane] = 0;
	__syncthreads();

	// this thread's cells: t0 .. t0 + C - 1
	int t0 = -16 + ((lane * C + 16) & M);
	int tb[C], cu[C], cv[C], cx[C], cy[C], cx2[C], cy2[C], cs[C], cH[C];
#pragma unroll
	for (int c = 0; c < C; ++c) {
		tb[c] = t0 + c >= 0 && t0 + c < T ? (int)stg[t0 + c] : 0;
		cu[c] = cv[c] = cx[c] = cy[c] = -q - e, cx2[c] = cy2[c] = -q2 - e2, cs[c] = 0, cH[c] = DP_NEG_INF;
	}
	unsigned long long wkey = 0;                              // the wave's exact-maximum key of the step before (uniform)
	int last_st = -1, last_en = -1, H0 = 0, last_H0_t = 0;
	int p_st = 0, p_en = 0, p_st0 = 0, p_en0 = 0;
	const int n_r = qlen + tlen - 1;
	WG_T(tm1);
	int r_done = 0;
	for (int r = 0;; ++r) {
		r_done = r;
		// ---- the cells as step r - 1 left them, for whoever reads another thread's
		lds_u64p bndb = bnd + (r & 1) * NW, slb = sl + (r & 1) * NW, oneb = one + (r & 1) * 4;
		const unsigned top = (unsigned)(cx[C - 1] & 0xff) | (unsigned)(cv[C - 1] & 0xff) << 8 | (unsigned)(cx2[C - 1] & 0xff) << 16;
		if ((lane & 63) == 63) bndb[wv] = (unsigned long long)top | (unsigned long long)(unsigned)cH[C - 1] << 32;
		if ((lane & 63) == 0) slb[wv] = wkey;
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const int t = t0 + c;
			if (!approx_max) {
				if (t == p_st0) oneb[0] = (unsigned long long)(unsigned)cH[c] << 32;
				if (t == tlen - 1) oneb[1] = (unsigned long long)(unsigned)cH[c] << 32;
			} else {
				if (t == last_H0_t) oneb[2] = (unsigned)(cv[c] & 0xff);
				if (t == last_H0_t + 1) oneb[3] = (unsigned)(cu[c] & 0xff);
			}
		}
		// the step's barrier: the LDS words above must have landed -- the direction bytes on their way to memory need not
		// (__syncthreads would wait for them too: a round trip to L2 per anti-diagonal, several microseconds beside kernels
		// that keep the memory system busy); the barrier behind the loop waits for all of them
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		// ---- what ksw2 does after the cells of an anti-diagonal, for step r - 1
		if (r > 0) {
			const int rp = r - 1;
			if (!approx_max) {
				// the largest of the waves' keys: a row of 16 lanes reads them, a prefix maximum along the row
				unsigned long long k = slb[lane & (NW - 1)];
#define MNC_KEY_STEP(ctrl) { const unsigned lo = (unsigned)MNC_DPPW(0, (int)(unsigned)k, ctrl, 0xf), hi = (unsigned)MNC_DPPW(0, (int)(unsigned)(k >> 32), ctrl, 0xf); \
	const unsigned long long o = (unsigned long long)hi << 32 | lo; k = o > k ? o : k; }
				if constexpr (NW > 1) MNC_KEY_STEP(0x111)
				if constexpr (NW > 2) MNC_KEY_STEP(0x112)
				if constexpr (NW > 4) MNC_KEY_STEP(0x114)
				if constexpr (NW > 8) MNC_KEY_STEP(0x118)
#undef MNC_KEY_STEP
				const unsigned klo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k, NW - 1), khi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(k >> 32), NW - 1);
				const int max_H = (int)(khi ^ 0x80000000u);
				const unsigned mr = ~klo;
				const int en1 = p_st0 + (p_en0 - p_st0) / 4 * 4;
				int max_t;
				if (mr == 0) max_t = p_en0;
				else if (mr < 1u + 4u * 0x1000000u) { const unsigned kk = mr - 1u; max_t = p_st0 + (int)((kk & 0xffffffu) * 4u + (kk >> 24)); }
				else max_t = en1 + (int)(mr - 1u - 4u * 0x1000000u);
				if (rp - p_st0 == qlen - 1) {
					const int h = uni_hi(oneb[0]);
					if (h > z_mqe) z_mqe = h, z_mqe_t = p_st0;
				}
				// ksw_apply_zdrop
				if (max_H > z_max) z_max = max_H, z_max_t = max_t, z_max_q = rp - max_t;
				else if (max_t >= z_max_t && rp - max_t >= z_max_q) {
					const int tl = max_t - z_max_t, ql = (rp - max_t) - z_max_q;
					const int l = tl > ql ? tl - ql : ql - tl;
					if (zdrop >= 0 && z_max - max_H > zdrop + l * e2) { z_zdropped = 1; break; }
				}
				if (rp == qlen + tlen - 2 && p_en0 == tlen - 1) z_score = uni_hi(oneb[1]);
			} else {
				if (rp > 0) {
					const int d0 = I8(uni_lo(oneb[2])), d1 = I8(uni_lo(oneb[3]));
					if (last_H0_t >= p_st0 && last_H0_t <= p_en0 && last_H0_t + 1 >= p_st0 && last_H0_t + 1 <= p_en0) {
						if (d0 > d1) H0 += d0;
						else H0 += d1, ++last_H0_t;
					} else if (last_H0_t >= p_st0 && last_H0_t <= p_en0) {
						H0 += d0;
					} else {
						++last_H0_t, H0 += d1;
					}
				} else H0 = I8(uni_lo(oneb[2])) - qe, last_H0_t = 0;
				if (rp == qlen + tlen - 2 && p_en0 == tlen - 1) z_score = H0;
			}
			last_st = p_st, last_en = p_en;
		}
		if (r >= n_r) break;
		int st = 0, en = tlen - 1;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
		if (en > (r + wl) >> 1) en = (r + wl) >> 1;
		if (st > en) { z_zdropped = 1; break; }
		const int st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		// ---- the cell below this thread's first, as the last step left it: the lane below, or the wave below through LDS
		unsigned nb = (unsigned)MNC_DPPW(0, (int)top, 0x138, 0xf);          // wave_shr:1
		int nbH = MNC_DPPW(0, cH[C - 1], 0x138, 0xf);
		if ((lane & 63) == 0) { const unsigned long long b = bndb[(wv - 1) & (NW - 1)]; nb = (unsigned)b, nbH = (int)(unsigned)(b >> 32); }
		// ---- the window moves up with st: a thread whose cells have left it takes fresh ones at the top
		{
			const int base = st - 16;
			const int tn = base + ((lane * C - base) & M);
			if (tn != t0) {
				t0 = tn;
#pragma unroll
				for (int c = 0; c < C; ++c) {
					cu[c] = cv[c] = cx[c] = cy[c] = -q - e, cx2[c] = cy2[c] = -q2 - e2, cs[c] = 0, cH[c] = DP_NEG_INF;
					tb[c] = tn + c < T ? (int)stg[tn + c] : 0;
				}
			}
		}
		const int v_edge = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		// x, v, x2 of the cell below cell c: the neighbour's for c = 0 (ksw2's x1, x21, v1 at st), else this thread's own, OLD
		int xb = I8(nb), vb = I8(nb >> 8), x2b = I8(nb >> 16), Hb = nbH;
		if (t0 == st) {
			if (st > 0) {
				if (!(st - 1 >= last_st && st - 1 <= last_en)) xb = -q - e, x2b = -q2 - e2, vb = -q - e;
			} else xb = -q - e, x2b = -q2 - e2, vb = v_edge;
		}
		const bool in_row = t0 >= st && t0 <= en;             // all C cells, or none
		const int n16 = (en0 - st0) / 16 + 1, en1 = st0 + (en0 - st0) / 4 * 4;
		int h_best = DP_NEG_INF - 1;
		unsigned rank_best = 0xffffffffu;
		unsigned dpack = 0;
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const int t = t0 + c;
			const int x_old = cx[c], v_old = cv[c], x2_old = cx2[c], H_old = cH[c];
			if (en >= r && t == r) cy[c] = -q - e, cy2[c] = -q2 - e2, cu[c] = v_edge;
			// scores: 16-lane strides from st0
			if (t >= st0 && t < st0 + n16 * 16) {
				const int qb = sq[qlen - 1 - r + t];
				cs[c] = (tb[c] == 4 || qb == 4) ? sc_N : tb[c] == qb ? sc_mch : sc_mis;
			}
			// the cell (every cell of the anti-diagonal reads old values only)
			if (in_row) {
				int zz = cs[c];
				int a = I8(xb + vb), b = I8(cy[c] + cu[c]), a2 = I8(x2b + vb), b2 = I8(cy2[c] + cu[c]), d, tmp;
				if (!right) {
					d = a > zz ? 1 : 0;  zz = zz > a ? zz : a;
					d = b > zz ? 2 : d;  zz = zz > b ? zz : b;
					d = a2 > zz ? 3 : d; zz = zz > a2 ? zz : a2;
					d = b2 > zz ? 4 : d; zz = zz > b2 ? zz : b2;
				} else {
					d = zz > a ? 0 : 1;  zz = zz > a ? zz : a;
					d = zz > b ? d : 2;  zz = zz > b ? zz : b;
					d = zz > a2 ? d : 3; zz = zz > a2 ? zz : a2;
					d = zz > b2 ? d : 4; zz = zz > b2 ? zz : b2;
				}
				zz = zz < sc_mch ? zz : sc_mch;
				const int ut = cu[c];
				cu[c] = I8(zz - vb), cv[c] = I8(zz - ut);
				tmp = I8(zz - q), a = I8(a - tmp), b = I8(b - tmp);
				tmp = I8(zz - q2), a2 = I8(a2 - tmp), b2 = I8(b2 - tmp);
				if (!right) {
					cx[c] = I8((a > 0 ? a : 0) - qe);            d |= a > 0 ? 0x08 : 0;
					cy[c] = I8((b > 0 ? b : 0) - qe);            d |= b > 0 ? 0x10 : 0;
					cx2[c] = I8((a2 > 0 ? a2 : 0) - (q2 + e2));  d |= a2 > 0 ? 0x20 : 0;
					cy2[c] = I8((b2 > 0 ? b2 : 0) - (q2 + e2));  d |= b2 > 0 ? 0x40 : 0;
				} else {
					cx[c] = I8((0 > a ? 0 : a) - qe);            d |= 0 > a ? 0 : 0x08;
					cy[c] = I8((0 > b ? 0 : b) - qe);            d |= 0 > b ? 0 : 0x10;
					cx2[c] = I8((0 > a2 ? 0 : a2) - (q2 + e2));  d |= 0 > a2 ? 0 : 0x20;
					cy2[c] = I8((0 > b2 ? 0 : b2) - (q2 + e2));  d |= 0 > b2 ? 0 : 0x40;
				}
				dpack |= (unsigned)d << (8 * c);
			}
			if (!approx_max) {
				// H[en0] from the OLD H[en0 - 1]; H[t] += v[t] for t in [st0, en0); the maximum in the SSE scan's tie order
				int h = DP_NEG_INF - 1;
				unsigned rank = 0xffffffffu;
				if (r > 0) {
					if (t >= st0 && t < en0) {
						cH[c] += cv[c], h = cH[c];
						rank = t < en1 ? 1u + ((unsigned)(t - st0) & 3u) * 0x1000000u + ((unsigned)(t - st0) >> 2)
						               : 1u + 4u * 0x1000000u + (unsigned)(t - en1);
					} else if (t == en0) {
						cH[c] = en0 > 0 ? Hb + cu[c] : cH[c] + cv[c], h = cH[c], rank = 0;   // en0 comes first in the scan
					}
				} else if (t == 0) cH[c] = cv[c] - qe, h = cH[c], rank = 0;
				if (h > h_best || (h == h_best && rank < rank_best)) h_best = h, rank_best = rank;
			}
			xb = x_old, vb = v_old, x2b = x2_old, Hb = H_old;   // ... of the next cell up
		}
		if (in_row) {
			PP pr = p + (size_t)r * ncol + (size_t)(t0 - st);
			if constexpr (C == 4) *reinterpret_cast<uint32_t*>(&pr[0]) = dpack;
			else if constexpr (C == 2) *reinterpret_cast<uint16_t*>(&pr[0]) = (uint16_t)dpack;
			else pr[0] = (uint8_t)dpack;
		}
		if (!approx_max) {
			const int mh = wave_max_dpp(h_best);
			const int mrk = wave_max_dpp((int)(h_best == mh ? ~rank_best : 0u) ^ INT32_MIN);   // the smallest rank among the wave's best
			wkey = (unsigned long long)((unsigned)mh ^ 0x80000000u) << 32 | ((unsigned)mrk ^ 0x80000000u);
		}
		p_st = st, p_en = en, p_st0 = st0, p_en0 = en0;
	}
	__syncthreads();                                          // the direction bytes are written; the sequences' LDS is free
	WG_T(tm2);
	(void)r_done;
	int i0 = -1, j0 = -1;
	if (!z_zdropped && !(flag & EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
	else if (!z_zdropped && (flag & EZ_EXTZ_ONLY) && z_mqe + end_bonus > z_max) z_reach_end = 1, i0 = z_mqe_t, j0 = qlen - 1;
	else if (z_max_t >= 0 && z_max_q >= 0) i0 = z_max_t, j0 = z_max_q;
	int n_cigar = 0;
	if (i0 >= 0 && j0 >= 0) n_cigar = walk_wg<NW>(qlen, tlen, wl, wr, ncol, p, cig, i0, j0, flag, sq, bc);
	n_cigar = wg_bcast0<NW>(n_cigar);
	WG_T(tm3);
	WG_ADD(0, tm1 - tm0); WG_ADD(1, tm2 - tm1); WG_ADD(2, tm3 - tm2); WG_ADD(3, 1); WG_ADD(4, r_done);
	ez_out.max = z_max, ez_out.zdropped = z_zdropped, ez_out.max_q = z_max_q, ez_out.max_t = z_max_t, ez_out.mqe = z_mqe, ez_out.mqe_t = z_mqe_t;
	ez_out.score = z_score, ez_out.reach_end = z_reach_end, ez_out.n_cigar = n_cigar;
	st_order<false, NW>();
}

// ================================================================ ksw_extd2 on ONE wave, packed pairs in registers
// The same call once more (round 5): one wave, C consecutive cells per lane held as C / 2 packed pairs -- two cells per
// 32-bit register, the recurrence on VOP3P with tags in the low byte of every 16-bit lane (ksw_pk.h: the value byte IS
// ksw2's int8, the tags do the work of its comparison chains, the direction byte is the XOR of five tag bytes).  What
// stays as in ksw_wg: ksw2's arrays indexed by the target position t, cell t in lane (t mod 64 C) / C for as long as it
// is inside the window [st - 16, st - 16 + 64 C), the 16-lane rounding of both ends, stale cells outside the band, scores
// refreshed on [st0, st0 + 16 n) only, the exact-maximum scan's tie order, the approximate-maximum walk.  What goes: the
// barrier and the LDS words between waves (the cell below a lane's first comes by ONE DPP rotation -- the window is
// circular, lane 0's neighbour is lane 63), the int8 emulation (~4 vector instructions per cell in 32-bit lanes), the
// target's codes in LDS (a lane reads its C codes when it takes new cells: every 64 C / 16 anti-diagonals or so).
// Per anti-diagonal and PAIR of cells: 3 v_alignbit (the cells below), 28 for the cell (kpk::cell_pair), 1 + 3 + 2 for the
// query window, the score and its stale-cell mask; per lane a dozen more and, with the exact maximum, 2 per cell for H.
//   C = 16: anti-diagonals of up to 961 cells (any band minimap2 uses: 751 + rounding); C = 8 / 4: up to 449 / 193 -- the
//   small calls, with their registers (and twice / four times the waves a SIMD).
// Calls with an ambiguous base, or scores outside kpk::params_fit, keep the forms above.
constexpr int WP_NEG = -(1 << 26);                        // H of a cell no anti-diagonal has reached (ksw2: -2^30; only its sign and size matter)
constexpr int wp_lds_bytes(int seq, int nw = 1) { return (nw > 1 ? 256 : 64) + (seq > WG_TILE * WG_TILE ? seq : WG_TILE * WG_TILE); }
template <int C, int NW = 1> __device__ __forceinline__ bool wp_fits(int qlen, int tlen, int w, int seq)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	int width = qlen < tlen ? qlen : tlen;
	if (w + 1 < width) width = w + 1;
	return width + 30 + 16 + 17 <= 64 * NW * C && (qlen + 15) / 16 * 16 + 32 <= seq;
}
#define MNC_DPP_ROR1(v) __builtin_amdgcn_update_dpp(0, (v), 0x13C, 0xf, 0xf, false)      // wave_ror:1 -- lane i reads lane i - 1, lane 0 lane 63

template <int C, bool RIGHT, bool APPROX, int NW, class PP, class CP>
__device__ __attribute__((noinline)) void ksw_wp(int qlen, int tlen, const uint8_t *sf_g, const uint8_t *qr_g, lds_u8p lds, PP p, CP cig,
                                                 int q, int e, int q2, int e2, int sc_mch, int sc_mis, int w, int zdrop, int end_bonus, int flag, Ez &ez_out)
{
	constexpr int CW = 64 * NW * C, M = CW - 1, NP = C / 2, CB = C - 1;
	static_assert(NW == 1 || NW == 2 || NW == 4, "waves on one call: one, or one a SIMD");
	static_assert(C == 4 || C == 8 || C == 16, "cells per lane: a divisor of 16, whole pairs");
	{
		auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
		auto unip = [&](auto ptr) { const unsigned long long v = (unsigned long long)ptr; return (decltype(ptr))((unsigned long long)(unsigned)uni((int)(unsigned)v) | (unsigned long long)(unsigned)uni((int)(unsigned)(v >> 32)) << 32); };
		qlen = uni(qlen), tlen = uni(tlen), q = uni(q), e = uni(e), q2 = uni(q2), e2 = uni(e2), sc_mch = uni(sc_mch), sc_mis = uni(sc_mis);
		w = uni(w), zdrop = uni(zdrop), end_bonus = uni(end_bonus), flag = uni(flag);
		sf_g = unip(sf_g), qr_g = unip(qr_g), p = unip(p), cig = unip(cig);
		lds = (lds_u8p)(unsigned)uni((int)(unsigned)(unsigned long long)lds);
	}
	// NW > 1 (round 5, for a pass with a handful of calls: a lone call's length is its steps times a wave's instructions a
	// step): the call's cells over NW waves, thread `tid` of the workgroup where the one-wave form has lane `tid` -- cell t
	// in thread (t mod 64 NW C) / C.  What crosses a wave's edge goes through LDS words and barriers that wait for LDS only:
	// [1] the top cell of each wave's last lane, for the first lane of the next wave; [2] each wave's maximum and the four
	// single cells ksw2's bookkeeping reads; [3] (when the maximum's position is asked for) each wave's best rank.  Every
	// wave keeps ksw2's per-call state itself, from the same words: all take the same branches and meet at every barrier.
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
	lds_i32p bc = (lds_i32p)lds;                              // 16 words for the walk
	lds_i32p xw = bc + 16;                                    // [NW][4] x, v, x2, H of the top cell of a wave's last lane
	lds_i32p red = xw + 16;                                   // [NW] the waves' maxima; [NW][2] their best (rank, t)
	lds_i32p one = red + 16;                                  // [4] H[st0], H[tlen - 1], v[last_H0_t], u[last_H0_t + 1]
	lds_u8p sq = (lds_u8p)(bc + (NW > 1 ? 64 : 16));          // the reversed query's codes; afterwards the walk's tile
#define MNC_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
	const kpk::Consts K = kpk::make_consts<RIGHT>(q, e, q2, e2, sc_mch, sc_mis);
	int z_max = 0, z_zdropped = 0, z_max_q = -1, z_max_t = -1, z_mqe_t = -1, z_mqe = DP_NEG_INF, z_score = DP_NEG_INF, z_reach_end = 0;
	const int qe = q + e;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int wl = w, wr = w;
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int ncol = n_col_ * 16;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const int T = (tlen + 15) / 16 * 16, Q = (qlen + 15) / 16 * 16 + 32;
	WG_T(tm0);
	__syncthreads();                                          // (the LDS may still hold the previous call's walk)
	for (int i = tid; i < Q / 4; i += 64 * NW) reinterpret_cast<lds_u32p>(sq)[i] = reinterpret_cast<const uint32_t*>(qr_g)[i];
	__syncthreads();

	// ---- this lane's cells t0 .. t0 + C - 1, pair k = cells t0 + 2 k (low half), t0 + 2 k + 1 (high half)
	typedef int hvec_t __attribute__((ext_vector_type(C)));
	typedef uint32_t pvec_t __attribute__((ext_vector_type(NP)));
	pvec_t U, V, X, Y, X2, Y2, S, TB, QB;
	hvec_t H;                                              // (vectors: a wave-uniform index into them is an indirect register access, not a chain of compares)
	auto take_cells = [&](int tn, int r) {                    // the arrays' initial values; the target's codes; the query window of step r
#pragma unroll
		for (int k = 0; k < NP; ++k) U[k] = V[k] = K.iuv, X[k] = K.ix, Y[k] = K.iy, X2[k] = K.ix2, Y2[k] = K.iy2, S[k] = K.is;
		if (!APPROX) {
#pragma unroll
			for (int c = 0; c < C; ++c) H[c] = WP_NEG;
		}
		uint32_t tw[C / 4];
#pragma unroll
		for (int j = 0; j < C / 4; ++j) tw[j] = tn >= 0 && tn < T ? reinterpret_cast<const uint32_t*>(sf_g + tn)[j] : 0u;
#pragma unroll
		for (int j = 0; j < C / 4; ++j) TB[2 * j] = (tw[j] & 0xffu) | (tw[j] & 0xff00u) << 8, TB[2 * j + 1] = (tw[j] >> 16 & 0xffu) | (tw[j] >> 24) << 16;
#pragma unroll
		for (int k = 0; k < NP; ++k) {
			int i0 = qlen - 1 - r + tn + 2 * k, i1 = i0 + 1;
			i0 = i0 < 0 ? 0 : i0 > Q - 1 ? Q - 1 : i0, i1 = i1 < 0 ? 0 : i1 > Q - 1 ? Q - 1 : i1;
			QB[k] = (uint32_t)sq[i0] | (uint32_t)sq[i1] << 16;
		}
	};
	int t0 = -16 + ((tid * C + 16) & M);
	take_cells(t0, 0);
	// one cell's value out of a packed array: lane and pair are wave-uniform
	auto cell_value = [&](const pvec_t &A, int t) -> int {   // (one wave)
		const int L = (t & M) / C, kk = (t & CB) >> 1;
		const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)A[kk], L);
		return kpk::value_of(v, t & 1);
	};
	auto cell_to_lds = [&](const pvec_t &A, int t, int slot) {   // (several waves: the owner writes, everybody reads behind the barrier)
		if (tid == (t & M) / C) one[slot] = kpk::value_of(A[(t & CB) >> 1], t & 1);
	};
	auto lds_uniform = [&](lds_i32p ptr) { return __builtin_amdgcn_readfirstlane(*ptr); };
	int last_st = -1, last_en = -1, H0 = 0, last_H0_t = 0, p_st0 = 0;
	const int n_r = qlen + tlen - 1;
	WG_T(tm1);
	int r_done = 0;
	for (int r = 0; r < n_r; ++r) {
		r_done = r;
		int st = 0, en = tlen - 1;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
		if (en > (r + wl) >> 1) en = (r + wl) >> 1;
		if (st > en) { z_zdropped = 1; break; }
		const int st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		const int v_edge = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		// ---- the cell below this lane's first, as the last step left it (its values in the HIGH halves)
		if (NW > 1) {                                          // [1]
			if (lane == 63) {
				xw[wv * 4 + 0] = (int)X[NP - 1], xw[wv * 4 + 1] = (int)V[NP - 1], xw[wv * 4 + 2] = (int)X2[NP - 1];
				if (!APPROX) xw[wv * 4 + 3] = H[C - 1];
			}
			MNC_LDS_BARRIER();
		}
		uint32_t nbX = (uint32_t)MNC_DPP_ROR1((int)X[NP - 1]), nbV = (uint32_t)MNC_DPP_ROR1((int)V[NP - 1]), nbX2 = (uint32_t)MNC_DPP_ROR1((int)X2[NP - 1]);
		int nbH = 0;
		if (!APPROX) nbH = MNC_DPP_ROR1(H[C - 1]);
		if (NW > 1 && lane == 0) {
			const int pw = (wv + NW - 1) & (NW - 1);
			nbX = (uint32_t)xw[pw * 4 + 0], nbV = (uint32_t)xw[pw * 4 + 1], nbX2 = (uint32_t)xw[pw * 4 + 2];
			if (!APPROX) nbH = xw[pw * 4 + 3];
		}
		// ---- the query window moves down one base; the window of cells moves up with st
		if (r > 0) {
			int qi = qlen - 1 - r + t0;
			qi = qi < 0 ? 0 : qi > Q - 1 ? Q - 1 : qi;
			const uint32_t qn = sq[qi];
#pragma unroll
			for (int k = NP - 1; k > 0; --k) QB[k] = kpk::shift16(QB[k], QB[k - 1]);
			QB[0] = QB[0] << 16 | qn;
		}
		{
			const int base = st - 16;
			const int tn = base + ((tid * C - base) & M);
			if (tn != t0) { t0 = tn; take_cells(tn, r); }
		}
		if (t0 == st) {                                        // ksw2's x1, x21, v1
			if (st > 0) {
				if (!(st - 1 >= last_st && st - 1 <= last_en)) nbX = K.ix, nbX2 = K.ix2, nbV = K.iuv;
			} else nbX = K.ix, nbX2 = K.ix2, nbV = kpk::lane(v_edge, 0);
		}
		if (en >= r) {                                         // y[r], y2[r], u[r]: the cell the virtual row enters at
			const int kk = (r & CB) >> 1;
			const uint32_t m = tid == (r & M) / C ? ((r & 1) ? 0xffff0000u : 0x0000ffffu) : 0u;
			const uint32_t ue = kpk::lane(v_edge, 0);
			Y[kk] = kpk::bitsel(m, K.iy, Y[kk]), Y2[kk] = kpk::bitsel(m, K.iy2, Y2[kk]), U[kk] = kpk::bitsel(m, ue, U[kk]);
		}
		// ---- scores, fresh on [st0, lim) only: both ends share their offset in a block of C cells (lim - st0 is a multiple
		// of 16), so a lane's mask is one of {none, all, cells >= oc, cells < oc} of ONE wave-uniform pattern
		{
			const int lim = st0 + ((en0 - st0) / 16 + 1) * 16, oc = st0 & CB, sblk = st0 & ~CB, lblk = lim & ~CB;
			const bool is_lo = t0 == sblk, is_hi = t0 == lblk, is_mid = t0 > sblk && t0 < lblk;
			const uint32_t ma = (is_lo || is_hi) ? 0xffffffffu : 0u, mb = (is_hi || is_mid) ? 0xffffffffu : 0u;
#pragma unroll
			for (int k = 0; k < NP; ++k) {
				const uint32_t pat = (2 * k >= oc ? 0x0000ffffu : 0u) | (2 * k + 1 >= oc ? 0xffff0000u : 0u);    // uniform
				const uint32_t mask = (pat & ma) ^ mb;
				S[k] = kpk::bitsel(mask, kpk::scores(K, TB[k], QB[k]), S[k]);
			}
		}
		// ---- the cells: all C of a lane are inside [st, en] or none (both ends are multiples of 16, C divides 16)
		const bool in_row = t0 >= st && t0 <= en;
		if (in_row) {
			uint32_t d[NP];
#pragma unroll
			for (int k = NP - 1; k >= 0; --k) {                   // from the top: [k - 1] is still old
				const uint32_t xb = kpk::shift16(X[k], k ? X[k - 1] : nbX), vb = kpk::shift16(V[k], k ? V[k - 1] : nbV), x2b = kpk::shift16(X2[k], k ? X2[k - 1] : nbX2);
				uint32_t cu = U[k], cv = V[k], cx = X[k], cy = Y[k], cx2 = X2[k], cy2 = Y2[k];
				d[k] = kpk::cell_pair<RIGHT>(K, xb, vb, x2b, S[k], cu, cv, cx, cy, cx2, cy2);
				U[k] = cu, V[k] = cv, X[k] = cx, Y[k] = cy, X2[k] = cx2, Y2[k] = cy2;
			}
			uint32_t dw[C / 4];
#pragma unroll
			for (int j = 0; j < C / 4; ++j) dw[j] = __builtin_amdgcn_perm(d[2 * j + 1], d[2 * j], 0x06040200u);
			uint32_t *pr = reinterpret_cast<uint32_t*>(&p[(size_t)r * ncol + (size_t)(t0 - st)]);
			if constexpr (C == 16) *reinterpret_cast<uint4*>(pr) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
			else if constexpr (C == 8) *reinterpret_cast<uint2*>(pr) = make_uint2(dw[0], dw[1]);
			else pr[0] = dw[0];
		}
		if (!APPROX) {
			int max_H, max_t = 0;
			if (r > 0) {
				// H[en0] = the OLD H[en0 - 1] + u[en0]; H[t] += v[t] on [st0, en0); the maximum in the SSE scan's tie order.
				// Every lane adds v to all its cells: a cell above en0 is ASSIGNED when the band reaches it, and until then it
				// stays near WP_NEG; a cell that has just fallen below st0 is sent back there, so the maximum over all cells
				// of the wave is the maximum over [st0, en0].
				const int ce = en0 & CB, Le = (en0 & M) / C;
				const int he = (ce ? H[(ce - 1) & CB] : nbH) + kpk::value_of(U[ce >> 1], ce & 1);
#pragma unroll
				for (int c = 0; c < C; ++c) H[c] += kpk::value_of(V[c >> 1], c & 1);
				{                                                    // (one unconditional insert each: a conditional one copies the vector)
					const int hv = (en0 > 0 && tid == Le) ? he : H[ce];
					H[ce] = hv;
					const int cp = (st0 - 1) & CB, Lp = ((st0 - 1) & M) / C;
					const int pv = (st0 > p_st0 && tid == Lp) ? WP_NEG : H[cp];
					H[cp] = pv;
				}
				int tm = H[0];
#pragma unroll
				for (int c = 1; c < C; ++c) tm = tm > H[c] ? tm : H[c];
				max_H = wave_max_dpp(tm);
				if (NW > 1) {                                      // [2]
					if (lane == 0) red[wv] = max_H;
					if (tid == (st0 & M) / C) one[0] = H[st0 & CB];
					if (tid == ((tlen - 1) & M) / C) one[1] = H[(tlen - 1) & CB];
					MNC_LDS_BARRIER();
					max_H = lds_uniform(red);
#pragma unroll
					for (int k = 1; k < NW; ++k) { const int o = lds_uniform(red + k); max_H = max_H > o ? max_H : o; }
				}
				// where: only a new maximum or a possible Z-drop asks (ksw_apply_zdrop reads max_t in no other case)
				if (max_H > z_max || (zdrop >= 0 && z_max - max_H > zdrop)) {
					// every lane marks its cells that hold the maximum (one word); the few lanes that have any -- one, as a rule --
					// are looked at one by one on the scalar unit, their cells ranked in the scan's order
					const int en1 = st0 + (en0 - st0) / 4 * 4, base = st - 16;
					unsigned cmask = 0;
#pragma unroll
					for (int c = 0; c < C; ++c) cmask |= H[c] == max_H ? 1u << c : 0u;
					unsigned long long lanes = __ballot(cmask != 0);
					unsigned best = 0xffffffffu;
					while (lanes) {
						const int L = __builtin_ctzll(lanes);
						lanes &= lanes - 1;
						unsigned cm = (unsigned)__builtin_amdgcn_readlane((int)cmask, L);
						const int tl = base + (((wv * 64 + L) * C - base) & M);
						while (cm) {
							const int t = tl + __builtin_ctz(cm);
							cm &= cm - 1;
							const unsigned rank = t == en0 ? 0u : t < en1 ? 1u + ((unsigned)(t - st0) & 3u) * 0x1000000u + ((unsigned)(t - st0) >> 2) : 1u + 4u * 0x1000000u + (unsigned)(t - en1);
							if (t >= st0 && t <= en0 && rank < best) best = rank, max_t = t;
						}
					}
					if (NW > 1) {                                  // [3]
						if (lane == 0) red[4 + wv * 2] = (int)best, red[5 + wv * 2] = max_t;
						MNC_LDS_BARRIER();
#pragma unroll
						for (int k = 0; k < NW; ++k) {
							const unsigned ob = (unsigned)lds_uniform(red + 4 + k * 2);
							const int ot = lds_uniform(red + 5 + k * 2);
							if (k == 0 || ob < best) best = ob, max_t = ot;
						}
					}
				}
			} else {
				if (tid == 0) H[0] = kpk::value_of(V[0], 0) - qe;       // t0 == 0 there
				if (NW > 1) {
					if (tid == 0) one[0] = H[0];
					MNC_LDS_BARRIER();
					max_H = lds_uniform(one), max_t = 0;
				} else max_H = __builtin_amdgcn_readlane(H[0], 0), max_t = 0;
			}
			if (r - st0 == qlen - 1) {
				const int cs = st0 & CB, Ls = (st0 & M) / C;
				const int hs = NW > 1 ? lds_uniform(one) : __builtin_amdgcn_readlane(H[cs], Ls);
				if (hs > z_mqe) z_mqe = hs, z_mqe_t = st0;
			}
			// ksw_apply_zdrop
			if (max_H > z_max) z_max = max_H, z_max_t = max_t, z_max_q = r - max_t;
			else if (zdrop >= 0 && z_max - max_H > zdrop && max_t >= z_max_t && r - max_t >= z_max_q) {
				const int tl = max_t - z_max_t, ql = (r - max_t) - z_max_q;
				const int l = tl > ql ? tl - ql : ql - tl;
				if (z_max - max_H > zdrop + l * e2) { z_zdropped = 1; break; }
			}
			if (r == qlen + tlen - 2 && en0 == tlen - 1) {
				const int cs = (tlen - 1) & CB, Ls = ((tlen - 1) & M) / C;
				z_score = NW > 1 ? lds_uniform(one + 1) : __builtin_amdgcn_readlane(H[cs], Ls);
			}
		} else {
			if (NW > 1) {                                          // [2]: v[last_H0_t], u[last_H0_t + 1] of this step
				cell_to_lds(V, r > 0 ? last_H0_t : 0, 2), cell_to_lds(U, last_H0_t + 1, 3);
				MNC_LDS_BARRIER();
			}
			auto v_at = [&](int t) { return NW > 1 ? lds_uniform(one + 2) : cell_value(V, t); };
			auto u_at = [&](int t) { return NW > 1 ? lds_uniform(one + 3) : cell_value(U, t); };
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					const int d0 = v_at(last_H0_t), d1 = u_at(last_H0_t + 1);
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) {
					H0 += v_at(last_H0_t);
				} else {
					++last_H0_t, H0 += u_at(last_H0_t);
				}
			} else H0 = v_at(0) - qe, last_H0_t = 0;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) z_score = H0;
		}
		last_st = st, last_en = en, p_st0 = st0;
	}
	__syncthreads();                                          // the direction codes are written; the query's LDS is free
	WG_T(tm2);
	(void)r_done;
	int i0 = -1, j0 = -1;
	if (!z_zdropped && !(flag & EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
	else if (!z_zdropped && (flag & EZ_EXTZ_ONLY) && z_mqe + end_bonus > z_max) z_reach_end = 1, i0 = z_mqe_t, j0 = qlen - 1;
	else if (z_max_t >= 0 && z_max_q >= 0) i0 = z_max_t, j0 = z_max_q;
	int n_cigar = 0;
	if (i0 >= 0 && j0 >= 0) n_cigar = walk_wg<NW, PP, CP, RIGHT ? 2 : 1>(qlen, tlen, wl, wr, ncol, p, cig, i0, j0, flag, sq, bc);
	n_cigar = wg_bcast0<NW>(n_cigar);
	WG_T(tm3);
	WG_ADD(0, tm1 - tm0); WG_ADD(1, tm2 - tm1); WG_ADD(2, tm3 - tm2); WG_ADD(3, 1); WG_ADD(4, r_done);
	ez_out.max = z_max, ez_out.zdropped = z_zdropped, ez_out.max_q = z_max_q, ez_out.max_t = z_max_t, ez_out.mqe = z_mqe, ez_out.mqe_t = z_mqe_t;
	ez_out.score = z_score, ez_out.reach_end = z_reach_end, ez_out.n_cigar = n_cigar;
	st_order<false, NW>();
#undef MNC_LDS_BARRIER
}

// mm_test_zdrop on a finished gap-filling CIGAR: 0 fine, 1 the score drops by more than zdrop, 2 and
// the dropped stretch aligns to its own reverse complement.  One lane; sequences come from `mem`.
template <class SU8, class CP>
__device__ int test_zdrop_lane0(const Batch &B, int qlen, int tlen, SU8 sf, SU8 qr, int n_cigar, CP cig,
                                int sc_mch, int sc_mis, int sc_N, int32_t *sw_H, int32_t *sw_E)
{
	int score = 0, mx = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
	int pos00 = -1, pos01 = -1, pos10 = -1, pos11 = -1;
	auto upd = [&](int sc, int ii, int jj) {
		if (sc < mx) {
			const int li = ii - max_i, lj = jj - max_j;
			const int diff = li > lj ? li - lj : lj - li;
			const int z = mx - sc - diff * B.gap_e;
			if (z > max_zdrop) max_zdrop = z, pos00 = max_i, pos01 = ii + 1, pos10 = max_j, pos11 = jj + 1;
		} else mx = sc, max_i = ii, max_j = jj;
	};
	for (int k = 0; k < n_cigar; ++k) {
		const uint32_t op = cig[k] & 0xf, len = cig[k] >> 4;
		if (op == 0) {
			for (uint32_t l = 0; l < len; ++l) {
				const int ct = sf[i + l], cq = qr[qlen - 1 - (j + (int)l)];
				score += (ct == 4 || cq == 4) ? sc_N : ct == cq ? sc_mch : sc_mis;
				upd(score, i + (int)l, j + (int)l);
			}
			i += len, j += len;
		} else {
			score -= B.gap_q + B.gap_e * (int)len;
			if (op == 1) j += len; else i += len;
			upd(score, i, j);
		}
	}
	const int q_len = pos11 - pos10, t_len = pos01 - pos00;
	if (max_zdrop > B.zdrop_inv && q_len < B.max_gap && t_len < B.max_gap) {
		// ksw_ll_i16 of the reverse complement of the query stretch against the target stretch: the
		// best local score with one affine gap cost (only the score matters here)
		int best = 0;
		for (int jj = 0; jj <= q_len; ++jj) sw_H[jj] = 0, sw_E[jj] = 0;
		for (int ii = 0; ii < t_len; ++ii) {
			int f = 0, diag = 0;
			const int ct = sf[pos00 + ii];
			for (int jj = 0; jj < q_len; ++jj) {
				const int c0 = qr[qlen - 1 - (pos11 - jj - 1)];
				const int cq = c0 >= 4 ? 4 : 3 - c0;
				int h = diag + ((ct == 4 || cq == 4) ? sc_N : ct == cq ? sc_mch : sc_mis);
				diag = sw_H[jj + 1];
				h = h > sw_E[jj + 1] ? h : sw_E[jj + 1];
				h = h > f ? h : f;
				h = h > 0 ? h : 0;
				sw_H[jj + 1] = h;
				best = best > h ? best : h;
				int t = h - (B.gap_q + B.gap_e);
				t = t > 0 ? t : 0;
				sw_E[jj + 1] = sw_E[jj + 1] - B.gap_e > t ? sw_E[jj + 1] - B.gap_e : t;
				f = f - B.gap_e > t ? f - B.gap_e : t;
			}
		}
		if (best > 32767) best = 32767;
		if (best >= B.min_sc * B.sc_a && best >= B.min_dp_max) return 2;
	}
	return max_zdrop > B.zdrop ? 1 : 0;
}

// workspace of one workgroup: [state bytes | H | CIGAR scratch | local-SW rows | direction bytes]
struct AlignWs { size_t state, h, cig, sw, p, total; };
__host__ __device__ __forceinline__ AlignWs align_ws(long long state_max, long long p_max, long long cig_max)
{
	AlignWs w;
	w.state = 0;
	w.h = (size_t)(state_max + 255) / 256 * 256;
	w.cig = w.h + (size_t)(state_max / 3 + 255) / 256 * 256;            // H: 4 T <= state / 3
	w.sw = w.cig + (size_t)cig_max * 4;
	w.p = w.sw + (size_t)cig_max * 8;
	w.total = (w.p + (size_t)p_max + 4095) / 4096 * 4096;
	return w;
}

// LDS of a workgroup: [state: lds_bytes | direction bytes: lds_p | CIGAR: lds_cig words].  Pass 0 runs
// with a small all-in-LDS layout (many workgroups per CU: the kernel is a chain of LDS round
// trips, only other waves hide them) and hands what does not fit to pass 2's list.
// NW: waves per call.  The launches for the long calls (the workspace layouts) put four waves on one: a call of 1 500 x
// 3 000 bases is 4 500 anti-diagonals of up to 751 cells, 27 ms on a single wave that a batch of divergent reads waits for.
// C > 0: the calls that fit run with their cells in registers (ksw_wg<NW, C>, `lds_bytes` = LDS bytes per sequence there).
// regime 1: this launch only works when the pass has at most `regime_n` calls, regime 2: only when it has more -- the long
// calls are launched in both forms (few calls: eight waves with two cells a thread, every call as fast as it goes; many: four waves with
// four cells a thread, four times as many side by side), and the count is only known on the device.
// one queue of calls (`big_pass`) on this workgroup's workspace slot `ws`
// PK: the cells as packed 16-bit pairs in registers (ksw_wp: one wave a call, or NW waves on one call)
template <int NW, int C, bool PK = (NW == 1 && C > 0)>
__device__ __forceinline__ void dp_align_queue(const Batch &B, uint8_t *smem, uint8_t *ws, long long state_max, long long p_max, long long cig_max,
                                               int lds_bytes, int lds_p, int lds_cig, int big_pass, int regime, int regime_n)
{
	const int lane = threadIdx.x;
	const AlignWs W = align_ws(state_max, p_max, cig_max);
	const int sc_mch = B.sc_a, sc_mis = -B.sc_b, sc_N = -B.sc_ambi;
	// pass 0: the segments of the round that are neither large nor another kernel's and fit the LDS
	// layout; 3: those that do not; 1: the large ones; 5: the few largest; 2: what the banded kernels handed back (4: those
	// of them that need the large workspace)
	if (big_pass != 0) __builtin_amdgcn_s_setprio(3);         // a few long calls on single waves beside chip-filling kernels: first in line for issue
	const unsigned long long n_items = big_pass == 0 ? B.dp_ctr[20] : big_pass == 1 ? B.dp_ctr[6] : big_pass == 3 ? B.dp_ctr[28] : big_pass == 4 ? B.dp_ctr[56] : big_pass == 5 ? B.dp_ctr[58] : B.dp_ctr[12];
	const int ctr_q = big_pass == 0 ? 21 : big_pass == 1 ? 7 : big_pass == 3 ? 29 : big_pass == 4 ? 57 : big_pass == 5 ? 59 : 15;
	if ((regime == 1 && n_items > (unsigned long long)regime_n) || (regime == 2 && n_items <= (unsigned long long)regime_n)) return;   // the other form's
	for (;;) {
		unsigned long long qi = 0;
		if (lane == 0) {
			if (big_pass == 4) {
				// this queue is drained twice -- inside the window (what the long kernels have handed back by then) and behind
				// its join (the rest): a claim that fails must leave the counter at the calls really taken
				unsigned long long cur = atomicAdd(&B.dp_ctr[ctr_q], 0ULL);
				for (;;) {
					if (cur >= n_items) { qi = n_items; break; }
					const unsigned long long old = atomicCAS(&B.dp_ctr[ctr_q], cur, cur + 1ULL);
					if (old == cur) { qi = cur; break; }
					cur = old;
				}
			} else qi = atomicAdd(&B.dp_ctr[ctr_q], 1ULL);
		}
		qi = wg_bcast0_u64<NW>(qi);
		if (qi >= n_items) break;                              // every wave reaches this: the queue is finite
		const long long si = big_pass == 0 ? (long long)B.gen_list[qi] : big_pass == 1 ? (long long)B.big_list[qi] : big_pass == 3 ? (long long)B.mid_list[qi] : big_pass == 4 ? (long long)B.bigfb_list[qi] : big_pass == 5 ? (long long)B.huge_list[qi] : (long long)B.fill_fb[qi];
		Seg g = B.segs[si];
		Ez ez;
		ez.max = 0, ez.zdropped = 0, ez.max_q = ez.max_t = ez.mqe_t = -1, ez.mqe = ez.score = DP_NEG_INF, ez.reach_end = 0, ez.n_cigar = 0;
		int zdrop_code = 0;
		unsigned long long off = 0;
		if (g.qlen <= 0 || g.tlen <= 0) {
			// ksw_extd2 returns at once
		} else if ((long long)g.tlen * g.qlen > B.max_sw_mat || (g.flag & SEG_SKIPPED)) {
			ez.zdropped = 1;                                   // mm_align_pair: too large, treated as a Z-drop (SEG_SKIPPED: too large for this library: the read is reported as skipped)
		} else {
			const int T = (g.tlen + 15) / 16 * 16, Q = (g.qlen + 15) / 16 * 16 + 32;
			int ncw = g.qlen < g.tlen ? g.qlen : g.tlen;
			{ const int wb = g.w < 0 ? (g.tlen > g.qlen ? g.tlen : g.qlen) : g.w; ncw = ((ncw < wb + 1 ? ncw : wb + 1) + 15) / 16 + 1; }
			const long long p_bytes = ((long long)(g.qlen + g.tlen - 1) * ncw + 1) * 16;
			const bool in_lds = NW == 1 && C == 0 && !PK && 12 * T + Q <= lds_bytes;    // (one wave with C > 0: the packed form -- LDS holds the query and the walk's tile, the state is in registers, sequences and direction codes in the workspace)
			// small calls (the extensions of most reads) keep direction bytes and CIGAR in LDS as well
			const bool all_lds = in_lds && p_bytes <= lds_p && g.qlen + g.tlen + 2 <= lds_cig;
			if (big_pass == 0 && !all_lds) {                        // not for the small layout: pass 2 takes it
				if (lane == 0) { const unsigned long long k = atomicAdd(&B.dp_ctr[12], 1ULL); B.fill_fb[k] = (int32_t)si; }
				continue;
			}
			if (!all_lds && ((!in_lds && 12LL * T + Q > state_max) || p_bytes > p_max || g.qlen + g.tlen + 8 > cig_max)) {
				// a call that does not fit this launch's workspace slot must never get here (mnc_dp_plan sorts them);
				// should one, the batch fails instead of writing past the slot
				if (lane == 0) atomicMax(&B.dp_ctr[4], 9ULL);
				continue;
			}
			const uint8_t *read = B.bases + B.offsets[g.read];
			const int rlen = (int)(B.offsets[g.read + 1] - B.offsets[g.read]);
			const int64_t coff = B.seq_off[g.rid];
			auto run = [&](auto mem, auto H, auto pbuf, auto cg) {
				typedef typename PtrTraits<decltype(mem)>::u8 SU8;
				SU8 sf = (SU8)(mem + 7 * (size_t)T), qr = sf + T;
				WG_T(ts0);
				// target / reversed query; the left extension runs on both sequences reversed
				int any_ambi = 0;
				for (int i = lane; i < T; i += 64 * NW) { const int c = i < g.tlen ? tcode(B, coff, g.kind == 0 ? g.ts + g.tlen - 1 - i : g.ts + i) : 0; sf[i] = (uint8_t)c, any_ambi |= c >> 2; }
				for (int i = lane; i < Q; i += 64 * NW) {
					const int c = i < g.qlen ? qcode(read, rlen, g.rev, g.kind == 0 ? g.qs + i : g.qs + g.qlen - 1 - i) : 0;
					qr[i] = (uint8_t)c, any_ambi |= c >> 2;
				}
				st_order<false, NW>();
				WG_T(ts1);
				WG_ADD(6, ts1 - ts0);
				// the cells in registers (ksw_wg: NW waves, C cells a thread; one wave: ksw_wp, C cells a lane as packed pairs)
				// unless the call's anti-diagonals or sequences outgrow that form; debug_route bit 6: never (the workspace form
				// on this many waves, for the tests)
				bool on_wg = false;
				if constexpr (C > 0 && !PK) on_wg = !(B.debug_route & 64) && wg_fits<NW, C>(g.qlen, g.tlen, g.w, lds_bytes);
				// (the packed form: no ambiguous base in either sequence, scores that keep every intermediate inside int8)
				if constexpr (PK) {
					if (NW > 1) {                                   // one flag for the workgroup: every wave takes the same form
						__shared__ int s_ambi;
						__syncthreads();
						if (threadIdx.x == 0) s_ambi = 0;
						__syncthreads();
						if (any_ambi) s_ambi = 1;
						__syncthreads();
						any_ambi = s_ambi;
					}
					on_wg = !(B.debug_route & 64) && !__any(any_ambi) && wp_fits<C, NW>(g.qlen, g.tlen, g.w, lds_bytes) &&
					        kpk::params_fit(B.gap_q, B.gap_e, B.gap_q2, B.gap_e2, sc_mch, sc_mis, sc_N);
				}
				auto call = [&](int zdrop, int end_bonus, int flag) {
					if constexpr (C > 0 && !PK) {
						if (on_wg) {
							ksw_wg<NW, C>(g.qlen, g.tlen, (const uint8_t*)sf, (const uint8_t*)qr, (lds_u8p)smem, lds_bytes, pbuf, cg, B.gap_q, B.gap_e, B.gap_q2, B.gap_e2,
							              sc_mch, sc_mis, sc_N, g.w, zdrop, end_bonus, flag, ez);
							return;
						}
					}
					if constexpr (PK) {
						if (on_wg) {
							// as few cells a lane as hold the call's widest anti-diagonal: a step costs a lane's pairs plus a fixed
							// part, and a lone call's length is its steps times that (4 cells: up to 193 wide, 8: 449, 16: 961)
#define MNC_WP_CALL(CC, R, A) ksw_wp<CC, R, A, NW>(g.qlen, g.tlen, (const uint8_t*)sf, (const uint8_t*)qr, (lds_u8p)smem, pbuf, cg, B.gap_q, B.gap_e, B.gap_q2, B.gap_e2, sc_mch, sc_mis, g.w, zdrop, end_bonus, flag, ez)
#define MNC_WP_FLAGS(CC) do { if (flag & EZ_RIGHT) { if (flag & EZ_APPROX_MAX) MNC_WP_CALL(CC, true, true); else MNC_WP_CALL(CC, true, false); } \
else { if (flag & EZ_APPROX_MAX) MNC_WP_CALL(CC, false, true); else MNC_WP_CALL(CC, false, false); } } while (0)
							if constexpr (NW == 1 && C == 16) {
								if (wp_fits<4>(g.qlen, g.tlen, g.w, lds_bytes)) MNC_WP_FLAGS(4);
								else if (wp_fits<8>(g.qlen, g.tlen, g.w, lds_bytes)) MNC_WP_FLAGS(8);
								else MNC_WP_FLAGS(16);
							} else MNC_WP_FLAGS(C);
#undef MNC_WP_FLAGS
#undef MNC_WP_CALL
							return;
						}
					}
					ksw_wave<NW>(g.qlen, g.tlen, mem, H, pbuf, cg, B.gap_q, B.gap_e, B.gap_q2, B.gap_e2, sc_mch, sc_mis, sc_N, g.w, zdrop, end_bonus, flag, ez);
				};
				call(g.zdrop, g.kind == 1 ? -1 : B.end_bonus, g.flag);
				if (g.kind == 1) {
					// the kernel's last 16-lane score store may spill into the first 15 target bytes (as in the
					// SSE buffer, where those are dead by then): restore them for the walk and the second pass
					if (lane < 16) sf[lane] = lane < g.tlen ? (uint8_t)tcode(B, coff, g.ts + lane) : 0;
					st_order<false, NW>();
					WG_T(tz0);
					if (lane == 0) zdrop_code = test_zdrop_lane0(B, g.qlen, g.tlen, sf, qr, ez.n_cigar, cg, sc_mch, sc_mis, sc_N,
					                                              reinterpret_cast<int32_t*>(ws + W.sw), reinterpret_cast<int32_t*>(ws + W.sw) + cig_max);
					WG_T(tz1);
					WG_ADD(5, tz1 - tz0);
					zdrop_code = wg_bcast0<NW>(zdrop_code);
					if (zdrop_code != 0)                             // second pass: exact maximum, real Z-drop
						call(zdrop_code == 2 ? B.zdrop_inv : B.zdrop, -1, 0);
				}
				// the CIGAR goes to the segment pool
				if (lane == 0 && ez.n_cigar > 0) off = atomicAdd(&B.dp_ctr[1], (unsigned long long)ez.n_cigar);
				off = wg_bcast0_u64<NW>(off);
				if (ez.n_cigar > 0) {
					if ((long long)(off + ez.n_cigar) > B.cig_seg_cap) {
						if (lane == 0) atomicMax(&B.dp_ctr[4], 2ULL);
						ez.n_cigar = 0;
					} else for (int k = lane; k < ez.n_cigar; k += 64 * NW) B.cig_seg[off + k] = cg[k];
				}
			};
			const size_t h_off = (size_t)(8 * T + Q + 15) / 16 * 16;
			if constexpr (NW == 1 && C == 0) {
				if (all_lds)
					run((lds_i8p)smem, (lds_i32p)(smem + h_off), (lds_u8p)(smem + lds_bytes), (lds_u32p)(smem + lds_bytes + lds_p));
				else if (in_lds)
					run((lds_i8p)smem, (lds_i32p)(smem + h_off), ws + W.p, reinterpret_cast<uint32_t*>(ws + W.cig));
				else
					run(reinterpret_cast<int8_t*>(ws + W.state), reinterpret_cast<int32_t*>(ws + W.h), ws + W.p, reinterpret_cast<uint32_t*>(ws + W.cig));
			} else {
				(void)h_off;
				run(reinterpret_cast<int8_t*>(ws + W.state), reinterpret_cast<int32_t*>(ws + W.h), ws + W.p, reinterpret_cast<uint32_t*>(ws + W.cig));
			}
		}
		if (lane == 0) {
			Seg *o = B.segs + si;
			o->n_cigar = ez.n_cigar, o->zdropped = ez.zdropped, o->zdrop_code = zdrop_code;
			o->max = ez.max, o->max_t = ez.max_t, o->max_q = ez.max_q, o->score = ez.score, o->reach_end = ez.reach_end, o->mqe_t = ez.mqe_t;
			o->cig_off = (int64_t)off;
		}
		st_order<false, NW>();
	}
}

template <int NW, int C = 0, bool PK = (NW == 1 && C > 0)>
__global__ __launch_bounds__(64 * NW) void mnc_dp_align(Batch B, uint8_t *ws_all, long long state_max, long long p_max, long long cig_max,
                                                   int lds_bytes, int lds_p, int lds_cig, int big_pass, int regime = 0, int regime_n = 0)
{
	extern __shared__ __align__(16) uint8_t smem[];
	const AlignWs W = align_ws(state_max, p_max, cig_max);
	dp_align_queue<NW, C, PK>(B, smem, ws_all + (size_t)blockIdx.x * W.total, state_max, p_max, cig_max, lds_bytes, lds_p, lds_cig, big_pass, regime, regime_n);
}

// The long passes (1: large workspace, 3: small workspace but not LDS, 5: the few largest) as ONE launch of one-wave
// workgroups that stay until all three queues are empty (round 5).  As three launches one behind the other on their
// stream, the second and third found the chip taken by the persistent workgroups of the gap-filling tiers -- launched
// meanwhile -- and their waves got on a SIMD only as those retired.  A workgroup owns a workspace slot of one class and
// takes every call its slot holds: the few with the largest slots first those calls, then the large ones, then the rest.
struct LongWs { uint8_t *ws[3]; long long state[3], p[3], cig[3]; int n_wg[3]; };      // classes: 0 the largest slots, 1 large, 2 small
template <int C>
__global__ __launch_bounds__(64) void mnc_dp_align_long(Batch B, LongWs L, int lds_bytes)
{
	extern __shared__ __align__(16) uint8_t smem[];
	int cls = 0, idx = (int)blockIdx.x;
	while (cls < 2 && idx >= L.n_wg[cls]) idx -= L.n_wg[cls], ++cls;
	const AlignWs W = align_ws(L.state[cls], L.p[cls], L.cig[cls]);
	uint8_t *ws = L.ws[cls] + (size_t)idx * W.total;
	if (cls == 0) dp_align_queue<1, C>(B, smem, ws, L.state[cls], L.p[cls], L.cig[cls], lds_bytes, 0, 0, 5, 0, 0);
	if (cls <= 1) dp_align_queue<1, C>(B, smem, ws, L.state[cls], L.p[cls], L.cig[cls], lds_bytes, 0, 0, 1, 0, 0);
	dp_align_queue<1, C>(B, smem, ws, L.state[cls], L.p[cls], L.cig[cls], lds_bytes, 0, 0, 3, 0, 0);
}

// ---- wave-wide scans with DPP (GFX9: row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143).
// A lane that receives nothing keeps `old`: the identity of the operation.
#define MNC_DPP(old, src, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (src), (ctrl), (rmask), 0xf, false)
__device__ __forceinline__ int dpp_incl_add(int v)
{
	v += MNC_DPP(0, v, 0x111, 0xf); v += MNC_DPP(0, v, 0x112, 0xf); v += MNC_DPP(0, v, 0x114, 0xf); v += MNC_DPP(0, v, 0x118, 0xf);
	v += MNC_DPP(0, v, 0x142, 0xa); v += MNC_DPP(0, v, 0x143, 0xc);
	return v;
}
__device__ __forceinline__ int dpp_max_all(int v)          // the maximum over the wave, in every lane
{
	const int lo = INT32_MIN;
	int o;
	o = MNC_DPP(lo, v, 0x111, 0xf); v = v > o ? v : o;  o = MNC_DPP(lo, v, 0x112, 0xf); v = v > o ? v : o;
	o = MNC_DPP(lo, v, 0x114, 0xf); v = v > o ? v : o;  o = MNC_DPP(lo, v, 0x118, 0xf); v = v > o ? v : o;
	o = MNC_DPP(lo, v, 0x142, 0xa); v = v > o ? v : o;  o = MNC_DPP(lo, v, 0x143, 0xc); v = v > o ? v : o;
	return __builtin_amdgcn_readlane(v, 63);
}
// inclusive scan of maps x -> max(x + a, b) under composition "earlier, then later":
// (a1, b1) then (a2, b2) = (a1 + a2, max(b1 + a2, b2)); identity (0, SC_NONE)
constexpr int SC_NONE = -(1 << 28);
__device__ __forceinline__ void dpp_scan_maps(int &a, int &b)
{
#define MNC_SCAN_STEP(ctrl, rmask) { const int pa = MNC_DPP(0, a, ctrl, rmask), pb = MNC_DPP(SC_NONE, b, ctrl, rmask); \
	const int nb = pb + a > b ? pb + a : b; a = pa + a, b = nb; }
	MNC_SCAN_STEP(0x111, 0xf) MNC_SCAN_STEP(0x112, 0xf) MNC_SCAN_STEP(0x114, 0xf) MNC_SCAN_STEP(0x118, 0xf)
	MNC_SCAN_STEP(0x142, 0xa) MNC_SCAN_STEP(0x143, 0xc)
#undef MNC_SCAN_STEP
}

// ================================================================ stitch: one region (the rest of mm_align1 + mm_update_extra)
// LDS: the region's query and target codes (when they fit), the joined CIGAR is built in the
// region pool.  One wave per region; lane 0 does the sequential bookkeeping, all lanes the walks.
constexpr int ST_CIG_MAX = 1024;                            // CIGAR words in LDS
constexpr int ST_EV_MAX = 1024;                             // score events (one byte each) buffered for one scan
constexpr int ST_EV_LONG = 8192, ST_LONG_Q = 61440;         // ... and in the launch for the regions of long reads; its longest query in LDS
constexpr unsigned long long ST_POOL_CHUNK = 4096;           // words of the region CIGAR pool a wave reserves at a time
constexpr int ST_EV_LIM = 127;                              // a gap that costs more is applied directly

__device__ __forceinline__ void append_op(uint32_t *c, int &n, uint32_t word)
{
	if (n > 0 && (c[n - 1] & 0xf) == (word & 0xf)) c[n - 1] += word >> 4 << 4;
	else c[n++] = word;
}

__global__ __launch_bounds__(64, 4) void mnc_dp_stitch(Batch B, const int32_t *work_list, int32_t *next_list, int seq_q_max, int seq_t_max,
                                                    int q_lo, int t_lo, int q_hi, int t_hi, int ev_max)
{
	// LDS (dynamic: the sequences are sized for the batch's longest read, so that many regions share a CU):
	// [joined CIGAR | score events | query codes | target codes]
	extern __shared__ __align__(16) uint8_t st_smem[];
	uint32_t *s_c = reinterpret_cast<uint32_t*>(st_smem);
	int8_t *s_d = reinterpret_cast<int8_t*>(st_smem + ST_CIG_MAX * 4);
	// (the target's codes four bits a base, eight to a word, base i at bits 4 (i & 7) of word i >> 3 -- the index's own layout of a
	// contig; the read's two bits a base, sixteen to a word, base i at bits 2 (i & 15): a read with an ambiguous base is read in place.
	// A byte a base was 11.4 of a region's 17.6 KB, nine regions a CU; now 4.5 of 9.6 KB, sixteen)
	uint32_t *s_q = reinterpret_cast<uint32_t*>(st_smem + ST_CIG_MAX * 4 + ev_max), *s_t = s_q + seq_q_max / 16;      // ev_max: a multiple of 16; seq_q_max: of 64
	const int lane = threadIdx.x;
	const unsigned long long n_work = B.dp_ctr[9];
	const bool chunked = n_work >= 4ull * gridDim.x;
	unsigned long long pool_off = 0, pool_left = 0;
	for (unsigned long long wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
		const int64_t rslot = work_list[wi];
		mnc_reg_t r = B.regs[rslot];
		RegDP d = B.regdp[rslot];
		if (d.state != 1) continue;
		{
			// a batch with long reads: several launches, by the LDS a region needs -- this one takes the regions that fit
			// (q_hi, t_hi) and did not fit the launch before it (q_lo, t_lo; -1: there was none)
			const int wq = d.qe0 - d.qs0, wt = d.re0 - d.rs0;
			if ((wq <= q_lo && wt <= t_lo) || wq > q_hi || wt > t_hi) continue;
		}
		const uint32_t rd = (uint32_t)d.read;
		const int64_t a_off = B.an_off[rd];
		const int qlen = d.qlen;
		const uint8_t *read = B.bases + d.read_off;
		const Anchor *a = B.ca + a_off;
		const int rev = d.rev;
		const int64_t coff = d.coff;
		const Seg *sg = B.segs + d.first_seg;
		// ---- which segments are joined: all of them, or up to the first gap filling that Z-dropped; how many
		// CIGAR words that is at most.  One segment per lane, 64 at a time.
		const int n_fill = d.n_seg - d.has_left - d.has_right;
		int stop = d.n_seg;
		bool dropped = false;
		long long total = 0;
		for (int k0 = 0; k0 < d.n_seg && !dropped; k0 += 64) {
			const int k = k0 + lane;
			const bool in = k < d.n_seg;
			int nc = in ? sg[k].n_cigar : 0;
			const bool zd = in && k >= d.has_left && k < d.has_left + n_fill && sg[k].zdropped != 0;
			const unsigned long long bz = __ballot(zd);
			if (bz) dropped = true, stop = k0 + __ffsll((long long)bz);       // the Z-dropped one is the last joined
			if (k >= stop) nc = 0;
			total += __builtin_amdgcn_readlane(dpp_incl_add(nc), 63);
		}
		// room in the region pool: a large batch takes it 4096 words at a time per wave (one counter for 100 000 regions
		// is a queue of atomics, each a round trip the region waits for)
		unsigned long long off = 0;
		if (total > 0 && (unsigned long long)total <= pool_left) off = pool_off, pool_off += total, pool_left -= total;
		else if (total > 0) {
			const unsigned long long want = chunked && total < ST_POOL_CHUNK ? ST_POOL_CHUNK : (unsigned long long)total;
			if (lane == 0) off = atomicAdd(&B.dp_ctr[2], want);
			off = (unsigned long long)__shfl((long long)off, 0);
			if ((long long)(off + want) > B.cig_reg_cap) {
				if (lane == 0) atomicMax(&B.dp_ctr[4], 3ULL);
				total = 0;
				continue;
			}
			pool_off = off + total, pool_left = want - total;
		}
		if ((long long)(off + total) > B.cig_reg_cap) {
			if (lane == 0) atomicMax(&B.dp_ctr[4], 3ULL);
			total = 0;
			continue;
		}
		// the joined CIGAR is built in LDS when it fits, else in place in the pool
		uint32_t *C = total <= ST_CIG_MAX ? s_c : B.cig_reg + off;
		int n_c = 0, flags = r.flags, dp_score = 0;
		int rs1 = d.rs, qs1 = d.qs, re1 = d.rs, qe1 = d.qs;
		int split_at = -1, split_inv = 0;
		// every value below is the same in all lanes; only the copies are shared out
		const bool c_lds = total <= ST_CIG_MAX;
		auto c_order = [&]() { if (c_lds) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); asm volatile("" ::: "memory"); } else mem_order(); };
		// reads of the joined CIGAR in the passes below: `C` is a generic pointer (flat loads); in LDS, say so
		auto CR = [&](int kk) -> uint32_t { return c_lds ? s_c[kk] : C[kk]; };
		if (total > 0) flags |= REG_HAS_DP;
		{
			// the join: a segment whose first operation equals the last one before it adds its length there
			// and starts one word later.  Offsets by a prefix sum, all copies at once, then the additions.
			int carry_lo = -1;                                   // operation of the last word so far
			for (int k0 = 0; k0 < stop; k0 += 64) {
				const int k = k0 + lane;
				const bool in = k < stop;
				const int nk = in ? sg[k].n_cigar : 0;
				const int64_t ok = in ? sg[k].cig_off : 0;
				const uint32_t first = nk > 0 ? B.cig_seg[ok] : 0u, last = nk > 0 ? B.cig_seg[ok + nk - 1] : 0u;
				// score: the extensions add their best score, a gap filling its global score (its best one if it Z-dropped)
				int term = 0;
				if (in) {
					const bool is_fill = k >= d.has_left && k < d.has_left + n_fill;
					if (!is_fill) term = nk > 0 ? sg[k].max : 0;
					else term = sg[k].zdropped ? sg[k].max : sg[k].score;
				}
				dp_score += __builtin_amdgcn_readlane(dpp_incl_add(term), 63);
				const unsigned long long ne = __ballot(nk > 0);
				const unsigned long long below = ne & ((1ULL << lane) - 1ULL);
				const int prev = below ? 63 - __clzll((long long)below) : -1;
				const int lo_prev_lane = __shfl((int)(last & 0xf), prev < 0 ? 0 : prev);
				const int lo_prev = prev < 0 ? carry_lo : lo_prev_lane;
				const int mk = nk > 0 && lo_prev >= 0 && (int)(first & 0xf) == lo_prev ? 1 : 0;
				const int cnt = nk - mk;
				const int incl = dpp_incl_add(cnt);
				const int dst = n_c + incl - cnt;
				// four segments' words are fetched before any of them is stored: a store through `C` (LDS or the pool) orders
				// the loads behind it, and one segment at a time the copies were 23 memory round trips in a row per region
				for (int j = 0; j < 64 && k0 + j < stop; j += 4) {
					int nj[4], mj[4], dj[4];
					int64_t oj[4];
					uint32_t w[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						const int ju = j + u < 64 ? j + u : 63;
						nj[u] = j + u < 64 ? __shfl(nk, ju) : 0, mj[u] = __shfl(mk, ju), dj[u] = __shfl(dst, ju);
						oj[u] = (int64_t)__shfl((long long)ok, ju);
						w[u] = mj[u] + lane < nj[u] ? B.cig_seg[oj[u] + mj[u] + lane] : 0u;
					}
#pragma unroll
					for (int u = 0; u < 4; ++u) {
						if (mj[u] + lane < nj[u]) C[dj[u] + lane] = w[u];
						for (int c = mj[u] + lane + 64; c < nj[u]; c += 64) C[dj[u] + c - mj[u]] = B.cig_seg[oj[u] + c];
					}
				}
				c_order();
				if (mk) atomicAdd(&C[dst - 1], first >> 4 << 4);
				c_order();
				n_c += __builtin_amdgcn_readlane(incl, 63);
				if (ne) carry_lo = __shfl((int)(last & 0xf), 63 - __clzll((long long)ne));
			}
			if (d.has_left) {
				const Seg g = sg[0];
				rs1 = d.rs - (g.reach_end ? g.mqe_t + 1 : g.max_t + 1);
				qs1 = d.qs - (g.reach_end ? d.qs - d.qs0 : g.max_q + 1);
			}
			if (dropped) {
				const Seg g = sg[stop - 1];
				const int prs = g.ts, pqs = g.qs;                    // a gap filling starts where the one before it ended
				int j;
				for (j = g.ai - 1; j >= 0; --j)
					if ((int32_t)a[d.as1 + j].x <= prs + g.max_t) break;
				if (j < 0) j = 0;
				re1 = prs + (g.max_t + 1), qe1 = pqs + (g.max_q + 1);
				if (d.cnt1 - (j + 1) >= B.min_cnt) split_at = d.as1 + j + 1 - r.as, split_inv = g.zdrop_code == 2;
			} else {
				if (n_fill > 0) { const Seg g = sg[d.has_left + n_fill - 1]; re1 = g.ts + g.tlen, qe1 = g.qs + g.qlen; }
				if (d.has_right) {
					const Seg g = sg[d.n_seg - 1];
					re1 = d.re + (g.reach_end ? g.mqe_t + 1 : g.max_t + 1);
					qe1 = d.qe + (g.reach_end ? d.qe0 - d.qe : g.max_q + 1);
				}
			}
		}
		mem_order();

		// ---- a Z-drop splits the region: its tail is planned and aligned in the next round (mm_split_reg)
		if (split_at > 0 && split_at < r.cnt) {
			mnc_reg_t r2 = r;                                    // same values in every lane
			r2.id = -1, r2.flags = REG_SPLIT_R | (split_inv ? REG_SPLIT_INV : 0);
			r2.dp_score = r2.dp_max = r2.dp_max2 = r2.n_ambi = r2.n_cigar = 0;
			r2.cnt = r.cnt - split_at;
			r2.score = (int32_t)((double)__fmul_rn((float)r.score, __fdiv_rn((float)r2.cnt, (float)r.cnt)) + .499);
			r2.as = r.as + split_at;
			if (r.parent == r.id) r2.parent = -2;
			r.cnt -= r2.cnt, r.score -= r2.score;
			// coordinates and chain-level lengths of both parts (mm_reg_set_coor)
			for (int part = 0; part < 2; ++part) {
				mnc_reg_t &x = part ? r2 : r;
				const Anchor f0 = a[x.as], l0 = a[x.as + x.cnt - 1];
				const int q_span = (int)(f0.y >> 32 & 0xff);
				x.rev = (int32_t)(f0.x >> 63), x.rid = (int32_t)(f0.x << 1 >> 33);
				x.rs = (int32_t)f0.x + 1 > q_span ? (int32_t)f0.x + 1 - q_span : 0;
				x.re = (int32_t)l0.x + 1;
				if (!x.rev) x.qs = (int32_t)f0.y + 1 - q_span, x.qe = (int32_t)l0.y + 1;
				else x.qs = qlen - ((int32_t)l0.y + 1), x.qe = qlen - ((int32_t)f0.y + 1 - q_span);
				x.mlen = x.blen = q_span;
				for (int i = x.as + 1; i < x.as + x.cnt; ++i) {
					const int span = (int)(a[i].y >> 32 & 0xff);
					const int tl = (int32_t)a[i].x - (int32_t)a[i - 1].x, ql = (int32_t)a[i].y - (int32_t)a[i - 1].y;
					x.blen += tl > ql ? tl : ql;
					x.mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
				}
			}
			flags |= REG_SPLIT_L;
			if (lane == 0) {
				const int idx = atomicAdd(&B.reg_cnt[rd], 1);
				const int64_t nslot = reg_slot(B, rd) + idx;
				if (nslot >= reg_slot(B, rd + 1)) atomicMax(&B.dp_ctr[4], 5ULL);       // out of region slots: the batch is redone with more per read
				else {
					RegDP d2;
					memset(&d2, 0, sizeof(d2));
					// skeleton order: the tail right behind its head; the odd number between them is the inversion region's
					d2.read = (int32_t)rd, d2.order = d.order + 2, d2.state = 1, d2.head = (int32_t)rslot + 1;
					B.regs[nslot] = r2, B.regdp[nslot] = d2;
					const unsigned long long ni = atomicAdd(&B.dp_ctr[5], 1ULL);
					next_list[ni] = (int32_t)nslot;
				}
			}
		}
		r.rs = rs1, r.re = re1;
		if (rev) r.qs = qlen - qe1, r.qe = qlen - qs1;
		else r.qs = qs1, r.qe = qe1;
		r.flags = flags, r.dp_score = dp_score;

		if (flags & REG_HAS_DP) {
			// ---- sequences of [qs1, qe1) x [rs1, re1) into LDS (or read in place when too long)
			const int ql = qe1 - qs1, tl = re1 - rs1;
			const bool in_lds = ql <= seq_q_max && tl <= seq_t_max;
			const bool q_lds = in_lds && !B.ambig[rd];                // (two bits hold no ambiguous code: such a read's bases are read in place)
			bool acgt_only = false;                                   // no ambiguous base in the region's read or target words
			if (in_lds) {
				if (!B.ambig[rd]) {
					// a read of A C G T only: its bases from the sketch stage's 2-bit words (sixteen a word, the first in the top
					// bits).  A lane builds a word of sixteen codes from the 32 bits that hold them: on the read's own strand the
					// sixteen fields are turned round; on the other one they already lie last base first, and are complemented
					const int64_t g_lo = d.read_off + (rev ? qlen - qe1 : qs1), g_hi = g_lo + ql;
					for (int w = lane; w * 16 < ql; w += 64) {
						int64_t g0 = rev ? g_hi - 16 * w - 16 : g_lo + 16 * w;          // first of the sixteen bases on the read's strand
						int skip = 0;
						if (g0 < 0) skip = (int)-g0, g0 = 0;                           // (the batch's very first bases: the word's tail, beyond ql)
						const uint32_t hi = B.packed[g0 >> 4], lo = B.packed[(g0 >> 4) + 1];
						const uint32_t sh2 = 2 * (uint32_t)(g0 & 15);
						uint32_t f = sh2 ? __builtin_amdgcn_alignbit(hi, lo, 32 - sh2) : hi;    // base g0 in bits 31:30 .. g0 + 15 in 1:0
						if (rev) f = ~f >> (2 * skip);
						else {
							f = __builtin_bitreverse32(f);
							f = (f & 0x55555555u) << 1 | (f >> 1 & 0x55555555u);
						}
						s_q[w] = f;
					}
				}
				// the target's codes: the contig's own words, moved to the region's first base
				const int64_t o0 = coff + rs1;
				const uint32_t sh = (uint32_t)(o0 & 7) * 4;
				uint32_t t_amb = 0;
				for (int w = lane; w * 8 < tl; w += 64) {
					const bool more = sh != 0 && w * 8 + 8 - (int)(o0 & 7) < tl;    // (the next word holds bases of the region: never read past them)
					const uint32_t a0 = B.seq4[(o0 >> 3) + w], a1 = more ? B.seq4[(o0 >> 3) + w + 1] : 0u;
					const uint32_t v = sh ? __builtin_amdgcn_alignbit(a1, a0, sh) : a0;
					t_amb |= v & 0xccccccccu;                                // a code above 3 somewhere in these eight bases (or next to the region)
					s_t[w] = v;
				}
				acgt_only = !B.ambig[rd] && !__any(t_amb != 0);
			}
			c_order();
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
			auto Q = [&](int i) -> int { return q_lds ? (int)(s_q[i >> 4] >> (2 * (i & 15)) & 3u) : qcode(read, qlen, rev, qs1 + i); };
			auto Tg = [&](int i) -> int { return in_lds ? (int)(s_t[i >> 3] >> (4 * (i & 7)) & 15u) : tcode(B, coff, rs1 + i); };
			// eight codes from base `at` of the target's LDS array (reads one word past the one that holds the last code asked for);
			// of the read's: sixteen bits, spread to four bits a code
			auto lds8 = [&](const uint32_t *base, int at) -> uint32_t {
				const uint32_t *w = base + (at >> 3);
				return __builtin_amdgcn_alignbit(w[1], w[0], ((uint32_t)at & 7u) * 4u);
			};
			auto lds8q = [&](int at) -> uint32_t {
				const uint32_t *w = s_q + (at >> 4);
				uint32_t f = __builtin_amdgcn_alignbit(w[1], w[0], ((uint32_t)at & 15u) * 2u) & 0xffffu;
				f = (f | f << 8) & 0x00ff00ffu, f = (f | f << 4) & 0x0f0f0f0fu, f = (f | f << 2) & 0x33333333u;
				return f;
			};
			int qshift = 0, tshift = 0;
			// ---- mm_fix_cigar.  Its common work -- sliding every indel between two M runs to the left
			// as far as the bases repeat -- is independent per indel as long as no slide eats a whole
			// M run: the lanes do it 64 operations at a time.  Anything else (an I next to a D, a
			// leading gap, an M run eaten up) is rare and takes the sequential form on one lane.
			bool rare = false;
			if (n_c > 1) {
				bool odd = (C[0] & 0xf) != 0;
				int carry_q = 0, carry_t = 0;
				for (int k0 = 0; k0 < n_c; k0 += 64) {
					const int k = k0 + lane;
					const uint32_t wd = k < n_c ? CR(k) : 0;
					const uint32_t op = wd & 0xf;
					const int len = (int)(wd >> 4);
					const int dq = k < n_c && op != 2 ? len : 0, dt = k < n_c && op != 1 ? len : 0;
					const int iq = dpp_incl_add(dq), it = dpp_incl_add(dt);
					const int qoff = carry_q + iq - dq, toff = carry_t + it - dt;
					carry_q += __builtin_amdgcn_readlane(iq, 63), carry_t += __builtin_amdgcn_readlane(it, 63);
					int sl = 0;
					if (k < n_c) {
						if (len == 0) odd = true;
						if (op != 0) {
							const uint32_t nx = k + 1 < n_c ? CR(k + 1) : 0u, pv = k > 0 ? CR(k - 1) : 1u;
							if (k + 1 < n_c && (nx & 0xf) != 0) odd = true;          // I next to D
							if (k > 0 && k < n_c - 1 && (pv & 0xf) == 0 && (nx & 0xf) == 0) {
								const int prev_len = (int)(pv >> 4);
								if (op == 1) { while (sl < prev_len && Q(qoff - 1 - sl) == Q(qoff + len - 1 - sl)) ++sl; }
								else { while (sl < prev_len && Tg(toff - 1 - sl) == Tg(toff + len - 1 - sl)) ++sl; }
								if (sl == prev_len) odd = true;
							}
						}
						if (k < ev_max && sl <= ST_EV_LIM) s_d[k] = (int8_t)sl; else if (sl) odd = true;
					}
				}
				rare = __any(odd) != 0;
				c_order();
				if (!rare) {
					for (int k = lane; k < n_c; k += 64) {
						const int sl = k < ev_max ? (int)s_d[k] : 0;
						if (sl > 0) { atomicSub(&C[k - 1], (uint32_t)sl << 4); atomicAdd(&C[k + 1], (uint32_t)sl << 4); }
					}
				}
				c_order();
			}
			if (rare && lane == 0) {
				int toff = 0, qoff = 0;
				bool to_shrink = false;
				for (int k = 0; k < n_c; ++k) {
					const uint32_t op = C[k] & 0xf;
					const int len = (int)(C[k] >> 4);
					if (len == 0) to_shrink = true;
					if (op == 0) toff += len, qoff += len;
					else {
						if (k > 0 && k < n_c - 1 && (C[k - 1] & 0xf) == 0 && (C[k + 1] & 0xf) == 0) {
							int l;
							const int prev_len = (int)(C[k - 1] >> 4);
							if (op == 1) { for (l = 0; l < prev_len; ++l) if (Q(qoff - 1 - l) != Q(qoff + len - 1 - l)) break; }
							else { for (l = 0; l < prev_len; ++l) if (Tg(toff - 1 - l) != Tg(toff + len - 1 - l)) break; }
							if (l > 0) C[k - 1] -= (uint32_t)l << 4, C[k + 1] += (uint32_t)l << 4, qoff -= l, toff -= l;
							if (l == prev_len) to_shrink = true;
						}
						if (op == 1) qoff += len; else toff += len;
					}
				}
				for (int k = 0; k < n_c - 2; ++k) {
					if ((C[k] & 0xf) > 0 && (C[k] & 0xf) + (C[k + 1] & 0xf) == 3) {
						uint32_t s3[3] = { 0, 0, 0 };
						int l;
						for (l = k; l < n_c; ++l) {
							const uint32_t op = C[l] & 0xf;
							if (op == 1 || op == 2 || C[l] >> 4 == 0) s3[op] += C[l] >> 4;
							else break;
						}
						if (s3[1] > 0 && s3[2] > 0 && l - k > 2) {
							C[k] = s3[1] << 4 | 1, C[k + 1] = s3[2] << 4 | 2;
							for (k += 2; k < l; ++k) C[k] &= 0xf;
							to_shrink = true;
						}
						k = l;
					}
				}
				if (to_shrink) {
					int l = 0;
					for (int k = 0; k < n_c; ++k) if (C[k] >> 4 != 0) C[l++] = C[k];
					n_c = l;
					l = 0;
					for (int k = 0; k < n_c; ++k)
						if (k == n_c - 1 || (C[k] & 0xf) != (C[k + 1] & 0xf)) C[l++] = C[k];
						else C[k + 1] += C[k] >> 4 << 4;
					n_c = l;
				}
				if ((C[0] & 0xf) == 1 || (C[0] & 0xf) == 2) {
					const int l = (int)(C[0] >> 4);
					if ((C[0] & 0xf) == 1) { if (r.rev) r.qe -= l; else r.qs += l; qshift = l; }
					else r.rs += l, tshift = l;
					--n_c;
					for (int k = 0; k < n_c; ++k) C[k] = C[k + 1];
				}
			}
			n_c = __shfl(n_c, 0), qshift = __shfl(qshift, 0), tshift = __shfl(tshift, 0);
			r.qs = __shfl(r.qs, 0), r.qe = __shfl(r.qe, 0), r.rs = __shfl(r.rs, 0);
			c_order();
			// ---- mm_update_extra: s := max(s + d, 0) per base (and per gap) is the map x -> max(x + d, 0);
			// the score events of the region are laid out in LDS and scanned 64 at a time
			int s_run = 0, s_max = 0, mlen = 0, blen = 0, n_ambi = 0, ev = 0;
			int c_amb = 0, c_diff = 0, c_gamb = 0;                    // per-lane counts (M columns; gap bases), summed at the end
			auto flush = [&]() {
				c_order();
				for (int c0 = 0; c0 < ev; c0 += 64) {
					const int i = c0 + lane;
					const bool act = i < ev;
					int fa = act ? (int)s_d[i] : 0, fb = act ? 0 : SC_NONE;
					dpp_scan_maps(fa, fb);
					const int sv = s_run + fa > fb ? s_run + fa : fb;
					const int mv = dpp_max_all(act ? sv : 0);
					s_max = s_max > mv ? s_max : mv;
					s_run = __builtin_amdgcn_readlane(sv, 63);
				}
				ev = 0;
				c_order();
			};
			int toff = tshift, qoff = qshift;
			// one operation at a time, the lanes sharing its bases: long M runs, gaps too costly for an event
			auto seq_ops = [&](int k_lo, int k_hi) {
			for (int k = k_lo; k < k_hi; ++k) {
				const uint32_t op = C[k] & 0xf;
				const int len = (int)(C[k] >> 4);
				if (op == 0) {
					for (int pos = 0; pos < len;) {
						if (ev == ev_max) flush();
						const int take = len - pos < ev_max - ev ? len - pos : ev_max - ev;
						for (int i = lane; i < take; i += 64) {
							const int cq = Q(qoff + pos + i), ct = Tg(toff + pos + i);
							int dlt;
							if (ct > 3 || cq > 3) ++c_amb, dlt = -B.sc_ambi;
							else if (ct != cq) ++c_diff, dlt = -B.sc_b;
							else dlt = B.sc_a;
							s_d[ev + i] = (int8_t)dlt;
						}
						ev += take, pos += take;
					}
					blen += len, mlen += len;
					toff += len, qoff += len;
				} else {
					for (int i = lane; i < len; i += 64) c_gamb += (op == 1 ? Q(qoff + i) : Tg(toff + i)) > 3;
					blen += len;
					if (ev == ev_max) flush();
					const int cost = B.gap_q + B.gap_e * len;
					if (cost > ST_EV_LIM) {                               // does not fit an event: applied directly
						flush();
						s_run = s_run - cost > 0 ? s_run - cost : 0;
					} else {
						if (lane == 0) s_d[ev] = (int8_t)-cost;
						++ev;
					}
					if (op == 1) qoff += len; else toff += len;
				}
			}
			};
			// The usual CIGAR (hundreds of short operations): every lane takes a contiguous share of the operations
			// and composes their maps one after the other in registers -- x -> max(x + A, Bv) after an event d is
			// A += d, Bv = max(Bv + d, 0), and the largest value on the way is max(x + MA, MB) with the running
			// maxima of A and Bv -- then ONE scan over the 64 lane maps gives every lane the score it starts from.
			// No event array, no scan per 64 events.  A CIGAR with few or very long operations (a clean read:
			// "5000M") would leave most lanes idle: it takes the event form below.
			bool lane_form = n_c >= 128;
			if (lane_form) {
				int longest = 0;
				for (int k = lane; k < n_c; k += 64) { const int len = (int)(CR(k) >> 4); longest = longest > len ? longest : len; }
				lane_form = dpp_max_all(longest) <= 384;
			}
			if (lane_form) {
				// shares of equal length in bases, cut at operation boundaries: lane l starts at the first operation whose
				// bases begin at or after l * per_u (found by the lane that owns the operation, 64 operations at a time)
				int T = 0;
				for (int k = lane; k < n_c; k += 64) T += (int)(CR(k) >> 4);
				T = __builtin_amdgcn_readlane(dpp_incl_add(T), 63);
				const int per_u = T / 64 + 1;
				int32_t *s_b = reinterpret_cast<int32_t*>(s_d);                     // [64][3]: first operation, query and target offset there
				s_b[lane * 3] = n_c, s_b[lane * 3 + 1] = 0, s_b[lane * 3 + 2] = 0;
				c_order();
				{
					int cP = 0, cQ = qshift, cT = tshift, prev_last = -1;           // prev_last: where the operation before this chunk began
					for (int k0 = 0; k0 < n_c; k0 += 64) {
						const int k = k0 + lane;
						const uint32_t wd = k < n_c ? CR(k) : 0, op = wd & 0xf;
						const int len = (int)(wd >> 4);
						const int dq = k < n_c && op != 2 ? len : 0, dt = k < n_c && op != 1 ? len : 0;
						const int il = dpp_incl_add(len), iq = dpp_incl_add(dq), it = dpp_incl_add(dt);
						const int Pk = cP + il - len, Pq = cQ + iq - dq, Pt = cT + it - dt;
						int before = __shfl_up(Pk, 1);
						if (lane == 0) before = prev_last;
						if (k < n_c) {
							// the lanes l with before < l * per_u <= Pk start here
							const int l_lo = before < 0 ? 0 : before / per_u + 1, l_hi = Pk / per_u;
							for (int l = l_lo; l <= l_hi && l < 64; ++l) s_b[l * 3] = k, s_b[l * 3 + 1] = Pq, s_b[l * 3 + 2] = Pt;
						}
						prev_last = __builtin_amdgcn_readlane(Pk, 63);
						cP += __builtin_amdgcn_readlane(il, 63), cQ += __builtin_amdgcn_readlane(iq, 63), cT += __builtin_amdgcn_readlane(it, 63);
					}
				}
				c_order();
				int k = s_b[lane * 3], qo = s_b[lane * 3 + 1], to = s_b[lane * 3 + 2];
				const int k_end = lane < 63 ? s_b[lane * 3 + 3] : n_c;
				const bool mine = k < k_end;
				// one base per lane and turn, whatever operation it belongs to: the lanes do not wait for each other's runs
				int A = 0, Bv = SC_NONE, MA = SC_NONE, MB = SC_NONE, my_blen = 0, my_mlen = 0;
				int len = 0, pos = 0;
				uint32_t op = 0;
				bool done = !mine;
				// `C` is a generic pointer (LDS or the pool): a flat load per turn costs more than the turn.  The next word
				// comes early and, when the CIGAR is in LDS, as an LDS read.
				auto cword = [&](int kk) -> uint32_t { return kk >= k_end ? 0u : c_lds ? s_c[kk] : C[kk]; };
				uint32_t nxt = cword(k);
				// scores that fit a signed four-bit field: a turn's eight events as one word (the fast form below)
				const bool nib_ok = acgt_only && B.sc_a >= 0 && B.sc_a <= 7 && B.sc_b >= 1 && B.sc_b <= 8;
				for (;;) {
					// (a gap whose bases need no turn is applied on the spot and the next operation fetched in the same turn)
					while (!done && pos == len) {
						if (k >= k_end) done = true;
						else {
							const uint32_t wd = nxt;
							nxt = cword(++k);
							op = wd & 0xf, len = (int)(wd >> 4), pos = 0;
							my_blen += len;
							if (op == 0) my_mlen += len;
							else {
								const int dlt = -(B.gap_q + B.gap_e * len);
								A += dlt, Bv = Bv + dlt > 0 ? Bv + dlt : 0;
								MA = MA > A ? MA : A, MB = MB > Bv ? MB : Bv;
								if (acgt_only) {                                   // nothing to count in its bases: no turn for them
									if (op == 1) qo += len; else to += len;
									pos = len;
								}
							}
						}
					}
					if (!__any(!done)) break;
					if (nib_ok) {
						// no ambiguous code in the region (only M runs have turns: gaps are applied when fetched) and scores that fit
						// four signed bits: the eight events of a turn are built as ONE word -- a for a base inside the run, less
						// a + b where the two codes differ, 0 beyond the run's end -- and a base costs one field extract and the map's
						// five operations instead of fifteen
						if (!done && pos < len) {
							const int n = len - pos < 8 ? len - pos : 8;
							pos += n;
							const uint32_t x8 = lds8q(qo) ^ lds8(s_t, to);
							uint32_t y = x8 | x8 >> 1;
							y |= y >> 2;
							const uint32_t vm = n < 8 ? 0x11111111u & ((1u << (4 * n)) - 1u) : 0x11111111u;
							y &= vm;
							c_diff += __popc(y);
							const uint32_t D = vm * (uint32_t)B.sc_a + y * (uint32_t)(16 - B.sc_a - B.sc_b);
#pragma unroll
							for (int b8 = 0; b8 < 8; ++b8) {
								const int dlt = (int)(D << (28 - 4 * b8)) >> 28;
								A += dlt, Bv = Bv + dlt > 0 ? Bv + dlt : 0;
								MA = MA > A ? MA : A, MB = MB > Bv ? MB : Bv;
							}
							qo += n, to += n;
						}
					} else if (q_lds) {
						// up to eight bases per lane and turn, from two aligned LDS words each: the turn's time is the LDS round
						// trip, not the arithmetic
						// one body for every operation, without branches: a gap's bases only count ambiguous codes (its score
						// event was applied when it was fetched), and a base beyond the operation's end is an event of 0 -- which
						// leaves the map as it is once a real event has been applied (Bv >= 0 from then on), and the first base
						// of a turn is always real
						if (!done && pos < len) {
							const int n = len - pos < 8 ? len - pos : 8;
							pos += n;
							const uint32_t q8 = lds8q(qo), t8 = lds8(s_t, to);
							const bool is_m = op == 0;
							const int sc_match = is_m ? B.sc_a : 0, sc_mis = is_m ? -B.sc_b : 0, sc_amb = is_m ? -B.sc_ambi : 0;
							// an M run compares the two; an insertion looks at the query's codes only, a deletion at the target's
							const uint32_t qv = op == 2 ? 0u : q8, tv = op == 1 ? 0u : t8;
							const uint32_t x8 = qv ^ tv, a8 = (qv | tv) & 0xccccccccu;
#pragma unroll
							for (int b8 = 0; b8 < 8; ++b8) {
								const bool valid = b8 < n;
								const bool amb = valid && (a8 >> (4 * b8) & 0xfu) != 0, dif = valid && (x8 >> (4 * b8) & 0xfu) != 0;
								const int dlt = !valid ? 0 : amb ? sc_amb : dif ? sc_mis : sc_match;
								c_amb += is_m && amb ? 1 : 0, c_diff += is_m && !amb && dif ? 1 : 0, c_gamb += !is_m && amb ? 1 : 0;
								A += dlt, Bv = Bv + dlt > 0 ? Bv + dlt : 0;
								MA = MA > A ? MA : A, MB = MB > Bv ? MB : Bv;
							}
							qo += op == 2 ? 0 : n, to += op == 1 ? 0 : n;
						}
					} else if (!done && pos < len) {
						++pos;
						if (op == 0) {
							const int cq = Q(qo), ct = Tg(to);
							int dlt;
							if (ct > 3 || cq > 3) ++c_amb, dlt = -B.sc_ambi;
							else if (ct != cq) ++c_diff, dlt = -B.sc_b;
							else dlt = B.sc_a;
							A += dlt, Bv = Bv + dlt > 0 ? Bv + dlt : 0;
							MA = MA > A ? MA : A, MB = MB > Bv ? MB : Bv;
							++qo, ++to;
						} else if (op == 1) c_gamb += Q(qo) > 3, ++qo;
						else c_gamb += Tg(to) > 3, ++to;
					}
				}
				// the score every lane starts from: the lanes before it, composed, applied to 0
				int fa = A, fb = Bv;
				dpp_scan_maps(fa, fb);
				const int x_out = fa > fb ? fa : fb;                                 // max(0 + fa, fb)
				int x_in = __shfl_up(x_out, 1);
				if (lane == 0) x_in = 0;
				const int top = mine ? (x_in + MA > MB ? x_in + MA : MB) : 0;
				s_max = dpp_max_all(top > 0 ? top : 0);
				blen = __builtin_amdgcn_readlane(dpp_incl_add(my_blen), 63), mlen = __builtin_amdgcn_readlane(dpp_incl_add(my_mlen), 63);
			}
			// 64 operations at a time, one lane each: offsets by prefix sums, every lane lays out its
			// operation's events (an M run: one per base; a gap: one) and counts
			for (int k0 = 0; k0 < n_c && !lane_form; k0 += 64) {
				const int k = k0 + lane, k1 = k0 + 64 < n_c ? k0 + 64 : n_c;
				const uint32_t wd = k < n_c ? CR(k) : 0;
				const uint32_t op = wd & 0xf;
				const int len = (int)(wd >> 4);
				const int cost = B.gap_q + B.gap_e * len;
				const int dq = k < n_c && op != 2 ? len : 0, dt = k < n_c && op != 1 ? len : 0, dev = k >= n_c ? 0 : op == 0 ? len : 1;
				const int iq = dpp_incl_add(dq), it = dpp_incl_add(dt), ie = dpp_incl_add(dev);
				const int tot_ev = __builtin_amdgcn_readlane(ie, 63);
				const int big = dpp_max_all(k < n_c ? (op == 0 ? len : cost > ST_EV_LIM ? 1 << 20 : 0) : 0);
				if (big > 256 || tot_ev > ev_max) { seq_ops(k0, k1); continue; }
				if (ev + tot_ev > ev_max) flush();
				if (k < n_c) {
					const int qo = qoff + iq - dq, to = toff + it - dt, eo = ev + ie - dev;
					if (op == 0) {
						for (int i = 0; i < len; ++i) {
							const int cq = Q(qo + i), ct = Tg(to + i);
							int dlt;
							if (ct > 3 || cq > 3) ++c_amb, dlt = -B.sc_ambi;
							else if (ct != cq) ++c_diff, dlt = -B.sc_b;
							else dlt = B.sc_a;
							s_d[eo + i] = (int8_t)dlt;
						}
					} else {
						for (int i = 0; i < len; ++i) c_gamb += (op == 1 ? Q(qo + i) : Tg(to + i)) > 3;
						s_d[eo] = (int8_t)-cost;
					}
				}
				blen += __builtin_amdgcn_readlane(dpp_incl_add(k < n_c ? len : 0), 63);
				mlen += __builtin_amdgcn_readlane(dpp_incl_add(k < n_c && op == 0 ? len : 0), 63);
				qoff += __builtin_amdgcn_readlane(iq, 63), toff += __builtin_amdgcn_readlane(it, 63), ev += tot_ev;
			}
			if (!lane_form) flush();
			for (int sft = 32; sft > 0; sft >>= 1) c_amb += __shfl_xor(c_amb, sft), c_diff += __shfl_xor(c_diff, sft), c_gamb += __shfl_xor(c_gamb, sft);
			blen -= c_amb + c_gamb, mlen -= c_amb + c_diff, n_ambi = c_amb + c_gamb;
			r.mlen = mlen, r.blen = blen, r.n_ambi += n_ambi, r.dp_max = s_max, r.n_cigar = n_c;
		}
		if (C == s_c) for (int k = lane; k < n_c; k += 64) B.cig_reg[off + k] = s_c[k];
		if (lane == 0) {
			d.state = 2, d.cig_off = (int64_t)off, d.n_cigar = n_c;
			B.regs[rslot] = r, B.regdp[rslot] = d;
		}
		mem_order();
	}
}

// ================================================================ the inversion between two halves (mm_align1_inv)
// A region split by a Z-drop whose inversion test was positive leaves a tail with REG_SPLIT_INV.  Once head and tail are
// both aligned, the stretch between them -- query [r1.qe, r2.qs) on the halves' strand, target [r1.re, r2.rs) -- is
// aligned on the OTHER strand of the read: minimap2 runs ksw_ll_i16 (striped int16 Smith-Waterman, one affine gap cost)
// on the two sequences reversed, takes the END it reports as the start of the alignment, and extends from there with
// ksw_extd2.  This kernel does the first half and leaves a region of its own (REG_INV, no seeds) whose one extension the
// next round's plan / alignment / stitch kernels run like any other.
//
// The local alignment, one wave per candidate: column by column over the (reversed) target, 64 query positions per step;
// the vertical gap state F[j] = max_{k<j} (H[k] - q - e (j - k)) is a prefix maximum of h[k] + e k over the column (opening
// a gap from a cell that was itself reached through a gap is never better than extending that gap), the horizontal one E
// is kept per position.  What minimap2 consumes are the score and the end coordinates with the tie rules of the striped
// layout (oracle/mm_ksw.c: orc_ksw_ll_i16 / orc_local_end): the query padded to a multiple of eight positions that score 0;
// te = the LAST column whose maximum equals the best score, qe = that column's position j with the largest
// (j % slen, j / slen), slen = padded length / 8.
constexpr int INV_NEG = -(1 << 29);

__global__ __launch_bounds__(64) void mnc_dp_inv(Batch B, const int32_t *work_list, int32_t *next_list, int32_t *ws_all, int ws_half)
{
	const int lane = threadIdx.x;
	const unsigned long long n_work = B.dp_ctr[9];
	int32_t *Hc = ws_all + (size_t)blockIdx.x * 2 * ws_half, *Ec = Hc + ws_half;
	for (unsigned long long wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
		const int64_t tslot = work_list[wi];
		const mnc_reg_t r2 = B.regs[tslot];
		const RegDP d2 = B.regdp[tslot];
		if (!(r2.flags & REG_SPLIT_INV) || d2.state != 2 || d2.head <= 0) continue;
		const int64_t hslot = d2.head - 1;
		const mnc_reg_t r1 = B.regs[hslot];
		{
			// what precedes the tail in the skeleton's array is the inversion region behind the head, if it has one (aligned by
			// now: this kernel follows the round's stitch), and not the head
			const int ia = B.regdp[hslot].inv_after;
			if (ia > 0 && (B.regs[ia - 1].flags & REG_HAS_DP)) continue;
		}
		if (!(r1.flags & REG_SPLIT_L) || !(r2.flags & REG_SPLIT_R)) continue;
		if (r1.id != r1.parent && r1.parent != -2) continue;    // primaries only (-2: a tail of one)
		if (r2.id != r2.parent && r2.parent != -2) continue;
		if (r1.rid != r2.rid || r1.rev != r2.rev) continue;
		const int ql = r1.rev ? r1.qs - r2.qe : r2.qs - r1.qe, tl = r2.rs - r1.re;
		if (ql < B.min_sc || ql > B.max_gap || tl < B.min_sc || tl > B.max_gap) continue;
		const uint32_t rd = (uint32_t)d2.read;
		const int qlen = (int)(B.offsets[rd + 1] - B.offsets[rd]);
		const uint8_t *read = B.bases + B.offsets[rd];
		const int s = r1.rev ? 0 : 1;                           // the other strand
		const int base = r1.rev ? r2.qe : qlen - r2.qs;         // the stretch's first base on it
		const int64_t coff = B.seq_off[r1.rid];
		const int L = (ql + 7) / 8 * 8, slen = L / 8;
		if (L + 1 > ws_half) { if (lane == 0) atomicMax(&B.dp_ctr[4], 9ULL); continue; }   // cannot happen: ql <= max_gap sizes the scratch
		const int q = B.gap_q, e = B.gap_e, a = B.sc_a, bmis = -B.sc_b, scN = -B.sc_ambi;
		for (int j = lane; j <= L; j += 64) Hc[j] = 0, Ec[j] = 0;
		mem_order();
		int best = 0, te = -1, qe = -1;
		for (int i = 0; i < tl; ++i) {
			const int tb = tcode(B, coff, r1.re + tl - 1 - i);  // the target, reversed
			int carry = INV_NEG, prev_last_old = 0, cmax = 0;
			for (int j0 = 0; j0 < L; j0 += 64) {
				const int j = j0 + lane;
				const bool in = j < L;
				const int old = in ? Hc[j] : 0;                  // H of the previous column
				int diag = __shfl_up(old, 1);
				if (lane == 0) diag = prev_last_old;
				prev_last_old = __shfl(old, 63);
				int sc = 0;
				if (j < ql) {
					const int qb = qcode(read, qlen, s, base + ql - 1 - j);   // the query, reversed
					sc = (tb > 3 || qb > 3) ? scN : tb == qb ? a : bmis;
				}
				int h = diag + sc;
				const int ein = in ? Ec[j] : 0;
				h = h > ein ? h : ein;
				h = h > 0 ? h : 0;
				// F: exclusive prefix maximum of h[k] + e k over the column
				int v = in ? h + e * j : INV_NEG, incl = v;
#pragma unroll
				for (int sft = 1; sft < 64; sft <<= 1) { const int o = __shfl_up(incl, sft); if (lane >= sft) incl = incl > o ? incl : o; }
				int excl = __shfl_up(incl, 1);
				if (lane == 0) excl = INV_NEG;
				excl = excl > carry ? excl : carry;
				const int last = __shfl(incl, 63);
				carry = carry > last ? carry : last;
				const int f = excl - q - e * j;
				h = h > f ? h : f;
				if (in) {
					int en = ein - e, t2 = h - q - e;
					en = en > t2 ? en : t2;
					Hc[j] = h, Ec[j] = en > 0 ? en : 0;
					cmax = cmax > h ? cmax : h;
				}
			}
#pragma unroll
			for (int sft = 32; sft > 0; sft >>= 1) { const int o = __shfl_xor(cmax, sft); cmax = cmax > o ? cmax : o; }
			mem_order();
			if (cmax >= best) {                                 // `imax >= gmax`: the last column with the best score
				best = cmax, te = i;
				int key = -1;
				for (int j = lane; j < L; j += 64)
					if (Hc[j] == best) { const int k = (j % slen) * 8 + j / slen; key = key > k ? key : k; }
#pragma unroll
				for (int sft = 32; sft > 0; sft >>= 1) { const int o = __shfl_xor(key, sft); key = key > o ? key : o; }
				qe = key < 0 ? -1 : (key >> 3) + (key & 7) * slen;
			}
		}
		if (best < B.min_dp_max || qe < 0) continue;
		const int q_off = ql - (qe + 1), t_off = tl - (te + 1);
		if (base + q_off < 0) continue;                         // in front of the read's first base (never: a tail holds three seeds)
		if (lane == 0) {
			const int idx = atomicAdd(&B.reg_cnt[rd], 1);
			const int64_t nslot = reg_slot(B, rd) + idx;
			if (nslot >= reg_slot(B, rd + 1)) atomicMax(&B.dp_ctr[4], 5ULL);
			else {
				mnc_reg_t ri;
				memset(&ri, 0, sizeof(ri));
				ri.id = -1, ri.parent = -1, ri.rid = r1.rid, ri.rev = s, ri.flags = REG_INV;
				RegDP di;
				memset(&di, 0, sizeof(di));
				di.read = (int32_t)rd, di.order = d2.order + 1, di.state = 1;
				di.rid = r1.rid, di.rev = s, di.qlen = qlen;
				di.rs = di.re = di.rs0 = r1.re + t_off, di.qs = di.qe = di.qs0 = base + q_off;
				di.re0 = r1.re + tl, di.qe0 = base + ql;
				B.regs[nslot] = ri, B.regdp[nslot] = di;
				B.regdp[tslot].inv_after = (int32_t)nslot + 1;
				const unsigned long long ni = atomicAdd(&B.dp_ctr[5], 1ULL);
				next_list[ni] = (int32_t)nslot;
			}
		}
	}
}

// ================================================================ round bookkeeping (one thread)
__global__ void mnc_dp_round(Batch B, int first)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	if (first) B.dp_ctr[8] = 0;
	B.dp_ctr[3] = B.dp_ctr[8];            // the align queue starts at this round's first segment
	B.dp_ctr[9] = B.dp_ctr[5];            // regions to plan / stitch this round
	B.dp_ctr[5] = 0;
	B.dp_ctr[6] = 0, B.dp_ctr[7] = 0;
	for (int k = 10; k < 48; ++k) B.dp_ctr[k] = 0;
	B.dp_ctr[56] = B.dp_ctr[57] = B.dp_ctr[58] = B.dp_ctr[59] = B.dp_ctr[60] = B.dp_ctr[61] = B.dp_ctr[62] = B.dp_ctr[63] = 0;
	B.dp_ctr[54] = 0;                     // queue of the 42-cell tier (its list length is [30], its anti-diagonals [52])
	B.dp_ctr[53] = 0;                     // regions left to mnc_dp_plan_long this round
	if (first) for (int k = 48; k < 64; ++k) B.dp_ctr[k] = 0;   // banded kernel: list lengths 10 / 11 / 12 (tier 1, tier 2, handed back), queues 13 / 14 / 15;
	                                                 // extension kernel: lists 16 / 17, queues 18 / 19; literal kernel's first pass: list 20, queue 21; banded kernel, 128 cells: list 22, queue 23; extension kernel, 128 / 256 cells: lists 24 / 25, queues 26 / 27
}
__global__ void mnc_dp_round_end(Batch B)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	B.dp_ctr[8] = B.dp_ctr[0];
}

// ================================================================ launches
void launch_dp_gather(const Batch &B, hipStream_t st)
{
	if (B.n_reads) hipLaunchKernelGGL(mnc_dp_gather, dim3(B.n_reads), dim3(64), 0, st, B);
}
int dp_gather_long_prepare(int lds_anchors)
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_dp_gather_long), hipFuncAttributeMaxDynamicSharedMemorySize, lds_anchors * 8);
	if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}
void launch_dp_gather_long(const Batch &B, const uint32_t *lists, const ClassSpans &spans, int lds_anchors, hipStream_t st)
{
	const uint32_t count = spans.start[spans.n];
	if (count) hipLaunchKernelGGL(mnc_dp_gather_long, dim3(count), dim3(64), (size_t)lds_anchors * 8, st, B, lists, spans, lds_anchors);
}
void launch_dp_round(const Batch &B, int first, hipStream_t st) { hipLaunchKernelGGL(mnc_dp_round, dim3(1), dim3(1), 0, st, B, first); }
void launch_dp_round_end(const Batch &B, hipStream_t st) { hipLaunchKernelGGL(mnc_dp_round_end, dim3(1), dim3(1), 0, st, B); }
constexpr int PLAN_LONG = 512;                // chained anchors of a read from which its regions are planned a wave each (a 5 kb read has ~200 per chain)
constexpr int PLAN_LDS_ANCHORS = 6144;        // ... out of LDS up to this many (20 bytes each: 120 KB, one wave a CU), from memory beyond
constexpr int PLAN_LDS_MID = 2560;            // most of them (reads up to ~30 kb) in a launch of their own with 50 KB a wave: three a CU
int dp_plan_prepare()
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_dp_plan_long), hipFuncAttributeMaxDynamicSharedMemorySize, PLAN_LDS_ANCHORS * 20);
	if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}
// `wave_form`: 1 the batch holds reads long enough for PLAN_LONG chained anchors (the caller knows its longest read): their
// regions go to mnc_dp_plan_long; 2 every region does (a test switch: the wave form on ordinary reads); 0 none
void launch_dp_plan(const Batch &B, const int32_t *work_list, unsigned max_work, int wave_form, int state_max, long long p_max, int cig_max,
                    long long big_state, long long big_p, long long big_cig, long long huge_state, long long huge_p, long long huge_cig, hipStream_t st)
{
	if (!max_work) return;
	const bool long_reads = wave_form != 0;
	hipLaunchKernelGGL(mnc_dp_plan, dim3((max_work + 63) / 64), dim3(64), 0, st, B, work_list, wave_form == 2 ? 1 : wave_form == 1 ? PLAN_LONG : 0,
	                   state_max, p_max, cig_max, big_state, big_p, big_cig, huge_state, huge_p, huge_cig);
	if (long_reads) {
		const PlanLimits lim = { state_max, p_max, cig_max, big_state, big_p, big_cig, huge_state, huge_p, huge_cig };
		hipLaunchKernelGGL(mnc_dp_plan_long, dim3(256 * 3), dim3(64), (size_t)PLAN_LDS_MID * 20, st, B, PLAN_LDS_MID, 0, PLAN_LDS_MID + 1, lim);
		hipLaunchKernelGGL(mnc_dp_plan_long, dim3(256), dim3(64), (size_t)PLAN_LDS_ANCHORS * 20, st, B, PLAN_LDS_ANCHORS, PLAN_LDS_MID + 1, INT32_MAX, lim);
	}
}
size_t dp_align_ws_bytes(long long state_max, long long p_max, long long cig_max) { return align_ws(state_max, p_max, cig_max).total; }
// the literal kernel's long calls with the cells in registers (ksw_wg): <8, 2> for few calls, <4, 4> for many
constexpr int ALIGN_SEQ_WIDE = 32768, ALIGN_SEQ_NARROW = 16384;   // LDS bytes per sequence in the two forms
constexpr int ALIGN_SEQ_PACKED = 16384;                           // ... and of the query in the packed one-wave form (a longer query: the workspace form)
// the form for few calls: eight waves x two cells a thread (measured against sixteen x one: an anti-diagonal costs 8 x 370
// instead of 16 x 260 vector instructions, and a workgroup needs half a CU instead of a whole one -- a config-4 block 33.1
// instead of 35.1 ms, 30 000 reads with 13 % errors 37 instead of 39 ms: tools/ab_wide.sh)
#ifndef MNC_WIDE_NW
#define MNC_WIDE_NW 8
#define MNC_WIDE_C 2
#endif
constexpr int WIDE_NW = MNC_WIDE_NW, WIDE_C = MNC_WIDE_C;
constexpr int ALIGN_FEW = 64;                                     // up to this many calls in a pass: the wide form, a workgroup (half a CU's registers) per call
int dp_align_prepare(int lds_bytes)
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_dp_align<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
	if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_dp_align<WIDE_NW, WIDE_C>), hipFuncAttributeMaxDynamicSharedMemorySize, wg_lds_bytes<WIDE_NW>(ALIGN_SEQ_WIDE));
	if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_dp_align<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, wg_lds_bytes<4>(ALIGN_SEQ_NARROW));
	if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_dp_align<1, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, wp_lds_bytes(ALIGN_SEQ_PACKED));
	if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_dp_align_long<16>), hipFuncAttributeMaxDynamicSharedMemorySize, wp_lds_bytes(ALIGN_SEQ_PACKED));
	if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_dp_align<4, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, wp_lds_bytes(ALIGN_SEQ_PACKED, 4));
	if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}
// the long passes 5, 1, 3 as one launch of one-wave workgroups, each on a slot of its class (mnc_dp_align_long)
bool dp_align_long_packed(const Batch &B)
{
	static const bool old_forms = getenv("MNC_KSW_FORMS") && !strcmp(getenv("MNC_KSW_FORMS"), "old");
	return !old_forms && !(B.debug_route & (32 | 64 | 128 | 256 | 512));
}
void launch_dp_align_long(const Batch &B, uint8_t *ws_huge, int n_huge, long long st_huge, long long p_huge, long long cig_huge,
                          uint8_t *ws_big, int n_big, long long st_big, long long p_big, long long cig_big,
                          uint8_t *ws_small, int n_small, long long st_small, long long p_small, long long cig_small, hipStream_t st)
{
	LongWs L;
	L.ws[0] = ws_huge, L.n_wg[0] = n_huge, L.state[0] = st_huge, L.p[0] = p_huge, L.cig[0] = cig_huge;
	L.ws[1] = ws_big, L.n_wg[1] = n_big, L.state[1] = st_big, L.p[1] = p_big, L.cig[1] = cig_big;
	L.ws[2] = ws_small, L.n_wg[2] = n_small, L.state[2] = st_small, L.p[2] = p_small, L.cig[2] = cig_small;
	hipLaunchKernelGGL((mnc_dp_align_long<16>), dim3(n_huge + n_big + n_small), dim3(64), (size_t)wp_lds_bytes(ALIGN_SEQ_PACKED), st, B, L, ALIGN_SEQ_PACKED);
}
// `forms`: 3 both forms of a long pass (the call count picks one on the device), 1 | 4 the four-wave form alone for every call.
void launch_dp_align(const Batch &B, uint8_t *ws, int n_wg, long long state_max, long long p_max, long long cig_max,
                     int lds_state, int lds_p, int lds_cig, int big_pass, hipStream_t st, int forms)
{
	// The long calls (passes 1, 3, 4, 5: direction bytes in the workspace) on several waves each, cells in registers (ksw_wg):
	// both forms are launched, the pass's call count -- known on the device only -- decides which of them works.
	// debug_route bit 5: one wave each, as the rest; 7: four waves with the cells in the workspace (round 3's form); 8:
	// always the wide form; 9: always the four-wave form; (6: the launches below, but cells in the workspace)
	// (pass 2, what the banded kernels handed back, likewise: it runs behind the window's join, the chip is its own)
	const bool long_pass = big_pass >= 1;
	// round 5: every pass with its direction bytes in the workspace on ONE wave per call, sixteen cells a lane as packed
	// pairs (ksw_wp).  MNC_KSW_FORMS=old (or any of the debug_route bits that name an older form) brings those back.
	static const bool old_forms = getenv("MNC_KSW_FORMS") && !strcmp(getenv("MNC_KSW_FORMS"), "old");
	if (long_pass && !old_forms && !(B.debug_route & (32 | 64 | 128 | 256 | 512))) {
		// two forms, the pass's call count -- known on the device only -- picks one: up to ALIGN_FEW calls four waves on each
		// (a lone call's length is its steps times a wave's instructions a step: a quarter of the cells a wave), more one wave each
		if (forms & 1) {
			hipLaunchKernelGGL((mnc_dp_align<4, 4, true>), dim3(n_wg < ALIGN_FEW ? n_wg : ALIGN_FEW), dim3(256), (size_t)wp_lds_bytes(ALIGN_SEQ_PACKED, 4), st, B, ws, state_max, p_max, cig_max,
			                   ALIGN_SEQ_PACKED, 0, 0, big_pass, 1, ALIGN_FEW);
			hipLaunchKernelGGL((mnc_dp_align<1, 16>), dim3(n_wg), dim3(64), (size_t)wp_lds_bytes(ALIGN_SEQ_PACKED), st, B, ws, state_max, p_max, cig_max,
			                   ALIGN_SEQ_PACKED, 0, 0, big_pass, 2, ALIGN_FEW);
		}
		return;
	}
	if (long_pass && !(B.debug_route & (32 | 128))) {
		// (forms bit 2: the four-wave form alone, whatever the count)
		const int regime16 = (B.debug_route & 256) ? 0 : (B.debug_route & 512) ? -1 : 1;
		const int regime4 = ((B.debug_route & 512) || ((forms & 4) && !(B.debug_route & 256))) ? 0 : (B.debug_route & 256) ? -1 : 2;
		// (a workgroup of the wide form takes half a CU's registers at once -- a whole CU's in its first, sixteen-wave shape --;
		// its launch is as small as its regime: placing hundreds of such workgroups beside the tiers costs time even when
		// they find nothing to do, 0.7 ms of a 100 000-read batch with 768 of them)
		if (regime16 >= 0 && (forms & 2))
			hipLaunchKernelGGL((mnc_dp_align<WIDE_NW, WIDE_C>), dim3(n_wg < ALIGN_FEW ? n_wg : ALIGN_FEW), dim3(64 * WIDE_NW), (size_t)wg_lds_bytes<WIDE_NW>(ALIGN_SEQ_WIDE), st, B, ws, state_max, p_max, cig_max,
			                   ALIGN_SEQ_WIDE, 0, 0, big_pass, regime16, ALIGN_FEW);
		if (regime4 >= 0 && (forms & 1) && big_pass == 2 && !(B.debug_route & 512))
			// what the banded kernels handed back is many small calls as a rule (1 300 per 30 000 reads at 16 % errors): one wave
			// each out of LDS, two thousand side by side, unless they are few
			hipLaunchKernelGGL((mnc_dp_align<1>), dim3(n_wg), dim3(64), (size_t)lds_state + lds_p + lds_cig * 4, st, B, ws, state_max, p_max, cig_max,
			                   lds_state, lds_p, lds_cig, big_pass, regime4, ALIGN_FEW);
		else if (regime4 >= 0 && (forms & 1))
			hipLaunchKernelGGL((mnc_dp_align<4, 4>), dim3(n_wg), dim3(256), (size_t)wg_lds_bytes<4>(ALIGN_SEQ_NARROW), st, B, ws, state_max, p_max, cig_max,
			                   ALIGN_SEQ_NARROW, 0, 0, big_pass, regime4, ALIGN_FEW);
	} else if (long_pass && !(B.debug_route & 32)) {
		if (forms & 1) hipLaunchKernelGGL((mnc_dp_align<4>), dim3(n_wg), dim3(256), 0, st, B, ws, state_max, p_max, cig_max, 0, 0, 0, big_pass, 0, 0);
	} else if (forms & 1)
		hipLaunchKernelGGL((mnc_dp_align<1>), dim3(n_wg), dim3(64), (size_t)lds_state + lds_p + lds_cig * 4, st, B, ws, state_max, p_max, cig_max,
		                   lds_state, lds_p, lds_cig, big_pass, 0, 0);
}
void launch_dp_stitch(const Batch &B, const int32_t *work_list, int32_t *next_list, int max_read_len, int n_wg, hipStream_t st)
{
	// query codes of a whole read; a region's target span is longer by its deletions: a quarter more
	auto dims = [](int len, int q_cap, int &q_max, int &t_max) {
		q_max = (len + 63) / 64 * 64;
		if (q_max > q_cap) q_max = q_cap;
		if (q_max < 256) q_max = 256;
		t_max = (q_max + q_max / 4 + 127) / 64 * 64;
	};
	auto launch = [&](int q_max, int t_max, int q_lo, int t_lo, int q_hi, int t_hi, int ev_max, int wgs) {
		const size_t lds = (size_t)ST_CIG_MAX * 4 + ev_max + q_max / 4 + t_max / 2 + 16;   // two bits a base of the read, four of the contig; + 16: the stitch reads a word past the one with a region's last base
		hipLaunchKernelGGL(mnc_dp_stitch, dim3(wgs), dim3(64), lds, st, B, work_list, next_list, q_max, t_max, q_lo, t_lo, q_hi, t_hi, ev_max);
	};
	constexpr int ST_SMALL = 6144;                   // up to here one launch: nine regions per CU (17.6 .. 20 KB of LDS each)
	int q_max, t_max;
	dims(max_read_len, ST_LONG_Q, q_max, t_max);
	if (q_max <= ST_SMALL) { launch(q_max, t_max, -1, -1, INT32_MAX, INT32_MAX, ST_EV_MAX, n_wg); return; }
	// long reads in the batch: a launch per LDS class, each region in the smallest that holds it -- 6 k query bases (nine
	// regions per CU), 12 k (four), 24 k (two), then whatever the batch's longest read needs (up to 60 k query bases: one
	// region per CU; beyond that the bases are read in place).  The long classes also have room for one slide per CIGAR
	// operation of mm_fix_cigar's parallel form (a 60 kb read with 10 % errors has ~8 000 of them).  With two classes only,
	// half the regions of a nanopore-like length mix (median 6 kb) ran one per CU: 5.4 ms of a 34 ms batch.
	int q_lo = -1, t_lo = -1, share = 1;
	for (int cls = ST_SMALL; cls < q_max; cls *= 2, share *= 2) {
		int q_c, t_c;
		dims(cls, cls, q_c, t_c);
		launch(q_c, t_c, q_lo, t_lo, q_c, t_c, cls / 6 < ST_EV_MAX ? ST_EV_MAX : (cls / 6 + 15) / 16 * 16, n_wg / share);
		q_lo = q_c, t_lo = t_c;
	}
	launch(q_max, t_max, q_lo, t_lo, INT32_MAX, INT32_MAX, ST_EV_LONG, n_wg / 4);
}
// what the waves of one round's launches (one per LDS class) can hold back of the
// region pool: the chunk each reserved last
size_t dp_stitch_pool_slack(int n_wg) { return (size_t)(2 * n_wg + n_wg / 4) * ST_POOL_CHUNK; }   // (n_wg + n_wg / 2 + n_wg / 4 + ... for the classes, n_wg / 4 for the last)
size_t dp_inv_ws_words(int max_gap) { return (size_t)(max_gap + 8 + 64) * 2; }   // H and E of one column, per workgroup
void launch_dp_inv(const Batch &B, const int32_t *work_list, int32_t *next_list, int32_t *ws, int n_wg, hipStream_t st)
{
	hipLaunchKernelGGL(mnc_dp_inv, dim3(n_wg), dim3(64), 0, st, B, work_list, next_list, ws, B.max_gap + 8 + 64);
}
int dp_stitch_prepare()
{
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_dp_stitch), hipFuncAttributeMaxDynamicSharedMemorySize,
	                                   ST_CIG_MAX * 4 + ST_EV_LONG + ST_LONG_Q / 4 + (ST_LONG_Q + ST_LONG_Q / 4 + 127) / 64 * 64 / 2 + 16);
	if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}

#ifdef MNC_WG_TIMING
} // namespace mnc
extern "C" int mnc_debug_wg_cycles(long long *out8, int reset)
{
	unsigned long long z[8] = {0};
	if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(mnc::g_wg_cycles), sizeof(z)) != hipSuccess) return -1;
	if (reset && hipMemcpyToSymbol(HIP_SYMBOL(mnc::g_wg_cycles), z, sizeof(z)) != hipSuccess) return -1;
	return 0;
}
namespace mnc {
#endif
} // namespace mnc
