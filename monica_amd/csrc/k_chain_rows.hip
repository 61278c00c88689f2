// Stage kernel K4+K5, row form: chaining DP and backtrack with the anchors of a read held in
// LDS -- gfx950.
//
// Replaces mm_chain_dp() inside index.map(seq) (monica/genomes/aligner.py:193,215; SURVEY.md
// Appendix A.5) for reads with at most 4096 anchors and fewer than 65 536 bases; longer
// reads take the sequential kernels of k_chain.hip.
//
// Mapping.  One wave = two half-waves of 32 lanes = two reads.  A half-wave evaluates 32
// candidate predecessors j = jb, jb-1, ... of its current anchor i per step; a chain anchor
// needs about 27 candidates before minimap2's max_skip rule stops the scan, i.e. one step.
// The order-dependent parts of the sequential loop are reproduced exactly with scans over
// the half-wave (DPP row_shr inside a 16-lane row, row_bcast:15 into the upper row):
//   * running maximum (strict '>' updates)      -> exclusive prefix-max
//   * n_skip (decrement-with-floor / increment) -> prefix sum + prefix max (see below)
//   * break at the first lane where n_skip > max_skip, argmax = first lane at the maximum
// The two halves advance independently (no lock-step over i).
//
// LDS per anchor, 14 bytes: one 64-bit word {p:16, f:16, t:16, v:16}, the low 32 bits of the
// reference coordinate, the 16-bit query position.  t[] holds the "seen for anchor i" stamp
// of the skip rule; v[] the peak score along the chain.  Scores fit 16 bits because a chain
// score never exceeds the read length.
#include "device.h"

namespace mnc {

constexpr int ROWS = 2;                       // reads per wave
constexpr int RW = 64 / ROWS;                 // lanes per read
constexpr uint32_t NONE16 = 0xffffu;
constexpr int NEG = -(1 << 24);
#ifndef INT32_MIN
#define INT32_MIN (-2147483647 - 1)
#endif

// ---------------------------------------------------------------- DPP row primitives
// VOP2 with a DPP source: lanes whose source lane falls outside the row are disabled
// (bound_ctrl:0), i.e. keep their value -- exactly "combine with the identity".  hipcc pads
// nothing inside an asm statement, so every op carries the two wait states a DPP read needs
// after a VALU write of the same register.
#define MNC_DPP_OP(NAME, INSN, CTRL) \
	__device__ __forceinline__ int NAME(int v) { \
		asm("s_nop 1\n\t" INSN " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(v)); return v; }
MNC_DPP_OP(max_shr1, "v_max_i32_dpp", "row_shr:1") MNC_DPP_OP(max_shr2, "v_max_i32_dpp", "row_shr:2")
MNC_DPP_OP(max_shr4, "v_max_i32_dpp", "row_shr:4") MNC_DPP_OP(max_shr8, "v_max_i32_dpp", "row_shr:8")
MNC_DPP_OP(add_shr1, "v_add_u32_dpp", "row_shr:1") MNC_DPP_OP(add_shr2, "v_add_u32_dpp", "row_shr:2")
MNC_DPP_OP(add_shr4, "v_add_u32_dpp", "row_shr:4") MNC_DPP_OP(add_shr8, "v_add_u32_dpp", "row_shr:8")
MNC_DPP_OP(max_ror1, "v_max_i32_dpp", "row_ror:1") MNC_DPP_OP(max_ror2, "v_max_i32_dpp", "row_ror:2")
MNC_DPP_OP(max_ror4, "v_max_i32_dpp", "row_ror:4") MNC_DPP_OP(max_ror8, "v_max_i32_dpp", "row_ror:8")
#undef MNC_DPP_OP
// lane 15 of rows 0 and 2 combined into every lane of rows 1 and 3 (row_mask 0xa)
__device__ __forceinline__ int max_bcast15(int v)
{
	asm("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v)); return v;
}
__device__ __forceinline__ int add_bcast15(int v)
{
	asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v)); return v;
}

// scans / reductions over one 32-lane half
__device__ __forceinline__ int row_incl_max(int v) { return max_bcast15(max_shr8(max_shr4(max_shr2(max_shr1(v))))); }
__device__ __forceinline__ int row_incl_add(int v) { return add_bcast15(add_shr8(add_shr4(add_shr2(add_shr1(v))))); }
__device__ __forceinline__ int row_all_max(int v)
{
	v = max_ror1(max_ror2(max_ror4(max_ror8(v))));                       // every 16-lane row: its maximum
	const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
	return max((int)sw[0], (int)sw[1]);                                  // pair the two rows of a half
}

// lane l <- lane l-1 of the half; lane 0 of the half gets `fill`
__device__ __forceinline__ int row_shift1(int v, int fill, int lr)
{
	int r = fill;
	asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v));
	return lr == 0 ? fill : r;
}

__device__ __forceinline__ uint32_t row_ballot(bool pred, int row)
{
	return (uint32_t)(__ballot(pred) >> (row * RW));
}

__device__ __forceinline__ void lds_order()
{
	// LDS operations of one wave execute in issue order; this only pins the compiler
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------- the kernel
// word layout: bits 0-15 p, 16-31 f, 32-47 t, 48-63 v
__global__ __launch_bounds__(64) void mnc_chain_rows(Batch B, const uint32_t *list, uint32_t count, int NM, int store_fp)
{
	extern __shared__ __align__(16) uint8_t smem[];
	const int lane = threadIdx.x, row = lane / RW, lr = lane % RW;
	uint8_t *rbase = smem + (size_t)row * ((size_t)NM * 14 + 64 * 8);
	uint64_t *W = reinterpret_cast<uint64_t*>(rbase);                  // NM words
	uint32_t *xlo = reinterpret_cast<uint32_t*>(rbase + (size_t)NM * 8);
	uint16_t *qp = reinterpret_cast<uint16_t*>(rbase + (size_t)NM * 12);
	uint64_t *ubuf = reinterpret_cast<uint64_t*>(rbase + (size_t)NM * 14);   // 64 chain-end keys
	uint16_t *W16 = reinterpret_cast<uint16_t*>(W);
	uint32_t *W32 = reinterpret_cast<uint32_t*>(W);

	const uint32_t li = blockIdx.x * ROWS + row;
	const bool has = li < count;
	const uint32_t r = has ? list[li] : 0;
	const int64_t a_off = has ? B.an_off[r] : 0;
	const int n = has ? (int)(B.an_off[r + 1] - a_off) : 0;
	const Anchor *ga = B.a + a_off;
	const int span = KMER;
	const int max_gap = B.max_gap, bw = B.bw, max_skip = B.max_skip, max_iter = B.max_iter;
	const double avg_span = (double)(float)KMER;

	// ---- load: low coordinate, query position, segment-start flag (kept in f until f[i] is set)
	for (int idx = lr; idx < n; idx += RW) {
		const Anchor e = ga[idx];
		const uint32_t hi = (uint32_t)(e.x >> 32);
		const uint32_t phi = idx ? (uint32_t)(ga[idx - 1].x >> 32) : ~hi;
		xlo[idx] = (uint32_t)e.x;
		qp[idx] = (uint16_t)e.y;
		W[idx] = (uint64_t)NONE16 | (uint64_t)(hi != phi ? 1u : 0u) << 16;
	}
	lds_order();

	// ---- DP: every row walks its own i.  Row-uniform state lives in VGPRs.
	// The predecessor window of anchor i is { j < i : same (strand, contig) segment,
	// x_i - x_j <= max_gap, i - j <= max_iter }: contiguous because anchors are sorted, so it
	// is tested per lane instead of keeping minimap2's running start index.
	bool active = n > 0, fresh = true;
	int i = 0, seg = 0, jb = -1, max_f = span, max_j = -1, ns_prev = 0;
	uint32_t xi = active ? xlo[0] : 0;
	int qi = active ? (int)qp[0] : 0;
	int pend_i = -1, pend_f = 0;
	uint32_t pend_vp = 0;
	while (__any(active)) {
		// ---- one step: RW candidates j = jb - lr
		const int j = jb - lr;
		const int lo = max(seg, i - max_iter);
		const bool inb = active && j >= lo;
		const int jc = inb ? j : 0;
		const uint64_t w = W[jc];
		const uint32_t xj = xlo[jc];
		const int qj = (int)qp[jc];
		// operands of the next anchor, fetched early (used when this anchor completes)
		const int inext = min(i + 1, NM - 1);
		const uint32_t nxi = xlo[inext];
		const int nqi = (int)qp[inext];
		const uint32_t nflag = (W32[2 * inext] >> 16) & 1u;

		const uint32_t pj = (uint32_t)w & 0xffffu;
		const int fj = (int)((uint32_t)w >> 16);
		const uint32_t dru = xi - xj;
		const bool inwin = inb && dru <= (uint32_t)max_gap;
		const int dr = (int)dru;
		const int dq = qi - qj;
		const int dd = dr > dq ? dr - dq : dq - dr;
		const bool ev = inwin && dr != 0 && dq > 0 && dq <= max_gap && dd <= bw;
		const int mind = dq < dr ? dq : dr;
		// gap cost (int)(dd * .01 * avg_span) + (ilog2(dd) >> 1): IEEE double products, as minimap2
		const int gap = (int)((double)dd * .01 * avg_span) + (dd > 0 ? (31 - __clz(dd)) >> 1 : 0);
		const int sc = ev ? (mind > span ? span : mind) - gap + fj : NEG;
		if (ev && pj != NONE16) W16[4 * pj + 2] = (uint16_t)i;         // t[p[j]] = i
		lds_order();
		const bool tflag = ev && W16[4 * jc + 2] == (uint16_t)i;       // t[j] == i
		// running maximum: strict '>' against everything before this lane
		const int incl = row_incl_max(sc);
		const int excl = max(row_shift1(incl, NEG, lr), max_f);
		const bool improve = ev && sc > excl;
		// n_skip after each lane: maps x -> max(x + a, b) with (a,b) = (-1,0) on an improvement,
		// (+1,-inf) on a seen non-improvement, identity otherwise; with S = prefix sum of a,
		// the value is max(S, S + max_{k<=l, improve_k}(-S_k)).  The carry enters at lane 0.
		int a = improve ? -1 : (tflag ? 1 : 0);
		// carry: n_skip after the last lane of the previous step of this anchor
		const int c0 = __builtin_amdgcn_readlane(ns_prev, RW - 1), c1 = __builtin_amdgcn_readlane(ns_prev, 2 * RW - 1);
		if (lr == 0 && !fresh) a += row ? c1 : c0;
		const int S = row_incl_add(a);
		const int M = row_incl_max(improve ? -S : NEG);
		const int ns = max(S, S + M);
		const bool brk = tflag && !improve && ns > max_skip;
		const uint32_t bm = row_ballot(brk, row);
		const int bl = bm ? __ffs((int)bm) - 1 : RW - 1;
		// best candidate among the lanes the sequential loop reaches; first lane wins ties
		const int mk = row_all_max(lr <= bl && ev ? sc * RW + (RW - 1 - lr) : INT32_MIN);
		if ((mk >> 5) > max_f) max_f = mk >> 5, max_j = jb - (RW - 1 - (mk & (RW - 1)));
		const uint32_t wm = row_ballot(inwin, row);
		const bool done = active && (bm != 0 || (wm >> (RW - 1)) == 0 || jb - RW < lo);
		ns_prev = ns, fresh = false;
		if (done) {
			if (lr == 0) {
				// v[] of the previous anchor completes now (its operand was fetched a step ago)
				if (pend_i >= 0) W16[4 * pend_i + 3] = (uint16_t)max((uint32_t)pend_f, pend_vp);
				W32[2 * i] = (max_j >= 0 ? (uint32_t)max_j : NONE16) | (uint32_t)max_f << 16;
				lds_order();
				pend_vp = max_j >= 0 ? (uint32_t)W16[4 * max_j + 3] : 0u;
				pend_i = i, pend_f = max_f;
			}
			++i;
			if (i >= n) active = false;
			xi = nxi, qi = nqi;
			if (nflag) seg = i;
			jb = i - 1, max_f = span, max_j = -1, fresh = true;
		} else if (active) jb -= RW;
		lds_order();
	}
	if (lr == 0 && pend_i >= 0) W16[4 * pend_i + 3] = (uint16_t)max((uint32_t)pend_f, pend_vp);
	lds_order();

	if (store_fp) {                                                  // stage dumps for the parity tests
		for (int idx = lr; idx < n; idx += RW) {
			const uint64_t w = W[idx];
			const uint32_t p = (uint32_t)w & 0xffffu;
			B.f[a_off + idx] = (int32_t)((uint32_t)w >> 16);
			B.p[a_off + idx] = p == NONE16 ? -1 : (int32_t)p;
			B.v[a_off + idx] = (int32_t)(w >> 48);
		}
	}

	// ---- backtrack.  (A) which anchors are somebody's predecessor
	for (int idx = lr; idx < n; idx += RW) W16[4 * idx + 2] = 0;
	lds_order();
	for (int idx = lr; idx < n; idx += RW) {
		const uint32_t p = (uint32_t)W[idx] & 0xffffu;
		if (p != NONE16) W16[4 * p + 2] = 1;
	}
	lds_order();
	// (B) chain ends with peak >= min_sc; walk each back to its peak
	int n_u = 0;
	uint64_t *gu = B.u + a_off;
	const int n_up = (n + RW - 1) / RW * RW;
	for (int base = 0; base < n_up; base += RW) {
		const int idx = base + lr;
		bool is_end = false;
		uint64_t key = 0;
		if (idx < n) {
			uint64_t w = W[idx];
			is_end = ((uint32_t)(w >> 32) & 0xffffu) == 0 && (int)(w >> 48) >= B.min_sc;
			if (is_end) {
				int jj = idx;
				while ((uint32_t)((uint32_t)w >> 16) < (uint32_t)(w >> 48)) {     // f < v
					const uint32_t p = (uint32_t)w & 0xffffu;
					if (p == NONE16) { jj = -1; break; }
					jj = (int)p, w = W[jj];
				}
				if (jj < 0) jj = idx, w = W[idx];
				key = (uint64_t)((uint32_t)w >> 16) << 32 | (uint32_t)jj;
			}
		}
		const uint32_t bits = row_ballot(is_end, row);
		if (is_end) {
			const int pos = n_u + __popc(bits & ((1u << lr) - 1u));
			if (pos < 64) ubuf[pos] = key;
			gu[pos] = key;
		}
		n_u += __popc(bits);
	}
	lds_order();
	// (C) order the ends: score descending, then index descending (keys are distinct)
	const bool u_in_lds = n_u <= 64;
	if (u_in_lds) {
		uint64_t mine[4];
		int rank[4];
#pragma unroll
		for (int s = 0; s < 4; ++s) {
			const int e = lr + RW * s;
			mine[s] = e < n_u ? ubuf[e] : 0, rank[s] = 0;
		}
		for (int k = 0; k < n_u; ++k) {
			const uint64_t o = ubuf[k];
#pragma unroll
			for (int s = 0; s < 4; ++s) rank[s] += o > mine[s];
		}
		lds_order();
#pragma unroll
		for (int s = 0; s < 4; ++s) if (lr + RW * s < n_u) ubuf[rank[s]] = mine[s];
	} else if (lr == 0 && n_u > 0) {
		// rare: many chain ends; heap-sort ascending in HBM and read it backwards below
		for (int start = n_u / 2 - 1; start >= 0; --start) {
			int root = start;
			for (;;) {
				int c = 2 * root + 1;
				if (c >= n_u) break;
				if (c + 1 < n_u && gu[c] < gu[c + 1]) ++c;
				if (gu[root] >= gu[c]) break;
				uint64_t x = gu[root]; gu[root] = gu[c], gu[c] = x;
				root = c;
			}
		}
		for (int end = n_u - 1; end > 0; --end) {
			uint64_t x = gu[0]; gu[0] = gu[end], gu[end] = x;
			int root = 0;
			for (;;) {
				int c = 2 * root + 1;
				if (c >= end) break;
				if (c + 1 < end && gu[c] < gu[c + 1]) ++c;
				if (gu[root] >= gu[c]) break;
				uint64_t y = gu[root]; gu[root] = gu[c], gu[c] = y;
				root = c;
			}
		}
	}
	// (D) clear the "used" marks
	for (int idx = lr; idx < n; idx += RW) W16[4 * idx + 2] = 0;
	lds_order();
	// (E) best-first backtrack by the row's first lane; chain records in backtrack order
	if (lr == 0 && has) {
		ChainRec *out = B.chains_tmp + a_off / 3;
		int k = 0;
		for (int e = 0; e < n_u; ++e) {
			const uint64_t key = u_in_lds ? ubuf[e] : gu[n_u - 1 - e];
			const int peak = (int)(key >> 32);
			const int last = (int)(uint32_t)key;
			int j = last, first = last, cnt = 0, mlen = span, blen = span, score = -1;
			uint32_t nx = 0;
			int nq = 0;
			for (;;) {
				const uint64_t w = W[j];
				if (cnt > 0 && ((uint32_t)(w >> 32) & 0xffffu) != 0) {       // reached a used anchor
					const int rest = peak - (int)((uint32_t)w >> 16);
					if (rest >= B.min_sc) score = rest;
					break;
				}
				const uint32_t cx = xlo[j];
				const int cq = (int)qp[j];
				if (cnt > 0) {
					const int tl = (int)(nx - cx), ql = nq - cq;
					blen += tl > ql ? tl : ql;
					mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
				}
				W16[4 * j + 2] = 1;
				nx = cx, nq = cq, first = j, ++cnt;
				const uint32_t p = (uint32_t)w & 0xffffu;
				if (p == NONE16) { score = peak; break; }
				j = (int)p;
			}
			if (score >= 0 && cnt >= B.min_cnt) {
				ChainRec c;
				const Anchor af = ga[first], al = ga[last];
				c.x0 = af.x, c.y0 = af.y, c.x1 = al.x, c.y1 = al.y;
				c.score = score, c.cnt = cnt, c.mlen = mlen, c.blen = blen, c.as = 0, c.pad = k;
				out[k++] = c;
			}
		}
		B.n_chain[r] = k;
	}
}

int chain_rows_prepare(size_t max_lds)
{
	// dynamic LDS above 64 KiB has to be opted into once per function
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_chain_rows),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
	if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}

size_t chain_rows_lds_bytes(int NM) { return (size_t)ROWS * ((size_t)NM * 14 + 64 * 8); }

void launch_chain_rows(const Batch &B, const uint32_t *list, uint32_t count, int NM, int store_fp, hipStream_t st)
{
	if (count == 0) return;
	const unsigned blocks = (count + ROWS - 1) / ROWS;
	hipLaunchKernelGGL(mnc_chain_rows, dim3(blocks), dim3(64), chain_rows_lds_bytes(NM), st, B, list, count, NM, store_fp);
}

} // namespace mnc
