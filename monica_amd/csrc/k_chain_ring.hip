// Stage kernels K4 (chaining DP, ring form) and K5 (backtrack, LDS form) -- gfx950.
//
// Replaces mm_chain_dp() inside index.map(seq) (monica/genomes/aligner.py:193,215; SURVEY.md
// Appendix A.5).
//
// K4 mapping.  One wave = two half-waves of 32 lanes = two reads.  A half-wave evaluates 32
// candidate predecessors j = jb, jb-1, ... of its current anchor i per step; a chain anchor
// needs about 27 candidates before minimap2's max_skip rule stops the scan, i.e. one step.
// The order-dependent parts of the sequential loop are reproduced exactly with scans over the
// half-wave (DPP row_shr inside a 16-lane row, row_bcast:15 into the upper row):
//   * running maximum (strict '>' updates)      -> inclusive prefix-max of (score, lane) keys
//   * n_skip (decrement-with-floor / increment) -> prefix sum + prefix max (see below)
//   * break at the first lane where n_skip > max_skip; argmax = the key scan at that lane
// The two halves advance independently (no lock-step over i).
//
// K4 memory.  The DP only ever looks a short way back, so a read keeps just a RING of its
// anchors in LDS: blocks of 32, the block being filled, PAST completed blocks behind it and
// the prefetched next block (128 entries x 24 bytes with PAST = 2).  A completed block's f/p/v
// go to HBM at once (coalesced); a block's skip-rule stamps t[] follow when it is evicted.
// The rare candidate or stamp that lies behind the ring is read / written in HBM instead, so
// there is no limit on anchors per read and LDS no longer caps occupancy.  (The first form of
// this kernel held all anchors of a read in LDS: 2.75 waves/SIMD, 8.2 ms; profiles/README.md.)
//
// K5.  The backtrack needs random access to a whole read, so it is its own kernel with the
// read's {p | owner, f | v} in LDS (8 bytes per anchor, size classes; coordinates are read from
// HBM only for the anchors a chain owns), two reads per wave; it has no sequential walk (see
// mnc_chain_tail).  Reads beyond the largest class
// take the sequential mnc_chain_backtrack of k_chain.hip.
#include "device.h"

namespace mnc {

constexpr int ROWS = 2;                       // reads per wave
constexpr int RW = 64 / ROWS;                 // lanes per read = ring block size
constexpr uint32_t NONE16 = 0xffffu;
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
// the builtin takes the condition as it is (the __any / __ballot wrappers go through an integer)
__device__ __forceinline__ bool any64(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0; }

__device__ __forceinline__ uint32_t row_ballot(bool pred, int row)
{
	return (uint32_t)(__builtin_amdgcn_ballot_w64(pred) >> (row * RW));
}

__device__ __forceinline__ void lds_order()
{
	// LDS operations of one wave execute in issue order; this only pins the compiler
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	asm volatile("" ::: "memory");
}

// ================================================================ K4: DP over a ring
// PAST = completed blocks kept behind the current one (2 in production; 0 in the stress build
// the tests use to drive every look-back through the HBM fall-back).
constexpr int DP_WAVES = 4;                   // waves per workgroup (no workgroup barrier is used)

// first DPP step of an inclusive scan, out of place: lanes without a source lane combine with 0
// (bound_ctrl), which is the identity for the non-negative keys and for sums
__device__ __forceinline__ int max_shr1_from(int v)
{
	int r;
	asm("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r) : "v"(v));
	return r;
}

template <int PAST>
__global__ __launch_bounds__(64 * DP_WAVES) void mnc_chain_dp_ring(Batch B, const uint32_t *lists, ClassSpans spans)
{
	constexpr int NBLK = PAST + 2;                                       // past + current + next
	constexpr int RING = NBLK * RW;
	static_assert((RING & (RING - 1)) == 0 && RW == 32, "ring slots are taken with a mask");
	auto slot = [](int x) { return (int)((uint32_t)x & (uint32_t)(RING - 1)); };
	// one 24-byte record per ring slot: every field of a candidate comes from one address
	struct Slot {
		int p, f;                                                        // p = -2 marks a segment start until set
		int x, q;                                                        // low 32 bits of x, query position
		int t, v;
	};
	__shared__ Slot s_rows[ROWS * DP_WAVES][RING];
	const int lane = threadIdx.x & 63, row = lane / RW, lr = lane % RW;
	const int slot_row = (threadIdx.x >> 6) * ROWS + row;
	Slot *S = s_rows[slot_row];

	// one launch for every read of the batch, in size-class order (neighbours in a wave have
	// similar anchor counts): ordinal li -> (class c, index in its list)
	const uint32_t li = blockIdx.x * (ROWS * DP_WAVES) + slot_row;
	const bool has = li < spans.start[spans.n];
	uint32_t r = 0;
	if (has) {
		int c = 0;
		while (li >= spans.start[c + 1]) ++c;
		r = lists[(size_t)c * spans.stride + (li - spans.start[c])];
	}
	const int64_t a_off = has ? B.an_off[r] : 0;
	const int n = has ? (int)(B.an_off[r + 1] - a_off) : 0;
	const Anchor *ga = B.a + a_off;
	int32_t *gf = B.f + a_off, *gp = B.p + a_off, *gv = B.v + a_off, *gt = B.t + a_off;
	const int span = KMER;
	const int max_gap = B.max_gap, bw = B.bw, max_skip = B.max_skip, max_iter = B.max_iter;
	__shared__ uint8_t s_gap[GAP_LUT];                                   // gap costs stay below 256 for dd < GAP_LUT
	for (int k = threadIdx.x; k < GAP_LUT; k += 64 * DP_WAVES) s_gap[k] = (uint8_t)B.gap_lut[k];
	__syncthreads();

	// block b of the read -> its ring slots.  An anchor whose nearest neighbour below (in the
	// sorted order) is on another (strand, contig) or more than max_gap away has no predecessor
	// at all: it gets its final f = v = span, p = -1 here and the DP walks past it.  Returns the
	// half's mask of the other ("busy") anchors of the block.
	auto load_block = [&](int b) -> uint32_t {
		const int idx = b * RW + lr;
		bool busy = false;
		if (idx < n) {
			const Anchor e = ga[idx];
			if (idx > 0) {
				const uint64_t px = ga[idx - 1].x;
				busy = (uint32_t)(e.x >> 32) == (uint32_t)(px >> 32) && (uint32_t)e.x - (uint32_t)px <= (uint32_t)max_gap;
			}
			Slot w;
			w.p = -1, w.f = busy ? 0 : span, w.x = (int)(uint32_t)e.x, w.q = (int)(uint32_t)e.y, w.t = 0, w.v = busy ? 0 : span;
			S[slot(idx)] = w;
		}
		return row_ballot(busy, row);
	};
	uint32_t busy_cur = load_block(0), busy_next = load_block(1);
	lds_order();

	// ---- DP: every half walks its own i.  Half-uniform state lives in VGPRs.
	// The predecessor window of anchor i is { j < i : same (strand, contig) segment,
	// x_i - x_j <= max_gap, i - j <= max_iter }: contiguous because anchors are sorted, so it
	// is tested per lane instead of keeping minimap2's running start index.
	// Scores enter the scans as key = (score + KEY_BIAS) * 32 + (31 - lane) > 0 (a score is at
	// least span + 1 - max gap cost > -KEY_BIAS), 0 = no candidate: maxima of keys break ties
	// towards the first lane, and 0 is the identity the DPP scans fill in.
	constexpr int KEY_BIAS = 256;
	bool fresh = true;
	int i = 0, seg = 0, jb = -1, max_f = span, max_j = -1, ns_prev = 0, ring_lo = 0;   // seg: the last anchor <= i without a predecessor
	uint32_t xi = n > 0 ? (uint32_t)S[0].x : 0;
	int qi = n > 0 ? S[0].q : 0;
	const uint64_t last_lane = 1ULL << (row * RW + RW - 1);
	int key_lane = KEY_BIAS * RW + RW - 1 - lr, perm_half = (lane & RW) << 2;
	asm volatile("" : "+v"(key_lane), "+v"(perm_half));                  // keep them in registers: one v_lshl_add / v_lshl_or at their use
	Slot *me = S;                                                        // slot of anchor i
	for (;;) {
		const bool active = i < n;
		if (!any64(active)) break;
		// ---- one step: RW candidates j = jb - lr
		const int j = jb - lr;
		const int lo = max(seg, i - max_iter);
		const bool inb = active && j >= lo;
		const bool far = inb && j < ring_lo;                              // behind the ring: HBM
		const Slot *c = S + slot(j);                                      // any slot will do for lanes outside [lo, jb]
		int pj = c->p, fj = c->f, xj = c->x, qj = c->q;
		if (far) {
			// volatile: the compiler must not merge these with the LDS reads into flat loads
			const volatile uint32_t *ge = reinterpret_cast<const volatile uint32_t*>(ga + (uint32_t)j);
			xj = (int)ge[0], qj = (int)ge[2];
			pj = *(const volatile int32_t*)(gp + (uint32_t)j), fj = *(const volatile int32_t*)(gf + (uint32_t)j);
		}
		// the next anchor to work on: past the run of anchors without predecessors that follows i
		// (inside this block; the block boundary itself is always visited).  Its operands are
		// fetched early (used when this anchor completes).
		const int ik = i & (RW - 1);
		// busy anchors after i, with a sentinel bit at the block boundary: branch-free
		const uint32_t rest = ((busy_cur >> 1) >> ik) | (0x80000000u >> ik);
		const int inext = i + 1 + __builtin_ctz(rest);
		Slot *nx = S + slot(inext);
		const int nxx = nx->x, nxq = nx->q;

		uint32_t dru = xi - (uint32_t)xj;
		const int dr = (int)dru;
		if (!inb) dru = 0xffffffffu;
		asm volatile("" : "+v"(dru));                                     // one compare gives the window mask
		const bool inwin = dru <= (uint32_t)max_gap;
		const uint64_t wm = __builtin_amdgcn_ballot_w64(inwin);
		const int dq = qi - qj;
		const int dd = dr > dq ? dr - dq : dq - dr;
		const bool ev = inwin && dr != 0 && dq > 0 && dq <= max_gap && dd <= bw;
		const int mind = dq < dr ? dq : dr;
		// gap cost (int)(dd * .01 * avg_span) + (ilog2(dd) >> 1): a table the host fills with IEEE
		// double products, as minimap2 evaluates it (dd <= bw < GAP_LUT wherever `ev` holds)
		const int gap = s_gap[min(dd, GAP_LUT - 1)];
		const int sc = (mind > span ? span : mind) - gap + fj;
		const int key = ev ? (sc << 5) + key_lane : 0;
		// t[p[j]] = i, then t[j] == i
		const bool mark = ev && pj >= 0;
		const bool mark_far = mark && pj < ring_lo;
		if (mark && !mark_far) S[slot(pj)].t = i;
		if (mark_far) {                                                   // rare: the stamp goes to HBM, ahead of the reads below
			*(volatile int32_t*)(gt + (uint32_t)pj) = i;
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
		}
		lds_order();
		int tj = c->t;
		if (far) tj = *(const volatile int32_t*)(gt + (uint32_t)j);
		const bool tflag = ev && tj == i;
		// running maximum, strict '>' against everything before this lane: keys are distinct and
		// order equal scores by lane, so a lane beats all earlier ones iff it is the inclusive maximum
		const int incl = max_bcast15(max_shr8(max_shr4(max_shr2(max_shr1_from(key)))));
		const bool improve = ev && incl == key && sc > max_f;
		// n_skip after each lane: maps x -> max(x + a, b) with (a,b) = (-1,0) on an improvement,
		// (+1,-inf) on a seen non-improvement, identity otherwise; with S = prefix sum of a,
		// the value is max(S, S + max_{k<=l, improve_k}(-S_k)).  The carry enters at lane 0.
		int a = improve ? -1 : (tflag ? 1 : 0);
		if (any64(active && !fresh)) {                                   // rare: an anchor needing a second step
			const int c0 = __builtin_amdgcn_readlane(ns_prev, RW - 1), c1 = __builtin_amdgcn_readlane(ns_prev, 2 * RW - 1);
			if (lr == 0 && !fresh) a += row ? c1 : c0;
		}
		const int Sa = row_incl_add(a);
		const int M = row_incl_max(-Sa);                                  // over all lanes: the floor never binds off an improvement
		const int ns = max(Sa, Sa + M);
		// first lane where the sequential loop breaks (lane RW-1 if none): one compare gives the mask
		int nsb = tflag && !improve ? ns : INT32_MIN;
		asm volatile("" : "+v"(nsb));
		const uint32_t bm = row_ballot(nsb > max_skip, row);
		const int bl = __builtin_ctz(bm | 0x80000000u);
		// best candidate among the lanes the sequential loop reaches (first lane wins ties) =
		// the inclusive maximum at the break lane
		const int mk = __builtin_amdgcn_ds_bpermute((bl << 2) | perm_half, incl);
		const int best = (mk >> 5) - KEY_BIAS;                            // -KEY_BIAS when there was no candidate
		if (best > max_f) max_f = best, max_j = jb - (RW - 1 - (mk & (RW - 1)));
		const bool done = active && (bm != 0 || (wm & last_lane) == 0 || jb - RW < lo);
		ns_prev = ns, fresh = false;
		if (done) {
			if (lr == 0) {
				// v[i] = max(f[i], v[p[i]]): the operand was written at an earlier step
				me->p = max_j, me->f = max_f;
				const int vp = max_j < 0 ? 0 : max_j >= ring_lo ? S[slot(max_j)].v : gv[(uint32_t)max_j];
				me->v = max(max_f, vp);
			}
			// every anchor walked past has no predecessor; neither has a non-busy block start
			if (inext != i + 1) seg = inext - 1;
			i = inext;
			me = nx;
			xi = (uint32_t)nxx, qi = nxq;
			jb = i - 1, max_f = span, max_j = -1, fresh = true;
			if (any64((i & (RW - 1)) == 0)) {                             // a block boundary (one step in 32): keep it a branch
				if ((i & (RW - 1)) == 0) {
					if (!(busy_next & 1u)) seg = i;
					busy_cur = busy_next;
				}
			}
			if (i < n && (i & (RW - 1)) == 0) {                           // entering block nb
				const int nb = i >> 5;
				lds_order();
				const int fi = (nb - 1) * RW + lr;                        // the block just completed -> HBM
				const Slot *w = S + slot(fi);
				gf[fi] = w->f, gp[fi] = w->p, gv[fi] = w->v;
				const int ob = nb - PAST - 1;                             // block about to lose its slots
				if (ob >= 0) gt[ob * RW + lr] = S[slot(ob * RW + lr)].t;
				lds_order();
				busy_next = load_block(nb + 1);
				ring_lo = max(0, nb - PAST) * RW;
			}
		} else if (active) jb -= RW;
		lds_order();
	}
	lds_order();
	if (n > 0) {                                                         // the last (partial) block
		const int fi = (n - 1) / RW * RW + lr;
		if (fi < n) {
			const Slot *w = S + slot(fi);
			gf[fi] = w->f, gp[fi] = w->p, gv[fi] = w->v;
		}
	}
}

// ================================================================ K5: backtrack in LDS
// minimap2 backtracks best-first: chain ends ordered by peak score, each walked from its peak
// towards the root, stopping at the first anchor an earlier walk has marked (rejected walks
// mark too).  An anchor is therefore taken by the best-ranked start among its descendants
// (itself included): owner(x) = min rank over the starts in x's subtree of the p[] forest, and
// the walk of rank e covers exactly the anchors it owns -- a path from its start upwards.
// Owners come from one backward sweep (p[x] < x): a block of 32 anchors first settles its
// in-block paths by pointer doubling, then hands its owners to the parents in earlier blocks.
// Counts and the match / block lengths of a chain are sums over its (parent, child) pairs, so
// they are accumulated per anchor as well.  Nothing here is sequential in the chain length.
//
// LDS per anchor: word0 = p | owner << 16 (during (A)-(C) the high half is the "is a
// predecessor" flag), word1 = f | v << 16; per read 64 chain-end keys and 64 accumulators.  Reads with more than 64 chain ends take the
// sequential walk below instead.
constexpr int TAIL_ENDS_MAX = 64;
// chain ends (and accumulators) held in LDS per read: short reads rarely have many, and the
// kernel's speed is set by its LDS footprint
__host__ __device__ __forceinline__ int tail_ends(int NM) { return NM <= 512 ? 32 : TAIL_ENDS_MAX; }
constexpr uint32_t OWN_NONE = 0xffffu;

__global__ __launch_bounds__(64) void mnc_chain_tail(Batch B, const uint32_t *lists, ClassSpans spans, int NM)
{
	extern __shared__ __align__(16) uint8_t smem[];
	const int lane = threadIdx.x, row = lane / RW, lr = lane % RW;
	const int TAIL_ENDS = tail_ends(NM);
	uint8_t *rbase = smem + (size_t)row * ((size_t)NM * 8 + TAIL_ENDS * 32);
	uint32_t *W0 = reinterpret_cast<uint32_t*>(rbase);                  // p | owner << 16
	uint32_t *W1 = W0 + NM;                                             // f | v << 16
	uint64_t *ubuf = reinterpret_cast<uint64_t*>(rbase + (size_t)NM * 8);   // TAIL_ENDS chain-end keys
	uint32_t *acc = reinterpret_cast<uint32_t*>(ubuf + TAIL_ENDS);      // TAIL_ENDS x {cnt, mlen, blen, top, stop f, -}
	uint16_t *H0 = reinterpret_cast<uint16_t*>(W0);

	// the reads of one or several adjacent size classes: ordinal -> (class, index in its list)
	const uint32_t li = blockIdx.x * ROWS + row;
	const bool has = li < spans.start[spans.n];
	uint32_t r = 0;
	if (has) {
		int c = 0;
		while (li >= spans.start[c + 1]) ++c;
		r = lists[(size_t)c * spans.stride + (li - spans.start[c])];
	}
	const int64_t a_off = has ? B.an_off[r] : 0;
	const int n = has ? (int)(B.an_off[r + 1] - a_off) : 0;
	const Anchor *ga = B.a + a_off;
	const int span = KMER;
	const int n_up = (n + RW - 1) / RW * RW;

	// ---- load the DP result of the read
	// (four blocks of loads in flight per lane)  The anchors' coordinates stay in HBM: only the
	// anchors a chain owns need them, in (E3), next to their parents'.
	for (int i0 = lr; i0 < n; i0 += 4 * RW) {
		int p[4], f[4], v[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int idx = i0 + k * RW;
			if (idx < n) p[k] = B.p[a_off + idx], f[k] = B.f[a_off + idx], v[k] = B.v[a_off + idx];
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int idx = i0 + k * RW;
			if (idx < n) {
				W0[idx] = p[k] < 0 ? NONE16 : (uint32_t)p[k];
				W1[idx] = ((uint32_t)f[k] & 0xffffu) | (uint32_t)v[k] << 16;
			}
		}
	}
	lds_order();

	// (A) which anchors are somebody's predecessor
	for (int idx = lr; idx < n; idx += RW) {
		const uint32_t p = W0[idx] & 0xffffu;
		if (p != NONE16) H0[2 * p + 1] = 1;
	}
	lds_order();
	// (B) chain ends with peak >= min_sc; walk each back to its peak
	int n_u = 0;
	uint64_t *gu = B.u + a_off;
	for (int base = 0; base < n_up; base += RW) {
		const int idx = base + lr;
		bool is_end = false;
		uint64_t key = 0;
		if (idx < n) {
			uint32_t w0 = W0[idx], w1 = W1[idx];
			is_end = (w0 >> 16) == 0 && (int)(w1 >> 16) >= B.min_sc;
			if (is_end) {
				int jj = idx;
				while ((w1 & 0xffffu) < (w1 >> 16)) {                          // f < v
					const uint32_t p = w0 & 0xffffu;
					if (p == NONE16) { jj = -1; break; }
					jj = (int)p, w0 = W0[jj], w1 = W1[jj];
				}
				if (jj < 0) jj = idx, w1 = W1[idx];
				key = (uint64_t)(w1 & 0xffffu) << 32 | (uint32_t)jj;
			}
		}
		const uint32_t bits = row_ballot(is_end, row);
		if (is_end) {
			const int pos = n_u + __popc(bits & ((1u << lr) - 1u));
			if (pos < TAIL_ENDS) ubuf[pos] = key;
			gu[pos] = key;
		}
		n_u += __popc(bits);
	}
	lds_order();
	const bool u_in_lds = n_u <= TAIL_ENDS;
	ChainRec *out = B.chains_tmp + a_off / 3;

	if (u_in_lds) {
		// (C) order the ends: score descending, then index descending; equal keys (two ends
		// behind one peak) keep their order, so ranks are distinct
		{
			uint64_t mine[2];
			int rank[2];
#pragma unroll
			for (int s = 0; s < 2; ++s) {
				const int e = lr + RW * s;
				mine[s] = e < n_u ? ubuf[e] : 0, rank[s] = 0;
			}
			for (int k = 0; k < n_u; ++k) {
				const uint64_t o = ubuf[k];
#pragma unroll
				for (int s = 0; s < 2; ++s) rank[s] += o > mine[s] || (o == mine[s] && k < lr + RW * s);
			}
			lds_order();
#pragma unroll
			for (int s = 0; s < 2; ++s) if (lr + RW * s < n_u) ubuf[rank[s]] = mine[s];
		}
		// (D) no owners yet; clear the accumulators
		for (int idx = lr; idx < n; idx += RW) H0[2 * idx + 1] = (uint16_t)OWN_NONE;
		for (int k = lr; k < TAIL_ENDS * 6; k += RW) acc[k] = 0;
		lds_order();
		// (E1) a start owns itself unless a better rank starts there too
		for (int e = lr; e < n_u; e += RW) {
			const uint32_t sidx = (uint32_t)ubuf[e];
			atomicMin(&W0[sidx], (uint32_t)e << 16 | (W0[sidx] & 0xffffu));
		}
		lds_order();
		// (E2) owners, blocks from the last to the first
		for (int base = n_up - RW; base >= 0; base -= RW) {
			const int x = base + lr;
			const uint32_t w0 = x < n ? W0[x] : (OWN_NONE << 16 | NONE16);
			const uint32_t px = w0 & 0xffffu;
			uint32_t own = w0 >> 16;
			if (row_ballot(own != OWN_NONE, row) == 0) continue;             // nothing to hand on in this block
			int jl = px != NONE16 && (int)px >= base ? (int)px - base : -1;  // lane of the in-block parent
#pragma unroll
			for (int round = 0; round < 5; ++round) {
				// hand the owner to the ancestor 2^round steps up, then jump twice as far
				const int src = (lane & RW) | (jl < 0 ? lr : jl);
				const uint32_t tp = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)px);      // p of that ancestor
				const int jj = __builtin_amdgcn_ds_bpermute(src << 2, jl);
				if (jl >= 0 && own != OWN_NONE) atomicMin(&W0[base + jl], own << 16 | tp);
				lds_order();
				own = x < n ? W0[x] >> 16 : OWN_NONE;
				jl = jl < 0 ? -1 : jj;
			}
			if (px != NONE16 && (int)px < base && own != OWN_NONE)
				atomicMin(&W0[px], own << 16 | (W0[px] & 0xffffu));
			lds_order();
		}
		// (E3) per-anchor share of its chain: count, (parent, child) lengths, top anchor.  A lane's
		// anchors (32 apart) mostly belong to one chain: sums are kept in registers while the owner
		// stays the same, so the LDS atomics (every lane on the same three words) are few.
		{
			uint32_t c_own = OWN_NONE, c_cnt = 0, c_ml = 0, c_bl = 0;
			auto flush = [&]() {
				if (c_own == OWN_NONE) return;
				uint32_t *A = acc + c_own * 6;
				atomicAdd(&A[0], c_cnt);
				if (c_bl) { atomicAdd(&A[1], c_ml); atomicAdd(&A[2], c_bl); }
			};
			for (int x = lr; x < n; x += RW) {
				const uint32_t w0 = W0[x], own = w0 >> 16, px = w0 & 0xffffu;
				if (own == OWN_NONE) continue;
				if (own != c_own) { flush(); c_own = own, c_cnt = c_ml = c_bl = 0; }
				++c_cnt;
				const bool same = px != NONE16 && (W0[px] >> 16) == own;
				if (same) {
					const Anchor ac = ga[x], ap = ga[px];
					const int tl = (int)((uint32_t)ac.x - (uint32_t)ap.x), ql = (int)(uint32_t)ac.y - (int)(uint32_t)ap.y;
					c_ml += (uint32_t)(tl > span && ql > span ? span : tl < ql ? tl : ql);
					c_bl += (uint32_t)(tl > ql ? tl : ql);
				} else {
					uint32_t *A = acc + own * 6;
					A[3] = (uint32_t)x;                                        // the walk stops above this anchor
					A[4] = px == NONE16 ? 0xffffffffu : (W1[px] & 0xffffu);
				}
			}
			flush();
		}
		lds_order();
		// (E4) chain records in rank order
		int k = 0;
		for (int e0 = 0; e0 < n_u; e0 += RW) {
			const int e = e0 + lr;
			bool ok = false;
			uint32_t cnt = 0, sidx = 0;
			int score = 0;
			if (e < n_u) {
				const uint64_t key = ubuf[e];
				const int peak = (int)(key >> 32);
				sidx = (uint32_t)key;
				const uint32_t *A = acc + e * 6;
				cnt = A[0];
				score = A[4] == 0xffffffffu ? peak : peak - (int)A[4];
				ok = (W0[sidx] >> 16) == (uint32_t)e && (int)cnt >= B.min_cnt && (A[4] == 0xffffffffu || score >= B.min_sc);
			}
			const uint32_t bits = row_ballot(ok, row);
			if (ok) {
				const uint32_t *A = acc + e * 6;
				const int kk = k + __popc(bits & ((1u << lr) - 1u));
				ChainRec c;
				const Anchor af = ga[A[3]], al = ga[sidx];
				c.x0 = af.x, c.y0 = af.y, c.x1 = al.x, c.y1 = al.y;
				c.score = score, c.cnt = (int)cnt, c.mlen = span + (int)A[1], c.blen = span + (int)A[2], c.as = (int)sidx, c.pad = kk;
				out[kk] = c;
			}
			k += __popc(bits);
		}
		if (lr == 0 && has) B.n_chain[r] = k;
		return;
	}

	// ---- rare: more than TAIL_ENDS chain ends.  Heap-sort them in HBM (ascending; read
	// backwards) and walk sequentially on the half's first lane.
	if (lr == 0 && n_u > 0) {
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
	for (int idx = lr; idx < n; idx += RW) H0[2 * idx + 1] = 0;           // marks now mean "used by a chain"
	lds_order();
	if (lr == 0 && has) {
		int k = 0;
		for (int e = 0; e < n_u; ++e) {
			const uint64_t key = gu[n_u - 1 - e];
			const int peak = (int)(key >> 32);
			const int last = (int)(uint32_t)key;
			int j = last, first = last, cnt = 0, mlen = span, blen = span, score = -1;
			uint32_t nx = 0;
			int nq = 0;
			for (;;) {
				const uint32_t w0 = W0[j];
				if (cnt > 0 && (w0 >> 16) != 0) {                              // reached a used anchor
					const int rest = peak - (int)(W1[j] & 0xffffu);
					if (rest >= B.min_sc) score = rest;
					break;
				}
				const Anchor aj = ga[j];
				const uint32_t cx = (uint32_t)aj.x;
				const int cq = (int)(uint32_t)aj.y;
				if (cnt > 0) {
					const int tl = (int)(nx - cx), ql = nq - cq;
					blen += tl > ql ? tl : ql;
					mlen += tl > span && ql > span ? span : tl < ql ? tl : ql;
				}
				H0[2 * j + 1] = 1;
				nx = cx, nq = cq, first = j, ++cnt;
				const uint32_t p = w0 & 0xffffu;
				if (p == NONE16) { score = peak; break; }
				j = (int)p;
			}
			if (score >= 0 && cnt >= B.min_cnt) {
				ChainRec c;
				const Anchor af = ga[first], al = ga[last];
				c.x0 = af.x, c.y0 = af.y, c.x1 = al.x, c.y1 = al.y;
				c.score = score, c.cnt = cnt, c.mlen = mlen, c.blen = blen, c.as = last, c.pad = k;
				out[k++] = c;
			}
		}
		B.n_chain[r] = k;
	}
}

int chain_tail_prepare(size_t max_lds)
{
	// dynamic LDS above 64 KiB has to be opted into once per function
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mnc_chain_tail),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
	if (e != hipSuccess) { set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return MNC_ERR_HIP; }
	return MNC_OK;
}

size_t chain_tail_lds_bytes(int NM) { return (size_t)ROWS * ((size_t)NM * 8 + (size_t)tail_ends(NM) * 32); }

void launch_chain_dp_ring(const Batch &B, const uint32_t *lists, const ClassSpans &spans, int stress, hipStream_t st)
{
	const uint32_t count = spans.start[spans.n];
	if (count == 0) return;
	const unsigned per = ROWS * DP_WAVES, blocks = (count + per - 1) / per;
	if (stress) hipLaunchKernelGGL(mnc_chain_dp_ring<0>, dim3(blocks), dim3(64 * DP_WAVES), 0, st, B, lists, spans);
	else hipLaunchKernelGGL(mnc_chain_dp_ring<2>, dim3(blocks), dim3(64 * DP_WAVES), 0, st, B, lists, spans);
}

void launch_chain_tail(const Batch &B, const uint32_t *lists, const ClassSpans &spans, int NM, hipStream_t st)
{
	const uint32_t count = spans.start[spans.n];
	if (count == 0) return;
	const unsigned blocks = (count + ROWS - 1) / ROWS;
	hipLaunchKernelGGL(mnc_chain_tail, dim3(blocks), dim3(64), chain_tail_lds_bytes(NM), st, B, lists, spans, NM);
}

} // namespace mnc
