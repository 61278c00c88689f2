// Stage kernels K8b / K8c: gap filling between two seeds and the extensions, the common cases of
// the base-level alignment stage -- gfx950.
//
// Replaces the first (approximate-maximum) pass of mm_align1's gap filling: ksw_extd2_sse as a
// GLOBAL alignment of ~200 x 200 bases with a band (1.5 bw + 1 = 751) that never clips such a
// matrix, then mm_test_zdrop on its CIGAR; and the left / right extension calls of the same
// function (monica/genomes/aligner.py:193, 215).
//
// ksw2 fills the whole matrix because its band is wider than the matrix.  The alignment it
// reports lies close to the main diagonal, so the gap-filling kernels fill only a diagonal band and
// then PROVE that the band held the answer: a path that leaves a band of half-width b around the
// diagonals 0 .. tlen - qlen has at least b + 1 gap bases more than it needs in each direction,
// which bounds its score by U = a (matches left) - gap(b + 1 ...) (dp_band_bound); when the
// banded score is strictly above U, every co-optimal path lies inside the band, every value on
// them equals the full matrix's, and the backtrack takes the same turns (ties are between
// co-optimal paths, all inside).  Otherwise the segment goes to the next, wider tier and in the end
// to the literal kernel of k_align.hip (also a CIGAR whose walk shows a Z-drop: that needs the
// exact second pass).
//
// In this file, in order:
//   mnc_dp_fill<64, 4, 2047>   the band in int32, one cell per lane and register -- the round's first
//                              form of the kernel, kept for gaps of 512 .. 2047 bases (256-cell band)
//   mnc_dp_fillp<16|32|64>     the same band on packed 16-bit pairs (32 / 64 / 128 cells): the
//                              batch's dominant kernel; 16 forward passes per wave, one walk per lane
//   mnc_dp_ext<LANES, CPL>     extensions step by step: full anti-diagonals, the maximum of each in
//                              the SSE scan's tie order, ksw_apply_zdrop -- what the packed extension
//                              kernel hands back, and extensions of 257 .. 512 bases (8 cells per lane)
//   mnc_dp_extp<LANES, CPL, RGT>  extensions on packed pairs, one cell per query base
// Direction bytes stream to HBM; scores, gap states and sequences stay in registers / LDS.
#include "device.h"

namespace mnc {

constexpr int FILL_NEG = -(1 << 28);
constexpr int FILL_WIN = 16;                            // rows of direction bytes held in LDS by the backtrack
constexpr int FILL_MIN_BAND = 8;                        // narrower bands are not worth a try
constexpr int FILL_CIG_MAX = 384;                       // CIGAR operations of a gap filling kept in LDS
constexpr size_t FILL_P_SLOT = (size_t)(2 * FILL_MAX_LEN + FILL_WIN) * 64;   // direction bytes of one wave

__device__ __forceinline__ void fill_order()
{
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	asm volatile("" ::: "memory");
}
__device__ __forceinline__ void fill_order_mem()
{
	__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
	asm volatile("" ::: "memory");
}

__device__ __forceinline__ int fill_nt4(uint8_t c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': case 'U': case 'u': return 3;
	default: return 4;
	}
}

// base `pq` of strand `rev` of a read as a code 0 .. 3 (4: anything else).  A read of A C G T only (B.ambig, from the sketch
// stage) is read from that stage's 2-bit words: the switch over the text's letters is a chain of divergent branches, and in
// the packed kernels it was half of what a forward pass costs beside its anti-diagonals
__device__ __forceinline__ int fill_qcode(const Batch &B, bool acgt, int64_t roff, const uint8_t *read, int rlen, int rev, int pq)
{
	const int p = rev ? rlen - 1 - pq : pq;
	if (acgt) {
		const int64_t g = roff + p;
		const int c = (int)(B.packed[g >> 4] >> (30 - 2 * (int)(g & 15)) & 3u);
		return rev ? 3 - c : c;
	}
	const int c = fill_nt4(read[p]);
	return rev ? (c < 4 ? 3 - c : 4) : c;
}

// sixteen bases of strand `rev` of a read of A C G T only, from strand position pq on, as 2-bit codes (base pq + j at bits 2 j):
// on the read's own strand the sixteen fields of the sketch stage's word are turned round; on the other one they lie last base
// first already, and are complemented.  Bases beyond the read's end are whatever the neighbouring words hold.
__device__ __forceinline__ uint32_t fill_qcodes16(const Batch &B, int64_t roff, int rlen, int rev, int pq)
{
	int64_t g0 = rev ? roff + rlen - 16 - pq : roff + pq;         // the first of the sixteen bases in the batch's text
	int skip = 0;
	if (g0 < 0) skip = (int)-g0, g0 = 0;
	const uint32_t hi = B.packed[g0 >> 4], lo = B.packed[(g0 >> 4) + 1];
	const uint32_t sh2 = 2 * (uint32_t)(g0 & 15);
	uint32_t f = sh2 ? __builtin_amdgcn_alignbit(hi, lo, 32 - sh2) : hi;    // base g0 in bits 31:30 .. g0 + 15 in 1:0
	if (rev) return ~f >> (2 * skip);
	f = __builtin_bitreverse32(f);
	return (f & 0x55555555u) << 1 | (f >> 1 & 0x55555555u);
}

__host__ __device__ __forceinline__ int fill_gap(int l, int q, int e, int q2, int e2)
{
	const int g1 = q + e * l, g2 = q2 + e2 * l;
	return g1 < g2 ? g1 : g2;
}

// band around the diagonals 0 .. d (d = tlen - qlen): the 2 * cells offsets i - j in [kmin, kmax] -- on
// an even anti-diagonal the cells hold the even offsets kmin .. kmax - 1, on an odd one the odd offsets
// kmin + 1 .. kmax (kmin even, so that the parity of a step means the same for every segment of a
// wave): every lane is inside the band on every step.  b = what is left on the narrower side.
__host__ __device__ __forceinline__ void fill_band(int n, int m, int cells, int &b, int &kmin, int &kmax)
{
	const int d = n - m, ad = d < 0 ? -d : d;
	b = (2 * cells - 2 - ad) / 2;
	kmin = (d < 0 ? d : 0) - b;
	if (kmin & 1) --kmin;
	kmax = kmin + 2 * cells - 1;
}

// the best score any global path can have that leaves the band (see the header)
__host__ __device__ __forceinline__ int dp_band_bound(int n, int m, int kmin, int kmax, int a, int q, int e, int q2, int e2)
{
	const int d = n - m;
	int U = FILL_NEG;
	{   // above kmax: at least kmax + 1 target-only bases, and kmax + 1 - d query-only ones to come back
		const int D0 = kmax + 1, I0 = kmax + 1 - d, left = n - D0;
		if (left >= 0 && m - I0 >= 0) {
			const int u = a * left - fill_gap(D0, q, e, q2, e2) - fill_gap(I0, q, e, q2, e2);
			U = U > u ? U : u;
		}
	}
	{   // below kmin
		const int I0 = 1 - kmin, D0 = I0 + d, left = m - I0;
		if (left >= 0 && n - D0 >= 0) {
			const int u = a * left - fill_gap(I0, q, e, q2, e2) - fill_gap(D0, q, e, q2, e2);
			U = U > u ? U : u;
		}
	}
	return U;
}

// One anti-diagonal of the banded fill.  ODD = 0: t0 stays, the upper neighbour is the previous
// cell of the band; 1: t0 advances, the left neighbour is the next cell.  `H` holds H of the same
// cells two steps back (same parity) and receives this step's.
template <int LANES, int CPL, int ODD, int MAXLEN>
__device__ __forceinline__ void fill_step(const int n, const int m, const int rows, const int kmin, const int a, const int bmis, const int scN,
                                          const int q, const int e, const int q2, const int e2, const uint8_t *st, const uint8_t *sq,
                                          int r, int L, int (&H)[CPL], int (&En)[CPL], int (&E2n)[CPL],
                                          int (&Fn)[CPL], int (&F2n)[CPL], int &Sc, uint8_t *prow)
{
	const int t0 = (r + kmin + 1) >> 1;
	int sE[CPL], sE2[CPL], sF[CPL], sF2[CPL];
	if (!ODD) {
#pragma unroll
		for (int k = 0; k < CPL; ++k) {
			sE[k] = __builtin_amdgcn_update_dpp(FILL_NEG, En[k], 0x138, 0xf, 0xf, false);      // wave_shr:1
			sE2[k] = __builtin_amdgcn_update_dpp(FILL_NEG, E2n[k], 0x138, 0xf, 0xf, false);
			if (k > 0) {
				const int pe = __builtin_amdgcn_readlane(En[k > 0 ? k - 1 : 0], 63), pe2 = __builtin_amdgcn_readlane(E2n[k > 0 ? k - 1 : 0], 63);
				if (L == 0) sE[k] = pe, sE2[k] = pe2;
			} else if (LANES == 32 && L == 0) sE[k] = sE2[k] = FILL_NEG;                      // lane 32 got the other segment's
			sF[k] = Fn[k], sF2[k] = F2n[k];
		}
	} else {
#pragma unroll
		for (int k = 0; k < CPL; ++k) {
			sF[k] = __builtin_amdgcn_update_dpp(FILL_NEG, Fn[k], 0x130, 0xf, 0xf, false);      // wave_shl:1
			sF2[k] = __builtin_amdgcn_update_dpp(FILL_NEG, F2n[k], 0x130, 0xf, 0xf, false);
			if (k < CPL - 1) {
				const int nf = __builtin_amdgcn_readlane(Fn[k + 1 < CPL ? k + 1 : k], 0), nf2 = __builtin_amdgcn_readlane(F2n[k + 1 < CPL ? k + 1 : k], 0);
				if (L == LANES - 1) sF[k] = nf, sF2[k] = nf2;
			} else if (LANES == 32 && L == LANES - 1) sF[k] = sF2[k] = FILL_NEG;
			sE[k] = En[k], sE2[k] = E2n[k];
		}
	}
#pragma unroll
	for (int k = 0; k < CPL; ++k) {
		// Every lane computes on every step: a cell outside the matrix holds garbage that no cell
		// inside ever reads (its upper / left / diagonal neighbours are inside too, or the virtual
		// row / column below replaces them), and the band has no inside edge to guard (fill_band).
		const int t = t0 + L + LANES * k, j = r - t;
		const bool act = r < rows && (unsigned)t < (unsigned)n && (unsigned)j < (unsigned)m;
		const int tc = t < 0 ? 0 : t > MAXLEN ? MAXLEN : t, jc = j < 0 ? 0 : j > MAXLEN ? MAXLEN : j;
		const int ct = st[tc], cq = sq[jc];
		const int sc = (ct == 4 || cq == 4) ? scN : ct == cq ? a : bmis;
		int hd = H[k], vE = sE[k], vE2 = sE2[k], vF = sF[k], vF2 = sF2[k];
		if (t == 0 || j == 0) {                                  // virtual row / column: gaps from the corner
			hd = t == 0 && j == 0 ? 0 : -fill_gap(t == 0 ? j : t, q, e, q2, e2);
			if (t == 0) { const int hb = -fill_gap(j + 1, q, e, q2, e2); vE = hb - q - e, vE2 = hb - q2 - e2; }
			if (j == 0) { const int hb = -fill_gap(t + 1, q, e, q2, e2); vF = hb - q - e, vF2 = hb - q2 - e2; }
		}
		int z = hd + sc, d;
		d = vE > z ? 1 : 0;  z = z > vE ? z : vE;
		d = vF > z ? 2 : d;  z = z > vF ? z : vF;
		d = vE2 > z ? 3 : d; z = z > vE2 ? z : vE2;
		d = vF2 > z ? 4 : d; z = z > vF2 ? z : vF2;
		const int o1 = z - q, o2 = z - q2;
		d |= vE > o1 ? 0x08 : 0;  En[k] = (vE > o1 ? vE : o1) - e;
		d |= vF > o1 ? 0x10 : 0;  Fn[k] = (vF > o1 ? vF : o1) - e;
		d |= vE2 > o2 ? 0x20 : 0; E2n[k] = (vE2 > o2 ? vE2 : o2) - e2;
		d |= vF2 > o2 ? 0x40 : 0; F2n[k] = (vF2 > o2 ? vF2 : o2) - e2;
		if (act) {
			prow[LANES * k] = (uint8_t)d;
			if (r == rows - 1) Sc = z;                          // the corner: the global score
		}
		H[k] = z;
	}
}

// The int32 form of the banded kernel (one cell per lane and register, absolute 32-bit scores, the
// ambiguity score included): what the packed kernel below cannot hold -- gaps between seeds of more
// than 511 bases, whose scores outgrow 12 bits.  64 lanes x CPL cells (a band of 128 or 256 cells),
// sequences up to MAXLEN.  Proof failures go to `next_list` (the wider tier, or the literal kernel),
// CIGARs whose walk shows a large score drop to `fb_list` (the literal kernel runs minimap2's exact
// second pass).
template <int LANES, int CPL, int MAXLEN, int CIGMAX>
__global__ __launch_bounds__(64) void mnc_dp_fill(Batch B, const int32_t *list, int ctr_n, int ctr_q, int32_t *next_list, int ctr_next,
                                                  int32_t *fb_list, int ctr_fb, uint8_t *p_all)
{
	constexpr int SEGS = 64 / LANES, W = LANES * CPL, ROWB = 64 * CPL;   // cells per segment and step; bytes per step of the wave
	__shared__ uint8_t s_t[SEGS][MAXLEN + 1], s_q[SEGS][MAXLEN + 1];
	__shared__ __align__(16) uint8_t s_win[SEGS][FILL_WIN * W];
	__shared__ uint32_t s_cg[SEGS][CIGMAX];                  // the gap filling's CIGAR (beyond: handed on)
	const int lane = threadIdx.x, sg = lane / LANES, L = lane % LANES, lead = sg * LANES;
	const bool leader = L == 0;
	const int a = B.sc_a, bmis = -B.sc_b, scN = -B.sc_ambi, q = B.gap_q, e = B.gap_e, q2 = B.gap_q2, e2 = B.gap_e2;
	uint8_t *p_wave = p_all + (size_t)blockIdx.x * ((size_t)(2 * MAXLEN + FILL_WIN) * 64 * CPL);
	const unsigned long long n_items = B.dp_ctr[ctr_n];
	// a long call is one wave's serial work for milliseconds, beside chip-filling kernels that have four waves on
	// every SIMD: with equal priority it would get a fifth of the issue slots and five times the latency
	if (MAXLEN > FILL_MAX_LEN) __builtin_amdgcn_s_setprio(3);
	for (;;) {
		unsigned long long q0 = 0;
		if (lane == 0) q0 = atomicAdd(&B.dp_ctr[ctr_q], (unsigned long long)SEGS);
		q0 = (unsigned long long)__shfl((long long)q0, 0);
		if (q0 >= n_items) break;                              // every wave gets here: the queue is finite
		const bool has = q0 + sg < n_items;
		const long long si = has ? (long long)list[q0 + sg] : -1;
		struct { int32_t tlen, qlen, ts, qs, read, rid, rev; } g = { 0, 0, 0, 0, 0, 0, 0 };
		if (has) {
			const Seg *gs = B.segs + si;
			g.tlen = gs->tlen, g.qlen = gs->qlen, g.ts = gs->ts, g.qs = gs->qs, g.read = gs->read, g.rid = gs->rid, g.rev = gs->rev;
		}
		const int n = g.tlen, m = g.qlen;
		int b, kmin, kmax;
		fill_band(n, m, W, b, kmin, kmax);
		bool ok = has && n >= 1 && m >= 1 && n <= MAXLEN && m <= MAXLEN && b >= FILL_MIN_BAND;
		// ---- sequences
		if (ok) {
			const uint8_t *read = B.bases + B.offsets[g.read];
			const int rlen = (int)(B.offsets[g.read + 1] - B.offsets[g.read]);
			const int64_t coff = B.seq_off[g.rid] + g.ts;
			for (int i = L; i < n; i += LANES) {
				const int64_t o = coff + i;
				s_t[sg][i] = (uint8_t)(B.seq4[o >> 3] >> ((o & 7) * 4) & 15u);
			}
			const bool acgt = !B.ambig[g.read];
			const int64_t roff = B.offsets[g.read];
			for (int i = L; i < m; i += LANES) s_q[sg][i] = (uint8_t)fill_qcode(B, acgt, roff, read, rlen, g.rev, g.qs + i);
		}
		fill_order();
		const int rows = ok ? n + m - 1 : 0;
		int max_rows = rows;
		if (SEGS == 2) { const int o = __shfl_xor(max_rows, 32); max_rows = max_rows > o ? max_rows : o; }
		// ---- forward: one anti-diagonal per step; cell c = L + LANES * k of the band holds t = t0 + c
		int H1[CPL], H2[CPL], En[CPL], E2n[CPL], Fn[CPL], F2n[CPL], Sc = FILL_NEG;
#pragma unroll
		for (int k = 0; k < CPL; ++k) H1[k] = H2[k] = En[k] = E2n[k] = Fn[k] = F2n[k] = FILL_NEG;
		uint8_t *prow = p_wave + lead * CPL + L;
		// two steps per iteration: an even one (the upper neighbour is the previous cell) and an odd one
		// (the left neighbour is the next cell); H of two steps back is the register of the same parity
		const uint8_t *st = &s_t[sg][0], *sq = &s_q[sg][0];
		int Heven[CPL], Hodd[CPL];
#pragma unroll
		for (int k = 0; k < CPL; ++k) Heven[k] = Hodd[k] = FILL_NEG;
		(void)H1; (void)H2;
		int r = 0;
		for (; r + 1 < max_rows; r += 2, prow += 2 * ROWB) {
			fill_step<LANES, CPL, 0, MAXLEN>(n, m, rows, kmin, a, bmis, scN, q, e, q2, e2, st, sq, r, L, Heven, En, E2n, Fn, F2n, Sc, prow);
			fill_step<LANES, CPL, 1, MAXLEN>(n, m, rows, kmin, a, bmis, scN, q, e, q2, e2, st, sq, r + 1, L, Hodd, En, E2n, Fn, F2n, Sc, prow + ROWB);
		}
		if (r < max_rows) fill_step<LANES, CPL, 0, MAXLEN>(n, m, rows, kmin, a, bmis, scN, q, e, q2, e2, st, sq, r, L, Heven, En, E2n, Fn, F2n, Sc, prow);
		// ---- the proof: every path that leaves the band scores at most U
		int S = FILL_NEG;
		{
			int lc = ok ? n - 1 - ((rows - 1 + kmin + 1) >> 1) : 0;
			lc = lc < 0 ? 0 : lc >= W ? W - 1 : lc;
			S = __shfl(Sc, lead + lc % LANES);
			const int U = dp_band_bound(n, m, kmin, kmax, a, q, e, q2, e2);
			if (ok && !(S > U)) ok = false;
		}
		bool to_next = has && !ok, to_fb = false;
		fill_order_mem();                                      // the direction bytes are in memory before the walk reads them
		// ---- backtrack through a window of FILL_WIN rows held in LDS
		int bi = n - 1, bj = m - 1, state = 0, n_c = 0, wlo = rows;
		uint32_t cur = 0;
		bool walking = ok;
		for (;;) {
			const int seg_r = __shfl(walking ? bi + bj : -1, lead), seg_wlo = __shfl(wlo, lead);
			if (!__any(seg_r >= 0)) break;
			if (seg_r >= 0 && seg_r < seg_wlo) {                   // refill: rows [lo, lo + FILL_WIN), 16 bytes per lane and cell
				const int lo = seg_r - (FILL_WIN - 1) > 0 ? seg_r - (FILL_WIN - 1) : 0;
#pragma unroll
				for (int k = 0; k < CPL; ++k) {
					const int byte0 = (L + LANES * k) * 16, row = lo + byte0 / W, col = byte0 % W;
					const uint4 v = *reinterpret_cast<const uint4*>(p_wave + (size_t)row * ROWB + lead * CPL + col);
					*reinterpret_cast<uint4*>(&s_win[sg][byte0]) = v;
				}
				wlo = lo;
			}
			fill_order();
			if (leader && walking) {
				while (bi >= 0 && bj >= 0) {
					const int r = bi + bj;
					if (r < wlo) break;
					const int idx = bi - ((r + kmin + 1) >> 1);
					if (idx < 0 || idx >= W) { walking = false, to_next = true; break; }   // cannot happen after the proof
					const uint32_t tmp = s_win[sg][(r - wlo) * W + idx];
					if (state == 0) state = tmp & 7;
					else if (!(tmp >> (state + 2) & 1)) state = 0;
					if (state == 0) state = tmp & 7;
					uint32_t op;
					if (state == 0) op = 0, --bi, --bj;
					else if (state == 1 || state == 3) op = 2, --bi;
					else op = 1, --bj;
					if (cur != 0 && (cur & 0xf) == op) cur += 1u << 4;
					else {
						if (cur != 0) {
							if (n_c >= CIGMAX - 4) { walking = false, to_fb = true; break; }   // more operations than the scratch holds
							s_cg[sg][n_c++] = cur;
						}
						cur = 1u << 4 | op;
					}
				}
				if (walking && (bi < 0 || bj < 0)) {
					if (bi >= 0) { if (cur != 0 && (cur & 0xf) == 2) cur += (uint32_t)(bi + 1) << 4; else { if (cur != 0) s_cg[sg][n_c++] = cur; cur = (uint32_t)(bi + 1) << 4 | 2; } }
					if (bj >= 0) { if (cur != 0 && (cur & 0xf) == 1) cur += (uint32_t)(bj + 1) << 4; else { if (cur != 0) s_cg[sg][n_c++] = cur; cur = (uint32_t)(bj + 1) << 4 | 1; } }
					if (cur != 0) s_cg[sg][n_c++] = cur;
					walking = false;
				}
			}
			fill_order();
		}
		// ---- mm_test_zdrop: a walk over the CIGAR (stored last operation first).  The drop it looks for, over any
		// stretch of the walk, is (gap costs q + e len) - a (M columns) of the stretch + what its columns that do not
		// match cost against matches (a + b per mismatch, a + 1 per ambiguous base).  The first part is at most the
		// largest sum over contiguous operations of (gap: + q + e len, M run: - a len) -- one pass, no sequences --
		// and the second at most its total over the whole walk, which follows from the score:
		// a (M columns) - S - (two-piece gap costs).  Below the threshold the answer is 0 without touching the bases.
		bool walk = leader && ok && !to_next && !to_fb;
		if (walk) {
			int mcols = 0, g2 = 0, kd = 0, kbest = 0;
			for (int k = 0; k < n_c; ++k) {
				const uint32_t op = s_cg[sg][k] & 0xf;
				const int len = (int)(s_cg[sg][k] >> 4);
				if (op == 0) mcols += len, kd = kd - a * len > 0 ? kd - a * len : 0;
				else g2 += fill_gap(len, q, e, q2, e2), kd += q + e * len, kbest = kbest > kd ? kbest : kd;
			}
			const int lost = a * mcols - S - g2;                   // what the columns that do not match cost against matches
			const int neg = kbest + lost;
			if (neg <= (B.zdrop_inv < B.zdrop ? B.zdrop_inv : B.zdrop)) walk = false;
		}
		if (walk) {
			int score = 0, mx = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
			for (int k = n_c - 1; k >= 0; --k) {
				const uint32_t op = s_cg[sg][k] & 0xf;
				const int len = (int)(s_cg[sg][k] >> 4);
				if (op == 0) {
					for (int l = 0; l < len; ++l) {
						const int ct = s_t[sg][i + l], cq = s_q[sg][j + l];
						score += (ct == 4 || cq == 4) ? scN : ct == cq ? a : bmis;
						if (score < mx) {
							const int li = i + l - max_i, lj = j + l - max_j;
							const int z = mx - score - (li > lj ? li - lj : lj - li) * e;
							max_zdrop = max_zdrop > z ? max_zdrop : z;
						} else mx = score, max_i = i + l, max_j = j + l;
					}
					i += len, j += len;
				} else {
					score -= q + e * len;
					if (op == 1) j += len; else i += len;
					if (score < mx) {
						const int li = i - max_i, lj = j - max_j;
						const int z = mx - score - (li > lj ? li - lj : lj - li) * e;
						max_zdrop = max_zdrop > z ? max_zdrop : z;
					} else mx = score, max_i = i, max_j = j;
				}
			}
			// above the smaller threshold the answer may be 1 or 2: the literal kernel decides and reruns
			if (max_zdrop > (B.zdrop_inv < B.zdrop ? B.zdrop_inv : B.zdrop)) to_fb = true;
		}
		const bool s_next = __shfl((int)to_next, lead) != 0, s_fb = __shfl((int)to_fb, lead) != 0;
		const int s_nc = __shfl(n_c, lead);
		if (has && (s_next || s_fb)) {
			if (leader) {
				const bool literal = s_fb || next_list == fb_list;      // the last tier hands everything to the literal kernel
				if (literal && (B.segs[si].flag & SEG_NEEDS_BIG_WS)) { const unsigned long long k = atomicAdd(&B.dp_ctr[56], 1ULL); B.bigfb_list[k] = (int32_t)si; }
				else if (literal) { const unsigned long long k = atomicAdd(&B.dp_ctr[ctr_fb], 1ULL); fb_list[k] = (int32_t)si; }
				else { const unsigned long long k = atomicAdd(&B.dp_ctr[ctr_next], 1ULL); next_list[k] = (int32_t)si; }
			}
		} else if (has) {
			unsigned long long off = 0;
			if (leader) off = atomicAdd(&B.dp_ctr[1], (unsigned long long)s_nc);
			off = (unsigned long long)__shfl((long long)off, lead);
			int wrote = s_nc;
			if ((long long)(off + s_nc) > B.cig_seg_cap) {
				if (leader) atomicMax(&B.dp_ctr[4], 2ULL);
				wrote = 0;
			} else for (int k = L; k < s_nc; k += LANES) B.cig_seg[off + k] = s_cg[sg][s_nc - 1 - k];
			if (leader) {
				Seg *o = B.segs + si;
				o->n_cigar = wrote, o->zdropped = 0, o->zdrop_code = 0;
				o->max = 0, o->max_t = -1, o->max_q = -1, o->score = S, o->reach_end = 0, o->mqe_t = -1;
				o->cig_off = (int64_t)off;
			}
		}
		fill_order_mem();
	}
}

// ================================================================ the same fill, two cells per lane and instruction
// Scores of a gap filling span less than 2^12 (at most 511 matches of 2; at least -4 per base and one
// long gap), so a cell fits 16 bits with four spare: the kernel keeps ((score + bias) << 4 | tag) in the
// halves of a register and runs the recurrence on VOP3P pairs (v_pk_add/sub/max_i16).  The tag makes
// the maximum itself pick ksw2's winner among equal scores -- H 15, E 7, F 3, E2 1, F2 0: the first of
// ksw2's comparison chain wins a tie -- and likewise "a gap is opened rather than extended on a tie":
// the opening candidate carries H's tag, and the bit that tells it from the extending one is bit 3, 2,
// 1, 0 for E, F, E2, F2, so the four flags of the direction byte are one bit-field insert each.
// Cell c = 2 L + h of the band: lane L, half h; the neighbour cells' gap states are a DPP move of the
// neighbouring lane and one v_alignbit.  The virtual row and column only touch the band during its
// first max(-kmin, kmax) steps: in the plain frame those run a variant of the step with the overrides, the
// rest without; in the drifting frame (minimap2's scores) row 0 and column 0 of the matrix ARE the virtual
// row and column -- a base in front of both sequences -- and every step is the bare one (mnc_dp_fillp, V).
// A segment with an ambiguous base goes to the literal kernel.
typedef short pk_s16 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_adds(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(pk_s16, x), __builtin_bit_cast(pk_s16, y))); }
__device__ __forceinline__ uint32_t pk_subs(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(pk_s16, x), __builtin_bit_cast(pk_s16, y))); }
__device__ __forceinline__ uint32_t pk_maxs(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(pk_s16, x), __builtin_bit_cast(pk_s16, y))); }
__device__ __forceinline__ uint32_t pk_subsu(uint32_t x, uint32_t y) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(pk_u16, x), __builtin_bit_cast(pk_u16, y))); }
__device__ __forceinline__ uint32_t pk_madu(uint32_t x, uint32_t y, uint32_t z) { return __builtin_bit_cast(uint32_t, (pk_u16)(__builtin_bit_cast(pk_u16, x) * __builtin_bit_cast(pk_u16, y) + __builtin_bit_cast(pk_u16, z))); }
__device__ __forceinline__ uint32_t pk_rep(int v) { return ((uint32_t)v & 0xffffu) * 0x10001u; }
__device__ __forceinline__ uint32_t pk_bfi(uint32_t mask, uint32_t x, uint32_t y) { return (x & mask) | (y & ~mask); }

constexpr uint32_t PK_NEG = 0x80008000u;
// x = tag(E) ^ tag(F) ^ tag(E2) ^ tag(F2) with tags 15 (opened) or 7 / 3 / 1 / 0: bit 0 = 1 ^ d, bit 1 = c ^ d,
// bit 2 = 1 ^ b ^ c ^ d, bit 3 = a ^ b ^ c ^ d for a, b, c, d = "E, F, E2, F2 opened".  Returns a << 3 | b << 2 | c << 1 | d.
__device__ __forceinline__ uint32_t fillp_opened(uint32_t x)
{
	const uint32_t d = (x & 1u) ^ 1u, c = (x >> 1 & 1u) ^ d, b = (x >> 2 & 1u) ^ 1u ^ c ^ d, a = (x >> 3 & 1u) ^ b ^ c ^ d;
	return a << 3 | b << 2 | c << 1 | d;
}
constexpr uint32_t PK_TAG_H = 0x000f000fu, PK_TAG_E = 0x00070007u, PK_TAG_F = 0x00030003u, PK_TAG_E2 = 0x00010001u;   // F2: 0
struct PkConst { uint32_t kmatch, kmis, q8, q28, e8, e28, de28, dstep; int bias; };

// the span of the scores a band of `cells` can hold, and the bias that centres it in 12 bits
__host__ __device__ __forceinline__ bool fillp_bias(int cells, int a, int bmis_abs, int q, int e, int q2, int e2, int &bias)
{
	const int hi = a * FILL_MAX_LEN, lo = -(bmis_abs * FILL_MAX_LEN + fill_gap(2 * cells + 2, q, e, q2, e2) + (q + e > q2 + e2 ? q + e : q2 + e2));
	bias = -(hi + lo) / 2;
	return hi + bias < 2040 && lo + bias > -2040;
}
// The drifting frame (fillp_step<DRIFT>): every value of anti-diagonal r is kept as (value + e r).  A diagonal move spans two
// anti-diagonals, so it adds a + 2 e for a match and b + 2 e for a mismatch -- nothing at all when the mismatch penalty is
// 2 e (minimap2's map-ont scores: b = 4, e = 2) --, extending a gap of the first piece costs e - e = 0 and one of the
// second piece e2 - e: three instructions of a step disappear (the mismatch term of the score, the two subtractions of e).
// Comparisons within an anti-diagonal are not affected (all candidates carry the same drift); the true score of the last
// cell is what is kept less e r.  Span: a cell (i, j) holds at most a min(i, j) + a + e (i + j) <= (a + 2 e) FILL_MAX_LEN,
// and at least what the straight path to it gives, -(b - 2 e) min(i, j) - gap(|i - j|) + e |i - j| >= -(q2 + e2) - ...: the
// bias below centres [-(q + e + q2 + e2) - 64, (a + 2 e) FILL_MAX_LEN]; anything lower clamps as before.
__host__ __device__ __forceinline__ bool fillp_drifts(int bmis_abs, int e, int e2) { return bmis_abs == 2 * e && e >= e2; }
__host__ __device__ __forceinline__ bool fillp_bias_drift(int a, int q, int e, int q2, int e2, int &bias)
{
	const int hi = (a + 2 * e) * FILL_MAX_LEN, lo = -(q + e + q2 + e2 + 64);
	bias = -(hi + lo) / 2;
	return hi + bias < 2040 && lo + bias > -2040;
}

// one base of a sequence word into a half of the pair register, straight from the LDS word (k = 0 .. 3: the byte)
template <int K> __device__ __forceinline__ uint32_t fillp_next_q(uint32_t Q2, uint32_t w) { return __builtin_amdgcn_perm(Q2, w, 0x05040c00u + K); }      // old low half -> high, byte K -> low
template <int K> __device__ __forceinline__ uint32_t fillp_next_t(uint32_t T2, uint32_t w) { return __builtin_amdgcn_perm(w, T2, 0x0c040302u + K * 0x10000u); }   // old high half -> low, byte K -> high

#ifndef MNC_FILLP_FRAME2
#define MNC_FILLP_FRAME2 1
#endif
template <int LANES, int ODD, bool EDGE, bool CAPTURE = true, bool DRIFT = false, int KB = -1>
__device__ __forceinline__ void fillp_step(const PkConst &K, const int r, const int L, const int rows_m1, const uint8_t *st, const uint8_t *sq,
                                           const int q, const int e, const int q2, const int e2,
                                           int &t_lo, int &tn, int &jn, uint32_t &T2, uint32_t &Q2, uint32_t &Hs,
                                           uint32_t &E, uint32_t &E2, uint32_t &F, uint32_t &F2, uint32_t &Sc,
                                           uint32_t &nb1, uint32_t &nb2, uint32_t &acc, const uint32_t wq = 0, const uint32_t wt = 0)
{
	constexpr int SHR = LANES == 16 ? 0x111 : 0x138, SHL = LANES == 16 ? 0x101 : 0x130;   // row_shr:1 / wave_shr:1
	constexpr bool SEG_EDGES = LANES != 16 && LANES != 64;      // segments inside the reach of a wave shift: their edge lanes get the neighbouring segment's
	uint32_t vE, vE2, vF, vF2;
	// nb1 / nb2 receive the neighbouring lane's register; the lane at the band's edge has no source and
	// keeps what they held: PK_NEG, from before the loop
	if (!ODD) {
		nb1 = (uint32_t)__builtin_amdgcn_update_dpp((int)nb1, (int)E, SHR, 0xf, 0xf, false);
		nb2 = (uint32_t)__builtin_amdgcn_update_dpp((int)nb2, (int)E2, SHR, 0xf, 0xf, false);
		if (SEG_EDGES && L == 0) nb1 = nb2 = PK_NEG;                 // it got the other segment's
		vE = __builtin_amdgcn_alignbit(E, nb1, 16), vE2 = __builtin_amdgcn_alignbit(E2, nb2, 16);
		vF = F, vF2 = F2;
		// (old low half -> high, the new base -> low; KB >= 0: the base is byte KB & 3 of a word the caller loaded -- no extraction)
		if constexpr (KB >= 0) Q2 = fillp_next_q<KB & 3>(Q2, wq);
		else Q2 = __builtin_amdgcn_perm(Q2, (uint32_t)sq[jn], 0x05040100u), ++jn;
	} else {
		nb1 = (uint32_t)__builtin_amdgcn_update_dpp((int)nb1, (int)F, SHL, 0xf, 0xf, false);
		nb2 = (uint32_t)__builtin_amdgcn_update_dpp((int)nb2, (int)F2, SHL, 0xf, 0xf, false);
		if (SEG_EDGES && L == LANES - 1) nb1 = nb2 = PK_NEG;
		vF = __builtin_amdgcn_alignbit(nb1, F, 16), vF2 = __builtin_amdgcn_alignbit(nb2, F2, 16);
		vE = E, vE2 = E2;
		if constexpr (KB >= 0) T2 = fillp_next_t<KB & 3>(T2, wt);   // (old high half -> low, the new base -> high)
		else T2 = __builtin_amdgcn_perm((uint32_t)st[tn], T2, 0x05040302u), ++tn;
		if (EDGE) ++t_lo;
	}
	uint32_t hd = Hs;
	if (EDGE) {
		// the virtual row / column: gaps from the corner.  Cell (0, j) sits on step r = j, cell (t, 0) on r = t:
		// the same three numbers for both, by the step alone.  (DRIFT: the diagonal source belongs to anti-diagonal
		// r - 2, the gap states to r)
		const int hb = -fill_gap(r + 1, q, e, q2, e2) + K.bias + (DRIFT ? e * r : 0);
		const uint32_t h0 = pk_rep(((r == 0 ? 0 : -fill_gap(r, q, e, q2, e2)) + K.bias + (DRIFT ? e * (r - 2) : 0)) << 4) | PK_TAG_H;
		const uint32_t g1 = pk_rep((hb - q - e) << 4), g2 = pk_rep((hb - q2 - e2) << 4);
		const uint32_t mt = t_lo == 0 ? 0x0000ffffu : t_lo == -1 ? 0xffff0000u : 0u;
		const uint32_t mj = t_lo == r ? 0x0000ffffu : t_lo == r - 1 ? 0xffff0000u : 0u;
		hd = pk_bfi(mt | mj, h0, hd);
		vE = pk_bfi(mt, g1 | PK_TAG_E, vE), vE2 = pk_bfi(mt, g2 | PK_TAG_E2, vE2);
		vF = pk_bfi(mj, g1 | PK_TAG_F, vF), vF2 = pk_bfi(mj, g2, vF2);
	}
	uint32_t z;
	if (DRIFT) z = pk_madu(pk_subsu(0x00010001u, T2 ^ Q2), K.kmatch, hd);        // + (a + 2 e) where the bases are equal, nothing where not
	else z = pk_adds(hd, pk_madu(pk_subsu(0x00010001u, T2 ^ Q2), K.kmatch, K.kmis));   // 1 - min(1, x): the bases are equal
	uint32_t o2sub = K.q28;
	constexpr bool FRAME2 = DRIFT && MNC_FILLP_FRAME2 && LANES <= 32;   // (not the 128-cell tier: the margin below is a few points there)
	if (FRAME2) {
		// the second gap piece drifts by e2 a step, not by e: extending it costs nothing either.  Its two states are compared with
		// each other as they are and brought to H's frame by one add of (e - e2) r; the opening candidate goes the other way,
		// (e - e2) r off together with q2.  Values of cells no good path visits clamp earlier in this frame (they lie lower by
		// (e - e2) r), and a state nobody has written reads as the floor + (e - e2) r in H's frame: both stay below every cell of
		// a path that passes the band proof as long as S > a (n + m) / 2 + floor, which the proof's S > U gives with a margin of
		// 255 points at 64 cells ("the second piece's frame")
		const uint32_t Dr = (uint32_t)r * K.dstep;                 // ((e - e2) r) << 4 in both halves (below 2^16: no carry between them)
		z = pk_maxs(z, vE), z = pk_maxs(z, vF), z = pk_maxs(z, pk_adds(pk_maxs(vE2, vF2), Dr));
		o2sub = K.q28 + Dr;
	} else z = pk_maxs(z, vE), z = pk_maxs(z, vF), z = pk_maxs(z, vE2), z = pk_maxs(z, vF2);
	const uint32_t zt = z | PK_TAG_H;
	const uint32_t o1 = pk_subs(zt, K.q8), o2 = pk_subs(zt, o2sub);
	const uint32_t mE = pk_maxs(vE, o1), mF = pk_maxs(vF, o1), mE2 = pk_maxs(vE2, o2), mF2 = pk_maxs(vF2, o2);
	if (FRAME2) {
		E = mE & 0xfff7fff7u, F = mF & 0xfff3fff3u, E2 = mE2 & 0xfff1fff1u, F2 = mF2 & 0xfff0fff0u;
	} else if (DRIFT) {
		E = mE & 0xfff7fff7u, F = mF & 0xfff3fff3u;                 // (extending costs e - e)
		E2 = pk_adds(mE2, K.de28) & 0xfff1fff1u, F2 = pk_adds(mF2, K.de28) & 0xfff0fff0u;   // (e2 - e)
	} else {
		E = pk_subs(mE, K.e8) & 0xfff7fff7u;                       // tag 15 or 7 -> 7
		F = pk_subs(mF, K.e8) & 0xfff3fff3u;                       // 15 or 3 -> 3
		E2 = pk_subs(mE2, K.e28) & 0xfff1fff1u;
		F2 = pk_subs(mF2, K.e28) & 0xfff0fff0u;
	}
	// direction byte: bits 0-3 the winner's tag; bits 4-7 the XOR of the four gap states' tags (15 where a gap
	// was opened, else 7 / 3 / 1 / 0 for E / F / E2 / F2), from which the walk solves the four "opened" bits:
	// three cheap instructions instead of four field inserts (fillp_opened) -- the XOR of three of them is one v_bitop3
	const uint32_t x3 = (uint32_t)__builtin_amdgcn_bitop3_b32((int)mE, (int)mF, (int)mE2, 0x96);
	const uint32_t d = pk_bfi(0x000f000fu, z, (x3 ^ mF2) << 4);
	// two steps' direction bytes per register: (even step: cells 2 L, 2 L + 1; odd step: likewise)
	// (the even step leaves its word as it is; the odd one picks the four bytes out of both -- steps come in pairs)
	acc = ODD ? __builtin_amdgcn_perm(d, acc, 0x06040200u) : d;
	if (CAPTURE) Sc = r == rows_m1 ? zt : Sc;                  // the corner's score: only the blocks that hold a segment's last step look for it
	Hs = zt;
}

// A wave takes FILLP_G x SEGS segments at a time: FILLP_G forward passes (SEGS segments side by side, as
// above), their direction bytes kept in the wave's slot of HBM -- per 16 steps a lane stores the 32
// bytes of its two cells, so a walk that stays near one diagonal reads a line many times -- and then
// ONE walk phase in which every lane backtracks a segment of its own: a walk is a few dozen
// instructions per CIGAR column whatever the number of active lanes, and with 4 of 64 it used to cost
// as much issue time as the fill itself.
// LANES need not divide 64: with 21 lanes a segment (a band of 42 cells) a wave holds three, lane 63 idles.
constexpr int FILLP_G = 16, FILLP_G_MAX = 21;
template <int LANES> struct FillpShape {
	static constexpr int SEGS = 64 / LANES;
	static constexpr int G = SEGS == 3 ? 21 : FILLP_G;                    // passes per group: 3 x 21 = 63 walks in one walk phase
};
constexpr int FILLP_BLOCKS = (2 * FILL_MAX_LEN + 15) / 16 + 1;           // 16-step blocks of direction bytes per pass
constexpr size_t FILLP_PASS_BYTES = (size_t)FILLP_BLOCKS * 64 * 32;
constexpr size_t FILLP_SLOT = FILLP_G_MAX * FILLP_PASS_BYTES;

#ifndef MNC_FILLP_WAVES
#define MNC_FILLP_WAVES 4      // at most 128 registers a lane: four waves a SIMD (the drifting frame took 164 left alone: three)
#endif
template <int LANES, bool DRIFT>
__global__ __launch_bounds__(64, MNC_FILLP_WAVES) void mnc_dp_fillp(Batch B, const int32_t *list, int ctr_n, int ctr_q, int32_t *next_list, int ctr_next,
                                                   int32_t *fb_list, int ctr_fb, uint8_t *p_all, uint32_t *cig_all)
{
	constexpr int SEGS = FillpShape<LANES>::SEGS, G_MAX = FillpShape<LANES>::G, W = 2 * LANES;
	// PAD + 1 (the first real base of the drifting frame's sequences, see V below) is a multiple of eight: whole words go in
	constexpr int PAD = ((2 * W + 8) & ~7) - 1, SEQ = (FILL_MAX_LEN + 1 + 4 * W + 32 + 7) & ~7;
	static_assert(SEGS * G_MAX <= 64, "one walk per lane");
	// The drifting frame has no step with overrides.  Both sequences get a base in front that matches nothing (V = 1): row 0 and
	// column 0 of the matrix are then ksw2's virtual row and column, and the recurrence itself fills them -- H(0, 0) = 0 is given to the
	// corner cell as its diagonal source (the two bases in front differ: nothing is added), the gap states it opens run along
	// row 0 and column 0 as -(q + e l) and -(q2 + e2 l), and H there is their maximum: what the overrides wrote, by the same
	// comparisons and with the same tags.  A pass runs two steps longer; the steps with overrides cost 1 079 instead of 503
	// instructions per sixteen, three or four blocks of every pass (a tenth of the kernel).
	constexpr int V = DRIFT ? 1 : 0;
	__shared__ __align__(16) uint8_t s_t[SEGS][SEQ], s_q[SEGS][SEQ];   // bases at [PAD + i]; what lies around them feeds cells outside the matrix only
	__shared__ __align__(16) uint8_t s_chunk[64][32];         // the walk: the 32 direction bytes a lane is reading from
	__shared__ int32_t s_n[64], s_m[64], s_kmin[64], s_S[64], s_si[64], s_state[64];   // per segment of the group; state 0 none, 1 walk, 2 next tier, 3 literal kernel
	__shared__ int32_t s_item[64];                             // the group's segments, shortest first
	const int lane = threadIdx.x;
	const bool live = lane < SEGS * LANES;                     // lanes beyond the last whole segment compute along on segment 0's bases, unseen
	const int sg = live ? lane / LANES : 0, L = live ? lane % LANES : 0, lead = sg * LANES;
	const bool leader = live && L == 0;
	const int a = B.sc_a, bmis = -B.sc_b, q = B.gap_q, e = B.gap_e, q2 = B.gap_q2, e2 = B.gap_e2;
	PkConst K;
	K.kmatch = pk_rep((DRIFT ? a + 2 * e : a - bmis) << 4), K.kmis = pk_rep(bmis << 4), K.q8 = pk_rep(q << 4), K.q28 = pk_rep(q2 << 4), K.e8 = pk_rep(e << 4), K.e28 = pk_rep(e2 << 4);
	K.de28 = pk_rep((e - e2) << 4), K.dstep = K.de28;
	const bool fits = DRIFT ? fillp_bias_drift(a, q, e, q2, e2, K.bias) : fillp_bias(W, a, -bmis > B.sc_ambi ? -bmis : B.sc_ambi, q, e, q2, e2, K.bias);
	uint8_t *p_wave = p_all + (size_t)blockIdx.x * FILLP_SLOT;
	uint32_t *cg = cig_all + ((size_t)blockIdx.x * 64 + lane) * FILL_CIG_MAX;
	const unsigned long long n_items = B.dp_ctr[ctr_n];
	const unsigned long long segmask = LANES == 64 ? ~0ULL : ((1ULL << (LANES & 63)) - 1) << lead;
	const int thr = B.zdrop_inv < B.zdrop ? B.zdrop_inv : B.zdrop;
	// few segments (a micro-batch): fewer forward passes per group, so that every workgroup has some
	// ... and when a workgroup gets between one and two groups: two of equal size (20 passes' worth per workgroup as 16 + 4
	// leaves three quarters of the workgroups idle while the others run their second group -- as 10 + 10 nobody waits)
	int g_eff = (int)((n_items + (unsigned long long)gridDim.x * SEGS - 1) / ((unsigned long long)gridDim.x * SEGS));
	if (g_eff > G_MAX && g_eff < 2 * G_MAX) g_eff = (g_eff + 1) / 2;   // (with more rounds the queue evens the load out, and full groups keep every lane of the walk phase busy)
	g_eff = g_eff < 1 ? 1 : g_eff > G_MAX ? G_MAX : g_eff;
	for (;;) {
		unsigned long long q0 = 0;
		if (lane == 0) q0 = atomicAdd(&B.dp_ctr[ctr_q], (unsigned long long)(g_eff * SEGS));
		q0 = (unsigned long long)__shfl((long long)q0, 0);
		if (q0 >= n_items) break;                              // every wave gets here: the queue is finite
		s_state[lane] = 0;
		// the group's segments in the order of their lengths: the SEGS segments of a pass run as many steps as the
		// longest of them, and neighbours in this order differ by a few steps instead of by up to a hundred
		{
			int my_si = -1, my_rows = INT32_MAX;
			if (lane < g_eff * SEGS && q0 + lane < n_items) {
				my_si = list[q0 + lane];
				const Seg *gs = B.segs + my_si;
				my_rows = gs->tlen + gs->qlen;
			}
			int rank = 0;
			for (int j = 0; j < 64; ++j) {
				const int rj = __shfl(my_rows, j);
				rank += (rj < my_rows || (rj == my_rows && j < lane)) ? 1 : 0;
			}
			s_item[rank] = my_si;
		}
		fill_order();
		// ================================================ forward passes
		for (int u = 0; u < g_eff; ++u) {
			if (q0 + (unsigned long long)u * SEGS >= n_items) break;
			const long long si = live ? s_item[u * SEGS + sg] : -1;
			const bool has = si >= 0;
			struct { int32_t tlen, qlen, ts, qs, read, rid, rev; } g = { 0, 0, 0, 0, 0, 0, 0 };
			if (has) {
				const Seg *gs = B.segs + si;
				g.tlen = gs->tlen, g.qlen = gs->qlen, g.ts = gs->ts, g.qs = gs->qs, g.read = gs->read, g.rid = gs->rid, g.rev = gs->rev;
			}
			const int n = g.tlen, m = g.qlen;
			int b, kmin, kmax;
			fill_band(n, m, W, b, kmin, kmax);
			bool ok = has && fits && n >= 1 && m >= 1 && n <= FILL_MAX_LEN && m <= FILL_MAX_LEN && b >= FILL_MIN_BAND;
			bool ambiguous = false;
			if (ok) {
				const uint8_t *read = B.bases + B.offsets[g.read];
				const int rlen = (int)(B.offsets[g.read + 1] - B.offsets[g.read]);
				const int64_t coff = B.seq_off[g.rid] + g.ts;
				const bool acgt = !B.ambig[g.read];
				const int64_t roff = B.offsets[g.read];
				if (V) {
					// whole words: eight bases of the contig's 4-bit words, sixteen of the read's 2-bit words per lane and turn, spread to a
					// byte a base and stored eight bytes at a time (what is written past a sequence's end feeds cells outside the matrix)
					const uint32_t sh = (uint32_t)(coff & 7) * 4;
#pragma unroll 1
					for (int i = 8 * L; i < n; i += 8 * LANES) {
						const bool more = sh != 0 && i + 8 - (int)(coff & 7) < n;    // (the next word holds bases of the segment: never read past them)
						const uint32_t a0 = B.seq4[((coff + i) >> 3)], a1 = more ? B.seq4[((coff + i) >> 3) + 1] : 0u;
						const uint32_t v = sh ? __builtin_amdgcn_alignbit(a1, a0, sh) : a0;
						ambiguous |= (v & 0xccccccccu & (n - i < 8 ? (1u << (4 * (n - i))) - 1u : ~0u)) != 0;
						uint2 w;
						w.x = v & 0xffffu, w.x = (w.x | w.x << 8) & 0x00ff00ffu, w.x = (w.x | w.x << 4) & 0x03030303u;
						w.y = v >> 16, w.y = (w.y | w.y << 8) & 0x00ff00ffu, w.y = (w.y | w.y << 4) & 0x03030303u;
						*reinterpret_cast<uint2*>(&s_t[sg][PAD + 1 + i]) = w;
					}
					if (acgt) {
#pragma unroll 1
						for (int i = 16 * L; i < m; i += 16 * LANES) {
							const uint32_t f = fill_qcodes16(B, roff, rlen, g.rev, g.qs + i);
							uint32_t x[4];
#pragma unroll
							for (int k = 0; k < 4; ++k) {
								const uint32_t b8 = f >> (8 * k) & 0xffu;
								x[k] = (b8 | b8 << 12) & 0x000f000fu, x[k] = (x[k] | x[k] << 6) & 0x03030303u;
							}
							*reinterpret_cast<uint2*>(&s_q[sg][PAD + 1 + i]) = make_uint2(x[0], x[1]);
							*reinterpret_cast<uint2*>(&s_q[sg][PAD + 1 + i + 8]) = make_uint2(x[2], x[3]);
						}
					} else {
#pragma unroll 1
						for (int i = L; i < m; i += LANES) {
							const int c = fill_qcode(B, false, roff, read, rlen, g.rev, g.qs + i);
							ambiguous |= c > 3;
							s_q[sg][PAD + 1 + i] = (uint8_t)(c & 3);
						}
					}
					if (L == 0) s_t[sg][PAD] = 4, s_q[sg][PAD] = 5;
				} else {
#pragma unroll 1
					for (int i = L; i < n; i += LANES) {
						const int64_t o = coff + i;
						const uint32_t c = B.seq4[o >> 3] >> ((o & 7) * 4) & 15u;
						ambiguous |= c > 3;
						s_t[sg][PAD + i] = (uint8_t)(c & 3);
					}
#pragma unroll 1
					for (int i = L; i < m; i += LANES) {
						const int c = fill_qcode(B, acgt, roff, read, rlen, g.rev, g.qs + i);
						ambiguous |= c > 3;
						s_q[sg][PAD + i] = (uint8_t)(c & 3);
					}
				}
			}
			const bool seg_amb = (__ballot(ambiguous && live) & segmask) != 0;
			const bool to_fb = ok && seg_amb;                       // the literal kernel scores an ambiguous base
			ok = ok && !seg_amb;
			fill_order();
			const int rows = ok ? n + m - 1 + 2 * V : 0;
			int max_rows = (rows + 15) & ~15, edge_rows = ok && !V ? ((-kmin > kmax ? -kmin : kmax) + 2 + 15) & ~15 : 0;
			{
				// the pass runs as long as its longest segment: the maximum over the segments' first lanes
				int mr = 0, er = 0;
#pragma unroll
				for (int s2 = 0; s2 < SEGS; ++s2) {
					const int o = __builtin_amdgcn_readlane(max_rows, s2 * LANES), oe = __builtin_amdgcn_readlane(edge_rows, s2 * LANES);
					mr = mr > o ? mr : o, er = er > oe ? er : oe;
				}
				max_rows = mr, edge_rows = er;
			}
			// the first 16-step block that holds the last step of one of the pass's segments (they are sorted by length:
			// the few blocks from there on compare every step with the segment's last)
			int cap_from = ok ? (rows - 1) & ~15 : INT32_MAX;
			{
				int cf = INT32_MAX;
#pragma unroll
				for (int s2 = 0; s2 < SEGS; ++s2) { const int o = __builtin_amdgcn_readlane(cap_from, s2 * LANES); cf = cf < o ? cf : o; }
				cap_from = cf < max_rows ? cf : max_rows;
			}
			if (edge_rows > max_rows) edge_rows = max_rows;
			const int kmin_run = ok ? kmin : -2 * W;                // a segment that sits out: indices inside the arrays all the same
			// one anti-diagonal per step; cells 2 L, 2 L + 1 of the band hold t = t0 + 2 L (+ 1)
			const uint8_t *st = &s_t[sg][0], *sq = &s_q[sg][0];
			int t_lo = (kmin_run >> 1) + 2 * L;
			uint32_t T2 = (uint32_t)st[PAD + t_lo] | (uint32_t)st[PAD + t_lo + 1] << 16;
			uint32_t Q2 = (uint32_t)sq[PAD - t_lo - 1] | (uint32_t)sq[PAD - t_lo - 2] << 16;
			int tn = PAD + t_lo + 2, jn = PAD - t_lo;
			uint32_t He = PK_NEG | PK_TAG_H, Ho = He, E = PK_NEG | PK_TAG_E, F = PK_NEG | PK_TAG_F, E2 = PK_NEG | PK_TAG_E2, F2 = PK_NEG, Sc = PK_NEG;
			uint32_t nbe1 = PK_NEG, nbe2 = PK_NEG, nbf1 = PK_NEG, nbf2 = PK_NEG;
			if (V && ok) {
				// the corner: cell -kmin / 2 of step 0 (offset 0); its H comes out as what it is given here
				const int cc = -kmin >> 1;
				if (L == cc >> 1) He = pk_bfi(cc & 1 ? 0xffff0000u : 0x0000ffffu, pk_rep(K.bias << 4) | PK_TAG_H, He);
			}
			const int rows_m1 = rows - 1;
			uint4 *pblk = reinterpret_cast<uint4*>(p_wave + (size_t)u * FILLP_PASS_BYTES + lane * 32);
			int r = 0;
			for (; r < edge_rows; r += 16, pblk += 2048 / 16) {
				uint32_t acc[8];
#pragma unroll
				for (int k = 0; k < 8; ++k) {
					fillp_step<LANES, 0, true, true, DRIFT>(K, r + 2 * k, L, rows_m1, st, sq, q, e, q2, e2, t_lo, tn, jn, T2, Q2, He, E, E2, F, F2, Sc, nbe1, nbe2, acc[k]);
					fillp_step<LANES, 1, true, true, DRIFT>(K, r + 2 * k + 1, L, rows_m1, st, sq, q, e, q2, e2, t_lo, tn, jn, T2, Q2, Ho, E, E2, F, F2, Sc, nbf1, nbf2, acc[k]);
				}
				pblk[0] = make_uint4(acc[0], acc[1], acc[2], acc[3]), pblk[1] = make_uint4(acc[4], acc[5], acc[6], acc[7]);
			}
			// the bare steps: the eight query and the eight target bases of a block as two LDS words each, a base moved into
			// its pair register by the step's own v_perm (no extraction)
#define MNC_FILLP_PAIR(CAP, k) \
	fillp_step<LANES, 0, false, CAP, DRIFT, k>(K, r + 2 * k, L, rows_m1, st, sq, q, e, q2, e2, t_lo, tn, jn, T2, Q2, He, E, E2, F, F2, Sc, nbe1, nbe2, acc[k], k < 4 ? wq.x : wq.y, 0u); \
	fillp_step<LANES, 1, false, CAP, DRIFT, k>(K, r + 2 * k + 1, L, rows_m1, st, sq, q, e, q2, e2, t_lo, tn, jn, T2, Q2, Ho, E, E2, F, F2, Sc, nbf1, nbf2, acc[k], 0u, k < 4 ? wt.x : wt.y);
#define MNC_FILLP_BLOCK(CAP) { \
	uint32_t acc[8]; \
	uint2 wq, wt; \
	__builtin_memcpy(&wq, sq + jn, 8), __builtin_memcpy(&wt, st + tn, 8); \
	jn += 8, tn += 8; \
	MNC_FILLP_PAIR(CAP, 0) MNC_FILLP_PAIR(CAP, 1) MNC_FILLP_PAIR(CAP, 2) MNC_FILLP_PAIR(CAP, 3) \
	MNC_FILLP_PAIR(CAP, 4) MNC_FILLP_PAIR(CAP, 5) MNC_FILLP_PAIR(CAP, 6) MNC_FILLP_PAIR(CAP, 7) \
	pblk[0] = make_uint4(acc[0], acc[1], acc[2], acc[3]), pblk[1] = make_uint4(acc[4], acc[5], acc[6], acc[7]); }
			for (; r < cap_from; r += 16, pblk += 2048 / 16) MNC_FILLP_BLOCK(false)
			for (; r < max_rows; r += 16, pblk += 2048 / 16) MNC_FILLP_BLOCK(true)
#undef MNC_FILLP_BLOCK
#undef MNC_FILLP_PAIR
			// the proof: every path that leaves the band scores at most U
			int S = FILL_NEG;
			{
				int lc = ok ? n - 1 + V - ((rows - 1 + kmin + 1) >> 1) : 0;
				lc = lc < 0 ? 0 : lc >= W ? W - 1 : lc;
				const uint32_t v = (uint32_t)__shfl((int)Sc, lead + (lc >> 1));
				S = ((int)(int16_t)(lc & 1 ? v >> 16 : v & 0xffffu) >> 4) - K.bias - (DRIFT ? e * (rows - 1) : 0);
				const int U = dp_band_bound(n, m, kmin, kmax, a, q, e, q2, e2);
				if (ok && !(S > U)) ok = false;
			}
			if (leader && has) {
				const int w = u * SEGS + sg;
				s_n[w] = n + V, s_m[w] = m + V, s_kmin[w] = kmin, s_S[w] = S, s_si[w] = (int32_t)si;   // (the walk's matrix: with row and column 0)
				s_state[w] = ok ? 1 : to_fb ? 3 : 2;
			}
			fill_order();                                          // the next pass overwrites the sequences
		}
		fill_order_mem();                                      // the direction bytes are in memory before the walks read them
		// ================================================ one walk per lane
		int state_w = lane < g_eff * SEGS ? s_state[lane] : 0;
		const int n = s_n[lane], m = s_m[lane], kmin = s_kmin[lane], S = s_S[lane];
		const long long si = s_si[lane];
		int n_c = 0;
		{
			const uint8_t *pu = p_wave + (size_t)(lane / SEGS) * FILLP_PASS_BYTES + (lane % SEGS) * LANES * 32;
			int bi = n - 1, bj = m - 1, state = 0, cid = -1, mcols = 0, g2 = 0, kd = 0, kbest = 0;   // kd / kbest: see mnc_dp_fill's mm_test_zdrop
			uint32_t cur = 0, hm32 = 0;
			bool walking = state_w == 1;
			while (walking && bi >= V && bj >= V) {
				const int r = bi + bj, idx = bi - ((r + kmin + 1) >> 1);
				if (idx < 0 || idx >= W) { state_w = 2; walking = false; break; }   // cannot happen after the proof
				const int c = (r >> 4) * 64 + (idx >> 1);
				if (c != cid) {
					const uint4 *src = reinterpret_cast<const uint4*>(pu + (size_t)c * 32);
					const uint4 v0 = src[0], v1 = src[1];
					*reinterpret_cast<uint4*>(&s_chunk[lane][0]) = v0, *reinterpret_cast<uint4*>(&s_chunk[lane][16]) = v1;
					cid = c;
					// which of the chunk's 32 cells were won by the diagonal (tag 15: bit 3), as four 8-bit masks -- byte
					// (row parity * 2 + cell) holds a bit per pair of rows: a run of matches is read off with one count
					const uint32_t w8[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
					hm32 = 0;
#pragma unroll
					for (int k = 0; k < 8; ++k) hm32 |= (w8[k] >> 3 & 0x01010101u) << k;
				}
				if (state == 0) {
					// on the diagonal: the steps back along it stay in this lane's column, two rows apart -- all those the
					// chunk still holds with the diagonal as their winner go in one turn (a 15-base match run: two turns)
					const int rho = r & 15, bp = rho >> 1;
					const uint32_t m8 = hm32 >> (8 * ((rho & 1) * 2 + (idx & 1))) & 0xffu;
					const uint32_t zeros = ~m8 & ((2u << bp) - 1u);
					int run = zeros ? bp - (31 - __clz((int)zeros)) : bp + 1;
					const int lim = (bi < bj ? bi : bj) + 1 - V;
					run = run < lim ? run : lim;
					if (run > 0) {
						if (cur != 0 && (cur & 0xf) == 0) cur += (uint32_t)run << 4;
						else {
							if (cur != 0) {
								if (n_c >= FILL_CIG_MAX - 4) { state_w = 3; walking = false; break; }
								const int len = (int)(cur >> 4);
								g2 += fill_gap(len, q, e, q2, e2), kd += q + e * len, kbest = kbest > kd ? kbest : kd;   // a gap: `cur` was not an M run
								cg[n_c++] = cur;
							}
							cur = (uint32_t)run << 4;
						}
						bi -= run, bj -= run;
						continue;
					}
				}
				const uint32_t raw = s_chunk[lane][(r & 15) * 2 + (idx & 1)];
				// bits 0-3: the winner's tag (H 15, E 7, F 3, E2 1, F2 0); bits 4-7: which gap states were opened here (fillp_opened)
				if (state != 0 && (fillp_opened(raw >> 4) >> (4 - state) & 1u)) state = 0;
				if (state == 0) { const uint32_t tag = raw & 15u; state = tag >= 8u ? 0 : tag == 7u ? 1 : tag == 3u ? 2 : tag == 1u ? 3 : 4; }
				uint32_t op;
				if (state == 0) op = 0, --bi, --bj;
				else if (state == 1 || state == 3) op = 2, --bi;
				else op = 1, --bj;
				if (cur != 0 && (cur & 0xf) == op) cur += 1u << 4;
				else {
					if (cur != 0) {
						if (n_c >= FILL_CIG_MAX - 4) { state_w = 3; walking = false; break; }   // more operations than the scratch holds
						const int len = (int)(cur >> 4);
						if ((cur & 0xf) == 0) mcols += len, kd = kd - a * len > 0 ? kd - a * len : 0;
						else g2 += fill_gap(len, q, e, q2, e2), kd += q + e * len, kbest = kbest > kd ? kbest : kd;
						cg[n_c++] = cur;
					}
					cur = 1u << 4 | op;
				}
			}
			if (walking) {
				auto push = [&](uint32_t w) {
					const int len = (int)(w >> 4);
					if ((w & 0xf) == 0) mcols += len, kd = kd - a * len > 0 ? kd - a * len : 0;
					else g2 += fill_gap(len, q, e, q2, e2), kd += q + e * len, kbest = kbest > kd ? kbest : kd;
					cg[n_c++] = w;
				};
				if (bi >= V) { if (cur != 0 && (cur & 0xf) == 2) cur += (uint32_t)(bi + 1 - V) << 4; else { if (cur != 0) push(cur); cur = (uint32_t)(bi + 1 - V) << 4 | 2; } }
				if (bj >= V) { if (cur != 0 && (cur & 0xf) == 1) cur += (uint32_t)(bj + 1 - V) << 4; else { if (cur != 0) push(cur); cur = (uint32_t)(bj + 1 - V) << 4 | 1; } }
				if (cur != 0) push(cur);
				// mm_test_zdrop: the drop it looks for is at most (largest sum over contiguous operations of gap costs
				// less a per M column) + (what all columns that do not match cost against matches, from the score); see
				// mnc_dp_fill.  Above the threshold: the walk over the bases below
				const int lost = a * mcols - S - g2;
				const int neg = kbest + lost;
				if (neg > thr) state_w = 4;
			}
		}
		fill_order_mem();                                      // a lane reads its CIGAR back below
		if (state_w == 4) {
			// the walk over the CIGAR (stored last operation first), bases from memory: rare
			const Seg *gs = B.segs + si;
			const uint8_t *read = B.bases + B.offsets[gs->read];
			const int rlen = (int)(B.offsets[gs->read + 1] - B.offsets[gs->read]), rev = gs->rev, qs = gs->qs;
			const int64_t coff = B.seq_off[gs->rid] + gs->ts;
			int score = 0, mx = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0, max_zdrop = 0;
			for (int k = n_c - 1; k >= 0; --k) {
				const uint32_t w = cg[k], op = w & 0xf;
				const int len = (int)(w >> 4);
				if (op == 0) {
					for (int l = 0; l < len; ++l) {
						const int64_t o = coff + i + l;
						const int ct = (int)(B.seq4[o >> 3] >> ((o & 7) * 4) & 15u), pq = qs + j + l;
						int cq = fill_nt4(read[rev ? rlen - 1 - pq : pq]);
						if (rev) cq = 3 - cq;
						score += ct == cq ? a : bmis;
						if (score < mx) {
							const int li = i + l - max_i, lj = j + l - max_j;
							const int z = mx - score - (li > lj ? li - lj : lj - li) * e;
							max_zdrop = max_zdrop > z ? max_zdrop : z;
						} else mx = score, max_i = i + l, max_j = j + l;
					}
					i += len, j += len;
				} else {
					score -= q + e * len;
					if (op == 1) j += len; else i += len;
					if (score < mx) {
						const int li = i - max_i, lj = j - max_j;
						const int z = mx - score - (li > lj ? li - lj : lj - li) * e;
						max_zdrop = max_zdrop > z ? max_zdrop : z;
					} else mx = score, max_i = i, max_j = j;
				}
			}
			// above the smaller threshold the answer may be 1 or 2: the literal kernel decides and reruns
			state_w = max_zdrop > thr ? 3 : 1;
		}
		// ================================================ results
		if (state_w == 2) { const unsigned long long k = atomicAdd(&B.dp_ctr[ctr_next], 1ULL); next_list[k] = (int32_t)si; }
		if (state_w == 3) { const unsigned long long k = atomicAdd(&B.dp_ctr[ctr_fb], 1ULL); fb_list[k] = (int32_t)si; }
		{
			const int mine = state_w == 1 ? n_c : 0;
			int incl = mine;
#pragma unroll
			for (int sft = 1; sft < 64; sft <<= 1) { const int o = __shfl_up(incl, sft); if (lane >= sft) incl += o; }
			const int total = __shfl(incl, 63);
			unsigned long long base = 0;
			if (lane == 0 && total > 0) base = atomicAdd(&B.dp_ctr[1], (unsigned long long)total);
			base = (unsigned long long)__shfl((long long)base, 0);
			if ((long long)(base + total) > B.cig_seg_cap) {
				if (lane == 0) atomicMax(&B.dp_ctr[4], 2ULL);
				if (state_w == 1) { Seg *o = B.segs + si; o->n_cigar = 0, o->zdropped = 0, o->zdrop_code = 0, o->cig_off = 0; }
			} else if (state_w == 1) {
				const unsigned long long off = base + (unsigned long long)(incl - mine);
				for (int k = 0; k < n_c; ++k) B.cig_seg[off + k] = cg[n_c - 1 - k];
				Seg *o = B.segs + si;
				o->n_cigar = n_c, o->zdropped = 0, o->zdrop_code = 0;
				o->max = 0, o->max_t = -1, o->max_q = -1, o->score = S, o->reach_end = 0, o->mqe_t = -1;
				o->cig_off = (int64_t)off;
			}
		}
		fill_order_mem();
	}
}

// ================================================================ extensions
// The left and right extension of a region (ksw_extd2 with KSW_EZ_EXTZ_ONLY): from the corner
// outwards until the Z-drop, reporting the best cell (or the end of the query).  For the flanks of
// most reads the matrix is small and the band (751) never clips it, so the kernel's result is the
// plain two-piece affine DP's; an anti-diagonal has at most min(qlen, tlen) cells: one lane each
// (32 lanes: two segments per wave).  Beyond what the gap-filling kernel does, every step finds
// the maximum of its anti-diagonal in the SSE scan's tie order (the last cell first, then four
// interleaved lanes, then the tail: DPP max-reduce of (H, rank) keys), updates the best cell or
// tests the Z-drop against it, and tracks the best score in the query's last row.  The left
// extension runs on reversed sequences with gaps right-aligned (ties go to the later candidate).
template <int LANES, int CPL, int MAXLEN>
__global__ __launch_bounds__(64) void mnc_dp_ext(Batch B, const int32_t *list, int ctr_n, int ctr_q, int32_t *fb_list, int ctr_fb, uint8_t *p_all)
{
	constexpr int SEGS = 64 / LANES, W = LANES * CPL, ROWB = 64 * CPL, SEQ = 2 * MAXLEN + 2;
	static_assert(CPL == 1 || LANES == 64, "several cells per lane: one segment per wave");
	__shared__ uint8_t s_t[SEGS][SEQ], s_q[SEGS][SEQ];
	__shared__ __align__(16) uint8_t s_win[SEGS][FILL_WIN * W];
	__shared__ uint32_t s_cg[SEGS][SEQ];
	const int lane = threadIdx.x, sg = lane / LANES, L = lane % LANES, lead = sg * LANES;
	const bool leader = L == 0;
	const int a = B.sc_a, bmis = -B.sc_b, scN = -B.sc_ambi, q = B.gap_q, e = B.gap_e, q2 = B.gap_q2, e2 = B.gap_e2;
	uint8_t *p_wave = p_all + (size_t)blockIdx.x * ((size_t)(2 * MAXLEN + FILL_WIN) * 64 * CPL);
	const unsigned long long n_items = B.dp_ctr[ctr_n];
	// a long call is one wave's serial work for milliseconds, beside chip-filling kernels that have four waves on
	// every SIMD: with equal priority it would get a fifth of the issue slots and five times the latency
	if (MAXLEN > FILL_MAX_LEN) __builtin_amdgcn_s_setprio(3);
	for (;;) {
		unsigned long long q0 = 0;
		if (lane == 0) q0 = atomicAdd(&B.dp_ctr[ctr_q], (unsigned long long)SEGS);
		q0 = (unsigned long long)__shfl((long long)q0, 0);
		if (q0 >= n_items) break;                              // every wave gets here: the queue is finite
		const bool has = q0 + sg < n_items;
		const long long si = has ? (long long)list[q0 + sg] : -1;
		struct { int32_t tlen, qlen, ts, qs, read, rid, rev, kind, zdrop, flag; } g = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
		if (has) {
			const Seg *gs = B.segs + si;
			g.tlen = gs->tlen, g.qlen = gs->qlen, g.ts = gs->ts, g.qs = gs->qs, g.read = gs->read, g.rid = gs->rid, g.rev = gs->rev;
			g.kind = gs->kind, g.zdrop = gs->zdrop, g.flag = gs->flag;
		}
		const int n = g.tlen, m = g.qlen;
		const int mn = n < m ? n : m;
		const bool ok = has && n >= 1 && m >= 1 && mn <= W && n + m - 1 <= 2 * MAXLEN;
		const int rgt = (g.flag & EZ_RIGHT) ? 1 : 0;          // gaps right-aligned: a later candidate wins a tie
		if (ok) {
			const uint8_t *read = B.bases + B.offsets[g.read];
			const int rlen = (int)(B.offsets[g.read + 1] - B.offsets[g.read]);
			const int64_t coff = B.seq_off[g.rid];
			const bool rv = g.kind == 0;                          // the left extension: both sequences reversed
			for (int i = L; i < n; i += LANES) {
				const int64_t o = coff + (rv ? g.ts + n - 1 - i : g.ts + i);
				s_t[sg][i] = (uint8_t)(B.seq4[o >> 3] >> ((o & 7) * 4) & 15u);
			}
			const bool acgt = !B.ambig[g.read];
			const int64_t roff = B.offsets[g.read];
			for (int i = L; i < m; i += LANES) s_q[sg][i] = (uint8_t)fill_qcode(B, acgt, roff, read, rlen, g.rev, rv ? g.qs + m - 1 - i : g.qs + i);
		}
		fill_order();
		const int rows = ok ? n + m - 1 : 0;
		int max_rows = rows;
		if (SEGS == 2) { const int o = __shfl_xor(max_rows, 32); max_rows = max_rows > o ? max_rows : o; }
		// per-segment state of ksw_extz_t (the same value in every lane of the segment)
		int ez_max = 0, ez_max_t = -1, ez_max_q = -1, ez_mqe = DP_NEG_INF, ez_mqe_t = -1, zdropped = 0;
		int H1[CPL], H2[CPL], En[CPL], E2n[CPL], Fn[CPL], F2n[CPL];
#pragma unroll
		for (int k = 0; k < CPL; ++k) H1[k] = H2[k] = En[k] = E2n[k] = Fn[k] = F2n[k] = FILL_NEG;
		uint8_t *prow = p_wave + lead * CPL + L;
		int done = 0;                                          // nothing left to find (see mnc_dp_extp's stop bound): the steps that remain are left out
		for (int r = 0; r < max_rows; ++r, prow += ROWB) {
			if (__all(zdropped || done || r >= rows)) break;
			// cell c = L + LANES * k holds t = st0 + c, st0 = max(0, r - m + 1): while r < m the upper
			// neighbour and the diagonal one are the previous cell; from r = m on the left neighbour is the
			// next cell (and the diagonal one, from r = m + 1 on)
			const int st0 = r - m + 1 > 0 ? r - m + 1 : 0, en0 = r < n - 1 ? r : n - 1;
			const int en1 = st0 + (en0 - st0) / 4 * 4;
			unsigned long long key = 0;
			int h_first = FILL_NEG;
			int sE_[CPL], sE2_[CPL], sF_[CPL], sF2_[CPL], hd_[CPL];
#pragma unroll
			for (int k = 0; k < CPL; ++k) {
				int uE = __builtin_amdgcn_update_dpp(FILL_NEG, En[k], 0x138, 0xf, 0xf, false), uE2 = __builtin_amdgcn_update_dpp(FILL_NEG, E2n[k], 0x138, 0xf, 0xf, false);
				int uH = __builtin_amdgcn_update_dpp(FILL_NEG, H2[k], 0x138, 0xf, 0xf, false);
				int dF = __builtin_amdgcn_update_dpp(FILL_NEG, Fn[k], 0x130, 0xf, 0xf, false), dF2 = __builtin_amdgcn_update_dpp(FILL_NEG, F2n[k], 0x130, 0xf, 0xf, false);
				int dH = __builtin_amdgcn_update_dpp(FILL_NEG, H2[k], 0x130, 0xf, 0xf, false);
				if (L == 0) {
					if (k == 0) uE = uE2 = uH = FILL_NEG;
					else uE = __builtin_amdgcn_readlane(En[k > 0 ? k - 1 : 0], 63), uE2 = __builtin_amdgcn_readlane(E2n[k > 0 ? k - 1 : 0], 63), uH = __builtin_amdgcn_readlane(H2[k > 0 ? k - 1 : 0], 63);
				}
				if (L == LANES - 1) {
					if (k == CPL - 1) dF = dF2 = dH = FILL_NEG;
					else dF = __builtin_amdgcn_readlane(Fn[k + 1 < CPL ? k + 1 : k], 0), dF2 = __builtin_amdgcn_readlane(F2n[k + 1 < CPL ? k + 1 : k], 0), dH = __builtin_amdgcn_readlane(H2[k + 1 < CPL ? k + 1 : k], 0);
				}
				sE_[k] = r < m ? uE : En[k], sE2_[k] = r < m ? uE2 : E2n[k];
				sF_[k] = r < m ? Fn[k] : dF, sF2_[k] = r < m ? F2n[k] : dF2;
				hd_[k] = r < m ? uH : r == m ? H2[k] : dH;
			}
#pragma unroll
			for (int k = 0; k < CPL; ++k) {
				int sE = sE_[k], sE2 = sE2_[k], sF = sF_[k], sF2 = sF2_[k], hd = hd_[k];
				const int t = st0 + L + LANES * k, j = r - t;
				const bool act = !zdropped && !done && r < rows && t <= en0;
				int Hn = FILL_NEG, nEn = FILL_NEG, nE2n = FILL_NEG, nFn = FILL_NEG, nF2n = FILL_NEG;
				if (act) {
					const int ct = s_t[sg][t], cq = s_q[sg][j];
					const int sc = (ct == 4 || cq == 4) ? scN : ct == cq ? a : bmis;
					if (t == 0 || j == 0) {
						hd = t == 0 && j == 0 ? 0 : -fill_gap(t == 0 ? j : t, q, e, q2, e2);
						if (t == 0) { const int hb = -fill_gap(j + 1, q, e, q2, e2); sE = hb - q - e, sE2 = hb - q2 - e2; }
						if (j == 0) { const int hb = -fill_gap(t + 1, q, e, q2, e2); sF = hb - q - e, sF2 = hb - q2 - e2; }
					}
					int z = hd + sc, d;
					d = sE + rgt > z ? 1 : 0;  z = z > sE ? z : sE;
					d = sF + rgt > z ? 2 : d;  z = z > sF ? z : sF;
					d = sE2 + rgt > z ? 3 : d; z = z > sE2 ? z : sE2;
					d = sF2 + rgt > z ? 4 : d; z = z > sF2 ? z : sF2;
					Hn = z;
					const int o1 = z - q, o2 = z - q2;
					d |= sE + rgt > o1 ? 0x08 : 0;  nEn = (sE > o1 ? sE : o1) - e;
					d |= sF + rgt > o1 ? 0x10 : 0;  nFn = (sF > o1 ? sF : o1) - e;
					d |= sE2 + rgt > o2 ? 0x20 : 0; nE2n = (sE2 > o2 ? sE2 : o2) - e2;
					d |= sF2 + rgt > o2 ? 0x40 : 0; nF2n = (sF2 > o2 ? sF2 : o2) - e2;
					prow[LANES * k] = (uint8_t)d;
					// the maximum of the anti-diagonal, ties in the SSE scan's order
					const unsigned rank = t == en0 ? 0u : t < en1 ? 1u + ((unsigned)(t - st0) & 3u) * 0x1000000u + ((unsigned)(t - st0) >> 2)
					                                            : 1u + 4u * 0x1000000u + (unsigned)(t - en1);
					const unsigned long long kk = (unsigned long long)(unsigned)(Hn + (1 << 30)) << 32 | (0xffffffffu - rank);
					key = key > kk ? key : kk;
				}
				if (k == 0) h_first = Hn;
				H2[k] = H1[k], H1[k] = Hn, En[k] = nEn, E2n[k] = nE2n, Fn[k] = nFn, F2n[k] = nF2n;
			}
			{
				// max-reduce inside the segment: rows of 16 lanes, then across them
#define MNC_KMAX(ctrl, rmask) { const unsigned lo32 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)key, ctrl, rmask, 0xf, false), \
	hi32 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(key >> 32), ctrl, rmask, 0xf, false); \
	const unsigned long long o = (unsigned long long)hi32 << 32 | lo32; key = key > o ? key : o; }
				MNC_KMAX(0x111, 0xf) MNC_KMAX(0x112, 0xf) MNC_KMAX(0x114, 0xf) MNC_KMAX(0x118, 0xf) MNC_KMAX(0x142, 0xa)
				if (LANES == 64) MNC_KMAX(0x143, 0xc)
#undef MNC_KMAX
			}
			const unsigned long long kseg = (unsigned long long)__shfl((long long)key, lead + LANES - 1);
			const int h_lead = __shfl(h_first, lead);              // H of the cell in the query's last row, when there is one
			if (!zdropped && !done && r < rows) {
				const int max_H = (int)(unsigned)(kseg >> 32) - (1 << 30);
				const unsigned mr = 0xffffffffu - (unsigned)kseg;
				int max_t;
				if (mr == 0) max_t = en0;
				else if (mr < 1u + 4u * 0x1000000u) { const unsigned k = mr - 1u; max_t = st0 + (int)((k & 0xffffffu) * 4u + (k >> 24)); }
				else max_t = en1 + (int)(mr - 1u - 4u * 0x1000000u);
				if (r - st0 == m - 1 && h_lead > ez_mqe) ez_mqe = h_lead, ez_mqe_t = st0;
				// ksw_apply_zdrop
				if (max_H > ez_max) ez_max = max_H, ez_max_t = max_t, ez_max_q = r - max_t;
				else if (max_t >= ez_max_t && r - max_t >= ez_max_q) {
					const int tl = max_t - ez_max_t, ql = (r - max_t) - ez_max_q;
					const int l = tl > ql ? tl - ql : ql - tl;
					if (g.zdrop >= 0 && ez_max - max_H > g.zdrop + l * e2) zdropped = 1;
				}
				// every 16 steps, once the query's last row has been reached: the bound of mnc_dp_extp (here cell c holds query base
				// m - 1 - c from step m - 1 on: it has c query bases left), and its condition that the Z-drop cannot fire later
				if ((r & 15) == 15 && r >= m && !zdropped) {
					int fv = FILL_NEG;
#pragma unroll
					for (int k = 0; k < CPL; ++k) {
						const int v = H1[k] > H2[k] ? H1[k] : H2[k], c = L + LANES * k;
						if (v > FILL_NEG / 2) fv = fv > v + a * c ? fv : v + a * c;
					}
#pragma unroll
					for (int sft = 1; sft < LANES; sft <<= 1) { const int o = __shfl_xor(fv, sft); fv = fv > o ? fv : o; }
					const int vc = a * m - fill_gap(r + 1, q, e, q2, e2);   // (from the virtual column: its cell beside step r + 1's first query base holds -gap(r + 1))
					const int bound = fv > vc ? fv : vc;
					const bool no_drop = g.zdrop < 0 || ez_max - max_H + 2 * q2 + e2 * (rows - 1 - r) <= g.zdrop;
					if (bound <= ez_max && bound <= ez_mqe && no_drop) done = 1;
				}
			}
		}
		// ---- where the backtrack starts
		int bi = -1, bj = -1, reach_end = 0;
		if (ok) {
			if (!zdropped && ez_mqe + B.end_bonus > ez_max) reach_end = 1, bi = ez_mqe_t, bj = m - 1;
			else if (ez_max_t >= 0 && ez_max_q >= 0) bi = ez_max_t, bj = ez_max_q;
		}
		fill_order_mem();
		int state = 0, n_c = 0, wlo = 1 << 30;
		uint32_t cur = 0;
		bool walking = ok && bi >= 0 && bj >= 0, bad = false;
		for (;;) {
			const int seg_r = __shfl(walking ? bi + bj : -1, lead), seg_wlo = __shfl(wlo, lead);
			if (!__any(seg_r >= 0)) break;
			if (seg_r >= 0 && seg_r < seg_wlo) {
				const int lo = seg_r - (FILL_WIN - 1) > 0 ? seg_r - (FILL_WIN - 1) : 0;
#pragma unroll
				for (int k = 0; k < CPL; ++k) {
					const int byte0 = (L + LANES * k) * 16, row = lo + byte0 / W, col = byte0 % W;
					const uint4 v = *reinterpret_cast<const uint4*>(p_wave + (size_t)row * ROWB + lead * CPL + col);
					*reinterpret_cast<uint4*>(&s_win[sg][byte0]) = v;
				}
				wlo = lo;
			}
			fill_order();
			if (leader && walking) {
				while (bi >= 0 && bj >= 0) {
					const int r = bi + bj;
					if (r < wlo) break;
					const int idx = bi - (r - m + 1 > 0 ? r - m + 1 : 0);
					if (idx < 0 || idx >= W) { walking = false, bad = true; break; }
					const uint32_t tmp = s_win[sg][(r - wlo) * W + idx];
					if (state == 0) state = tmp & 7;
					else if (!(tmp >> (state + 2) & 1)) state = 0;
					if (state == 0) state = tmp & 7;
					uint32_t op;
					if (state == 0) op = 0, --bi, --bj;
					else if (state == 1 || state == 3) op = 2, --bi;
					else op = 1, --bj;
					if (cur != 0 && (cur & 0xf) == op) cur += 1u << 4;
					else { if (cur != 0) s_cg[sg][n_c++] = cur; cur = 1u << 4 | op; }
				}
				if (walking && (bi < 0 || bj < 0)) {
					if (bi >= 0) { if (cur != 0 && (cur & 0xf) == 2) cur += (uint32_t)(bi + 1) << 4; else { if (cur != 0) s_cg[sg][n_c++] = cur; cur = (uint32_t)(bi + 1) << 4 | 2; } }
					if (bj >= 0) { if (cur != 0 && (cur & 0xf) == 1) cur += (uint32_t)(bj + 1) << 4; else { if (cur != 0) s_cg[sg][n_c++] = cur; cur = (uint32_t)(bj + 1) << 4 | 1; } }
					if (cur != 0) s_cg[sg][n_c++] = cur;
					walking = false;
				}
			}
			fill_order();
		}
		const bool s_bad = __shfl((int)(bad || (has && !ok)), lead) != 0;
		const int s_nc = __shfl(n_c, lead);
		if (has && s_bad) {
			if (leader && (g.flag & SEG_NEEDS_BIG_WS)) { const unsigned long long k = atomicAdd(&B.dp_ctr[56], 1ULL); B.bigfb_list[k] = (int32_t)si; }
			else if (leader) { const unsigned long long k = atomicAdd(&B.dp_ctr[ctr_fb], 1ULL); fb_list[k] = (int32_t)si; }
		} else if (has) {
			unsigned long long off = 0;
			if (leader && s_nc > 0) off = atomicAdd(&B.dp_ctr[1], (unsigned long long)s_nc);
			off = (unsigned long long)__shfl((long long)off, lead);
			int wrote = s_nc;
			if ((long long)(off + s_nc) > B.cig_seg_cap) {
				if (leader) atomicMax(&B.dp_ctr[4], 2ULL);
				wrote = 0;
			} else {
				// ksw_backtrack leaves the CIGAR last operation first; KSW_EZ_REV_CIGAR keeps that order
				const bool keep = (g.flag & EZ_REV_CIGAR) != 0;
				for (int k = L; k < s_nc; k += LANES) B.cig_seg[off + k] = s_cg[sg][keep ? k : s_nc - 1 - k];
			}
			if (leader) {
				Seg *o = B.segs + si;
				o->n_cigar = wrote, o->zdropped = zdropped, o->zdrop_code = 0;
				o->max = ez_max, o->max_t = ez_max_t, o->max_q = ez_max_q, o->score = DP_NEG_INF, o->reach_end = reach_end, o->mqe_t = ez_mqe_t;
				o->cig_off = (int64_t)off;
			}
		}
		fill_order_mem();
	}
}

// ================================================================ extensions, two cells per lane and instruction
// The same packed arithmetic for the extensions.  Cell c of a segment is QUERY base c for the whole
// run and holds target base t = r - c on step r: the upper neighbour is the cell itself one step
// back, the left and the diagonal ones are cell c - 1 (one / two steps back), the same on every
// step; the virtual column enters at cell 0 on every step, the virtual row at cell c = r.  A lane
// holds 2 CPL consecutive cells, so the shift by one cell is one DPP move and CPL v_alignbit.
//
// What ksw2 does per anti-diagonal -- the maximum, the best cell so far, the Z-drop test -- is not
// done per step: every cell keeps its own best H and the step it first reached it (packed), the best
// cell of the matrix follows at the end (first step, then the SSE scan's order within it), and
// the score of the query's last row (mqe) is cell m - 1's own best.  That is ksw2's result when its
// Z-drop never fires; every 16 steps the kernel checks a bound that rules the Z-drop out until the
// next check (best so far - maximum of this anti-diagonal <= zdrop - what 16 steps can change), and
// a segment that fails it goes to the step-by-step kernel above.  RGT: gaps right-aligned (the left
// extension, KSW_EZ_RIGHT): the later candidate wins a tie, so the tags are the other way round.
constexpr int EXTP_G = 16;
constexpr int EXTP_CIG_MAX = 1024;
constexpr int EXTP_ROWS = 2 * FILL_MAX_LEN;                 // n + m - 1 at most

// right-aligned gaps: x = XOR of the tags 1 / 3 / 7 / 15 (E / F / E2 / F2 extended) or 0 (opened): bit 3 = d, bit 2 = c ^ d,
// bit 1 = b ^ c ^ d, bit 0 = a ^ b ^ c ^ d.  Returns d << 3 | c << 2 | b << 1 | a ("extended" bits, E lowest).
__device__ __forceinline__ uint32_t extp_extended(uint32_t x)
{
	const uint32_t d = x >> 3 & 1u, c = (x >> 2 & 1u) ^ d, b = (x >> 1 & 1u) ^ c ^ d, a = (x & 1u) ^ b ^ c ^ d;
	return d << 3 | c << 2 | b << 1 | a;
}
template <int RGT> struct ExtpTag {
	static constexpr uint32_t H = RGT ? 0u : 0x000f000fu, E = RGT ? 0x00010001u : 0x00070007u, F = 0x00030003u,
	                          E2 = RGT ? 0x00070007u : 0x00010001u, F2 = RGT ? 0x000f000fu : 0u;
};

__host__ __device__ __forceinline__ bool extp_bias(int cells, int a, int bmis_abs, int q, int e, int q2, int e2, int &bias)
{
	const int hi = a * cells, lo = -(bmis_abs * cells + fill_gap(EXTP_ROWS + 32, q, e, q2, e2) + (q + e > q2 + e2 ? q + e : q2 + e2));
	bias = -(hi + lo) / 2;
	// cells that have not started hold what grew from the initial value: still below every real score
	return hi + bias < 2040 && lo + bias > -2040 && lo + bias > -2048 + a * cells + 16;
}

template <int LANES, int CPL, int RGT, bool EDGE, bool LIVE>
__device__ __forceinline__ void extp_step(const PkConst &K, const int r, const int L, const int rows, const uint8_t *st, int &tn,
                                          const int q, const int e, const int q2, const int e2, const uint32_t colmask,
                                          uint32_t (&T)[CPL], const uint32_t (&Q)[CPL], uint32_t (&H1)[CPL], uint32_t (&H2)[CPL],
                                          uint32_t (&E)[CPL], uint32_t (&E2)[CPL], uint32_t (&F)[CPL], uint32_t (&F2)[CPL],
                                          uint32_t (&best)[CPL], uint32_t (&bestR)[CPL], uint32_t &nbH, uint32_t &nbF, uint32_t &nbF2,
                                          uint32_t (&acc)[CPL], const int odd)
{
	typedef ExtpTag<RGT> TG;
	constexpr int SHR = LANES == 16 ? 0x111 : 0x138;           // row_shr:1 / wave_shr:1
	// the target bases move up one cell
	{
		const uint32_t nc = st[tn];
		++tn;
#pragma unroll
		for (int k = CPL - 1; k > 0; --k) T[k] = __builtin_amdgcn_alignbit(T[k], T[k - 1], 16);
		T[0] = __builtin_amdgcn_perm(T[0], nc, 0x05040100u);
	}
	// H two steps back, F and F2 one step back, of the cell below
	nbH = (uint32_t)__builtin_amdgcn_update_dpp((int)nbH, (int)H2[CPL - 1], SHR, 0xf, 0xf, false);
	nbF = (uint32_t)__builtin_amdgcn_update_dpp((int)nbF, (int)F[CPL - 1], SHR, 0xf, 0xf, false);
	nbF2 = (uint32_t)__builtin_amdgcn_update_dpp((int)nbF2, (int)F2[CPL - 1], SHR, 0xf, 0xf, false);
	uint32_t hd[CPL], vF[CPL], vF2[CPL];
#pragma unroll
	for (int k = CPL - 1; k > 0; --k)
		hd[k] = __builtin_amdgcn_alignbit(H2[k], H2[k - 1], 16), vF[k] = __builtin_amdgcn_alignbit(F[k], F[k - 1], 16), vF2[k] = __builtin_amdgcn_alignbit(F2[k], F2[k - 1], 16);
	hd[0] = __builtin_amdgcn_alignbit(H2[0], nbH, 16), vF[0] = __builtin_amdgcn_alignbit(F[0], nbF, 16), vF2[0] = __builtin_amdgcn_alignbit(F2[0], nbF2, 16);
	// the virtual column at cell 0 (t = r), and on the first steps the virtual row at cell c = r (t = 0)
	const int hb = -fill_gap(r + 1, q, e, q2, e2) + K.bias;
	const uint32_t h0 = pk_rep(((r == 0 ? 0 : -fill_gap(r, q, e, q2, e2)) + K.bias) << 4) | TG::H;
	const uint32_t g1 = pk_rep((hb - q - e) << 4), g2 = pk_rep((hb - q2 - e2) << 4);
	hd[0] = pk_bfi(colmask, h0, hd[0]), vF[0] = pk_bfi(colmask, g1 | TG::F, vF[0]), vF2[0] = pk_bfi(colmask, g2 | TG::F2, vF2[0]);
	uint32_t vE[CPL], vE2[CPL];
#pragma unroll
	for (int k = 0; k < CPL; ++k) vE[k] = E[k], vE2[k] = E2[k];
	if (EDGE) {
		// `r` is a multiple of 2 CPL plus a constant in the unrolled block: register and half are static
		const uint32_t rowmask = L == r / (2 * CPL) ? ((odd & 1) ? 0xffff0000u : 0x0000ffffu) : 0u;
		const int ks = (odd >> 1) % CPL;
#pragma unroll
		for (int k = 0; k < CPL; ++k)
			if (k == ks) hd[k] = pk_bfi(rowmask, h0, hd[k]), vE[k] = pk_bfi(rowmask, g1 | TG::E, vE[k]), vE2[k] = pk_bfi(rowmask, g2 | TG::E2, vE2[k]);
	}
	const uint32_t Rpk = pk_rep(r);
#pragma unroll
	for (int k = 0; k < CPL; ++k) {
		const uint32_t sc = pk_madu(pk_subsu(0x00010001u, T[k] ^ Q[k]), K.kmatch, K.kmis);
		uint32_t z = pk_adds(hd[k], sc);
		z = pk_maxs(z, vE[k]), z = pk_maxs(z, vF[k]), z = pk_maxs(z, vE2[k]), z = pk_maxs(z, vF2[k]);
		const uint32_t zt = RGT ? (z & 0xfff0fff0u) : (z | 0x000f000fu);
		const uint32_t o1 = pk_subs(zt, K.q8), o2 = pk_subs(zt, K.q28);
		const uint32_t mE = pk_maxs(vE[k], o1), mF = pk_maxs(vF[k], o1), mE2 = pk_maxs(vE2[k], o2), mF2 = pk_maxs(vF2[k], o2);
		uint32_t d;
		if (RGT) {
			E[k] = pk_subs(mE, K.e8) | TG::E, F[k] = pk_subs(mF, K.e8) | TG::F, E2[k] = pk_subs(mE2, K.e28) | TG::E2, F2[k] = pk_subs(mF2, K.e28) | TG::F2;
			d = mE ^ mF ^ mE2 ^ mF2;                                // tags 1 / 3 / 7 / 15 where extended, 0 where opened: extp_extended
		} else {
			E[k] = pk_subs(mE, K.e8) & 0xfff7fff7u, F[k] = pk_subs(mF, K.e8) & 0xfff3fff3u, E2[k] = pk_subs(mE2, K.e28) & 0xfff1fff1u, F2[k] = pk_subs(mF2, K.e28) & 0xfff0fff0u;
			d = mE ^ mF ^ mE2 ^ mF2;                                // fillp_opened
		}
		d = pk_bfi(0x000f000fu, z, d << 4);                          // bits 0-3 the winner's tag, bits 4-7 the XOR
		acc[k] = (odd & 1) ? __builtin_amdgcn_perm(d, acc[k], 0x06040100u) : __builtin_amdgcn_perm(d, d, 0x0c0c0200u);
		// every cell's best H and the step it first reached it
		const uint32_t zl = LIVE ? (r < rows ? zt : PK_NEG) : zt;
		const uint32_t nb = pk_maxs(best[k], zl), chg = nb ^ best[k];
		const uint32_t msk = __builtin_bit_cast(uint32_t, (pk_u16)(__builtin_bit_cast(pk_u16, pk_subsu(chg, 0x00010001u)) - __builtin_bit_cast(pk_u16, chg)));   // 0xffff where it changed
		bestR[k] = pk_bfi(msk, Rpk, bestR[k]), best[k] = nb;
		H2[k] = H1[k], H1[k] = zt;
	}
}

template <int LANES> __device__ __forceinline__ int seg_max_i32(int v)
{
#pragma unroll
	for (int sft = 1; sft < LANES; sft <<= 1) { const int o = __shfl_xor(v, sft); v = v > o ? v : o; }
	return v;
}
template <int LANES> __device__ __forceinline__ unsigned seg_min_u32(unsigned v)
{
#pragma unroll
	for (int sft = 1; sft < LANES; sft <<= 1) { const unsigned o = (unsigned)__shfl_xor((int)v, sft); v = v < o ? v : o; }
	return v;
}
__device__ __forceinline__ int pk_half(uint32_t v, int h) { return (int)(int16_t)(h ? v >> 16 : v & 0xffffu); }

#ifndef MNC_EXTP_WAVES
#define MNC_EXTP_WAVES 4
#endif
template <int LANES, int CPL, int RGT>
__global__ __launch_bounds__(64, MNC_EXTP_WAVES) void mnc_dp_extp(Batch B, const int32_t *list, int ctr_n, int ctr_q, uint8_t *p_all, uint32_t *cig_all)
{
	constexpr int SEGS = 64 / LANES, WC = 2 * CPL * LANES, PAD = WC + 2, SEQ = PAD + EXTP_ROWS + 48;
	constexpr bool LIVE = SEGS > 1;
	constexpr size_t PASS_BYTES = (size_t)FILLP_BLOCKS * 64 * 32 * CPL;
	typedef ExtpTag<RGT> TG;
	__shared__ uint8_t s_t[SEGS][SEQ], s_q[SEGS][WC];
	__shared__ __align__(16) uint8_t s_chunk[64][32];
	__shared__ int32_t s_n[64], s_m[64], s_bi[64], s_bj[64], s_si[64], s_state[64], s_max[64], s_maxt[64], s_maxq[64], s_mqet[64], s_reach[64], s_keep[64];
	const int lane = threadIdx.x, sg = lane / LANES, L = lane % LANES, lead = sg * LANES;
	const bool leader = L == 0;
	const int a = B.sc_a, bmis = -B.sc_b, q = B.gap_q, e = B.gap_e, q2 = B.gap_q2, e2 = B.gap_e2;
	PkConst K;
	K.kmatch = pk_rep((a - bmis) << 4), K.kmis = pk_rep(bmis << 4), K.q8 = pk_rep(q << 4), K.q28 = pk_rep(q2 << 4), K.e8 = pk_rep(e << 4), K.e28 = pk_rep(e2 << 4);
	const bool fits = extp_bias(WC, a, -bmis > B.sc_ambi ? -bmis : B.sc_ambi, q, e, q2, e2, K.bias);
	const int slack = 2 * q + 16 * e + 8 * a + 8 * (-bmis);    // what 16 steps can add to the best score so far and take from the anti-diagonal's maximum
	uint8_t *p_wave = p_all + (size_t)blockIdx.x * (EXTP_G * PASS_BYTES);
	uint32_t *cg = cig_all + ((size_t)blockIdx.x * 64 + lane) * EXTP_CIG_MAX;
	const unsigned long long n_items = B.dp_ctr[ctr_n];
	const unsigned long long segmask = LANES == 64 ? ~0ULL : ((1ULL << (LANES & 63)) - 1) << lead;
	const uint32_t colmask = L == 0 ? 0x0000ffffu : 0u;
	// few segments: smaller groups, so that every workgroup has some
	int g_eff = (int)((n_items + (unsigned long long)gridDim.x * SEGS - 1) / ((unsigned long long)gridDim.x * SEGS));
	if (g_eff > EXTP_G && g_eff < 2 * EXTP_G) g_eff = (g_eff + 1) / 2;
	g_eff = g_eff < 1 ? 1 : g_eff > EXTP_G ? EXTP_G : g_eff;
	for (;;) {
		unsigned long long q0 = 0;
		if (lane == 0) q0 = atomicAdd(&B.dp_ctr[ctr_q], (unsigned long long)(g_eff * SEGS));
		q0 = (unsigned long long)__shfl((long long)q0, 0);
		if (q0 >= n_items) break;                              // every wave gets here: the queue is finite
		s_state[lane] = 0;
		fill_order();
		for (int u = 0; u < g_eff; ++u) {
			if (q0 + (unsigned long long)u * SEGS >= n_items) break;
			const unsigned long long item = q0 + (unsigned long long)u * SEGS + sg;
			const bool has = item < n_items;
			const long long si = has ? (long long)list[item] : -1;
			struct { int32_t tlen, qlen, ts, qs, read, rid, rev, kind, zdrop, flag; } g = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
			if (has) {
				const Seg *gs = B.segs + si;
				g.tlen = gs->tlen, g.qlen = gs->qlen, g.ts = gs->ts, g.qs = gs->qs, g.read = gs->read, g.rid = gs->rid, g.rev = gs->rev;
				g.kind = gs->kind, g.zdrop = gs->zdrop, g.flag = gs->flag;
			}
			const int n = g.tlen, m = g.qlen;
			bool ok = has && fits && n >= 1 && m >= 1 && m <= WC && n + m - 1 <= EXTP_ROWS && ((g.flag & EZ_RIGHT) != 0) == (RGT != 0);
			const int rows = ok ? n + m - 1 : 0;
			int max_rows = rows, edge_rows = ok ? (m + 15) & ~15 : 0;
#pragma unroll
			for (int sft = LANES; sft < 64; sft <<= 1) {
				const int o = __shfl_xor(max_rows, sft), oe = __shfl_xor(edge_rows, sft);
				max_rows = max_rows > o ? max_rows : o, edge_rows = edge_rows > oe ? edge_rows : oe;
			}
			max_rows = __builtin_amdgcn_readfirstlane(max_rows), edge_rows = __builtin_amdgcn_readfirstlane(edge_rows);
			// ---- sequences: target bases at [PAD + t] (beyond the end: a base that matches nothing), query bases in registers
			bool ambiguous = false;
			if (ok) {
				const uint8_t *read = B.bases + B.offsets[g.read];
				const int rlen = (int)(B.offsets[g.read + 1] - B.offsets[g.read]);
				const int64_t coff = B.seq_off[g.rid];
				const bool rv = g.kind == 0;                          // the left extension: both sequences reversed
#pragma unroll 1
				for (int i = L; i < n; i += LANES) {
					const int64_t o = coff + (rv ? g.ts + n - 1 - i : g.ts + i);
					const uint32_t c = B.seq4[o >> 3] >> ((o & 7) * 4) & 15u;
					ambiguous |= c > 3;
					s_t[sg][PAD + i] = (uint8_t)(c & 3);
				}
#pragma unroll 1
				for (int i = n + L; i < max_rows + 16; i += LANES) s_t[sg][PAD + i] = 4;
				const bool acgt = !B.ambig[g.read];
				const int64_t roff = B.offsets[g.read];
#pragma unroll 1
				for (int i = L; i < WC; i += LANES) {
					int c = 5;
					if (i < m) {
						c = fill_qcode(B, acgt, roff, read, rlen, g.rev, rv ? g.qs + m - 1 - i : g.qs + i);
						ambiguous |= c > 3;
						c &= 3;
					}
					s_q[sg][i] = (uint8_t)c;
				}
			}
			const bool seg_amb = (__ballot(ambiguous) & segmask) != 0;
			ok = ok && !seg_amb;                                   // the step-by-step kernel scores an ambiguous base
			fill_order();
			const uint8_t *st = &s_t[sg][0];
			uint32_t T[CPL], Q[CPL], H1[CPL], H2[CPL], E[CPL], E2[CPL], F[CPL], F2[CPL], best[CPL], bestR[CPL];
#pragma unroll
			for (int k = 0; k < CPL; ++k) {
				const int c0 = 2 * CPL * L + 2 * k;
				Q[k] = (uint32_t)s_q[sg][c0] | (uint32_t)s_q[sg][c0 + 1] << 16;
				T[k] = (uint32_t)st[PAD - 1 - c0] | (uint32_t)st[PAD - 2 - c0] << 16;
				H1[k] = H2[k] = PK_NEG | TG::H, E[k] = PK_NEG | TG::E, E2[k] = PK_NEG | TG::E2, F[k] = PK_NEG | TG::F, F2[k] = PK_NEG | TG::F2;
				best[k] = PK_NEG | TG::H, bestR[k] = 0;
			}
			int tn = PAD - 2 * CPL * L;
			uint32_t nbH = PK_NEG, nbF = PK_NEG, nbF2 = PK_NEG;
			bool suspect = false;
			uint4 *pblk = reinterpret_cast<uint4*>(p_wave + (size_t)u * PASS_BYTES + (size_t)lane * 32 * CPL);
			for (int r = 0; r < max_rows; r += 16, pblk += 64 * 32 * CPL / 16) {
				uint32_t acc[8][CPL] = {};
				if (r < edge_rows) {
#pragma unroll
					for (int k = 0; k < 16; ++k)
						if (r + k < max_rows) extp_step<LANES, CPL, RGT, true, LIVE>(K, r + k, L, rows, st, tn, q, e, q2, e2, colmask, T, Q, H1, H2, E, E2, F, F2, best, bestR, nbH, nbF, nbF2, acc[k >> 1], k);
				} else {
#pragma unroll
					for (int k = 0; k < 16; ++k)
						if (r + k < max_rows) extp_step<LANES, CPL, RGT, false, LIVE>(K, r + k, L, rows, st, tn, q, e, q2, e2, colmask, T, Q, H1, H2, E, E2, F, F2, best, bestR, nbH, nbF, nbF2, acc[k >> 1], k);
				}
#pragma unroll
				for (int k = 0; k < CPL; ++k)
					pblk[2 * k] = make_uint4(acc[0][k], acc[1][k], acc[2][k], acc[3][k]), pblk[2 * k + 1] = make_uint4(acc[4][k], acc[5][k], acc[6][k], acc[7][k]);
				// ---- can the Z-drop fire before the next check?  best so far against this anti-diagonal's maximum
				const int rl = r + 15 < max_rows - 1 ? r + 15 : max_rows - 1;
				int gv = -(1 << 20), av = -(1 << 20);
#pragma unroll
				for (int k = 0; k < CPL; ++k)
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const int c = 2 * CPL * L + 2 * k + h;
						if (c < m) {
							const int bv = pk_half(best[k], h) >> 4, hv = pk_half(H1[k], h) >> 4;
							gv = gv > bv ? gv : bv;
							if ((unsigned)(rl - c) < (unsigned)n) av = av > hv ? av : hv;
						}
					}
				gv = seg_max_i32<LANES>(gv) - K.bias, av = seg_max_i32<LANES>(av) - K.bias;
				gv = gv > 0 ? gv : 0;
				if (ok && g.zdrop >= 0 && rl < rows && gv - av > g.zdrop - slack) suspect = true;
				// ---- nothing left to find?  Every later cell is reached from this block's last two anti-diagonals (the gap states a
				// cell hands on are below its own H, and lead to cells with no more query left than it has) or from the virtual
				// column: its score is at most H(cell c) + a (m - 1 - c) -- a diagonal step takes a query base, a gap gains nothing --
				// or a m - gap(step).  Once that bound is no more than the best score so far AND than the best of the query's
				// last row so far, no later cell changes either (an equal score does not replace the first one): the target window
				// is about twice the query flank, and from the step where the query's end passes the diagonal on there is little to find.
				bool fin = !ok || !has || r + 16 >= rows;
				{
					const int cm = ok ? m - 1 : 0, src = lead + cm / (2 * CPL), km = cm % (2 * CPL) / 2;
					uint32_t bq = 0;
#pragma unroll
					for (int k = 0; k < CPL; ++k) { const uint32_t x = (uint32_t)__shfl((int)best[k], src); if (k == km) bq = x; }
					int fv = -(1 << 20);
#pragma unroll
					for (int k = 0; k < CPL; ++k) {
						const uint32_t hm = pk_maxs(H1[k], H2[k]);
#pragma unroll
						for (int h = 0; h < 2; ++h) {
							const int c = 2 * CPL * L + 2 * k + h, fc = (pk_half(hm, h) >> 4) + a * (m - 1 - c);
							if (c < m) fv = fv > fc ? fv : fc;
						}
					}
					fv = seg_max_i32<LANES>(fv) - K.bias;
					if (!fin && r + 16 >= m) {                              // (every cell has started: the virtual row is behind)
						const int mq = (pk_half(bq, cm & 1) >> 4) - K.bias;
						const int vc = a * m - fill_gap(r + 16, q, e, q2, e2);   // (the virtual column's cell beside step r + 16's first query base holds -gap(r + 16))
						const int bound = fv > vc ? fv : vc;
						// ... and ksw2's Z-drop must not fire in the steps left out either (it would clear reach_end): the maximum of a
						// later anti-diagonal, k steps on, is at least this one's less two gap openings and k extensions (from its best
						// cell some way down and some way right, inside the matrix), the threshold at least zdrop
						const int left = rows - 1 - rl;
#ifdef MNC_EXT_STOP_IGNORES_ZDROP                                             // (a test build: tests/test_gpu_dp.py's flanks must then differ from the oracle)
						const bool no_drop = true;
						(void)left;
#else
						const bool no_drop = g.zdrop < 0 || gv - av + 2 * q2 + e2 * left <= g.zdrop;   // (a gap costs at most its second piece)
#endif
						fin = bound <= gv && bound <= mq && no_drop;
					}
				}
				if (!__any(!fin)) break;
			}
			// ---- the best cell: the first step that reached the best score, then the SSE scan's order on that step
			int ez_max = 0, ez_max_t = -1, ez_max_q = -1, ez_mqe = DP_NEG_INF, ez_mqe_t = -1;
			{
				int gb = -(1 << 20);
#pragma unroll
				for (int k = 0; k < CPL; ++k)
#pragma unroll
					for (int h = 0; h < 2; ++h) { const int bv = pk_half(best[k], h); if (2 * CPL * L + 2 * k + h < m && bv > gb) gb = bv; }
				gb = seg_max_i32<LANES>(gb);
				const int gmax = (gb >> 4) - K.bias;
				if (ok && gmax > 0) {
					unsigned rmin = 0xffffffffu;
#pragma unroll
					for (int k = 0; k < CPL; ++k)
#pragma unroll
						for (int h = 0; h < 2; ++h)
							if (2 * CPL * L + 2 * k + h < m && pk_half(best[k], h) == gb) { const unsigned rr = (h ? bestR[k] >> 16 : bestR[k] & 0xffffu); rmin = rmin < rr ? rmin : rr; }
					rmin = seg_min_u32<LANES>(rmin);
					const int rr = (int)rmin, st0 = rr - m + 1 > 0 ? rr - m + 1 : 0, en0 = rr < n - 1 ? rr : n - 1, en1 = st0 + (en0 - st0) / 4 * 4;
					unsigned key = 0xffffffffu;
#pragma unroll
					for (int k = 0; k < CPL; ++k)
#pragma unroll
						for (int h = 0; h < 2; ++h) {
							const int c = 2 * CPL * L + 2 * k + h;
							if (c < m && pk_half(best[k], h) == gb && (h ? bestR[k] >> 16 : bestR[k] & 0xffffu) == rmin) {
								const int t = rr - c;
								const unsigned rank = t == en0 ? 0u : t < en1 ? 256u * (1u + ((unsigned)(t - st0) & 3u)) + ((unsigned)(t - st0) >> 2) : 256u * 5u + (unsigned)(t - en1);
								const unsigned kk = rank << 10 | (unsigned)t;
								key = key < kk ? key : kk;
							}
						}
					key = seg_min_u32<LANES>(key);
					ez_max = gmax, ez_max_t = (int)(key & 1023u), ez_max_q = rr - ez_max_t;
				}
				// the query's last row: cell m - 1
				const int cm = ok ? m - 1 : 0, src = lead + cm / (2 * CPL), km = cm % (2 * CPL) / 2;
				uint32_t bq = 0, bqr = 0;
#pragma unroll
				for (int k = 0; k < CPL; ++k) {
					const uint32_t x = (uint32_t)__shfl((int)best[k], src), y = (uint32_t)__shfl((int)bestR[k], src);
					if (k == km) bq = x, bqr = y;
				}
				if (ok) ez_mqe = (pk_half(bq, cm & 1) >> 4) - K.bias, ez_mqe_t = (int)((cm & 1) ? bqr >> 16 : bqr & 0xffffu) - cm;
			}
			if (leader && has) {
				const int w = u * SEGS + sg;
				int bi = -1, bj = -1, reach_end = 0;
				if (ok && !suspect) {
					if (ez_mqe + B.end_bonus > ez_max) reach_end = 1, bi = ez_mqe_t, bj = m - 1;
					else if (ez_max_t >= 0 && ez_max_q >= 0) bi = ez_max_t, bj = ez_max_q;
				}
				s_n[w] = n, s_m[w] = m, s_bi[w] = bi, s_bj[w] = bj, s_si[w] = (int32_t)si, s_keep[w] = (g.flag & EZ_REV_CIGAR) != 0;
				s_max[w] = ez_max, s_maxt[w] = ez_max_t, s_maxq[w] = ez_max_q, s_mqet[w] = ez_mqe_t, s_reach[w] = reach_end;
				s_state[w] = ok && !suspect ? 1 : 2;
			}
			fill_order();
		}
		fill_order_mem();                                      // the direction bytes are in memory before the walks read them
		// ================================================ one walk per lane
		int state_w = lane < g_eff * SEGS ? s_state[lane] : 0;
		const int n = s_n[lane], m = s_m[lane];
		const long long si = s_si[lane];
		int n_c = 0;
		if (state_w == 1) {
			const uint8_t *pu = p_wave + (size_t)(lane / SEGS) * PASS_BYTES;
			const int slot = lane % SEGS;
			int bi = s_bi[lane], bj = s_bj[lane], state = 0, cid = -1;
			uint32_t cur = 0;
			bool walking = bi >= 0 && bj >= 0, bad = false;
			while (walking && bi >= 0 && bj >= 0) {
				const int r = bi + bj;
				const int c = ((r >> 4) * 64 + slot * LANES + bj / (2 * CPL)) * CPL + bj % (2 * CPL) / 2;
				if (c != cid) {
					const uint4 *src = reinterpret_cast<const uint4*>(pu + (size_t)c * 32);
					const uint4 v0 = src[0], v1 = src[1];
					*reinterpret_cast<uint4*>(&s_chunk[lane][0]) = v0, *reinterpret_cast<uint4*>(&s_chunk[lane][16]) = v1;
					cid = c;
				}
				const uint32_t raw = s_chunk[lane][(r & 15) * 2 + (bj & 1)], tag = raw & 15u;
				if (RGT) {
					if (state != 0 && !(extp_extended(raw >> 4) >> (state - 1) & 1u)) state = 0;     // not extended here: opened
					if (state == 0) state = tag == 0u ? 0 : tag == 1u ? 1 : tag == 3u ? 2 : tag == 7u ? 3 : 4;
				} else {
					if (state != 0 && (fillp_opened(raw >> 4) >> (4 - state) & 1u)) state = 0;
					if (state == 0) state = tag >= 8u ? 0 : tag == 7u ? 1 : tag == 3u ? 2 : tag == 1u ? 3 : 4;
				}
				uint32_t op;
				if (state == 0) op = 0, --bi, --bj;
				else if (state == 1 || state == 3) op = 2, --bi;
				else op = 1, --bj;
				if (cur != 0 && (cur & 0xf) == op) cur += 1u << 4;
				else {
					if (cur != 0) {
						if (n_c >= EXTP_CIG_MAX - 4) { bad = true; break; }
						cg[n_c++] = cur;
					}
					cur = 1u << 4 | op;
				}
			}
			if (bad) state_w = 2;
			else if (walking) {
				if (bi >= 0) { if (cur != 0 && (cur & 0xf) == 2) cur += (uint32_t)(bi + 1) << 4; else { if (cur != 0) cg[n_c++] = cur; cur = (uint32_t)(bi + 1) << 4 | 2; } }
				if (bj >= 0) { if (cur != 0 && (cur & 0xf) == 1) cur += (uint32_t)(bj + 1) << 4; else { if (cur != 0) cg[n_c++] = cur; cur = (uint32_t)(bj + 1) << 4 | 1; } }
				if (cur != 0) cg[n_c++] = cur;
			}
		}
		fill_order_mem();                                      // a lane reads its CIGAR back below
		if (state_w == 2) {
			// to the step-by-step kernel, by the length of its anti-diagonals
			const int mn = n < m ? n : m;
			const int ci = mn <= 32 ? 16 : mn <= 64 ? 17 : mn <= 128 ? 24 : 25;
			int32_t *lst = mn <= 32 ? B.ext_list1 : mn <= 64 ? B.ext_list2 : mn <= 128 ? B.ext_list3 : B.ext_list4;
			if (mn <= 256) { const unsigned long long k = atomicAdd(&B.dp_ctr[ci], 1ULL); lst[k] = (int32_t)si; }
			else { const unsigned long long k = atomicAdd(&B.dp_ctr[12], 1ULL); B.fill_fb[k] = (int32_t)si; }
		}
		{
			const int mine = state_w == 1 ? n_c : 0;
			int incl = mine;
#pragma unroll
			for (int sft = 1; sft < 64; sft <<= 1) { const int o = __shfl_up(incl, sft); if (lane >= sft) incl += o; }
			const int total = __shfl(incl, 63);
			unsigned long long base = 0;
			if (lane == 0 && total > 0) base = atomicAdd(&B.dp_ctr[1], (unsigned long long)total);
			base = (unsigned long long)__shfl((long long)base, 0);
			const bool over = (long long)(base + total) > B.cig_seg_cap;
			if (over && lane == 0) atomicMax(&B.dp_ctr[4], 2ULL);
			if (state_w == 1) {
				const unsigned long long off = base + (unsigned long long)(incl - mine);
				// ksw_backtrack leaves the CIGAR last operation first; KSW_EZ_REV_CIGAR keeps that order
				const bool keep = s_keep[lane] != 0;
				if (!over) for (int k = 0; k < n_c; ++k) B.cig_seg[off + k] = cg[keep ? k : n_c - 1 - k];
				Seg *o = B.segs + si;
				o->n_cigar = over ? 0 : n_c, o->zdropped = 0, o->zdrop_code = 0;
				o->max = s_max[lane], o->max_t = s_maxt[lane], o->max_q = s_maxq[lane], o->score = DP_NEG_INF, o->reach_end = s_reach[lane], o->mqe_t = s_mqet[lane];
				o->cig_off = (int64_t)off;
			}
		}
		fill_order_mem();
	}
}

size_t dp_extp_slot() { return (size_t)EXTP_G * FILLP_BLOCKS * 64 * 32 * 2; }
size_t dp_extp_cig_slot() { return (size_t)64 * EXTP_CIG_MAX * 4; }
// `cells`: 32 / 64 / 128 / 256 query bases at most; rgt: the left extensions (gaps right-aligned)
void launch_dp_extp(const Batch &B, int cells, int rgt, const int32_t *list, int ctr_n, int ctr_q, uint8_t *p_all, uint32_t *cig_all, int n_wg, hipStream_t st)
{
#define MNC_EXTP(LN, CP) do { if (rgt) hipLaunchKernelGGL((mnc_dp_extp<LN, CP, 1>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, p_all, cig_all); \
	else hipLaunchKernelGGL((mnc_dp_extp<LN, CP, 0>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, p_all, cig_all); } while (0)
	if (cells == 32) MNC_EXTP(16, 1);
	else if (cells == 64) MNC_EXTP(32, 1);
	else if (cells == 128) MNC_EXTP(64, 1);
	else MNC_EXTP(64, 2);
#undef MNC_EXTP
}

constexpr int LFILL_MAX_LEN = 2047, LFILL_CIG_MAX = 1024;
size_t dp_lfill_p_slot() { return (size_t)(2 * LFILL_MAX_LEN + FILL_WIN) * 64 * 4; }
// gaps between seeds of 512 .. 2047 bases: the int32 banded kernel with a band of 256 cells
void launch_dp_lfill(const Batch &B, const int32_t *list, int ctr_n, int ctr_q, int32_t *fb_list, int ctr_fb, uint8_t *p_all, int n_wg, hipStream_t st)
{
	hipLaunchKernelGGL((mnc_dp_fill<64, 4, LFILL_MAX_LEN, LFILL_CIG_MAX>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, fb_list, ctr_fb, fb_list, ctr_fb, p_all);
}
constexpr int LEXT_MAX_LEN = 1535;
size_t dp_lext_p_slot() { return (size_t)(2 * LEXT_MAX_LEN + FILL_WIN) * 64 * 8; }
// extensions of 257 .. 512 bases on the shorter side: the step-by-step kernel with eight cells per lane
void launch_dp_lext(const Batch &B, const int32_t *list, int ctr_n, int ctr_q, int32_t *fb_list, int ctr_fb, uint8_t *p_all, int n_wg, hipStream_t st)
{
	hipLaunchKernelGGL((mnc_dp_ext<64, 8, LEXT_MAX_LEN>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, fb_list, ctr_fb, p_all);
}
size_t dp_fill_p_slot() { return FILL_P_SLOT; }
size_t dp_fillp_slot() { return FILLP_SLOT; }
size_t dp_fillp_cig_slot() { return (size_t)64 * FILL_CIG_MAX * 4; }
void launch_dp_fill(const Batch &B, int lanes, const int32_t *list, int ctr_n, int ctr_q, int32_t *next_list, int ctr_next,
                    int32_t *fb_list, int ctr_fb, uint8_t *p_all, uint32_t *cig_all, int n_wg, hipStream_t st)
{
	// `lanes` = cells of the band
	// the drifting frame when the scores allow it (b = 2 e: minimap2's map-ont), debug_route bit 10: never
	bool drift = fillp_drifts(B.sc_b, B.gap_e, B.gap_e2) && !(B.debug_route & 1024);
	if (drift && MNC_FILLP_FRAME2 && lanes <= 64) {
		// the second piece's frame needs room between the 12-bit floor and what the band proof guarantees:
		// a (cells + 1) + gap(D0) + gap(I0) + one opening, with D0 + I0 = 2 (cells + 1); scores that do not leave it take the plain frame
		int bias = 0;
		const bool fits = fillp_bias_drift(B.sc_a, B.gap_q, B.gap_e, B.gap_q2, B.gap_e2, bias);
		const int floor_pts = 2048 + bias;                       // the floor is -floor_pts points
		const int qmax = B.gap_q > B.gap_q2 ? B.gap_q : B.gap_q2, emax = B.gap_e > B.gap_e2 ? B.gap_e : B.gap_e2;
		const int need = B.sc_a * (lanes + 1) + 2 * qmax + emax * 2 * (lanes + 1) + (B.gap_q2 + B.gap_e2) + 16;
		if (!fits || need >= floor_pts) drift = false;
	}
#define MNC_LAUNCH_FILLP(LN) do { if (drift) hipLaunchKernelGGL((mnc_dp_fillp<LN, true>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, next_list, ctr_next, fb_list, ctr_fb, p_all, cig_all); \
	else hipLaunchKernelGGL((mnc_dp_fillp<LN, false>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, next_list, ctr_next, fb_list, ctr_fb, p_all, cig_all); } while (0)
	if (lanes == 32) MNC_LAUNCH_FILLP(16);
	else if (lanes == FILL_MID_CELLS) MNC_LAUNCH_FILLP(FILL_MID_CELLS / 2);
	else if (lanes == 64) MNC_LAUNCH_FILLP(32);
	else MNC_LAUNCH_FILLP(64);
#undef MNC_LAUNCH_FILLP
}

void launch_dp_ext(const Batch &B, int lanes, const int32_t *list, int ctr_n, int ctr_q, int32_t *fb_list, int ctr_fb, uint8_t *p_all, int n_wg, hipStream_t st)
{
	if (lanes == 32) hipLaunchKernelGGL((mnc_dp_ext<32, 1, FILL_MAX_LEN>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, fb_list, ctr_fb, p_all);
	else if (lanes == 64) hipLaunchKernelGGL((mnc_dp_ext<64, 1, FILL_MAX_LEN>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, fb_list, ctr_fb, p_all);
	else if (lanes == 128) hipLaunchKernelGGL((mnc_dp_ext<64, 2, FILL_MAX_LEN>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, fb_list, ctr_fb, p_all);
	else hipLaunchKernelGGL((mnc_dp_ext<64, 4, FILL_MAX_LEN>), dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, fb_list, ctr_fb, p_all);
}

} // namespace mnc
