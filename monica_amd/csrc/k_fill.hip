// Stage kernel K8b: gap filling between two seeds, the common case of the base-level alignment
// stage -- gfx950.
//
// Replaces the first (approximate-maximum) pass of mm_align1's gap filling: ksw_extd2_sse as a
// GLOBAL alignment of ~200 x 200 bases with a band (1.5 bw + 1 = 751) that never clips such a
// matrix, then mm_test_zdrop on its CIGAR (monica/genomes/aligner.py:193, 215).
//
// ksw2 fills the whole matrix because its band is wider than the matrix.  The alignment it
// reports lies close to the main diagonal, so this kernel fills only a diagonal band and then
// PROVES that the band held the answer: a path that leaves a band of half-width b around the
// diagonals 0 .. tlen - qlen has at least b + 1 gap bases more than it needs in each direction,
// which bounds its score by U = a (matches left) - gap(b + 1 ...) (dp_band_bound); when the
// banded score is strictly above U, every co-optimal path lies inside the band, every value on
// them equals the full matrix's, and the backtrack takes the same turns (ties are between
// co-optimal paths, all inside).  Otherwise the segment goes to the next tier: 64 lanes instead
// of 32 per segment, then the literal kernel of k_align.hip (also for a CIGAR whose walk shows
// a Z-drop: that needs the exact second pass).
//
// A segment is one half-wave (or one wave): lane L holds cell t = t0(r) + L of anti-diagonal r
// (t = target index, t0 advances every other step), absolute int32 scores, two-piece affine
// gaps.  A cell needs H of the same lane two steps back (the diagonal), the gap states its upper
// neighbour produced (lane L - 1 or L, by the parity of the step) and those of its left
// neighbour (lane L or L + 1): all state is in registers, two values cross lanes per step.
// Direction bytes (which of H / E / F / E2 / F2 wins, with ksw2's priority; "the next cell's
// gap state extends this one" x 4) stream to HBM, 32 (64) contiguous bytes per segment and
// step; the backtrack walks them through a 16-row window in LDS that the whole half-wave refills.
#include "device.h"

namespace mnc {

constexpr int FILL_NEG = -(1 << 28);
constexpr int FILL_WIN = 16;                            // rows of direction bytes held in LDS by the backtrack
constexpr int FILL_MIN_BAND = 8;                        // narrower bands are not worth a try
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

__host__ __device__ __forceinline__ int fill_gap(int l, int q, int e, int q2, int e2)
{
	const int g1 = q + e * l, g2 = q2 + e2 * l;
	return g1 < g2 ? g1 : g2;
}

// band of half-width b around the diagonals 0 .. d (d = tlen - qlen): offsets i - j in [kmin, kmax]
__host__ __device__ __forceinline__ void fill_band(int n, int m, int lanes, int &b, int &kmin, int &kmax)
{
	const int d = n - m, ad = d < 0 ? -d : d;
	b = (2 * lanes - 2 - ad) / 2;
	kmin = (d < 0 ? d : 0) - b, kmax = (d > 0 ? d : 0) + b;
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

// LANES = 32: two segments per wave; 64: one.  Proof failures go to `next_list` (the wider tier, or
// the literal kernel), CIGARs whose walk shows a large score drop to `fb_list` (the literal kernel
// runs minimap2's exact second pass).
template <int LANES>
__global__ __launch_bounds__(64) void mnc_dp_fill(Batch B, const int32_t *list, int ctr_n, int ctr_q, int32_t *next_list, int ctr_next,
                                                  int32_t *fb_list, int ctr_fb, uint8_t *p_all)
{
	constexpr int SEGS = 64 / LANES;
	__shared__ uint8_t s_t[SEGS][FILL_MAX_LEN + 1], s_q[SEGS][FILL_MAX_LEN + 1];
	__shared__ __align__(16) uint8_t s_win[SEGS][FILL_WIN * LANES];
	__shared__ uint32_t s_cg[SEGS][2 * FILL_MAX_LEN + 2];
	const int lane = threadIdx.x, sg = lane / LANES, L = lane % LANES, lead = sg * LANES;
	const bool leader = L == 0;
	const int a = B.sc_a, bmis = -B.sc_b, scN = -B.sc_ambi, q = B.gap_q, e = B.gap_e, q2 = B.gap_q2, e2 = B.gap_e2;
	uint8_t *p_wave = p_all + (size_t)blockIdx.x * FILL_P_SLOT;
	const unsigned long long n_items = B.dp_ctr[ctr_n];
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
		fill_band(n, m, LANES, b, kmin, kmax);
		bool ok = has && n >= 1 && m >= 1 && n <= FILL_MAX_LEN && m <= FILL_MAX_LEN && b >= FILL_MIN_BAND;
		// ---- sequences
		if (ok) {
			const uint8_t *read = B.bases + B.offsets[g.read];
			const int rlen = (int)(B.offsets[g.read + 1] - B.offsets[g.read]);
			const int64_t coff = B.seq_off[g.rid] + g.ts;
			for (int i = L; i < n; i += LANES) {
				const int64_t o = coff + i;
				s_t[sg][i] = (uint8_t)(B.seq4[o >> 3] >> ((o & 7) * 4) & 15u);
			}
			for (int i = L; i < m; i += LANES) {
				const int pq = g.qs + i;
				const int c = fill_nt4(read[g.rev ? rlen - 1 - pq : pq]);
				s_q[sg][i] = (uint8_t)(g.rev ? (c < 4 ? 3 - c : 4) : c);
			}
		}
		fill_order();
		const int rows = ok ? n + m - 1 : 0;
		int max_rows = rows;
		if (SEGS == 2) { const int o = __shfl_xor(max_rows, 32); max_rows = max_rows > o ? max_rows : o; }
		// ---- forward: one anti-diagonal per step
		int H1 = FILL_NEG, H2 = FILL_NEG, En = FILL_NEG, E2n = FILL_NEG, Fn = FILL_NEG, F2n = FILL_NEG, Sc = FILL_NEG;
		uint8_t *prow = p_wave + lane;
		for (int r = 0; r < max_rows; ++r, prow += 64) {
			const int t0 = (r + kmin + 1) >> 1;
			// t0(r) = ceil((r + kmin) / 2) advances on the steps where r + kmin is odd: then the left
			// neighbour sits one lane up, else the upper one one lane down.  Uniform over the wave.
			// The two gap states that cross lanes this step: DPP wave shifts, one instruction each.
			int sE, sE2, sF, sF2;
			if (((r + kmin) & 1) == 0) {
				sE = __builtin_amdgcn_update_dpp(FILL_NEG, En, 0x138, 0xf, 0xf, false);      // wave_shr:1
				sE2 = __builtin_amdgcn_update_dpp(FILL_NEG, E2n, 0x138, 0xf, 0xf, false);
				if (LANES == 32 && L == 0) sE = sE2 = FILL_NEG;                                // lane 32 got the other segment's
				sF = Fn, sF2 = F2n;
			} else {
				sF = __builtin_amdgcn_update_dpp(FILL_NEG, Fn, 0x130, 0xf, 0xf, false);      // wave_shl:1
				sF2 = __builtin_amdgcn_update_dpp(FILL_NEG, F2n, 0x130, 0xf, 0xf, false);
				if (LANES == 32 && L == LANES - 1) sF = sF2 = FILL_NEG;
				sE = En, sE2 = E2n;
			}
			const int t = t0 + L, j = r - t;
			const bool act = r < rows && t >= 0 && j >= 0 && t < n && j < m && 2 * t - r <= kmax;
			int Hn = FILL_NEG, nEn = FILL_NEG, nE2n = FILL_NEG, nFn = FILL_NEG, nF2n = FILL_NEG;
			if (act) {
				const int ct = s_t[sg][t], cq = s_q[sg][j];
				const int sc = (ct == 4 || cq == 4) ? scN : ct == cq ? a : bmis;
				int hd = H2;
				if (t == 0 || j == 0) {                              // virtual row / column: gaps from the corner
					hd = t == 0 && j == 0 ? 0 : -fill_gap(t == 0 ? j : t, q, e, q2, e2);
					if (t == 0) { const int hb = -fill_gap(j + 1, q, e, q2, e2); sE = hb - q - e, sE2 = hb - q2 - e2; }
					if (j == 0) { const int hb = -fill_gap(t + 1, q, e, q2, e2); sF = hb - q - e, sF2 = hb - q2 - e2; }
				}
				int z = hd + sc, d;
				d = sE > z ? 1 : 0;  z = z > sE ? z : sE;
				d = sF > z ? 2 : d;  z = z > sF ? z : sF;
				d = sE2 > z ? 3 : d; z = z > sE2 ? z : sE2;
				d = sF2 > z ? 4 : d; z = z > sF2 ? z : sF2;
				Hn = z;
				const int o1 = z - q, o2 = z - q2;
				d |= sE > o1 ? 0x08 : 0;  nEn = (sE > o1 ? sE : o1) - e;
				d |= sF > o1 ? 0x10 : 0;  nFn = (sF > o1 ? sF : o1) - e;
				d |= sE2 > o2 ? 0x20 : 0; nE2n = (sE2 > o2 ? sE2 : o2) - e2;
				d |= sF2 > o2 ? 0x40 : 0; nF2n = (sF2 > o2 ? sF2 : o2) - e2;
				*prow = (uint8_t)d;
				if (r == rows - 1) Sc = Hn;                          // the corner: the global score
			}
			H2 = H1, H1 = Hn, En = nEn, E2n = nE2n, Fn = nFn, F2n = nF2n;
		}
		// ---- the proof: every path that leaves the band scores at most U
		int S = FILL_NEG;
		{
			const int lc = ok ? n - 1 - ((rows - 1 + kmin + 1) >> 1) : 0;
			S = __shfl(Sc, lead + (lc < 0 ? 0 : lc >= LANES ? LANES - 1 : lc));
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
			if (seg_r >= 0 && seg_r < seg_wlo) {                   // refill: rows [lo, lo + FILL_WIN), 16 bytes per lane
				const int lo = seg_r - (FILL_WIN - 1) > 0 ? seg_r - (FILL_WIN - 1) : 0;
				const int byte0 = L * 16, row = lo + byte0 / LANES, col = byte0 % LANES;
				const uint4 v = *reinterpret_cast<const uint4*>(p_wave + (size_t)row * 64 + lead + col);
				*reinterpret_cast<uint4*>(&s_win[sg][byte0]) = v;
				wlo = lo;
			}
			fill_order();
			if (leader && walking) {
				while (bi >= 0 && bj >= 0) {
					const int r = bi + bj;
					if (r < wlo) break;
					const int idx = bi - ((r + kmin + 1) >> 1);
					if (idx < 0 || idx >= LANES || 2 * bi - r > kmax) { walking = false, to_next = true; break; }   // cannot happen after the proof
					const uint32_t tmp = s_win[sg][(r - wlo) * LANES + idx];
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
		// ---- mm_test_zdrop: a walk over the CIGAR (stored last operation first)
		if (leader && ok && !to_next) {
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
				if (s_fb) { const unsigned long long k = atomicAdd(&B.dp_ctr[ctr_fb], 1ULL); fb_list[k] = (int32_t)si; }
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

size_t dp_fill_p_slot() { return FILL_P_SLOT; }
void launch_dp_fill(const Batch &B, int lanes, const int32_t *list, int ctr_n, int ctr_q, int32_t *next_list, int ctr_next,
                    int32_t *fb_list, int ctr_fb, uint8_t *p_all, int n_wg, hipStream_t st)
{
	if (lanes == 32) hipLaunchKernelGGL(mnc_dp_fill<32>, dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, next_list, ctr_next, fb_list, ctr_fb, p_all);
	else hipLaunchKernelGGL(mnc_dp_fill<64>, dim3(n_wg), dim3(64), 0, st, B, list, ctr_n, ctr_q, next_list, ctr_next, fb_list, ctr_fb, p_all);
}

} // namespace mnc
