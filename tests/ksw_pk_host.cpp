// Test program (CPU): the packed-pair form of ksw2's kernel (monica_amd/csrc/ksw_pk.h, compiled for the host with its
// plain-C++ emulation of the VOP3P operations) in ksw2's own ARRAY layout -- pairs of cells (2k, 2k + 1), anti-diagonal
// ranges rounded to 16, in-place updates from the top, scores refreshed on [st0, st0 + 16 n) only, the exact-maximum scan's
// tie order, the approximate-maximum walk, Z-drop, direction codes decoded by kpk::decode in ksw_backtrack -- against the
// oracle's literal int8 simulation (oracle/mm_ksw.c: orc_ksw_extd2) on random calls: every result field and every CIGAR.
// What the GPU kernel adds to this (cells in registers of one wave, the window that moves with st) is held to the oracle
// by the -m gpu tests; this program pins the arithmetic: tags, tie rules, the rewritten gap-state update, the code.
//
// usage: ksw_pk_host <n_cases> <seed>      prints "ok <n>" or the first mismatch; exit code 0 / 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "../monica_amd/csrc/ksw_pk.h"
extern "C" {
#include "../oracle/mm_oracle.h"
}

namespace {

constexpr int NEG_INF = -0x40000000;
struct Ez { int max = 0, zdropped = 0, max_q = -1, max_t = -1, mqe = NEG_INF, mqe_t = -1, score = NEG_INF, reach_end = 0; std::vector<uint32_t> cigar; };

uint64_t rng_state = 1;
uint64_t rnd() { rng_state += 0x9E3779B97F4A7C15ull; uint64_t z = rng_state; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
int rnd_int(int lo, int hi) { return lo + (int)(rnd() % (uint64_t)(hi - lo + 1)); }

void push_cigar(std::vector<uint32_t> &c, uint32_t op, int len)
{
	if (c.empty() || (c.back() & 0xf) != op) c.push_back((uint32_t)len << 4 | op);
	else c.back() += (uint32_t)len << 4;
}

template <bool RIGHT>
void run_packed(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int q, int e, int q2, int e2, int mch, int mis,
                int w, int zdrop, int end_bonus, int flag, Ez &ez)
{
	using namespace kpk;
	const Consts K = make_consts<RIGHT>(q, e, q2, e2, mch, mis);
	const bool approx_max = flag & ORC_EZ_APPROX_MAX;
	const int qe = q + e;
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	const int wl = w, wr = w;
	int n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	const int ncol = n_col_ * 16;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const int T = (tlen + 15) / 16 * 16, P = T / 2 + 16;           // pairs (a score range may run 15 cells past T)
	std::vector<uint32_t> U(P, K.iuv), V(P, K.iuv), X(P, K.ix), Y(P, K.iy), X2(P, K.ix2), Y2(P, K.iy2), S(P, K.is);
	std::vector<int> H(T + 32, NEG_INF);
	std::vector<uint8_t> p((size_t)(qlen + tlen) * ncol + 16, 0);
	std::vector<uint8_t> sf(T + 64, 0), qr(qlen + T + 96, 0);
	for (int t = 0; t < tlen; ++t) sf[t] = target[t];
	for (int t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	auto half = [](uint32_t v, int h) { return h ? v >> 16 : v & 0xffffu; };
	auto set_half = [](uint32_t &v, int h, uint32_t x) { v = h ? (v & 0xffffu) | x << 16 : (v & 0xffff0000u) | (x & 0xffffu); };
	auto val = [&](const std::vector<uint32_t> &A, int t) { return value_of(A[t >> 1], t & 1); };
	int last_st = -1, last_en = -1, H0 = 0, last_H0_t = 0;
	for (int r = 0; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
		if (en > (r + wl) >> 1) en = (r + wl) >> 1;
		if (st > en) { ez.zdropped = 1; break; }
		const int st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		const int v_edge = r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
		uint32_t x1, x21, v1;                                      // the cell below st, in the HIGH half
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = X[(st - 1) >> 1], x21 = X2[(st - 1) >> 1], v1 = V[(st - 1) >> 1];   // st - 1 is odd: the pair's high half
			else x1 = K.ix, x21 = K.ix2, v1 = K.iuv;
		} else x1 = K.ix, x21 = K.ix2, v1 = lane(v_edge, 0);
		if (en >= r) {
			set_half(Y[r >> 1], r & 1, half(K.iy, 0)), set_half(Y2[r >> 1], r & 1, half(K.iy2, 0));
			set_half(U[r >> 1], r & 1, half(lane(v_edge, 0), 0));
		}
		// scores on [st0, st0 + 16 n): per pair, both halves or one
		const int lim = st0 + ((en0 - st0) / 16 + 1) * 16;
		for (int k = st0 >> 1; k <= (lim - 1) >> 1; ++k) {
			uint32_t tb = 0, qb = 0;
			for (int h = 0; h < 2; ++h) {
				const int t = 2 * k + h, qi = qlen - 1 - r + t;
				set_half(tb, h, t < T ? sf[t] : 0), set_half(qb, h, qi >= 0 ? qr[qi] : 0);
			}
			const uint32_t sn = scores(K, tb, qb);
			const uint32_t mask = (2 * k >= st0 && 2 * k < lim ? 0xffffu : 0) | (2 * k + 1 >= st0 && 2 * k + 1 < lim ? 0xffff0000u : 0);
			S[k] = bitsel(mask, sn, S[k]);
		}
		// the cells, pairs from the top: [t - 1] is still old
		for (int k = en >> 1; k >= st >> 1; --k) {
			const bool first = k == st >> 1;
			const uint32_t xb = shift16(X[k], first ? x1 : X[k - 1]), vb = shift16(V[k], first ? v1 : V[k - 1]), x2b = shift16(X2[k], first ? x21 : X2[k - 1]);
			const uint32_t d = cell_pair<RIGHT>(K, xb, vb, x2b, S[k], U[k], V[k], X[k], Y[k], X2[k], Y2[k]);
			p[(size_t)r * ncol + (2 * k - st)] = (uint8_t)d, p[(size_t)r * ncol + (2 * k + 1 - st)] = (uint8_t)(d >> 16);
		}
		if (!approx_max) {
			int max_H, max_t;
			if (r > 0) {
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				int HH[4], tt[4];
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + val(U, en0) : H[en0] + val(V, en0);
				max_t = en0;
				for (int i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				int t;
				for (t = st0; t < en1; t += 4)
					for (int i = 0; i < 4; ++i) {
						H[t + i] += val(V, t + i);
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				for (int i = 0; i < 4; ++i) if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) { H[t] += val(V, t); if (H[t] > max_H) max_H = H[t], max_t = t; }
			} else H[0] = val(V, 0) - qe, max_H = H[0], max_t = 0;
			if (r - st0 == qlen - 1 && H[st0] > ez.mqe) ez.mqe = H[st0], ez.mqe_t = st0;
			bool stop = false;
			if (max_H > ez.max) ez.max = max_H, ez.max_t = max_t, ez.max_q = r - max_t;
			else if (max_t >= ez.max_t && r - max_t >= ez.max_q) {
				const int tl = max_t - ez.max_t, ql = (r - max_t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
				if (zdrop >= 0 && ez.max - max_H > zdrop + l * e2) ez.zdropped = 1, stop = true;
			}
			if (stop) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H[tlen - 1];
		} else {
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					const int d0 = val(V, last_H0_t), d1 = val(U, last_H0_t + 1);
					if (d0 > d1) H0 += d0;
					else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += val(V, last_H0_t);
				else ++last_H0_t, H0 += val(U, last_H0_t);
			} else H0 = val(V, 0) - qe, last_H0_t = 0;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H0;
		}
		last_st = st, last_en = en;
	}
	// ksw_backtrack on the decoded codes
	int i0 = -1, j0 = -1;
	if (!ez.zdropped && !(flag & ORC_EZ_EXTZ_ONLY)) i0 = tlen - 1, j0 = qlen - 1;
	else if (!ez.zdropped && (flag & ORC_EZ_EXTZ_ONLY) && ez.mqe + end_bonus > ez.max) ez.reach_end = 1, i0 = ez.mqe_t, j0 = qlen - 1;
	else if (ez.max_t >= 0 && ez.max_q >= 0) i0 = ez.max_t, j0 = ez.max_q;
	if (i0 >= 0 && j0 >= 0) {
		int i = i0, j = j0, state = 0;
		while (i >= 0 && j >= 0) {
			const int r = i + j;
			int st = 0, en = tlen - 1;
			if (st < r - qlen + 1) st = r - qlen + 1;
			if (en > r) en = r;
			if (st < (r - wr + 1) >> 1) st = (r - wr + 1) >> 1;
			if (en > (r + wl) >> 1) en = (r + wl) >> 1;
			st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
			int force_state = -1;
			if (i < st) force_state = 2;
			if (i > en) force_state = 1;
			const uint32_t tmp = force_state < 0 ? kpk::decode<RIGHT>(p[(size_t)r * ncol + (i - st)]) : 0;
			if (state == 0) state = tmp & 7;
			else if (!(tmp >> (state + 2) & 1)) state = 0;
			if (state == 0) state = tmp & 7;
			if (force_state >= 0) state = force_state;
			if (state == 0) push_cigar(ez.cigar, 0, 1), --i, --j;
			else if (state == 1 || state == 3) push_cigar(ez.cigar, 2, 1), --i;
			else push_cigar(ez.cigar, 1, 1), --j;
		}
		if (i >= 0) push_cigar(ez.cigar, 2, i + 1);
		if (j >= 0) push_cigar(ez.cigar, 1, j + 1);
		if (!(flag & ORC_EZ_REV_CIGAR)) std::reverse(ez.cigar.begin(), ez.cigar.end());
	}
}

} // namespace

int main(int argc, char **argv)
{
	const int n_cases = argc > 1 ? atoi(argv[1]) : 2000;
	rng_state = argc > 2 ? strtoull(argv[2], nullptr, 0) : 1;
	const int q = 4, e = 2, q2 = 24, e2 = 1, mch = 2, mis = -4;
	if (!kpk::params_fit(q, e, q2, e2, mch, mis, -1)) { printf("params do not fit\n"); return 1; }
	int8_t mat[25];
	orc_gen_simple_mat(5, mat, mch, -mis, 1);
	for (int c = 0; c < n_cases; ++c) {
		// shapes: small matrices, long thin ones, matrices the band clips (w far below the lengths), skewed ones
		const int shape = rnd_int(0, 5);
		int qlen, tlen, w;
		if (shape == 0) qlen = rnd_int(1, 40), tlen = rnd_int(1, 40), w = rnd_int(0, 50);
		else if (shape == 1) qlen = rnd_int(30, 400), tlen = rnd_int(30, 400), w = rnd_int(5, 120);
		else if (shape == 2) qlen = rnd_int(200, 900), tlen = qlen + rnd_int(-150, 300), w = rnd_int(20, 200);
		else if (shape == 3) qlen = rnd_int(1, 30), tlen = rnd_int(100, 600), w = rnd_int(1, 700);
		else if (shape == 4) qlen = rnd_int(100, 600), tlen = rnd_int(1, 30), w = rnd_int(1, 700);
		else qlen = rnd_int(50, 500), tlen = rnd_int(50, 1000), w = -1;
		if (tlen < 1) tlen = 1;
		std::vector<uint8_t> t(tlen), qv(qlen);
		for (auto &b : t) b = (uint8_t)rnd_int(0, 3);
		// the query: a noisy copy of a stretch of the target, junk, or a low-complexity sequence (many ties)
		const int kind = rnd_int(0, 3);
		const int err = rnd_int(0, 30);
		int ti = rnd_int(0, std::max(0, tlen - qlen));
		for (int j = 0; j < qlen; ++j) {
			if (kind == 3) { qv[j] = (uint8_t)(j / 3 % 2); continue; }
			if (kind == 2) { qv[j] = (uint8_t)rnd_int(0, 3); continue; }
			const int x = rnd_int(0, 99);
			if (x < err / 3) { qv[j] = (uint8_t)rnd_int(0, 3); continue; }            // insertion
			if (x < 2 * err / 3 && ti + 1 < tlen) ++ti;                               // deletion
			qv[j] = ti < tlen ? t[ti] : (uint8_t)rnd_int(0, 3);
			if (x >= 2 * err / 3 && x < err) qv[j] = (uint8_t)((qv[j] + 1 + rnd_int(0, 2)) & 3);
			++ti;
		}
		if (kind == 3) for (int i = 0; i < tlen; ++i) t[i] = (uint8_t)(i / 3 % 2);
		const int mode = rnd_int(0, 4);
		int flag = mode == 0 ? ORC_EZ_APPROX_MAX : mode == 1 ? 0 : mode == 2 ? ORC_EZ_EXTZ_ONLY : mode == 3 ? (ORC_EZ_EXTZ_ONLY | ORC_EZ_RIGHT | ORC_EZ_REV_CIGAR) : ORC_EZ_RIGHT;
		if (rnd_int(0, 3) == 0) flag ^= ORC_EZ_RIGHT;
		const int zdrop = rnd_int(0, 2) == 0 ? rnd_int(10, 120) : rnd_int(0, 1) ? 400 : 200;
		const int end_bonus = rnd_int(0, 1) ? -1 : rnd_int(0, 20);
		orc_extz_t oz;
		memset(&oz, 0, sizeof(oz));
		orc_ksw_extd2(qlen, qv.data(), tlen, t.data(), 5, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, &oz);
		Ez ez;
		if (flag & ORC_EZ_RIGHT) run_packed<true>(qlen, qv.data(), tlen, t.data(), q, e, q2, e2, mch, mis, w, zdrop, end_bonus, flag, ez);
		else run_packed<false>(qlen, qv.data(), tlen, t.data(), q, e, q2, e2, mch, mis, w, zdrop, end_bonus, flag, ez);
		const bool approx = flag & ORC_EZ_APPROX_MAX;
		bool ok = ez.zdropped == oz.zdropped && ez.score == oz.score && ez.reach_end == oz.reach_end && (int)ez.cigar.size() == oz.n_cigar;
		if (!approx) ok = ok && ez.max == (int)oz.max && ez.max_t == oz.max_t && ez.max_q == oz.max_q && ez.mqe == oz.mqe && ez.mqe_t == oz.mqe_t;
		for (int k = 0; ok && k < oz.n_cigar; ++k) ok = ez.cigar[k] == oz.cigar[k];
		if (!ok) {
			printf("MISMATCH case %d: qlen %d tlen %d w %d flag 0x%x zdrop %d end_bonus %d kind %d | packed: zdropped %d score %d max %d (%d, %d) mqe %d@%d reach %d n_cigar %zu | oracle: zdropped %d score %d max %d (%d, %d) mqe %d@%d reach %d n_cigar %d\n",
			       c, qlen, tlen, w, flag, zdrop, end_bonus, kind, ez.zdropped, ez.score, ez.max, ez.max_t, ez.max_q, ez.mqe, ez.mqe_t, ez.reach_end, ez.cigar.size(),
			       oz.zdropped, oz.score, (int)oz.max, oz.max_t, oz.max_q, oz.mqe, oz.mqe_t, oz.reach_end, oz.n_cigar);
			free(oz.cigar);
			return 1;
		}
		free(oz.cigar);
	}
	printf("ok %d\n", n_cases);
	return 0;
}
