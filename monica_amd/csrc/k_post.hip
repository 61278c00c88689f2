// Stage kernel K6: chains -> regions -> hierarchy -> long-join -> chain-level MAPQ ->
// monica's per-read decision and taxon counts -- gfx950.
//
// Replaces, for one index part: mm_gen_regs / mm_set_parent / mm_select_sub / mm_join_long /
// mm_set_mapq inside index.map(seq) and the Python that
// consumes the hits: the gate `hit.is_primary and hit.mapq >= mapping_quality`
// (monica/genomes/aligner.py:194,216), best_hit (aligner.py:328-339), the ambiguous /
// unmapped split (aligner.py:225-233, 264-265) and the count accumulation
// (aligner.py:247-263).
//
// One thread per read: a read has a handful of regions, so this stage is a few hundred
// scalar operations per read.  The float32 islands of minimap2 (overlap ratio, pri_ratio,
// MAPQ) are evaluated with IEEE single-precision operations in the reference's order
// (the library is built with -ffp-contract=off and correctly rounded division); logf()
// comes from a table the host fills with its own libm, so values are bit-identical to a
// host evaluation.
#include "device.h"

namespace mnc {

__device__ __forceinline__ uint64_t mix64(uint64_t key)
{
	key = ~key + (key << 21);
	key = key ^ key >> 24;
	key = (key + (key << 3)) + (key << 8);
	key = key ^ key >> 14;
	key = (key + (key << 2)) + (key << 4);
	key = key ^ key >> 28;
	key = key + (key << 31);
	return key;
}

__device__ __forceinline__ uint32_t wang32(uint32_t key)
{
	key += ~(key << 15);
	key ^=  (key >> 10);
	key +=  (key << 3);
	key ^=  (key >> 6);
	key += ~(key << 11);
	key ^=  (key >> 16);
	return key;
}

__device__ void sort_u64(uint64_t *a, int n)       // ascending; insertion for the common tiny case
{
	if (n <= 24) {
		for (int i = 1; i < n; ++i) {
			uint64_t x = a[i];
			int j = i - 1;
			while (j >= 0 && a[j] > x) { a[j + 1] = a[j]; --j; }
			a[j + 1] = x;
		}
		return;
	}
	for (int start = n / 2 - 1; start >= 0; --start) {
		int root = start;
		for (;;) {
			int c = 2 * root + 1;
			if (c >= n) break;
			if (c + 1 < n && a[c] < a[c + 1]) ++c;
			if (a[root] >= a[c]) break;
			uint64_t x = a[root]; a[root] = a[c], a[c] = x;
			root = c;
		}
	}
	for (int end = n - 1; end > 0; --end) {
		uint64_t x = a[0]; a[0] = a[end], a[end] = x;
		int root = 0;
		for (;;) {
			int c = 2 * root + 1;
			if (c >= end) break;
			if (c + 1 < end && a[c] < a[c + 1]) ++c;
			if (a[root] >= a[c]) break;
			uint64_t y = a[root]; a[root] = a[c], a[c] = y;
			root = c;
		}
	}
}

// coordinates of a region from its first / last anchor (mm_reg_set_coor)
__device__ __forceinline__ void set_coor(mnc_reg_t &r, const RegX &e, int qlen)
{
	const int32_t q_span = (int32_t)(e.y0 >> 32 & 0xff);
	r.rev = (int32_t)(e.x0 >> 63);
	r.rid = (int32_t)(e.x0 << 1 >> 33);
	r.rs = (int32_t)e.x0 + 1 > q_span ? (int32_t)e.x0 + 1 - q_span : 0;
	r.re = (int32_t)e.x1 + 1;
	if (!r.rev) {
		r.qs = (int32_t)e.y0 + 1 - q_span;
		r.qe = (int32_t)e.y1 + 1;
	} else {
		r.qs = qlen - ((int32_t)e.y1 + 1);
		r.qe = qlen - ((int32_t)e.y0 + 1 - q_span);
	}
}

// re-number ids after a compaction and re-point parents (mm_sync_regs)
__device__ void sync_regs(int n_regs, mnc_reg_t *regs, int32_t *tmp)
{
	if (n_regs <= 0) return;
	int max_id = -1;
	for (int i = 0; i < n_regs; ++i) max_id = max_id > regs[i].id ? max_id : regs[i].id;
	for (int i = 0; i <= max_id; ++i) tmp[i] = -1;
	for (int i = 0; i < n_regs; ++i) if (regs[i].id >= 0) tmp[regs[i].id] = i;
	for (int i = 0; i < n_regs; ++i) {
		mnc_reg_t &r = regs[i];
		r.id = i;
		if (r.parent == -2) r.parent = i;
		else if (r.parent >= 0 && tmp[r.parent] >= 0) r.parent = tmp[r.parent];
		else r.parent = -1;
	}
}

__global__ __launch_bounds__(64) void mnc_regions_decide(Batch B, RegX *regx_all, uint64_t *k64a_all,
                                                         uint64_t *k64b_all, mnc_hit_t *gated_all)
{
	const uint32_t rd = blockIdx.x * blockDim.x + threadIdx.x;
	if (rd >= B.n_reads) return;
	const int qlen = (int)(B.offsets[rd + 1] - B.offsets[rd]);
	int32_t assign = MNC_UNMAPPED, nhits = 0;
	mnc_hit_t best;
	best.rid = best.mapq = best.nm = best.mlen = 0;
	int n = B.n_chain[rd];
	int n_regs = 0;
	if (n > 0) {
		const int64_t slot = B.an_off[rd] / 3;
		ChainRec *ch = B.chains_tmp + slot;                     // backtrack order (pad = rank)
		mnc_reg_t *r = B.regs + slot;
		RegX *ex = regx_all + slot;
		uint64_t *ka = k64a_all + slot, *kb = k64b_all + slot;
		int32_t *w = B.tmp_i32 + slot * 4, *tmp = w + n;        // n ints each (4 per slot available)
		mnc_hit_t *gated = gated_all + slot;

		// ---------------- chains ordered by (first anchor x, rank); `as` = running anchor count
		// in that order (mm_chain_dp's final ordering; total order instead of an unstable sort)
		for (int i = 0; i < n; ++i) w[i] = i;
		for (int i = 1; i < n; ++i) {
			const int c = w[i];
			int j = i - 1;
			while (j >= 0 && (ch[w[j]].x0 > ch[c].x0 || (ch[w[j]].x0 == ch[c].x0 && w[j] > c))) { w[j + 1] = w[j]; --j; }
			w[j + 1] = c;
		}
		{
			int as = 0;
			for (int i = 0; i < n; ++i) { ch[w[i]].as = as; as += ch[w[i]].cnt; }
		}
		// ---------------- regions, sorted by score (desc) with the pseudo-random tie-break
		uint32_t hash = wang32((uint32_t)qlen) + wang32((uint32_t)B.seed);
		hash = wang32(hash);
		for (int i = 0; i < n; ++i) {
			const uint32_t h = (uint32_t)mix64((mix64(ch[i].x0) + mix64(ch[i].y0)) ^ (uint64_t)hash);
			ka[i] = ((uint64_t)(uint32_t)ch[i].score << 32 | (uint32_t)ch[i].cnt) ^ (uint64_t)h;
			kb[i] = (uint64_t)(uint32_t)ch[i].as << 32 | (uint32_t)i;   // `as` is unique: total order
		}
		// sort chain indices by (ka, as) descending: selection through insertion (n is small)
		for (int i = 0; i < n; ++i) w[i] = i;
		for (int i = 1; i < n; ++i) {
			const int c = w[i];
			int j = i - 1;
			while (j >= 0 && (ka[w[j]] < ka[c] || (ka[w[j]] == ka[c] && kb[w[j]] < kb[c]))) { w[j + 1] = w[j]; --j; }
			w[j + 1] = c;
		}
		for (int i = 0; i < n; ++i) {
			const ChainRec &c = ch[w[i]];
			mnc_reg_t x;
			x.id = i, x.parent = -1;
			x.score = x.score0 = (int32_t)(ka[w[i]] >> 32);
			x.hash = (uint32_t)ka[w[i]];
			x.cnt = c.cnt, x.as = c.as, x.mlen = c.mlen, x.blen = c.blen;
			x.subsc = 0, x.n_sub = 0, x.mapq = 0;
			RegX e;
			e.x0 = c.x0, e.y0 = c.y0, e.x1 = c.x1, e.y1 = c.y1;
			set_coor(x, e, qlen);
			r[i] = x, ex[i] = e;
		}
		n_regs = n;

		// ---------------- parent / secondary, subsc, n_sub (mm_set_parent)
		{
			uint64_t *cov = ka;
			int k = 1;
			w[0] = 0, r[0].parent = 0;
			for (int i = 1; i < n_regs; ++i) {
				mnc_reg_t &ri = r[i];
				const int si = ri.qs, ei = ri.qe;
				int n_cov = 0, uncov_len = 0, j;
				for (j = 0; j < k; ++j) {
					const mnc_reg_t &rp = r[w[j]];
					int sj = rp.qs, ej = rp.qe;
					if (ej <= si || sj >= ei) continue;
					if (sj < si) sj = si;
					if (ej > ei) ej = ei;
					cov[n_cov++] = (uint64_t)(uint32_t)sj << 32 | (uint32_t)ej;
				}
				j = k;
				if (n_cov > 0) {
					int x = si;
					sort_u64(cov, n_cov);
					for (int jj = 0; jj < n_cov; ++jj) {
						if ((int)(cov[jj] >> 32) > x) uncov_len += (int)(cov[jj] >> 32) - x;
						x = (int32_t)cov[jj] > x ? (int32_t)cov[jj] : x;
					}
					if (ei > x) uncov_len += ei - x;
					for (j = 0; j < k; ++j) {
						mnc_reg_t &rp = r[w[j]];
						const int sj = rp.qs, ej = rp.qe;
						if (ej <= si || sj >= ei) continue;
						const int mn = ej - sj < ei - si ? ej - sj : ei - si;
						const int mx = ej - sj > ei - si ? ej - sj : ei - si;
						const int ol = si < sj ? (ei < sj ? 0 : ei < ej ? ei - sj : ej - sj)
						                       : (ej < si ? 0 : ej < ei ? ej - si : ei - si);
						const float lhs = __fsub_rn(__fdiv_rn((float)ol, (float)mn), __fdiv_rn((float)uncov_len, (float)mx));
						if (lhs > B.mask_level) {
							ri.parent = rp.parent;
							rp.subsc = rp.subsc > ri.score ? rp.subsc : ri.score;
							if (ri.cnt >= rp.cnt) ++rp.n_sub;
							break;
						}
					}
				}
				if (j == k) w[k++] = i, ri.parent = i, ri.n_sub = 0;
			}
		}

		// ---------------- keep primaries and the best secondaries (mm_select_sub); the
		// in-place compaction reads r[p] after earlier slots may have been overwritten
		if (B.pri_ratio > 0.0f) {
			const int min_diff = KMER * 2;
			int k = 0, n_2nd = 0;
			for (int i = 0; i < n_regs; ++i) {
				const int p = r[i].parent;
				if (p == i) {
					r[k] = r[i], ex[k] = ex[i], ++k;
				} else if (((float)r[i].score >= __fmul_rn((float)r[p].score, B.pri_ratio) || r[i].score + min_diff >= r[p].score) && n_2nd < B.best_n) {
					if (!(r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re)) {
						r[k] = r[i], ex[k] = ex[i], ++k, ++n_2nd;
					}
				}
			}
			if (k != n_regs) sync_regs(k, r, tmp);
			n_regs = k;
		}

		// ---------------- long-join of adjacent co-linear primaries (mm_join_long)
		if (n_regs >= 2) {
			uint64_t *aux = ka;
			// squeeze: `as` becomes the running anchor count in original-`as` order
			for (int i = 0; i < n_regs; ++i) aux[i] = (uint64_t)(uint32_t)r[i].as << 32 | (uint32_t)i;
			sort_u64(aux, n_regs);
			int as = 0;
			for (int i = 0; i < n_regs; ++i) {
				mnc_reg_t &x = r[(int32_t)(uint32_t)aux[i]];
				x.as = as;
				as += x.cnt;
			}
			int n_aux = 0, n_drop = 0;
			for (int i = 0; i < n_regs; ++i)
				if (r[i].parent == i || r[i].parent < 0) aux[n_aux++] = (uint64_t)(uint32_t)r[i].as << 32 | (uint32_t)i;
			sort_u64(aux, n_aux);
			for (int i = n_aux - 1; i >= 1; --i) {
				const int i0 = (int32_t)(uint32_t)aux[i - 1], i1 = (int32_t)(uint32_t)aux[i];
				mnc_reg_t &r0 = r[i0], &r1 = r[i1];
				RegX &e0 = ex[i0];
				const RegX &e1 = ex[i1];
				if (r0.as + r0.cnt != r1.as) continue;
				if (r0.rid != r1.rid || r0.rev != r1.rev) continue;
				if (e1.x0 <= e0.x1 || (int32_t)e1.y0 <= (int32_t)e0.y1) continue;
				const int64_t dx = (int64_t)(e1.x0 - e0.x1);
				int max_gap = (int32_t)e1.y0 - (int32_t)e0.y1, min_gap = max_gap;
				max_gap = max_gap > dx ? max_gap : (int)dx;
				min_gap = min_gap < dx ? min_gap : (int)dx;
				if (max_gap > B.max_join_long || min_gap > B.max_join_short) continue;
				const float per = __fdiv_rn((float)B.min_join_flank_sc, (float)B.max_join_long);
				const int sc_thres = (int)((double)__fmul_rn(per, (float)max_gap) + .499);
				if (r0.score < sc_thres || r1.score < sc_thres) continue;
				const int min_flank_len = (int)__fmul_rn((float)max_gap, B.min_join_flank_ratio);
				if (r0.re - r0.rs < min_flank_len || r0.qe - r0.qs < min_flank_len) continue;
				if (r1.re - r1.rs < min_flank_len || r1.qe - r1.qs < min_flank_len) continue;
				// join: r0 absorbs r1
				{
					const int sp = (int)(e1.y0 >> 32 & 0xff);
					const int tl = (int32_t)e1.x0 - (int32_t)e0.x1;
					const int ql = (int32_t)e1.y0 - (int32_t)e0.y1;
					r0.blen += (tl > ql ? tl : ql) + (r1.blen - sp);
					r0.mlen += (tl > sp && ql > sp ? sp : tl < ql ? tl : ql) + (r1.mlen - sp);
				}
				r0.cnt += r1.cnt, r0.score += r1.score;
				e0.x1 = e1.x1, e0.y1 = e1.y1;
				{ const int32_t m = r0.mlen, b = r0.blen; set_coor(r0, e0, qlen); r0.mlen = m, r0.blen = b; }
				r1.cnt = 0;
				r1.parent = r0.id;
				++n_drop;
			}
			if (n_drop > 0) {
				for (int i = 0; i < n_regs; ++i) {
					mnc_reg_t &x = r[i];
					if (x.parent >= 0 && x.id != x.parent)
						if (r[x.parent].parent >= 0 && r[x.parent].parent != x.parent) x.parent = r[x.parent].parent;
				}
				int k = 0;
				for (int i = 0; i < n_regs; ++i) {          // mm_filter_regs: cnt < min_cnt
					if (r[i].cnt < B.min_cnt) continue;
					if (k < i) r[k] = r[i], ex[k] = ex[i];
					++k;
				}
				n_regs = k;
				sync_regs(n_regs, r, tmp);
			}
		}

		// ---------------- chain-level MAPQ (mm_set_mapq, branch without base-level DP)
		{
			long long sum_sc = 0;
			for (int i = 0; i < n_regs; ++i) if (r[i].parent == r[i].id) sum_sc += r[i].score;
			const float uniq_ratio = __fdiv_rn((float)sum_sc, (float)(sum_sc + (long long)B.rep_len[rd]));
			for (int i = 0; i < n_regs; ++i) {
				mnc_reg_t &x = r[i];
				if (x.parent == x.id) {
					const float pen_s1 = __fmul_rn(x.score > 100 ? 1.0f : __fmul_rn(0.01f, (float)x.score), uniq_ratio);
					float pen_cm = x.cnt > 10 ? 1.0f : __fmul_rn(0.1f, (float)x.cnt);
					pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
					const int subsc = x.subsc > B.min_sc ? x.subsc : B.min_sc;
					const float xr = __fdiv_rn((float)subsc, (float)x.score0);
					const int li = x.score < B.logf_n ? x.score : B.logf_n - 1;
					const int ls = x.n_sub + 1 < B.logf_n ? x.n_sub + 1 : B.logf_n - 1;
					float q = __fmul_rn(__fmul_rn(__fmul_rn(pen_cm, 40.0f), __fsub_rn(1.0f, xr)), B.logf_lut[li]);
					int mapq = (int)q;
					mapq -= (int)__fadd_rn(__fmul_rn(4.343f, B.logf_lut[ls]), .499f);
					mapq = mapq > 0 ? mapq : 0;
					x.mapq = mapq < 60 ? mapq : 60;
				} else x.mapq = 0;
			}
		}

		// ---------------- monica: gate, best_hit, decision (aligner.py:216-233)
		int bi = -1, ties = 0;
		for (int i = 0; i < n_regs; ++i) {
			const mnc_reg_t &x = r[i];
			if (x.id == x.parent && x.mapq >= B.min_mapq) {
				mnc_hit_t h;
				h.rid = x.rid, h.mapq = x.mapq, h.nm = x.blen - x.mlen, h.mlen = x.mlen;
				gated[nhits] = h;
				if (bi < 0) bi = nhits, ties = 1;
				else {
					const long long l = (long long)h.nm * gated[bi].mlen, rr = (long long)gated[bi].nm * h.mlen;
					if (l < rr) bi = nhits, ties = 1;
					else if (l == rr) bi = nhits, ++ties;
				}
				++nhits;
			}
		}
		if (nhits > 0) {                                  // best = the minimal hit, also when it is tied
			best = gated[bi];
			assign = (nhits == 1 || ties == 1) ? gated[bi].rid : MNC_AMBIGUOUS;
		}
	}
	B.n_reg[rd] = n_regs;
	B.assign[rd] = assign;
	if (B.best) B.best[rd] = best;
	B.nhits[rd] = nhits;
	if (B.counts && assign >= 0) {                          // aligner.py:247-263, all three modes
		const int g = B.contig_genome[assign];
		atomicAdd((unsigned long long*)&B.counts[g * 3 + 0], 1ULL);
		atomicAdd((unsigned long long*)&B.counts[g * 3 + 1], (unsigned long long)qlen);
		atomicAdd((unsigned long long*)&B.counts[g * 3 + 2], (unsigned long long)best.mlen);
	}
}

// ---------------------------------------------------------------- gather gated hits -> CSR
__global__ __launch_bounds__(256) void mnc_gather_hits(Batch B, const mnc_hit_t *gated_all,
                                                       const int64_t *hit_off, mnc_hit_t *out)
{
	const uint32_t rd = blockIdx.x * blockDim.x + threadIdx.x;
	if (rd >= B.n_reads) return;
	const int n = B.nhits[rd];
	if (n == 0) return;
	const mnc_hit_t *src = gated_all + B.an_off[rd] / 3;
	mnc_hit_t *dst = out + hit_off[rd];
	for (int i = 0; i < n; ++i) dst[i] = src[i];
}

void launch_regions(const Batch &B, void *regx, uint64_t *k64a, uint64_t *k64b, mnc_hit_t *gated, hipStream_t st)
{
	if (B.n_reads == 0) return;
	hipLaunchKernelGGL(mnc_regions_decide, dim3((B.n_reads + 63) / 64), dim3(64), 0, st, B,
	                   reinterpret_cast<RegX*>(regx), k64a, k64b, gated);
}

void launch_gather_hits(const Batch &B, const mnc_hit_t *gated, const int64_t *hit_off, mnc_hit_t *out, hipStream_t st)
{
	if (B.n_reads == 0) return;
	hipLaunchKernelGGL(mnc_gather_hits, dim3((B.n_reads + 255) / 256), dim3(256), 0, st, B, gated, hit_off, out);
}

} // namespace mnc
