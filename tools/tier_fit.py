"""Which band tier proves each gap filling of the bench batch, against the tier the planner sends it to first.
A tier of c cells proves a segment iff its banded score S (the final score, whatever tier found it) is strictly above the bound
U(c) of every path that leaves the band (k_fill.hip: dp_band_bound).  Cost of a try = anti-diagonals x cells / 32.
python tools/tier_fit.py [reads]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from monica_amd import _capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
names, seqs = synth.genome_set(20)
idx = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(idx, 0)
bases, offsets, truth = synth.reads(seqs, n, 5000, seed=synth.SEED_READS + 2)
eng.classify(bases, offsets, 60)
segs = eng.dump(_capi.DUMP_SEGS, _capi.SEG_DTYPE)
g = segs[(segs["kind"] == 1) & (segs["tlen"] <= 511) & (segs["qlen"] <= 511) & (segs["tlen"] >= 1) & (segs["qlen"] >= 1)]
N, M, S = g["tlen"].astype(np.int64), g["qlen"].astype(np.int64), g["score"].astype(np.int64)
a, q, e, q2, e2 = 2, 4, 2, 24, 1
def gap(l): return np.minimum(q + e * l, q2 + e2 * l)
def bound(cells):
    d = N - M; ad = np.abs(d)
    b = (2 * cells - 2 - ad) // 2
    kmin = np.where(d < 0, d, 0) - b
    kmin = kmin - (kmin & 1)
    kmax = kmin + 2 * cells - 1
    U = np.full(len(N), -(1 << 28), dtype=np.int64)
    D0, I0 = kmax + 1, kmax + 1 - d
    ok = (N - D0 >= 0) & (M - I0 >= 0)
    U = np.where(ok, np.maximum(U, a * (N - D0) - gap(D0) - gap(I0)), U)
    I1 = 1 - kmin; D1 = I1 + d
    ok = (M - I1 >= 0) & (N - D1 >= 0)
    U = np.where(ok, np.maximum(U, a * (M - I1) - gap(I1) - gap(D1)), U)
    return U, b
tiers = [32, 42, 64, 128]
unit = {32: 1.0, 42: 4.0 / 3.0, 64: 2.0, 128: 4.0}
proves = {}
for c in tiers:
    U, b = bound(c)
    proves[c] = (S > U) & (b >= 8)
steps = N + M - 1
first = np.full(len(N), 999)
for c in reversed(tiers):
    first = np.where(proves[c], c, first)
print("gap fillings", len(N), "proven first by tier:", {c: int((first == c).sum()) for c in tiers + [999]})
planned = g["big"] - 4        # 1 -> 32 cells (tier code 1 is stored as 3 + 1), 18 -> 42, 2 -> 64, 6 -> 128
code = g["big"]
ptier = np.select([code == 4, code == 21, code == 5, code == 9], [32, 42, 64, 128], default=0)
print("planned first tier:", {c: int((ptier == c).sum()) for c in tiers + [0]})
def cost_of(start):
    """total cost when segment i starts at tier start[i] and moves up until proven"""
    total = 0.0
    for c in tiers:
        tried = (start <= c) & (c <= np.maximum(first, start))        # from its first tier up to the one that proves it
        total += float((steps[tried] * unit[c]).sum())
    return total
cur = cost_of(np.where(ptier == 0, 999, ptier))
best = cost_of(np.where(first == 999, 999, first))
print("cost (anti-diagonals x cells / 32): planner %.4g, oracle (every segment straight to its proving tier) %.4g -> %.1f %% above" % (cur, best, 100 * (cur / best - 1)))
for c in tiers:
    sel = ptier == c
    print(f"  planned {c}: proven there {int((first[sel] == c).sum())}, an earlier tier would have done {int((first[sel] < c).sum())}, handed up {int((first[sel] > c).sum())}")
# a sweep of the two thresholds the planner uses (score per base the read is expected to reach)
mn = np.minimum(N, M)
for p1 in (28, 30, 32, 34, 36):
    for pm in (30, 32, 34, 36, 38):
        U32, b32 = bound(32); U42, b42 = bound(42); U64, b64 = bound(64); U128, b128 = bound(128)
        t = np.full(len(N), 128)
        ok64 = (b64 >= 8) & ~(U64 * 25 > mn * 32); t = np.where(ok64, 64, t)
        ok42 = (b42 >= 8) & ~(U42 * 25 > mn * pm) & ((2 * 42 - 2 - np.abs(N - M)) // 2 >= 12); t = np.where(ok42, 42, t)
        ok32 = ((62 - np.abs(N - M)) // 2 >= 12) & ~(U32 * 25 > mn * p1); t = np.where(ok32, 32, t)
        print(f"  thresholds {p1}/{pm}: cost {cost_of(t) / best:.4f} x oracle")
