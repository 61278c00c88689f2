"""What a region's anchor density says about the score its gap fillings reach: per error rate, the regions' density ratio
rho = cnt * (w + 1) / (2 * query span) (1 at no errors, (1 - eps)^k in expectation) and the gap fillings' score per base
of the shorter side.  python tools/pred_fit.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
names, seqs = synth.genome_set(20)
index = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(index, 0)
for sub, ins, dele in ((120, 90, 90), (280, 210, 210), (400, 300, 300), (500, 400, 400), (700, 450, 450)):
    bases, offsets, truth = synth.reads(seqs, 20000, 5000, seed=901, sub=sub, ins=ins, dele=dele)
    eng.classify(bases, offsets, 0)
    regs = eng.dump(_capi.DUMP_REGS, _capi.REG_DTYPE)
    segs = eng.dump(_capi.DUMP_SEGS, _capi.SEG_DTYPE)
    span = (regs["qe"] - regs["qs"]).astype(np.float64)
    ok = (regs["cnt"] >= 20) & (span > 500)
    rho = regs["cnt"][ok] * 11.0 / (2.0 * span[ok])
    g = segs[(segs["kind"] == 1) & (segs["tlen"] <= 511) & (segs["qlen"] <= 511) & (segs["tlen"] >= 1) & (segs["qlen"] >= 1) & (segs["score"] > -100000)]
    mn = np.minimum(g["tlen"], g["qlen"]).astype(np.float64)
    per_base = g["score"] / mn
    eps = (sub + ins + dele) / 1e4
    print(f"errors {100 * eps:4.1f} %: rho mean {rho.mean():.4f} (sd {rho.std():.4f}; (1 - eps)^15 = {(1 - eps) ** 15:.4f}) -> eps from rho {1 - rho.mean() ** (1 / 15):.4f}; "
          f"score per base of {len(g)} gap fillings: mean {per_base.mean():.3f}, 10th / 50th / 90th percentile {np.percentile(per_base, 10):.3f} / {np.percentile(per_base, 50):.3f} / {np.percentile(per_base, 90):.3f}")
