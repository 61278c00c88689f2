"""The literal kernel's long calls of one batch: sizes, anti-diagonals, how they ended.  python tools/big_calls.py sub ins del [n] [debug]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
sub, ins, dele = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 30000
dbg = int(sys.argv[5], 0) if len(sys.argv) > 5 else 0
names, seqs = synth.genome_set(20, min_len=2_000_000, max_len=7_000_000)
index = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(index, 0)
bases, offsets, truth = synth.reads(seqs, n, 5000, seed=778, sub=sub, ins=ins, dele=dele)
eng.set_debug(dbg)
eng.classify(bases, offsets, 60)
segs = eng.dump(_capi.DUMP_SEGS, _capi.SEG_DTYPE)
w = np.where(segs["w"] < 0, np.maximum(segs["tlen"], segs["qlen"]), segs["w"])
width = np.minimum(np.minimum(segs["qlen"], segs["tlen"]), w + 1)
ncw = ((width + 15) // 16 + 1) * 16
pbytes = (segs["qlen"].astype(np.int64) + segs["tlen"] - 1) * ncw
big = pbytes > (1 << 20)
print("segments", len(segs), "with more than 1 MB of direction bytes", int(big.sum()))
b = segs[big]
steps_full = b["qlen"] + b["tlen"] - 1
# anti-diagonals actually run: to the Z-drop (max's anti-diagonal + what it takes to drop) or to the end
order = np.argsort(-(steps_full.astype(np.int64) * width[big]))
print("kind qlen tlen w width steps zdropped max max_t max_q n_cigar")
for i in order[:25]:
    s = b[i]
    print(s["kind"], s["qlen"], s["tlen"], s["w"], int(width[big][i]), int(steps_full[i]), s["zdropped"], s["max"], s["max_t"], s["max_q"], s["n_cigar"])
print("sum of steps x width over the big calls: %.3g cells; longest %d steps" % (float((steps_full.astype(np.int64) * width[big]).sum()), int(steps_full.max()) if len(b) else 0))
for k in (0, 1, 2):
    print("kind", k, int((b["kind"] == k).sum()), "zdropped", int(((b["kind"] == k) & (b["zdropped"] != 0)).sum()))
