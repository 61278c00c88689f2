"""What the alignment stage's kernel calls look like on the bench batch: lengths of the extensions' query side,
of the gap fillings, and which kernel class each went to (python tools/seg_profile.py [reads])."""
import sys
import numpy as np
sys.path.insert(0, ".")
from monica_amd import _capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
names, seqs = synth.genome_set(20)
idx = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(idx, 0)
bases, offsets, truth = synth.reads(seqs, n, 5000, seed=synth.SEED_READS + 2)
eng.classify(bases, offsets, 60)
segs = eng.dump(_capi.DUMP_SEGS, _capi.SEG_DTYPE)
ext = segs[segs["kind"] != 1]
gap = segs[segs["kind"] == 1]
print("segments", len(segs), "extensions", len(ext), "gap fillings", len(gap))
q = ext["qlen"]
print("extension qlen percentiles 10/25/50/75/90/99:", np.percentile(q, [10, 25, 50, 75, 90, 99]).tolist())
edges = [0, 8, 16, 24, 32, 42, 48, 64, 84, 96, 128, 192, 256, 512, 100000]
h, _ = np.histogram(q, bins=edges)
cells = [(ext["qlen"][(q > lo) & (q <= hi)].astype(np.int64) * (ext["tlen"] + ext["qlen"] - 1)[(q > lo) & (q <= hi)]).sum() for lo, hi in zip(edges[:-1], edges[1:])]
for (lo, hi), c, w in zip(zip(edges[:-1], edges[1:]), h, cells):
    print(f"  qlen ({lo:4d}, {hi:6d}]  {c:8d} calls  {w / max(sum(cells), 1):6.3f} of the cell-steps")
print("extension tlen / qlen median:", float(np.median(ext["tlen"] / np.maximum(ext["qlen"], 1))))
print("extension max_q + 1 == qlen (best cell in the query's last row):", float(((ext["max_q"] + 1) == ext["qlen"]).mean()))
print("extension best cell: max_t - max_q percentiles 1/10/50/90/99:", np.percentile(ext["max_t"] - ext["max_q"], [1, 10, 50, 90, 99]).tolist())
g = gap["tlen"].astype(np.int64) + gap["qlen"]
print("gap filling tlen+qlen percentiles 10/50/90/99:", np.percentile(g, [10, 50, 90, 99]).tolist())
print("gap filling |tlen-qlen| percentiles 50/90/99:", np.percentile(np.abs(gap["tlen"] - gap["qlen"]), [50, 90, 99]).tolist())
print("gap filling score / min(tlen, qlen) percentiles 1/10/50/90:", np.percentile(gap["score"] / np.maximum(np.minimum(gap["tlen"], gap["qlen"]), 1), [1, 10, 50, 90]).tolist())
vals, cnts = np.unique(segs["big"], return_counts=True)
print("kernel class (4 + tier):", dict(zip(vals.tolist(), cnts.tolist())))
