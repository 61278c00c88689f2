"""Soak: the BASELINE config-2 batch classified repeatedly on one engine; decisions, best hits and gated hit lists
of every pass must equal the first pass's."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
names, seqs = synth.genome_set(20, min_len=2_000_000, max_len=7_000_000)
index = _capi.Index.from_seqs(names, seqs)
bases, offsets, truth = synth.reads(seqs, 100000, 5000, seed=synth.SEED_READS + 2)
eng = _capi.Engine(index, 0)
first = None
t0 = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for it in range(n):
    a, best, nh = eng.classify(bases, offsets, 60)
    off, hits = eng.fetch_hits()
    cur = (a.copy(), best.copy(), nh.copy(), off.copy(), hits.copy())
    if first is None:
        first = cur
    else:
        for x, y, nm in zip(cur, first, ("assign", "best", "nhits", "hit_off", "hits")):
            if not np.array_equal(x, y):
                print("pass", it, nm, "differs at", np.flatnonzero(x != y)[:5] if x.dtype.names is None else "records")
print("passes", n, "all equal to the first" , "in %.1f s" % (time.time() - t0))
