"""Parity of short reads (the extension kernels' share is largest there): N reads of 600 .. 1 500 bases at 5 / 12 % errors against the
CPU oracle -- decisions, best hits and gated hit counts read for read."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monica_amd import _capi, synth
from oracle import pyoracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
names, seqs = synth.genome_set(20, min_len=2_000_000, max_len=7_000_000)
index = _capi.Index.from_seqs(names, seqs)
oidx = pyoracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
oidx.opt.cigar = 1
eng = _capi.Engine(index, 0)
for rl in (600, 1000, 1500):
    for label, kw in (("5% errors", dict(sub=200, ins=150, dele=150)), ("12% errors", dict(sub=500, ins=350, dele=350))):
        bases, offsets, truth = synth.reads(seqs, n, rl, seed=900 + rl, **kw)
        t = time.time(); a, best, nh = eng.classify(bases, offsets, 60); tg = time.time() - t
        t = time.time(); oa, ob, onh, _ = oidx.classify(bases, offsets, 60, n_threads=16); tc = time.time() - t
        same = np.array_equal(a, oa) and np.array_equal(nh, onh) and all(np.array_equal(best[k], ob[k]) for k in _capi.HIT_DTYPE.names)
        print(rl, "bases", label, "reads", n, "equal", same, "mapped", int((a >= 0).sum()), "gpu %.2f s cpu %.1f s" % (tg, tc))
