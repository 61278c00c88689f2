"""Stage times of one batch with a long-tailed read length distribution (nanopore-like: log-normal around 6 kb, a few per
cent beyond 20 kb), and -- with `check` -- the same batch against the CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
check = len(sys.argv) > 2 and sys.argv[2] == "check"
names, seqs = synth.genome_set(20, min_len=2_000_000, max_len=7_000_000)
index = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(index, 0)
rng = np.random.default_rng(5)
lens = np.clip(np.exp(rng.normal(np.log(6000), 0.7, n)).astype(np.int64), 300, 60000)
parts, total = [], 0
for L in np.unique(lens // 2000):                      # reads in 2 kb length classes, each class from the generator
    k = int((lens // 2000 == L).sum())
    b, o, _ = synth.reads(seqs, k, int(L) * 2000 + 1000, seed=100 + int(L))
    parts.append((b, o))
bases = np.concatenate([p[0] for p in parts])
offsets = np.concatenate([[0]] + [p[1][1:] + sum(len(q[0]) for q in parts[:i]) for i, p in enumerate(parts)]).astype(np.int64)
perm_note = "sorted by length class"
print("reads %d, bases %.1f M, longest %d, mean %.0f (%s)" % (len(offsets) - 1, len(bases) / 1e6, int(np.diff(offsets).max()), np.diff(offsets).mean(), perm_note))
eng.classify(bases, offsets, 60)
eng.set_profiling(True); eng.timings(reset=True)
t = time.time(); a, best, nh = eng.classify(bases, offsets, 60); dt = time.time() - t
tm = eng.timings()
print("%.3f s = %.0f reads/s, %.1f Mbases/s; mapped %d" % (dt, (len(offsets) - 1) / dt, len(bases) / dt / 1e6, int((a >= 0).sum())))
print({k: round(v[0], 2) for k, v in tm.items() if v[1]})
if check:
    from oracle import pyoracle
    oidx = pyoracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    oidx.opt.cigar = 1
    oa, ob, onh, _ = oidx.classify(bases, offsets, 60, n_threads=16)
    print("equal to the oracle:", np.array_equal(a, oa) and np.array_equal(nh, onh) and all(np.array_equal(best[k], ob[k]) for k in _capi.HIT_DTYPE.names))
