"""Stage times of ONE block of BASELINE config 4 (python tools/shard_block_profile.py): 100 000 reads drawn from all 500
genomes classified against one index part of 62 genomes -- three quarters of the reads have no genome in the part."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth, dist as mdist

names, seqs = synth.genome_set(500)
lo, hi = mdist.shard_bounds(500, 0, 8)
idx = _capi.Index.from_seqs(names[lo:hi], seqs[lo:hi], device=0)
eng = _capi.Engine(idx, 0)
bases, offsets, truth = synth.reads(seqs, 100_000, 5000, seed=synth.SEED_READS + 4)
eng.classify(bases, offsets, 60)
eng.set_profiling(True); eng.timings(reset=True)
t = time.perf_counter(); a, best, nh = eng.classify(bases, offsets, 60); dt = time.perf_counter() - t
tm = eng.timings()
print("reads with their genome (or its diverged copy) in the part:", int(((truth >= lo) & (truth < hi)).sum()), "mapped", int((a >= 0).sum()), "call ms", round(dt * 1e3, 1))
print({k: round(v[0], 2) for k, v in tm.items() if v[1]})
print(eng.counters())
# ... and its alignment kernels one at a time (debug 0x10000: one stream, per-kernel timers)
eng.set_debug(0x10000)
eng.classify(bases, offsets, 60)
eng.timings(reset=True)
eng.classify(bases, offsets, 60)
tm = eng.timings()
print("one kernel at a time:", {k: round(v[0], 2) for k, v in tm.items() if v[1] and k.startswith("dp_")})
