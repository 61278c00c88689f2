"""What a small batch costs: wall time of one call and the stage table (events) at 400 / 1 500 / 3 000 / 6 000 / 12 500 /
30 000 / 100 000 reads of the bench's kind, host buffers in, host arrays out.  python tools/small_batches.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
names, seqs = synth.genome_set(20)
index = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(index, 0)
sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [400, 1500, 3000, 6000, 12500, 30000, 100000]
for n in sizes:
    bases, offsets, truth = synth.reads(seqs, n, 5000, seed=synth.SEED_READS + 2)
    bases, offsets = _capi.pinned_array(bases), _capi.pinned_array(offsets)
    for _ in range(3):
        eng.classify(bases, offsets, 60)
    reps = 20 if n <= 12500 else 5
    eng.set_profiling(False)
    t = time.perf_counter()
    for _ in range(reps):
        eng.classify(bases, offsets, 60)
    plain = (time.perf_counter() - t) / reps
    eng.set_profiling(True); eng.timings(reset=True)
    t = time.perf_counter()
    for _ in range(reps):
        eng.classify(bases, offsets, 60)
    prof = (time.perf_counter() - t) / reps
    tm = eng.timings()
    print(f"{n:7d} reads: {1e3 * plain:7.2f} ms a call ({1e3 * plain / n * 1e3:6.2f} us a read); with stage events {1e3 * prof:7.2f} ms; "
          f"stages {({k: round(v[0] / reps, 2) for k, v in tm.items() if v[1]})}")
