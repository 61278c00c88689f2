"""Which stage of a batch pays when the next batch's host-to-device copy starts `delay` ms into it
(python tools/e2e_delay.py): per-stage times of the running batch from the engine's own stage timers."""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
from monica_amd import _capi, synth

n = 100_000
names, seqs = synth.genome_set(20)
idx = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(idx, 0)
bases, offsets, _ = synth.reads(seqs, n, 5000, seed=synth.SEED_READS + 2)
bufs = [_capi.pinned_array(bases), _capi.pinned_array(bases)]
eng.classify(bufs[0], offsets, 60)
eng.set_profiling(True)


def announce(buf, t_ref, delay, box):
    if delay:
        time.sleep(delay)                                   # (a spinning thread would hold the interpreter lock and delay the CALLER)
    while not eng.prefetch_ptr(buf.ctypes.data, offsets.ctypes.data, n):
        time.sleep(0.0002)
    box.append((time.perf_counter() - t_ref) * 1e3)


for delay in (0.0, 0.001, 0.002, 0.003, 0.004, 0.005, 0.007, 0.010, 0.020):
    box = []
    announce(bufs[0], time.perf_counter(), 0, box)
    rows = []
    for k in range(5):
        t0 = time.perf_counter()
        th = threading.Thread(target=announce, args=(bufs[(k + 1) & 1], t0, delay, box))
        th.start()
        eng.classify_ptr(bufs[k & 1].ctypes.data, offsets.ctypes.data, n, 60)
        t1 = time.perf_counter()
        th.join()
        tm = eng.timings(reset=True)
        rows.append(((t1 - t0) * 1e3, box[-1], {k_: round(v[0], 2) for k_, v in tm.items() if v[1] and v[0] >= 0.05}))
    eng.classify_ptr(bufs[1].ctypes.data, offsets.ctypes.data, n, 60)
    r = rows[-1]
    print("delay ms", delay * 1e3, "call ms", [round(x[0], 2) for x in rows], "copy issued at ms", round(r[1], 2))
    print("   stages", r[2])
