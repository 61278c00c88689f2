"""Stage times and tier counters of one batch at a given error rate (sub, ins, del in 1e-4)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
sub, ins, dele, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 30000
names, seqs = (synth.genome_set_repeats if os.environ.get("MNC_GENOME_MODEL") == "repeats" else synth.genome_set)(20, min_len=2_000_000, max_len=7_000_000)
index = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(index, 0)
if len(sys.argv) > 5:
    eng.set_debug(int(sys.argv[5], 0))
bases, offsets, truth = synth.reads(seqs, n, 5000, seed=778, sub=sub, ins=ins, dele=dele)
eng.classify(bases, offsets, 60)
eng.set_profiling(True); eng.timings(reset=True)
t = time.time(); a, best, nh = eng.classify(bases, offsets, 60); dt = time.time() - t
tm = eng.timings(); c = eng.counters()
print("error %.1f %%: %d reads in %.3f s = %.0f reads/s; mapped %d" % ((sub + ins + dele) / 100, n, dt, n / dt, int((a >= 0).sum())))
print({k: round(v[0], 2) for k, v in tm.items() if v[1]})
print({k: v for k, v in c.items() if k.startswith("dp_")})
import ctypes
L = _capi.lib()
if hasattr(L, "mnc_debug_wg_cycles"):                       # a build with -DMNC_WG_TIMING (k_align.hip)
    out = (ctypes.c_longlong * 8)()
    L.mnc_debug_wg_cycles(out, 1)
    eng.classify(bases, offsets, 60)
    L.mnc_debug_wg_cycles(out, 0)
    v = list(out)
    print("ksw_wg thread-0 cycles: lds %d, steps %d, walk %d, calls %d, anti-diagonals %d, test_zdrop %d, call set-up %d -> %.0f cycles per anti-diagonal" %
          (v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[1] / max(v[4], 1)))
