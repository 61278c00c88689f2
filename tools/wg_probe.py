"""The literal kernel's long calls alone on the chip: reads whose flanks are junk (extensions of `flank` query bases
that run to their Z-drop), classified with the long calls on one wave (0x20), four waves on the workspace (0x80), and with the cells in registers
on sixteen waves (0x8), on four waves x four cells (0x1), or whichever the call count picks (0).  python tools/wg_probe.py [flank] [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
flank = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
modes = [int(x, 0) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 0x8, 0x1, 0x80, 0x20]
names, seqs = synth.genome_set(2, min_len=400_000, max_len=500_000, diverged_half=False)
index = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(index, 0)
rng = np.random.default_rng(5)
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
reads = []
for i in range(n):
    s = int(rng.integers(0, 300_000))
    reads.append(np.concatenate([ACGT[rng.integers(0, 4, flank)], seqs[i & 1][s:s + 3000], ACGT[rng.integers(0, 4, flank)]]))
offsets = np.zeros(n + 1, dtype=np.int64)
offsets[1:] = np.cumsum([len(r) for r in reads])
bases = np.concatenate(reads)
ref = None
for mode in modes:
    eng.set_debug(mode | 0x10000)
    eng.classify(bases, offsets, 0)
    t = time.perf_counter()
    for _ in range(3):
        a, best, nh = eng.classify(bases, offsets, 0)
    dt = (time.perf_counter() - t) / 3
    regs = eng.dump(_capi.DUMP_REGS, _capi.REG_DTYPE)
    key = (a.tobytes(), regs.tobytes())
    if ref is None:
        ref = key
    c = eng.counters()
    print(f"mode {mode:#x}: {dt * 1e3:8.2f} ms per batch; big {c['dp_literal_big']} mid {c['dp_literal_mid']} lext {c['dp_long_extensions']}; same as first: {key == ref}")
    import ctypes
    L = _capi.lib()
    if hasattr(L, "mnc_debug_wg_cycles"):                       # a build with -DMNC_WG_TIMING (k_align.hip)
        out = (ctypes.c_longlong * 8)()
        L.mnc_debug_wg_cycles(out, 1)
        eng.classify(bases, offsets, 0)
        L.mnc_debug_wg_cycles(out, 0)
        v = list(out)
        print("   thread-0 cycles: lds %d, steps %d, walk %d, calls %d, anti-diagonals %d, test_zdrop %d, call set-up %d -> %.0f cycles per anti-diagonal, %.0f walk cycles per call" %
              (v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[1] / max(v[4], 1), v[2] / max(v[3], 1)))
