"""The gap-filling tiers planned by the region's anchor density against the fixed thresholds (debug bits 8-15 / 24-30 = 32 / 34),
one engine, same batches: time per batch, tier counts, and that nothing in the results moves.  python tools/ab_fill_pred.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
names, seqs = synth.genome_set(20)
index = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(index, 0)
FIXED = 32 << 8 | 34 << 24
RATES = (dict(sub=120, ins=90, dele=90), dict(sub=280, ins=210, dele=210), dict(), dict(sub=500, ins=400, dele=400), dict(sub=700, ins=450, dele=450))
for label, n, kw in (("3 %", 30000, RATES[0]), ("7 %", 30000, RATES[1]), ("10 %", 100000, RATES[2]), ("13 %", 30000, RATES[3]), ("16 %", 30000, RATES[4]),
                     ("mixed 3-16 %", 30000, None)):
    if kw is None:                                                # every fifth of the batch at its own rate
        parts = [synth.reads(seqs, n // 5, 5000, seed=950 + i, **k) for i, k in enumerate(RATES)]
        bases = np.concatenate([p[0] for p in parts])
        offsets = np.arange(len(bases) // 5000 + 1, dtype=np.int64) * 5000
    else:
        bases, offsets, _ = synth.reads(seqs, n, 5000, seed=940, **kw)
    d_b, d_o = _capi.pinned_array(bases), _capi.pinned_array(offsets)
    out = {}
    for mode, dbg in (("density", 0), ("fixed", FIXED), ("density", 0), ("fixed", FIXED)):
        eng.set_debug(dbg)
        eng.classify(d_b, d_o, 60)
        t = time.perf_counter()
        for _ in range(3):
            a, best, nh = eng.classify(d_b, d_o, 60)
        dt = (time.perf_counter() - t) / 3
        regs = eng.dump(_capi.DUMP_REGS, _capi.REG_DTYPE)
        c = eng.counters()
        key = (a.tobytes(), nh.tobytes(), regs.tobytes())
        out.setdefault(mode, []).append((dt, key, {k: c[k] for k in ("dp_fill_tier1", "dp_fill_tier_mid", "dp_fill_tier2", "dp_fill_tier3", "dp_literal_mid")}))
    same = all(k[1] == out["density"][0][1] for v in out.values() for k in v)
    print(f"{label:13s} {len(offsets) - 1} reads: density {1e3 * min(x[0] for x in out['density']):7.2f} ms, fixed {1e3 * min(x[0] for x in out['fixed']):7.2f} ms; same results {same}; "
          f"tiers' lists (32 / 42 / 64 / 128 cells, literal) density {list(out['density'][0][2].values())} fixed {list(out['fixed'][0][2].values())}")
