"""One-off wide parity sweep: N reads of the BASELINE config-2 workload (and a noisier mix) classified on the GPU with
base-level alignment and by the CPU oracle; decisions, best hits and gated hit counts compared read for read."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monica_amd import _capi, synth
from oracle import pyoracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
model = sys.argv[2] if len(sys.argv) > 2 else "iid"            # "repeats": synth.genome_set_repeats (operons, insertion sequences, shared stretches)
rates = sys.argv[3:]                                           # e.g. "20": a 20 % row as well
seed_add = int(os.environ.get("SWEEP_SEED", "0"))              # other reads of the same kinds
read_len = int(os.environ.get("SWEEP_LEN", "5000"))
names, seqs = (synth.genome_set_repeats if model == "repeats" else synth.genome_set)(20, min_len=2_000_000, max_len=7_000_000)
index = _capi.Index.from_seqs(names, seqs)
oidx = pyoracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
oidx.opt.cigar = 1
eng = _capi.Engine(index, 0)
rows = [("10% errors", dict(seed=777)), ("16% errors", dict(seed=778, sub=700, ins=450, dele=450)), ("3% errors", dict(seed=779, sub=120, ins=90, dele=90))]
if "20" in rates:
    rows.append(("20% errors", dict(seed=780, sub=800, ins=600, dele=600)))
print("genome model", model, "mid_occ", index.mid_occ, "read length", read_len, "seeds +", seed_add)
for label, kw in rows:
    kw = dict(kw, seed=kw["seed"] + seed_add)
    bases, offsets, truth = synth.reads(seqs, n, read_len, **kw)
    t = time.time(); a, best, nh = eng.classify(bases, offsets, 60); tg = time.time() - t
    t = time.time(); oa, ob, onh, _ = oidx.classify(bases, offsets, 60, n_threads=16); tc = time.time() - t
    same = np.array_equal(a, oa) and np.array_equal(nh, onh) and all(np.array_equal(best[k], ob[k]) for k in _capi.HIT_DTYPE.names)
    print(label, "reads", n, "equal", same, "mapped", int((a >= 0).sum()), "gpu %.2f s cpu %.1f s" % (tg, tc))
