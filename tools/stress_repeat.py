"""Run-to-run determinism of the alignment stage: the same batches classified again and again; any field of
any region (or CIGAR) that differs from the first run is reported with the read it belongs to."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from monica_amd import _capi as capi, synth
import util
from test_gpu_fuzz import _random_world, _random_reads

worlds = []
for k in (0, 3, 7):
    rng = np.random.default_rng(0xF00D + k)
    names, seqs = _random_world(rng, k)
    idx = capi.Index.from_seqs(names, seqs)
    eng = capi.Engine(idx, 0)
    reads = _random_reads(rng, seqs, 70)
    bases, offsets = util.pack_reads(reads)
    worlds.append((k, idx, eng, bases, offsets))
# the noisy reads of test_gpu_dp
names, seqs = util.small_genomes()
idx = capi.Index.from_seqs(names, seqs)
eng = capi.Engine(idx, 0)
b, o, _ = synth.reads(seqs, 60, 3000, seed=31, sub=800, ins=600, dele=600)
worlds.append(("noisy20", idx, eng, b, o))
b, o, _ = synth.reads(seqs, 300, 5000, seed=0x5EED + 1)
worlds.append(("noisy10", idx, capi.Engine(idx, 0), b, o))

for w in worlds:
    w[2].set_debug(int(os.environ.get('MNC_DEBUG', '0'), 0))
first = {}
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
    for (k, idx, eng, bases, offsets) in worlds:
        a, best, nh = eng.classify(bases, offsets, 0)
        regs = eng.dump(capi.DUMP_REGS, capi.REG_DTYPE)
        reg_off = eng.dump(capi.DUMP_REG_OFFSETS, np.int64)
        cig = eng.dump(capi.DUMP_CIGARS, np.uint32)
        if k not in first:
            first[k] = (regs.copy(), reg_off.copy(), cig.copy())
            continue
        r0, o0, c0 = first[k]
        if len(regs) != len(r0) or not np.array_equal(reg_off, o0):
            bad += 1
            d = np.flatnonzero(np.diff(reg_off) != np.diff(o0))
            print("iter", it, "world", k, "region counts differ at reads", d[:5], np.diff(reg_off)[d[:5]], np.diff(o0)[d[:5]], "len", offsets[d[0] + 1] - offsets[d[0]])
            continue
        for name in capi.REG_DTYPE.names:
            if not np.array_equal(regs[name], r0[name]):
                bad += 1
                i = np.flatnonzero(regs[name] != r0[name])
                rd = np.searchsorted(reg_off, i[0], side="right") - 1
                print("iter", it, "world", k, "field", name, "region", i[:4], "read", rd, "len", offsets[rd + 1] - offsets[rd], regs[name][i[:4]], r0[name][i[:4]], "qs/qe", regs["qs"][i[0]], regs["qe"][i[0]], "rs/re", regs["rs"][i[0]], regs["re"][i[0]])
                break
print("mismatching runs:", bad)
