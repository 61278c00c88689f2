import numpy as np, ctypes as C, sys
sys.path.insert(0,'/root/repo')
from monica_amd import _capi, synth
import torch
names, seqs = synth.genome_set(20, min_len=2_000_000, max_len=7_000_000)
index = _capi.Index.from_seqs(names, seqs)
bases, offsets, truth = synth.reads(seqs, 100000, 5000, seed=synth.SEED_READS + 2)
eng = _capi.Engine(index, 0)
a,b,n = eng.classify(bases, offsets, 60)
c = np.zeros(24, dtype=np.int64)
_capi.check(_capi.lib().mnc_engine_get_counters(eng._h, c.ctypes.data, 24))
print("mid fills, mid ext, mid sum(t+q), mid max len:", c[16:20], " big fills, big ext, sum, max:", c[20:24])
