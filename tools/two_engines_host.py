"""Experiment: the PCIe-inclusive rate with E engines on the same device, each on its own host thread with its own page-locked
batch (mnc_classify_batch: H2D, kernels, D2H) -- does one engine's copy hide behind the other's kernels?"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monica_amd import _capi, synth
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
R = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
names, seqs = synth.genome_set(20, min_len=2_000_000, max_len=7_000_000)
index = _capi.Index.from_seqs(names, seqs)
class Slot:
    def __init__(self, k):
        bases, self.offsets, _ = synth.reads(seqs, R, 5000, seed=synth.SEED_READS + 2 + k)
        self.hb = _capi.pinned_array(bases)
        self.eng = _capi.Engine(index, 0)
        self.out = None
    def step(self):
        self.out = self.eng.classify(self.hb, self.offsets, 60)
slots = [Slot(k) for k in range(E)]
for s in slots:
    s.step(); s.step()
ref = [s.out[0].copy() for s in slots]
t = time.perf_counter()
for _ in range(K): slots[0].step()
one = time.perf_counter() - t
print("one engine: %.2f ms per batch, %.2f M reads/s" % (one / K * 1e3, R * K / one / 1e6))
def run(s):
    for _ in range(K): s.step()
th = [threading.Thread(target=run, args=(s,)) for s in slots]
t = time.perf_counter()
for x in th: x.start()
for x in th: x.join()
many = time.perf_counter() - t
print("%d engines: %.2f ms per batch, %.2f M reads/s" % (E, many / (K * E) * 1e3, R * K * E / many / 1e6))
print("results unchanged:", all(np.array_equal(s.out[0], r) for s, r in zip(slots, ref)))
