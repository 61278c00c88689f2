"""Experiment: E engines on the same device, each on its own host thread, each classifying its own resident batch K times;
whole-job reads/s against one engine doing the same number of batches."""
import os, sys, time, threading
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monica_amd import _capi, synth
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
R = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
dev = torch.device("cuda:0")
names, seqs = synth.genome_set(20, min_len=2_000_000, max_len=7_000_000)
index = _capi.Index.from_seqs(names, seqs)
ng = index.info().n_genomes
class Slot:
    def __init__(self, k):
        self.bases, self.offsets, _ = synth.reads(seqs, R, 5000, seed=synth.SEED_READS + 2 + k)
        self.eng = _capi.Engine(index, 0)
        self.eng.set_contract(_capi.CONTRACT_DP)
        self.d_bases = torch.from_numpy(self.bases).to(dev); self.d_off = torch.from_numpy(self.offsets).to(dev)
        self.d_assign = torch.empty(R, dtype=torch.int32, device=dev); self.d_best = torch.zeros(R * 4, dtype=torch.int32, device=dev)
        self.d_nhits = torch.zeros(R, dtype=torch.int32, device=dev); self.d_counts = torch.zeros(ng * 3, dtype=torch.int64, device=dev)
        self.total = int(self.offsets[-1])
    def step(self):
        self.eng.classify_device(self.d_bases.data_ptr(), self.d_off.data_ptr(), R, self.total, 5000, 60, self.d_assign.data_ptr(),
                                 self.d_best.data_ptr(), self.d_nhits.data_ptr(), self.d_counts.data_ptr())
        self.eng.sync()
slots = [Slot(k) for k in range(E)]
torch.cuda.synchronize()
for s in slots:
    s.step(); s.step()
ref = [s.d_assign.cpu().numpy().copy() for s in slots]
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
print("results unchanged:", all(np.array_equal(s.d_assign.cpu().numpy(), r) for s, r in zip(slots, ref)))
