"""Where the PCIe-inclusive call spends its time beyond the resident one (python tools/e2e_probe.py)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from monica_amd import _capi, synth

n = 100_000
names, seqs = synth.genome_set(20)
idx = _capi.Index.from_seqs(names, seqs)
eng = _capi.Engine(idx, 0)
bases, offsets, _ = synth.reads(seqs, n, 5000, seed=synth.SEED_READS + 2)
hb, hb2 = _capi.pinned_array(bases), _capi.pinned_array(bases)
dev = torch.device("cuda:0")
d_b, d_o = torch.from_numpy(bases).to(dev), torch.from_numpy(offsets).to(dev)
d_a = torch.empty(n, dtype=torch.int32, device=dev)
d_best = torch.zeros(n * 4, dtype=torch.int32, device=dev)
d_nh = torch.zeros(n, dtype=torch.int32, device=dev)


def t(f, k=4):
    f()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    return (time.perf_counter() - t0) / k * 1e3


def resident():
    eng.classify_device(d_b.data_ptr(), d_o.data_ptr(), n, int(offsets[-1]), 5000, 60, d_a.data_ptr(), d_best.data_ptr(), d_nh.data_ptr(), 0)
    eng.sync()


def plain():
    eng.classify_ptr(hb.ctypes.data, offsets.ctypes.data, n, 60)


def copy_only():
    assert eng.prefetch_ptr(hb.ctypes.data, offsets.ctypes.data, n)
    t0 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng.classify_ptr(hb2.ctypes.data, offsets.ctypes.data, n, 60)     # drops the announcement
    return dt


def prefetched_done():
    assert eng.prefetch_ptr(hb.ctypes.data, offsets.ctypes.data, n)
    time.sleep(0.03)                                                   # the copy has finished
    t0 = time.perf_counter()
    eng.classify_ptr(hb.ctypes.data, offsets.ctypes.data, n, 60)
    return time.perf_counter() - t0


print("resident ms", round(t(resident), 2))
print("plain host-buffer call ms", round(t(plain), 2))
print("H2D of the bases alone ms (device sync)", [round(copy_only() * 1e3, 2) for _ in range(3)])
print("call on a batch whose copy has finished ms", [round(prefetched_done() * 1e3, 2) for _ in range(3)])
t0 = time.perf_counter()
a = np.empty(n, dtype=np.int32); b = np.zeros(n, dtype=_capi.HIT_DTYPE); c = np.zeros(n, dtype=np.int32)
print("output arrays ms", round((time.perf_counter() - t0) * 1e3, 3))

# ---- does a host-to-device copy on another stream overlap the batch's kernels at all?
import threading
pin = torch.empty(len(bases), dtype=torch.uint8).pin_memory()
pin.numpy()[:] = bases
dst = torch.empty(len(bases), dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()


def copy_torch():
    with torch.cuda.stream(side):
        dst.copy_(pin, non_blocking=True)


torch.cuda.synchronize()
t0 = time.perf_counter(); copy_torch(); side.synchronize(); print("torch pinned H2D alone ms", round((time.perf_counter() - t0) * 1e3, 2))
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = threading.Thread(target=lambda: (time.sleep(0.005), copy_torch(), side.synchronize()))
    th.start()
    resident()
    t1 = time.perf_counter()
    th.join()
    t2 = time.perf_counter()
    print("resident batch with a torch H2D copy started 5 ms into it: batch done after ms", round((t1 - t0) * 1e3, 2), "copy thread done after ms", round((t2 - t0) * 1e3, 2))
# ... and the engine's own prefetch while a resident batch runs
for trial in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    box = []
    th = threading.Thread(target=lambda: (time.sleep(0.005), box.append(time.perf_counter()), eng.prefetch_ptr(hb.ctypes.data, offsets.ctypes.data, n), box.append(time.perf_counter())))
    th.start()
    resident()
    t1 = time.perf_counter()
    th.join()
    tcall = time.perf_counter()
    eng.classify_ptr(hb.ctypes.data, offsets.ctypes.data, n, 60)
    t2 = time.perf_counter()
    print("resident batch with mnc_engine_prefetch 5 ms into it: batch ms", round((t1 - t0) * 1e3, 2), "prefetch call took ms", round((box[1] - box[0]) * 1e3, 2),
          "the prefetched batch's own call ms", round((t2 - tcall) * 1e3, 2))

# ---- the bench's loop, with per-iteration detail
bufs = [hb, hb2]
eng.classify(bufs[0], offsets, 60)
log = []


def announce(buf, t_ref):
    tries = 0
    while not eng.prefetch_ptr(buf.ctypes.data, offsets.ctypes.data, n):
        tries += 1
        time.sleep(0.0005)
    log.append(("copy issued after ms", round((time.perf_counter() - t_ref) * 1e3, 2), "tries", tries))


announce(bufs[0], time.perf_counter())
for k in range(6):
    t0 = time.perf_counter()
    th = threading.Thread(target=announce, args=(bufs[(k + 1) & 1], t0))
    th.start()
    eng.classify_ptr(bufs[k & 1].ctypes.data, offsets.ctypes.data, n, 60)
    t1 = time.perf_counter()
    th.join()
    print("iteration", k, "call ms", round((t1 - t0) * 1e3, 2), log[-1])
