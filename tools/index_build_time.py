"""How long an index part takes from contig bases to a resident, probe-ready table (python tools/index_build_time.py [genomes]):
the host builder and the device builder (sketch + sort on the GPU), each followed by the upload + perfect-hash construction
that the first engine on the part pays.  The reference does this once per database part (aligner.py:45 `mp.Aligner(fn_idx_in=
database, fn_idx_out=index)`) and loads the part at every pass (aligner.py:59)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from monica_amd import _capi, synth

g = int(sys.argv[1]) if len(sys.argv) > 1 else 20
names, seqs = synth.genome_set(g)
mbp = sum(len(s) for s in seqs) / 1e6
import torch
torch.zeros(1, device="cuda:0")                      # the context exists (its creation is not the builder's)
_capi.Index.from_seqs(names[:2], seqs[:2], device=0)  # code objects loaded


def timed(f):
    t0 = time.perf_counter()
    r = f()
    return r, time.perf_counter() - t0


for trial in range(2):
    ih, th = timed(lambda: _capi.Index.from_seqs(names, seqs))
    eh, tuh = timed(lambda: _capi.Engine(ih, 0))
    del eh
    idv, td = timed(lambda: _capi.Index.from_seqs(names, seqs, device=0))
    ed, tud = timed(lambda: _capi.Engine(idv, 0))
    del ed
    same = all(np.array_equal(a, b) for a, b in zip(ih.dump(), idv.dump()))
    print(f"{g} genomes, {mbp:.1f} Mbp: host build {th:.3f} s + resident {tuh:.3f} s; device build {td:.3f} s + resident {tud:.3f} s; "
          f"same index: {same}")
    del ih, idv
