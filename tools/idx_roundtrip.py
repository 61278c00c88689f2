"""An index built on the device from a FASTA file, saved as .mmi and loaded again, `n` times in one process: every
round must load and hold the arrays of the host builder (what `aligner.indexer` + `index_loader` do per database part).
python tools/idx_roundtrip.py [n]"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from monica_amd import _capi, synth
import util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
names, seqs = util.small_genomes(4, 120_000, 150_000)
tmp = tempfile.mkdtemp()
fa = os.path.join(tmp, "database0.fna.gz")
synth.write_fasta(fa, names, seqs)
host = _capi.Index.build(fa, None, 15, 10, device=None)
want = host.dump()
bad = 0
garbage = os.environ.get("IDX_GARBAGE")                              # dirty the HBM the build is about to be given
if garbage:
    import torch
for i in range(n):
    if garbage:
        x = torch.randint(0, 256, (1 << 30,), dtype=torch.uint8, device="cuda:0") if i & 1 else torch.full((1 << 30,), 0xFF if i & 2 else 0xA5, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        del x
        torch.cuda.empty_cache()
    path = os.path.join(tmp, f"index{i}.mmi")
    try:
        built = _capi.Index.build(fa, None, 15, 10, device=0)
        built.save(path, mmi=True)
        loaded = _capi.Index.load(path)
        got = loaded.dump()
        if loaded.mid_occ != host.mid_occ or not (np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])):
            raise RuntimeError(f"loaded index differs: {len(got[0])} occurrences against {len(want[0])}")
    except Exception as e:                                           # noqa: BLE001 -- a diagnosis tool: say which and go on
        bad += 1
        print(f"round {i}: {type(e).__name__}: {e}", flush=True)
print(f"{n} rounds, {bad} failed")
sys.exit(1 if bad else 0)
