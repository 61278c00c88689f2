"""One sample file through `multi_threaded_aligner` twice; the second call's pipeline as a table: per batch and stage,
start and end in ms from the call's start.  MONICA_AMD_PIPE_TRACE=1 python tools/files_trace.py [reads]"""
import os, sys, tempfile, time, shutil
os.environ.setdefault("MONICA_AMD_PIPE_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
from monica_amd import aligner as al
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
names, seqs = synth.genome_set(20)
work = tempfile.mkdtemp(prefix="mnc_trace_")
try:
    query, out = os.path.join(work, "query"), os.path.join(work, "out")
    os.makedirs(query), os.makedirs(out)
    idx_path = os.path.join(work, "index1.mmi")
    _capi.Index.from_seqs(names, seqs).save(idx_path)
    bases, offsets, truth = synth.reads(seqs, n, 5000, seed=synth.SEED_READS + 2)
    fq = os.path.join(query, "sample.fastq")
    cwd = os.getcwd()
    for call in range(4):
        synth.write_fastq(fq, bases, offsets)
        al.PIPE_TRACE.clear()
        t0 = time.perf_counter()
        al.multi_threaded_aligner(query, [idx_path], mode="basic", n_threads=1, output_folder=out)
        t1 = time.perf_counter()
        os.chdir(cwd)
        if call:
            tr = sorted(al.PIPE_TRACE, key=lambda x: x[2])
            routes = [x for x in tr if x[0] == "route"]
            busy = sum(b - a for _, _, a, b in routes)
            marks = {x[0]: 1e3 * (x[3] - t0) for x in tr if x[0] in ("loaded", "joined", "closed", "hits_free", "return", "mapped", "update")}
            print(f"call {call}: {1e3 * (t1 - t0):.1f} ms = {n / (t1 - t0) / 1e6:.3f} M reads/s; routing busy {1e3 * busy:.1f} ms from {1e3 * (routes[0][2] - t0):.1f} to "
                  f"{1e3 * (routes[-1][3] - t0):.1f}, idle inside {1e3 * (routes[-1][3] - routes[0][2] - busy):.1f}; marks {({k: round(v, 1) for k, v in marks.items()})}")
            if os.environ.get("VERBOSE"):
                for stage, k, a, b in tr:
                    print(f"  {stage:9s} {k:6d} reads  {1e3 * (a - t0):7.1f} -> {1e3 * (b - t0):7.1f}  ({1e3 * (b - a):5.1f} ms)")
finally:
    shutil.rmtree(work, ignore_errors=True)
