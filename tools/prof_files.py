"""Where the wall time of multi_threaded_aligner goes on a 1 GB FASTQ (second call: index and engine cached)."""
import cProfile, pstats, os, sys, tempfile, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monica_amd import _capi, synth
from monica_amd import aligner as al
names, seqs = synth.genome_set(20, min_len=2_000_000, max_len=7_000_000)
work = tempfile.mkdtemp(prefix="mnc_files_")
query, out = os.path.join(work, "query"), os.path.join(work, "out")
os.makedirs(query), os.makedirs(out)
idx_path = os.path.join(work, "index1.mmi")
_capi.Index.from_seqs(names, seqs).save(idx_path)
bases, offsets, truth = synth.reads(seqs, 100000, 5000, seed=synth.SEED_READS + 2)
fq = os.path.join(query, "sample.fastq")
cwd = os.getcwd()
for it in range(3):
    synth.write_fastq(fq, bases, offsets)
    t0 = time.perf_counter()
    if it == 2:
        pr = cProfile.Profile(); pr.enable()
    al.multi_threaded_aligner(query, [idx_path], mode="basic", n_threads=1, output_folder=out)
    if it == 2:
        pr.disable()
    print("call", it, "wall", round(time.perf_counter() - t0, 3), {k: round(v, 3) for k, v in al.TIMINGS.get("sample", {}).items()})
    al.TIMINGS.clear()
    os.chdir(cwd)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
shutil.rmtree(work, ignore_errors=True)
