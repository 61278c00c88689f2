"""GPU test that replays the call sequence of the reference's test/test_aligner.py
(indexer -> multi_threaded_aligner -> alignment_to_data_frame -> normalizer) on a temp
directory with synthetic FASTQ files, single-part and two-part indexes, and checks the
returned dict and every side effect against the CPU oracle."""
import os
import pickle
from collections import Counter

import numpy as np
import pytest

from monica_amd import aligner, synth, fastq
import util

pytestmark = pytest.mark.gpu


def expected_from_oracle(oracle, parts, names, bases, offsets, mode, min_mapq=60):
    """aligner.py:184-263 on top of the oracle's per-part gated hit lists."""
    n = len(offsets) - 1
    per_read = [[] for _ in range(n)]
    for part_names, part_seqs, rid_base in parts:
        oidx = oracle.Index.from_seqs(part_names, [s.tobytes() for s in part_seqs])
        assign, best, nhits, flat = oidx.classify(bases, offsets, min_mapq)
        k = 0
        for r in range(n):
            for h in flat[k:k + nhits[r]]:
                per_read[r].append((part_names[int(h["rid"])], int(h["nm"]), int(h["mlen"])))
            k += nhits[r]
    counts, route = {}, []
    for r in range(n):
        hits = per_read[r]
        if not hits:
            route.append("unmapped")
            continue
        best = hits[0] if len(hits) == 1 else aligner.best_hit(hits)
        if not best:
            route.append("ambiguous")
            continue
        tax, acc = best[0].split(":")
        route.append("mapped:" + tax)
        amount = {"basic": 1, "query_length": int(offsets[r + 1] - offsets[r]), "matching": best[2]}[mode]
        counts.setdefault(tax, Counter()).update({acc: amount})
    return counts, route


def count_records(path):
    return sum(len(b) for b in fastq.read_batches(path)) if os.path.exists(path) and os.path.getsize(path) else 0


@pytest.mark.parametrize("n_parts,mode", [(1, "query_length"), (2, "basic"), (2, "matching")])
def test_reference_call_sequence(oracle, tmp_path, n_parts, mode):
    names, seqs = util.small_genomes(4, 150_000, 200_000)
    dbs = tmp_path / "databases"
    dbs.mkdir()
    if n_parts == 1:
        chunks = [(names, seqs, 0)]
    else:
        # genome i and its 3 %-diverged copy i+2 land in different parts, as monica's
        # chunking by size can do (database.py:70-92)
        chunks = [(names[:2], seqs[:2], 0), (names[2:], seqs[2:], 2)]
    for i, (cn, cs, _) in enumerate(chunks):
        synth.write_fasta(str(dbs / f"database{i}.fna.gz"), cn, cs)
    idx_dir = str(tmp_path / "indexes")
    paths = sorted(aligner.indexer(str(dbs), idx_dir))
    assert [os.path.basename(p) for p in paths] == [f"index{i}.mmi" for i in range(n_parts)]

    query = tmp_path / "query"
    query.mkdir()
    out = tmp_path / "output"
    out.mkdir()
    bases, offsets, truth = synth.reads(seqs, 240, 2500, seed=77)
    extra = util.edge_reads_small(seqs, np.random.default_rng(1))
    eb, eo = util.pack_reads([bases[offsets[i]:offsets[i + 1]] for i in range(240)] + extra)
    half = 130
    synth.write_fastq(str(query / "sampleA.pass.fastq"), eb[:eo[half]], eo[:half + 1], ids=[f"a{i} ch=1" for i in range(half)])
    n_b = len(eo) - 1 - half
    synth.write_fastq(str(query / "sampleB.fastq"), eb[eo[half]:], eo[half:] - eo[half], ids=[f"b{i}" for i in range(n_b)])
    (query / "empty.fastq").write_bytes(b"")

    cwd = os.getcwd()
    try:
        alignment = aligner.multi_threaded_aligner(str(query), paths, mode=mode, n_threads=2,
                                                   focus_species=["Genus1_species1"], output_folder=str(out))
    finally:
        os.chdir(cwd)

    parts = [(cn, cs, b) for cn, cs, b in chunks]
    want_a, route_a = expected_from_oracle(oracle, parts, names, eb[:eo[half]], eo[:half + 1], mode)
    want_b, route_b = expected_from_oracle(oracle, parts, names, eb[eo[half]:], eo[half:] - eo[half], mode)
    assert alignment == {"sampleA": want_a, "sampleB": want_b}
    assert sum(sum(c.values()) for c in want_a.values()) > 0

    # side effects of aligner.py:81-87, 208-211, 242-243, 265, 273, 278, 300
    for sample, route in (("sampleA.pass.fastq", route_a), ("sampleB.fastq", route_b)):
        assert not os.path.exists(query / sample)                       # consumed
        assert count_records(str(query / "mapped" / sample)) == sum(r.startswith("mapped") for r in route)
        assert count_records(str(query / "unmapped" / sample)) == route.count("unmapped")
        assert count_records(str(query / "ambiguous" / sample)) == route.count("ambiguous")
        assert count_records(str(query / "focus" / sample)) == route.count("mapped:Genus1_species1")
    first = next(fastq.read_batches(str(query / "mapped" / "sampleA.pass.fastq")))
    mapped_a = [r for r in route_a if r.startswith("mapped")]
    assert first.ids[0] == mapped_a[0].split(":")[1] and " ch=1" in first.headers[0]   # id replaced by tax_unit
    assert os.listdir(query / "hits") == []                                           # carried hits removed
    assert os.path.exists(query / "empty.fastq")                                      # empty files are skipped
    with open(out / "alignment.pkl", "rb") as f:
        assert pickle.load(f) == alignment

    lens = {n.split(":")[1]: len(s) for n, s in zip(names, seqs)}
    raw_df = aligner.alignment_to_data_frame(alignment, output_folder=str(out), filename="raw_monica.dataframe")
    norm = aligner.normalizer(alignment, genomes_length=lens)
    df = aligner.alignment_to_data_frame(norm, output_folder=str(out))
    assert raw_df.shape == df.shape and abs(float(df["sampleA"].sum()) - 1.0) < 1e-9
    assert aligner.any_result(alignment) == 1

    # a second invocation on a folder that received new reads merges into alignment.pkl
    synth.write_fastq(str(query / "sampleA.pass.fastq"), eb[:eo[20]], eo[:21], ids=[f"n{i}" for i in range(20)])
    try:
        again = aligner.multi_threaded_aligner(str(query), paths, mode="basic", n_threads=1, output_folder=str(out))
    finally:
        os.chdir(cwd)
    assert set(again) == {"sampleA", "sampleB"}
    assert count_records(str(query / "mapped" / "sampleA.pass.fastq")) >= sum(r.startswith("mapped") for r in route_a)


def test_focus_second_pass(oracle, tmp_path):
    """monica.py:455-467: reads routed to focus/ are classified again, by the same function,
    against strain-level indexes."""
    names, seqs = util.small_genomes(4, 150_000, 200_000)
    dbs = tmp_path / "databases"
    dbs.mkdir()
    synth.write_fasta(str(dbs / "database0.fna.gz"), names, seqs)
    paths = aligner.indexer(str(dbs), str(tmp_path / "indexes"))
    # two "strains" of genome 1: the genome itself and a 1 %-diverged copy
    strain_names = ["Genus1_species1_strainA:STRA.1", "Genus1_species1_strainB:STRB.1"]
    strain_seqs = [seqs[1], synth.diverge(seqs[1], 99, 10_000)]
    fdbs = tmp_path / "focus_databases"
    fdbs.mkdir()
    synth.write_fasta(str(fdbs / "database0.fna.gz"), strain_names, strain_seqs)
    focus_paths = aligner.indexer(str(fdbs), str(tmp_path / "focus_indexes"))

    query, out = tmp_path / "query", tmp_path / "output"
    query.mkdir(), out.mkdir()
    bases, offsets, truth = synth.reads(seqs, 200, 2500, seed=78)
    synth.write_fastq(str(query / "s.fastq"), bases, offsets, ids=[f"r{i}" for i in range(200)])
    cwd = os.getcwd()
    try:
        first = aligner.multi_threaded_aligner(str(query), paths, mode="basic", n_threads=1,
                                               focus_species=["Genus1_species1"], output_folder=str(out))
        n_focus = count_records(str(query / "focus" / "s.fastq"))
        assert n_focus == first["s"].get("Genus1_species1", Counter()).get("ACC000001.1", 0) and n_focus > 10
        focus_reads = list(fastq.read_batches(str(query / "focus" / "s.fastq")))[0]
        (out / "focus").mkdir()
        second = aligner.multi_threaded_aligner(str(query / "focus"), focus_paths, mode="basic", n_threads=1,
                                                output_folder=str(out / "focus"))
    finally:
        os.chdir(cwd)
    want, route = expected_from_oracle(oracle, [(strain_names, strain_seqs, 0)], strain_names, focus_reads.bases,
                                       focus_reads.offsets, "basic")
    assert second == {"s": want}
    assert focus_reads.ids[0].startswith("r")                              # focus/ keeps the original ids
    assert count_records(str(query / "focus" / "mapped" / "s.fastq")) == sum(r.startswith("mapped") for r in route)
    assert count_records(str(query / "focus" / "ambiguous" / "s.fastq")) == route.count("ambiguous")


def test_no_samples_returns_zero(tmp_path, capsys):
    cwd = os.getcwd()
    try:
        assert aligner.multi_threaded_aligner(str(tmp_path), ["x.mmi"], output_folder=str(tmp_path)) == 0
    finally:
        os.chdir(cwd)
    assert "No query files were provided" in capsys.readouterr().out


def test_index_loader_errors(tmp_path):
    bad = tmp_path / "index0.mmi"
    bad.write_bytes(b"garbage")
    with pytest.raises(Exception, match="Damaged or empty index"):
        aligner.index_loader(str(bad))
    assert aligner.index_loader(str(tmp_path / "notes.txt")) is None


def test_mappy_compat_map_yields_hits(oracle):
    from monica_amd import mappy_compat
    names, seqs = util.small_genomes(2, 80_000, 90_000)
    a = mappy_compat.Aligner(seq=seqs[0].tobytes())
    assert a and a.k == 15 and a.w == 10
    hits = list(a.map(seqs[0][1000:4000].tobytes().decode()))
    assert len(hits) == 1 and hits[0].is_primary and hits[0].mapq == 60 and hits[0].ctg == "N/A"
    assert hits[0].r_st <= 1020 and hits[0].r_en >= 3980 and hits[0].strand == 1 and hits[0].NM == hits[0].blen - hits[0].mlen
    assert not mappy_compat.Aligner(fn_idx_in="/nonexistent/file.fa")


@pytest.mark.timeout(300)
def test_a_failed_sample_leaves_the_engine_usable(oracle, tmp_path, monkeypatch):
    """A malformed record in the middle of a file raises where the reference's loop would (SeqIO.parse inside
    aligner.py:212) -- after batches have been announced to the engine ahead of their call.  The engine goes back to
    the pool; the next invocation (monica's loop calls again, aligner.py:65) must neither wait for the abandoned
    announcement nor take it for one of its own batches."""
    names, seqs = util.small_genomes(4, 150_000, 200_000)
    dbs = tmp_path / "databases"
    dbs.mkdir()
    synth.write_fasta(str(dbs / "database0.fna.gz"), names, seqs)
    paths = aligner.indexer(str(dbs), str(tmp_path / "indexes"))
    query, out = tmp_path / "query", tmp_path / "output"
    query.mkdir(), out.mkdir()
    monkeypatch.setattr(aligner, "BATCH_READS", 64)            # many small batches: several are in flight when it breaks
    bases, offsets, truth = synth.reads(seqs, 600, 2000, seed=79)
    fq = query / "s.fastq"
    synth.write_fastq(str(fq), bases, offsets, ids=[f"r{i}" for i in range(600)])
    good = fq.read_bytes()
    lines = good.split(b"\n")
    lines[4 * 400 + 3] = lines[4 * 400 + 3][:-7]               # record 400: quality shorter than the sequence
    fq.write_bytes(b"\n".join(lines))
    cwd = os.getcwd()
    try:
        with pytest.raises(ValueError):
            aligner.multi_threaded_aligner(str(query), paths, mode="basic", n_threads=1, output_folder=str(out))
        # same engine (pooled per device), a different file of the same shape: same batch sizes, recycled page-locked arrays
        for stale in ("mapped", "unmapped", "ambiguous"):
            p = query / stale / "s.fastq"
            if p.exists():
                p.unlink()
        b2, o2, _ = synth.reads(seqs, 600, 2000, seed=80)
        synth.write_fastq(str(fq), b2, o2, ids=[f"q{i}" for i in range(600)])
        got = aligner.multi_threaded_aligner(str(query), paths, mode="basic", n_threads=1, output_folder=str(out))
    finally:
        os.chdir(cwd)
    want, route = expected_from_oracle(oracle, [(names, seqs, 0)], names, b2, o2, "basic")
    assert got == {"s": want}
    assert count_records(str(query / "mapped" / "s.fastq")) == sum(r.startswith("mapped") for r in route)


def test_default_thread_count_with_many_samples(oracle, tmp_path, monkeypatch):
    """`n_threads=None` (aligner.py:65, 89: one worker per core in the reference) with more samples than workers: the
    default is capped (a worker here is a GPU feeder with an engine, pipeline threads and OpenMP teams of its own), an
    explicit count is kept, and the per-sample results do not depend on either."""
    names, seqs = util.small_genomes(4, 120_000, 150_000)
    dbs = tmp_path / "databases"
    dbs.mkdir()
    synth.write_fasta(str(dbs / "database0.fna.gz"), names, seqs)
    paths = sorted(aligner.indexer(str(dbs), str(tmp_path / "indexes")))
    bases, offsets, truth = synth.reads(seqs, 12 * 40, 2000, seed=5)
    sizes = []
    real_pool = aligner.ThreadPool
    monkeypatch.setattr(aligner, "ThreadPool", lambda n=None: (sizes.append(n), real_pool(n))[1])
    monkeypatch.setattr(aligner, "DEFAULT_MAX_WORKERS", 3)
    results = []
    cwd = os.getcwd()
    for n_threads in (None, 20, 1):
        query = tmp_path / f"query_{n_threads}"
        query.mkdir()
        for s in range(12):
            lo, hi = offsets[s * 40], offsets[(s + 1) * 40]
            synth.write_fastq(str(query / f"s{s:02d}.fastq"), bases[lo:hi], offsets[s * 40:(s + 1) * 40 + 1] - lo, ids=[f"r{i}" for i in range(40)])
        out = tmp_path / f"out_{n_threads}"
        out.mkdir()
        try:
            results.append(aligner.multi_threaded_aligner(str(query), paths, mode="basic", n_threads=n_threads, output_folder=str(out)))
        finally:
            os.chdir(cwd)
    assert sizes == [3, 12, 1]                  # None: the cap; 20 threads for 12 samples: one each; 1: as given
    assert results[0] == results[1] == results[2] and len(results[0]) == 12
    want, _ = expected_from_oracle(oracle, [(names, seqs, 0)], names, bases[:offsets[40]], offsets[:41], "basic")
    assert results[0]["s00"] == want


def test_a_read_beyond_the_device_limits_does_not_cost_its_sample(oracle, tmp_path, capsys):
    """aligner.py:212-279 never drops a sample for one record.  A FASTQ file holding a 1.1 Mb read among ordinary ones:
    the file is consumed, the ordinary reads are counted and routed as without it (the oracle's answers), the long
    read is written to unmapped/ and the run says what it did."""
    g = synth.genome(0x2A2A, 1_400_000)
    names, seqs = [synth.contig_name(0), synth.contig_name(1)], [g, synth.genome(0x2A2B, 200_000)]
    dbs = tmp_path / "databases"
    dbs.mkdir()
    synth.write_fasta(str(dbs / "database0.fna.gz"), names, seqs)
    paths = sorted(aligner.indexer(str(dbs), str(tmp_path / "indexes")))
    bases, offsets, truth = synth.reads(seqs, 60, 3000, seed=8)
    reads = [bases[offsets[i]:offsets[i + 1]] for i in range(60)]
    reads.insert(25, g[50_000:50_000 + 1_100_000].copy())
    eb, eo = util.pack_reads(reads)
    query, out = tmp_path / "query", tmp_path / "output"
    query.mkdir(), out.mkdir()
    synth.write_fastq(str(query / "long.fastq"), eb, eo, ids=[f"r{i}" for i in range(61)])
    cwd = os.getcwd()
    try:
        got = aligner.multi_threaded_aligner(str(query), paths, mode="basic", n_threads=1, output_folder=str(out))
    finally:
        os.chdir(cwd)
    want, route = expected_from_oracle(oracle, [(names, seqs, 0)], names, bases, offsets, "basic")
    assert got == {"long": want}
    assert not os.path.exists(query / "long.fastq")
    assert count_records(str(query / "mapped" / "long.fastq")) == sum(r.startswith("mapped") for r in route)
    assert count_records(str(query / "unmapped" / "long.fastq")) == route.count("unmapped") + 1
    unm = [b for b in fastq.read_batches(str(query / "unmapped" / "long.fastq"))]
    assert "r25" in [i for b in unm for i in b.ids]
    assert "1 read(s) beyond the device limits" in capsys.readouterr().out
