"""CPU tests of the host-side mirror of monica.genomes.aligner: the pure-Python pieces
(best_hit, alignment_update, normalizer, any_result, data-frame export, FASTQ I/O)."""
import os
import pickle
from collections import Counter

import numpy as np
import pytest

from monica_amd import aligner, fastq


def test_best_hit_truth_table():
    c, d, e = ("G_a:ACC1", 1, 10), ("G_b:ACC2", 1, 10), ("G_c:ACC3", 1, 20)
    assert aligner.best_hit([c, d]) == 0
    assert aligner.best_hit([c, ("G_b:ACC2", 2, 10)]) == c
    assert aligner.best_hit([("G_a:ACC1", 2, 10), d]) == d
    assert aligner.best_hit([("G_a:ACC1", 2, 10), d, ("G_c:ACC3", 1, 10)]) == 0
    assert aligner.best_hit([c, d, e]) == e
    assert aligner.best_hit([c]) == c


def test_best_hit_agrees_with_integer_restatement(capi):
    rng = np.random.default_rng(8)
    for _ in range(2000):
        n = int(rng.integers(1, 6))
        hits = [("x:y", int(rng.integers(0, 300)), int(rng.integers(1, 300))) for _ in range(n)]
        want = capi.best_hit([(h[1], h[2]) for h in hits])
        got = aligner.best_hit(hits)
        assert (got == 0 and want == -1) or (got is hits[want])


def test_alignment_update_merges_into_pickle(tmp_path):
    out = str(tmp_path)
    first = aligner.alignment_update([({"Genus_a": Counter({"ACC1": 5})}, "s1"), ({}, "s2")], out)
    assert first == {"s1": {"Genus_a": Counter({"ACC1": 5})}, "s2": {}}
    second = aligner.alignment_update([({"Genus_a": Counter({"ACC1": 2, "ACC9": 1}), "Genus_b": Counter({"ACC2": 7})}, "s1"),
                                       ({"Genus_c": Counter({"ACC3": 1})}, "s3")], out)
    assert second["s1"]["Genus_a"] == Counter({"ACC1": 7, "ACC9": 1})
    assert second["s1"]["Genus_b"] == Counter({"ACC2": 7}) and second["s3"] == {"Genus_c": Counter({"ACC3": 1})}
    with open(os.path.join(out, aligner.ALIGNMENT_PICKLE_FILENAME), "rb") as f:
        assert pickle.load(f) == second
    assert aligner.any_result(second) == 1 and aligner.any_result({"s": {}}) == 0


def test_normalizer_and_data_frame(tmp_path):
    al = {"s1": {"Genus_a": Counter({"ACC1": 1000, "ACC2": 3000}), "Genus_b": Counter({"ACC3": 500})}, "s2": {}}
    lens = {"ACC1": 1000, "ACC2": 2000, "ACC3": 1000}
    norm = aligner.normalizer(al, genomes_length=lens)
    tot = 1.0 + 1.5 + 0.5
    assert norm["s1"]["Genus_a"]["ACC1"] == pytest.approx(1.0 / tot)
    assert norm["s1"]["Genus_a"]["ACC2"] == pytest.approx(1.5 / tot)
    assert sum(v for c in norm["s1"].values() for v in c.values()) == pytest.approx(1.0)
    df = aligner.alignment_to_data_frame(norm, output_folder=str(tmp_path))
    assert os.path.exists(tmp_path / "monica.dataframe")
    assert df.shape == (3, 1) and list(df.columns) == ["s1"]


def test_fastq_reader_and_biopython_title_rule(tmp_path):
    p = tmp_path / "a.fastq"
    p.write_bytes(b"@r1 runid=7 ch=3\nACGT\n+\nIIII\n@r2\nAC\nGT\n+r2\nII\nII\n\n@r3 x\n\n+\n\n")
    batches = list(fastq.read_batches(str(p), max_reads=2))
    assert [len(b) for b in batches] == [2, 1]
    b = batches[0]
    assert b.ids == ["r1", "r2"] and b.offsets.tolist() == [0, 4, 8] and b.seq(1) == b"ACGT"
    assert fastq.format_record(b, 0) == b"@r1 runid=7 ch=3\nACGT\n+\nIIII\n"
    assert fastq.format_record(b, 0, new_id="Genus_a") == b"@Genus_a r1 runid=7 ch=3\nACGT\n+\nIIII\n"
    assert fastq.format_record(b, 1, new_id="r2") == b"@r2\nACGT\n+\nIIII\n"
    assert len(batches[1]) == 1 and batches[1].seq(0) == b""
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@r1\nACGT\n+\nII\n")
    with pytest.raises(ValueError):
        list(fastq.read_batches(str(bad)))


def test_module_surface_matches_reference():
    import inspect
    sig = inspect.signature(aligner.multi_threaded_aligner)
    assert list(sig.parameters)[:8] == ["query_folder", "indexes_paths", "mode", "mapping_quality", "overnight",
                                        "n_threads", "focus_species", "output_folder"]
    assert sig.parameters["mapping_quality"].default == 60 and sig.parameters["overnight"].default is False
    assert sig.parameters["mode"].default is None
    asig = inspect.signature(aligner.aligner)
    assert list(asig.parameters) == ["sample", "sample_name", "index", "mode", "hits_folder", "mapping_quality",
                                     "overnight", "focus_species", "mapped_folder", "unmapped_folder",
                                     "ambiguous_folder", "focus_folder", "last_index"]
    assert aligner.BEST_N == 15 and aligner.INDEX_NAME == ["index", ".mmi"]
    assert (aligner.MAPPED_FILES_FOLDER, aligner.UNMAPPED_FILES_FOLDER, aligner.AMBIGUOUS_FILES_FOLDER,
            aligner.HITS_FILES_FOLDER, aligner.FOCUS_FILES_FOLDER) == ("mapped", "unmapped", "ambiguous", "hits", "focus")
    for name in ("indexer", "index_loader", "alignment_update", "normalizer", "alignment_to_data_frame", "best_hit",
                 "any_result"):
        assert callable(getattr(aligner, name))
