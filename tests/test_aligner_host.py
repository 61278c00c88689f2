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
    assert sig.parameters["mode"].default is None and sig.parameters["n_threads"].default is None
    # n_threads goes to ThreadPool as it is (aligner.py:89: None = one worker per core)
    src = inspect.getsource(aligner.multi_threaded_aligner)
    assert "ThreadPool(n_threads)" in src
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


def test_fastq_errors_follow_biopython(tmp_path):
    cases = {
        b"r1\nACGT\n+\nIIII\n": "Records in Fastq files should start with '@' character",
        b"@r1\nACGT\n": "End of file without quality information.",
        b"@r1\nACGT\n+r2\nIIII\n": "Sequence and quality captions differ.",
        b"@r1\nAC GT\n+\nIIIII\n": "Whitespace is not allowed in the sequence.",
        b"@r1\nACGT\n+\nII\x07I\n": "Invalid character in quality string",
        b"@r1 d\nACGT\n+\nIIIII\n": "Lengths of sequence and quality values differs for r1 d (4 and 5).",
    }
    for i, (data, msg) in enumerate(cases.items()):
        p = tmp_path / f"bad{i}.fastq"
        p.write_bytes(data)
        with pytest.raises(ValueError) as ei:
            list(fastq.read_batches(str(p)))
        assert str(ei.value) == msg
    empty = tmp_path / "empty.fastq"
    empty.write_bytes(b"")
    assert list(fastq.read_batches(str(empty))) == []
    # a quality line that starts with '@' belongs to the record while characters are missing
    tricky = tmp_path / "tricky.fastq"
    tricky.write_bytes(b"@r1\nACGTAC\n+\n@II\n@II\n@r2\nAC\n+\n@I")
    (b,) = list(fastq.read_batches(str(tricky)))
    assert b.ids == ["r1", "r2"] and b.qual(0) == b"@II@II" and b.qual(1) == b"@I"


def test_route_writes_what_seqio_write_would(tmp_path):
    from monica_amd import _capi
    src = tmp_path / "s.fastq"
    src.write_bytes(b"@r1 ch=1\nACGT\n+\nIIII\n@r2\nGGGG\n+\n!!!!\n@r3 z\nTT\n+\n##\n@Genus_a k\nCC\n+\nII\n@r5\nAAAA\n+\nIIII\n")
    rd = _capi.FastqReader(str(src))
    assert rd.next() == 5
    paths = [str(tmp_path / n) for n in ("unmapped.fq", "ambiguous.fq", "mapped.fq", "focus.fq")]
    (tmp_path / "mapped.fq").write_bytes(b"@old\nA\n+\nI\n")             # append mode
    dest = [_capi.TO_MAPPED | _capi.TO_FOCUS, _capi.TO_UNMAPPED, _capi.TO_AMBIGUOUS, _capi.TO_MAPPED, 0]
    rd.route(dest, [1, -1, -1, 0, -1], ["Genus_a", "Genus_b"], paths)
    assert open(paths[0], "rb").read() == b"@r2\nGGGG\n+\n!!!!\n"
    assert open(paths[1], "rb").read() == b"@r3 z\nTT\n+\n##\n"
    assert open(paths[2], "rb").read() == b"@old\nA\n+\nI\n@Genus_b r1 ch=1\nACGT\n+\nIIII\n@Genus_a k\nCC\n+\nII\n"
    assert open(paths[3], "rb").read() == b"@r1 ch=1\nACGT\n+\nIIII\n"
    with pytest.raises(_capi.MncError):
        rd.route([_capi.TO_FOCUS, 0, 0, 0, 0], None, [], paths[:3] + [None])
    rd.close()


def test_many_records_parse_and_route_on_all_host_threads(tmp_path):
    """Large batches take the four-line fast path of the reader and the sliced, in-place writes of
    the router: both must give what the record-by-record forms give."""
    from monica_amd import _capi
    rng = np.random.default_rng(5)
    n = 9000
    recs = []
    for r in range(n):
        l = int(rng.integers(0, 60))
        seq = rng.choice(list(b"ACGTN"), l).astype(np.uint8).tobytes() if l else b""
        qual = rng.integers(33, 127, l).astype(np.uint8).tobytes() if l else b""
        title = (b"Genus_a" if r % 7 == 0 else b"r%d" % r) + (b" ch=%d" % (r % 512) if r % 3 else b"")
        recs.append((title, seq, qual))
    text = b"".join(b"@" + t + b"\n" + s_ + b"\n+" + (t if i % 11 == 0 else b"") + b"\n" + q + b"\n" for i, (t, s_, q) in enumerate(recs))
    src = tmp_path / "big.fastq"
    src.write_bytes(text[:-1])                                     # no line terminator at the end of the file
    assert list(_fastq_general_iterator(text)) == recs
    got = []
    for b in fastq.read_batches(str(src), max_reads=4000):
        got += [(b.headers[i].encode(), b.seq(i), b.qual(i)) for i in range(len(b))]
    assert got == recs
    rd = _capi.FastqReader(str(src))
    assert rd.next() == n
    dest = rng.choice([0, _capi.TO_UNMAPPED, _capi.TO_AMBIGUOUS, _capi.TO_MAPPED, _capi.TO_MAPPED | _capi.TO_FOCUS], n).astype(np.uint8)
    label = np.where(dest & _capi.TO_MAPPED, rng.integers(0, 2, n), -1).astype(np.int32)
    labels = ["Genus_a", "Genus_b"]
    paths = [str(tmp_path / k) for k in ("u.fq", "a.fq", "m.fq", "f.fq")]
    (tmp_path / "m.fq").write_bytes(b"@old\nA\n+\nI\n")
    rd.route(dest, label, labels, paths)
    rd.close()
    want = {k: b"" for k in range(4)}
    want[2] = b"@old\nA\n+\nI\n"
    for r, (t, s_, q) in enumerate(recs):
        for k in range(4):
            if not (dest[r] >> k) & 1:
                continue
            head = t
            if k == 2 and label[r] >= 0:
                lab = labels[label[r]].encode()
                first = t.split(None, 1)[0] if t.split() else b""
                head = lab if not t else (t if first == lab else lab + b" " + t)
            want[k] += b"@" + head + b"\n" + s_ + b"\n+\n" + q + b"\n"
    for k in range(4):
        assert open(paths[k], "rb").read() == want[k], k


def test_large_appends_through_a_mapping_equal_the_formatted_writes(tmp_path, monkeypatch):
    """SeqIO.write appends record after record (aligner.py:232-243, 265); here a batch's records for one file are
    written by all routing threads at once: formatted through small buffers (the default), or -- MNC_ROUTE_TEXT=1 --
    gathered by pwritev from the file's own bytes, or -- with MNC_ROUTE_MMAP=1 as well -- the file that takes most of a
    large batch has its new tail reserved, mapped and filled in place (buffered writes to ONE file wait for each other).  All three must leave the same
    bytes, also when the append starts in the middle of a page and when some records are not written as they were read
    (a '+' line with the title repeated, trailing blanks, CRLF)."""
    from monica_amd import _capi, synth
    rng = np.random.default_rng(11)
    n, L = 9000, 5000                                             # 90 MB of FASTQ: the mapped file's share is above the threshold
    names, seqs = synth.genome_set(1, min_len=300_000, max_len=300_001, diverged_half=False)
    bases, offsets, _ = synth.reads(seqs, n, L, seed=5)
    raw = bases.tobytes()
    parts = []
    for r in range(n):
        s_ = raw[offsets[r]:offsets[r + 1]]
        q = bytes(rng.integers(33, 127, 1).astype(np.uint8)) * len(s_)
        t = b"read%d ch=%d" % (r, r % 512)
        if r % 1000 == 7:
            parts.append(b"@" + t + b"  \n" + s_ + b"\n+" + t + b"\n" + q + b"\n")      # trailing blanks, the title repeated
        elif r % 1000 == 9:
            parts.append(b"@" + t + b"\r\n" + s_ + b"\r\n+\r\n" + q + b"\r\n")          # CRLF
        else:
            parts.append(b"@" + t + b"\n" + s_ + b"\n+\n" + q + b"\n")
    src = tmp_path / "big.fastq"
    src.write_bytes(b"".join(parts))
    dest = rng.choice([_capi.TO_UNMAPPED, _capi.TO_AMBIGUOUS, _capi.TO_MAPPED, _capi.TO_MAPPED, _capi.TO_MAPPED, _capi.TO_MAPPED,
                       _capi.TO_MAPPED | _capi.TO_FOCUS], n).astype(np.uint8)
    label = np.where(dest & _capi.TO_MAPPED, rng.integers(0, 2, n), -1).astype(np.int32)
    labels = ["Genus_a", "read17"]                                 # (read17: its own id -- that title stays as it is)
    results = {}
    for mode in ("mapped", "pwritev", "formatted"):
        monkeypatch.delenv("MNC_ROUTE_TEXT", raising=False)
        monkeypatch.delenv("MNC_ROUTE_MMAP", raising=False)
        if mode != "formatted":
            monkeypatch.setenv("MNC_ROUTE_TEXT", "1")
        if mode == "mapped":
            monkeypatch.setenv("MNC_ROUTE_MMAP", "1")
        out = tmp_path / mode
        out.mkdir()
        paths = [str(out / k) for k in ("u.fq", "a.fq", "m.fq", "f.fq")]
        (out / "m.fq").write_bytes(b"@old\nACG\n+\nIII\n")        # 15 bytes: the append starts inside a page
        for _ in range(2):                                         # monica's loop calls again: a second append
            rd = _capi.FastqReader(str(src))
            assert rd.next(n, 1 << 40) == n
            batch = rd.detach()
            batch.route(dest, label, labels, paths)
            batch.close(), rd.close()
        results[mode] = [open(p_, "rb").read() for p_ in paths]
    assert results["mapped"] == results["formatted"] and results["pwritev"] == results["formatted"]
    m = results["mapped"][2]
    assert m.startswith(b"@old\nACG\n+\nIII\n@")
    first = int(np.flatnonzero(dest & _capi.TO_MAPPED)[0])
    s0 = raw[offsets[first]:offsets[first + 1]]
    assert m[15:].startswith(b"@" + labels[label[first]].encode() + b" read%d ch=%d\n" % (first, first % 512) + s0 + b"\n+\n")
    assert len(m) == 15 + 2 * (len(m) - 15) // 2 and m.count(b"\r") == 0 and m.count(b"  \n") == 0
    assert sum(len(x) for x in results["mapped"]) > 150_000_000


def test_hitmap_is_sample_hits_with_best_hit(tmp_path):
    """Extending per-id lists part by part and reducing them with best_hit equals the carried summary."""
    import numpy as np
    from monica_amd import _capi
    rng = np.random.default_rng(5)
    n, parts = 300, 3
    ids = [f"read{i % 250}" for i in range(n)]                             # 50 duplicated ids
    src = tmp_path / "s.fastq"
    src.write_bytes(b"".join(f"@{i} x\nACGT\n+\nIIII\n".encode() for i in ids))
    names = [f"Genus_{c}:ACC{c}" for c in range(6)]
    idx = _capi.Index.from_seqs(names, ["ACGTTGCATGCATGACTGACTGATCGATCGTAGCTAGCTAGCATGCATGCAT" * 3] * 6)
    ref = {}                                                               # the reference's dict of lists
    hm = _capi.HitMap()
    for part in range(parts):
        if part == 1:                                                      # through the carried file
            hm.save(str(tmp_path / "h.pkl"))
            hm = _capi.HitMap(str(tmp_path / "h.pkl"))
        rd = _capi.FastqReader(str(src))
        base = 0
        while rd.next(max_reads=128):
            m = rd.n
            nh = rng.integers(0, 4, m).astype(np.int32)
            lists = [[(names[int(rng.integers(0, 6))], int(rng.integers(0, 4)), int(rng.integers(1, 4)) * 10)
                      for _ in range(k)] for k in nh]
            assign = np.full(m, _capi.UNMAPPED, dtype=np.int32)
            best = np.zeros(m, dtype=_capi.HIT_DTYPE)
            for r, hl in enumerate(lists):
                if not hl:
                    continue
                lo = min(h[1] / h[2] for h in hl)
                at = [h for h in hl if h[1] / h[2] == lo]
                b = at[-1]
                best[r] = (names.index(b[0]), 60, b[1], b[2])
                assign[r] = names.index(b[0]) if len(at) == 1 else _capi.AMBIGUOUS
            state = hm.update(rd, idx, assign, best, nh)
            hm_names = hm.names()
            for r, hl in enumerate(lists):
                rid = ids[base + r]
                if hl:
                    ref.setdefault(rid, []).extend(hl)
                if rid not in ref:
                    assert state[r, 0] == 0
                    continue
                cur = ref[rid]
                want = cur[0] if len(cur) == 1 else aligner.best_hit(cur)
                assert state[r, 0] == len(cur)
                if not want:
                    assert state[r, 4] == 1
                else:
                    assert state[r, 4] == 0 and hm_names[state[r, 3]] == want[0] and state[r, 2] == want[2]
            base += m
        rd.close()
    assert len(hm) == len(ref)


def test_database_builder_header_contract_and_chunks(tmp_path, monkeypatch):
    """database.multi_threaded_builder -> databaseN.fna.gz with `tax_unit:accession` headers
    (database.py:52-67), which the index builder turns into one genome per name."""
    import gzip
    from monica_amd import database, _capi
    gen = tmp_path / "genomes"
    gen.mkdir()
    monkeypatch.setattr(database, "GENOMES_PATH", str(gen))
    rng = np.random.default_rng(11)

    def write_genome(name, contigs):
        path = gen / name
        with gzip.open(path, "wt") as f:
            for title, n in contigs:
                seq = "".join("ACGT"[i] for i in rng.integers(0, 4, n))
                f.write(">" + title + "\n")
                for i in range(0, n, 70):
                    f.write(seq[i:i + 70] + "\n")
        return str(path)

    g1 = write_genome("g1.fna.gz", [("NZ_1.1 Genus a chromosome", 3000), ("NZ_1p.1 plasmid", 500)])
    g2 = write_genome("g2.fna.gz", [("NZ_2.1 Genus b", 2500)])
    g3 = write_genome("g3.fna.gz", [("NZ_3.1 Genus c", 2000)])
    genomes = [(g1, ("Genus_a", "ACC1.1")), (g2, ("Genus_b", "ACC2.1")), (g3, ("Genus_c", "ACC3.1"))]
    out = tmp_path / "dbs"
    path, lengths = database.multi_threaded_builder(genomes=genomes, max_chunk_size=1 << 20, databases_path=str(out),
                                                    keep_genomes=True, n_threads=2)
    assert path == str(out) and lengths == {"ACC1.1": 3500, "ACC2.1": 2500, "ACC3.1": 2000}
    assert sorted(os.listdir(out)) == ["database0.fna.gz"] and os.path.exists(gen / "database_created")
    with open(gen / "current_genomes_length.pkl", "rb") as f:
        assert pickle.load(f) == lengths
    with gzip.open(out / "database0.fna.gz", "rt") as f:
        text = f.read().splitlines()
    heads = [l for l in text if l.startswith(">")]
    assert heads == [">Genus_a:ACC1.1 NZ_1.1 Genus a chromosome", ">Genus_a:ACC1.1 NZ_1p.1 plasmid",
                     ">Genus_b:ACC2.1 NZ_2.1 Genus b", ">Genus_c:ACC3.1 NZ_3.1 Genus c"]
    assert max(len(l) for l in text if not l.startswith(">")) == 60
    idx = _capi.Index.build(str(out / "database0.fna.gz"))
    assert idx.contig_names == ["Genus_a:ACC1.1", "Genus_a:ACC1.1", "Genus_b:ACC2.1", "Genus_c:ACC3.1"]
    assert idx.genome_names == ["Genus_a:ACC1.1", "Genus_b:ACC2.1", "Genus_c:ACC3.1"] and idx.genome_lens == [3500, 2500, 2000]
    # the reference's chunking: a genome larger than the limit goes alone; the genome that closes a
    # chunk by not fitting is not carried over (database.py:84-90)
    sizes = [os.path.getsize(g) for g in (g1, g2, g3)]
    chunks = list(database._genomes_splitter(genomes, max_chunk_size=sizes[0] + 10))
    assert chunks == [[genomes[0]], [genomes[2]]]
    assert list(database._genomes_splitter(genomes, max_chunk_size=sizes[1] - 1))[0] == [genomes[0]]
    # keep_genomes falsy: the downloaded genomes are deleted after the build
    database.multi_threaded_builder(genomes=genomes, max_chunk_size=1 << 20, databases_path=str(out), n_threads=1)
    assert not [f for f in os.listdir(gen) if f.endswith(".fna.gz")]


def _fastq_general_iterator(text):
    """Biopython's FastqGeneralIterator + the quality check of FastqPhredIterator, transcribed
    (plain Python, line by line) as the behaviour the C++ reader has to reproduce."""
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    it = iter([l + b"\n" for l in lines])
    rstrip = lambda b: b.rstrip(b" \t\r\n\x0b\x0c")

    def readline():
        return next(it, b"")
    line = readline()
    if not line:
        return
    while line:
        if line[0:1] != b"@":
            raise ValueError("Records in Fastq files should start with '@' character")
        title = rstrip(line[1:])
        seq = rstrip(readline())
        while True:
            line = readline()
            if not line:
                raise ValueError("End of file without quality information.")
            if line[0:1] == b"+":
                second = rstrip(line[1:])
                if second and second != title:
                    raise ValueError("Sequence and quality captions differ.")
                break
            seq += rstrip(line)
        if b" " in seq or b"\t" in seq:
            raise ValueError("Whitespace is not allowed in the sequence.")
        qual = rstrip(readline())
        while True:
            line = readline()
            if not line:
                break
            if line[0:1] == b"@" and len(qual) >= len(seq):
                break
            qual += rstrip(line)
        if len(seq) != len(qual):
            raise ValueError("Lengths of sequence and quality values differs for %s (%i and %i)."
                             % (title.decode(errors="replace"), len(seq), len(qual)))
        if any(c < 33 or c > 126 for c in qual):
            raise ValueError("Invalid character in quality string")
        yield title, seq, qual


def test_fastq_reader_against_the_transcribed_biopython_rules(tmp_path):
    rng = np.random.default_rng(99)
    alphabet_q = bytes(range(33, 127))

    def chunks(b, rng):
        out, i = [], 0
        while i < len(b):
            step = int(rng.integers(1, max(2, len(b) + 1)))
            out.append(b[i:i + step])
            i += step
        return out or [b""]

    for case in range(400):
        text = b""
        for r in range(int(rng.integers(0, 6))):
            n = int(rng.integers(0, 40))
            seq = bytes(rng.choice(list(b"ACGTNacgt"), n)) if n else b""
            qual = bytes(rng.choice(list(alphabet_q), n)) if n else b""
            title = b"r%d" % r + (b" extra words" if rng.random() < 0.4 else b"")
            multi = rng.random() < 0.3
            text += b"@" + title + b"\n"
            text += b"\n".join(chunks(seq, rng) if multi else [seq]) + b"\n"
            text += b"+" + (title if rng.random() < 0.3 else b"") + b"\n"
            text += b"\n".join(chunks(qual, rng) if multi else [qual]) + b"\n"
        kind = rng.random()
        if kind < 0.10:
            text += b"\n\n"
        elif kind < 0.18 and text:
            at = int(rng.integers(0, len(text)))                       # damage one byte
            text = text[:at] + bytes([int(rng.choice(list(b"@+ \n\tA\x07")))]) + text[at + 1:]
        elif kind < 0.24 and text:
            text = text[: int(rng.integers(0, len(text)))]                # truncate
        elif kind < 0.28:
            text = text.replace(b"\n", b"\r\n")
        p = tmp_path / "f.fastq"
        p.write_bytes(text)
        try:
            want = list(_fastq_general_iterator(text))
            want_err = None
        except ValueError as e:
            want, want_err = None, str(e)
        try:
            got = []
            for b in fastq.read_batches(str(p), max_reads=int(rng.integers(1, 5)), max_bases=int(rng.choice([1, 7, 50, 1 << 29]))):
                got += [(b.headers[i].encode(), b.seq(i), b.qual(i)) for i in range(len(b))]
            got_err = None
        except ValueError as e:
            got, got_err = None, str(e)
        assert got_err == want_err, (case, text, got_err, want_err)
        if want is not None:
            assert got == want, (case, text)


def test_decisions_agree_with_real_mappy_when_it_is_installed():
    """PARITY UNPINNED: this image has no mappy, so the restatement cannot be checked against the
    real library here.  Where mappy 2.17 is installed next to a GPU this measures the agreement
    of the gated (ctg, NM, mlen) lists on a synthetic world (monica/genomes/aligner.py:193-195)."""
    mappy = pytest.importorskip("mappy")
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU as well")
    from monica_amd import mappy_compat, synth
    import tempfile
    names, seqs = synth.genome_set(4, min_len=150_000, max_len=250_000)
    with tempfile.TemporaryDirectory() as d:
        fa = os.path.join(d, "db.fna")
        synth.write_fasta(fa, names, seqs)
        theirs = mappy.Aligner(fn_idx_in=fa, preset="map-ont", best_n=15)
        ours = mappy_compat.Aligner(fn_idx_in=fa, preset="map-ont", best_n=15)
        bases, offsets, truth = synth.reads(seqs, 300, 4000, seed=77)
        raw, same = bases.tobytes(), 0
        for r in range(300):
            s = raw[offsets[r]:offsets[r + 1]].decode()
            a = sorted((h.ctg, h.NM, h.mlen) for h in theirs.map(s) if h.is_primary and h.mapq >= 60)
            b = sorted((h.ctg, h.NM, h.mlen) for h in ours.map(s) if h.is_primary and h.mapq >= 60)
            same += a == b
        assert same >= 285, f"only {same}/300 gated hit lists equal mappy's"


def test_indexer_can_write_minimap2s_own_format(tmp_path, monkeypatch):
    """aligner.py:45-46 writes `indexN.mmi` through mappy; with `mappy_compat.INDEX_FILE_FORMAT = "mmi"` the files are
    minimap2's format (readable by the reference's installation too), and `index_loader` (aligner.py:56-62) loads both."""
    from monica_amd import mappy_compat, synth
    import util
    names, seqs = util.small_genomes(2, 60_000, 80_000)
    dbs = tmp_path / "databases"
    dbs.mkdir()
    synth.write_fasta(str(dbs / "database0.fna.gz"), names, seqs)
    native = aligner.indexer(str(dbs), str(tmp_path / "native"))
    monkeypatch.setattr(mappy_compat, "INDEX_FILE_FORMAT", "mmi")
    mmi = aligner.indexer(str(dbs), str(tmp_path / "mmi"))
    assert [os.path.basename(p) for p in native + mmi] == ["index0.mmi", "index0.mmi"]
    assert open(native[0], "rb").read(6) == b"MNCIDX" and open(mmi[0], "rb").read(4) == b"MMI\x02"
    a, b = aligner.index_loader(native[0]), aligner.index_loader(mmi[0])
    assert a and b and a.seq_names == b.seq_names == names
    ha, ya = a.index.dump()
    hb, yb = b.index.dump()
    assert np.array_equal(ha, hb) and np.array_equal(ya, yb) and a.index.mid_occ == b.index.mid_occ


def test_index_files_interoperate_with_real_mappy_when_it_is_installed(tmp_path):
    """UNPINNED here (no mappy in this image): where mappy 2.17 is installed, an index it wrote loads as the index this
    library builds from the same FASTA, and an index this library wrote in the .mmi format loads in mappy
    (aligner.py:45-46, 59)."""
    mappy = pytest.importorskip("mappy")
    from monica_amd import mappy_compat, synth
    names, seqs = synth.genome_set(3, min_len=80_000, max_len=120_000)
    fa = str(tmp_path / "db.fna")
    synth.write_fasta(fa, names, seqs)
    theirs_path, ours_path = str(tmp_path / "theirs.mmi"), str(tmp_path / "ours.mmi")
    assert mappy.Aligner(fn_idx_in=fa, preset="map-ont", best_n=15, fn_idx_out=theirs_path)
    ours = mappy_compat.Aligner(fn_idx_in=fa, preset="map-ont", best_n=15)
    ours.index.save(ours_path, mmi=True)
    loaded = mappy_compat.Aligner(fn_idx_in=theirs_path)
    assert loaded and loaded.seq_names == ours.seq_names
    for x, y in zip(loaded.index.dump(), ours.index.dump()):
        assert np.array_equal(x, y)
    back = mappy.Aligner(fn_idx_in=ours_path)
    assert back and list(back.seq_names) == ours.seq_names
