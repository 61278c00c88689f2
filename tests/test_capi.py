"""CPU tests of the C-ABI library: it loads, exports every symbol the header declares, the
host-side index builder agrees with the oracle, the host-side monica helpers agree with the
oracle, and compute entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from monica_amd import synth
import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "monica_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mnc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(capi):
    declared = header_functions()
    assert len(declared) >= 30
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/monica_amd.h but not exported"
    assert sorted(capi.EXPORTS) == declared
    assert b"gfx950" in L.mnc_version()


def test_error_strings(capi):
    L = capi.lib()
    assert L.mnc_strerror(0) == b"ok"
    assert L.mnc_strerror(capi.ERR_FORMAT) == b"Damaged or empty index"     # aligner.py:61
    for code in range(-8, 0):
        assert L.mnc_strerror(code) != b"unknown error"


@pytest.fixture(scope="module")
def world(capi, oracle):
    names, seqs = util.small_genomes(4, 100_000, 140_000)
    # a genome split over two contigs sharing one name (database.py:59-64) and an N run
    s2 = seqs[2].copy()
    s2[5000:5040] = ord("N")
    names = names[:2] + [names[2], names[2], names[3]]
    seqs = seqs[:2] + [s2[:60_000], s2[60_000:], seqs[3]]
    idx = capi.Index.from_seqs(names, seqs)
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    return names, seqs, idx, oidx


def test_index_matches_oracle(capi, world):
    names, seqs, idx, oidx = world
    info = idx.info()
    assert (info.k, info.w) == (15, 10)
    assert info.n_contigs == 5 and info.n_genomes == 4
    assert info.n_keys == oidx.n_keys and info.n_occ == oidx.n_minimizers
    assert info.mid_occ == oidx.mid_occ
    h, y = idx.dump()
    oh, oy = oidx.dump()
    assert np.array_equal(h, oh) and np.array_equal(y, oy)
    assert idx.contig_names == names
    assert idx.contig_genome.tolist() == [0, 1, 2, 2, 3]
    assert idx.genome_lens[2] == len(seqs[2]) + len(seqs[3])       # database.py:57-65 sums contigs
    assert info.total_len == sum(len(s) for s in seqs)


def test_index_file_roundtrip_and_fasta_build(capi, world, tmp_path):
    names, seqs, idx, oidx = world
    p = str(tmp_path / "index0.mmi")
    idx.save(p)
    again = capi.Index.load(p)
    assert again.contig_names == idx.contig_names and again.mid_occ == idx.mid_occ
    a, b = again.dump(), idx.dump()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    for gz in (False, True):
        fa = str(tmp_path / ("database0.fna.gz" if gz else "database0.fna"))
        synth.write_fasta(fa, [n + " some description" for n in names], seqs, width=70, gz=gz)
        out = str(tmp_path / f"built{int(gz)}.mmi")
        built = capi.Index.build(fa, out)
        assert built.contig_names == names                               # name = text before whitespace
        c = built.dump()
        assert np.array_equal(c[0], b[0]) and np.array_equal(c[1], b[1])
        assert os.path.getsize(out) > 0
        assert capi.Index.load(out).info().n_keys == idx.info().n_keys


def test_damaged_or_empty_index_is_refused(capi, world, tmp_path):
    names, seqs, idx, _ = world
    p = str(tmp_path / "index0.mmi")
    idx.save(p)
    raw = open(p, "rb").read()
    for name, blob in (("trunc.mmi", raw[: len(raw) // 2]), ("junk.mmi", b"not an index at all" * 10), ("empty.mmi", b"")):
        q = str(tmp_path / name)
        open(q, "wb").write(blob)
        with pytest.raises(capi.MncError) as e:
            capi.Index.load(q)
        assert e.value.code == capi.ERR_FORMAT and "Damaged or empty index" in str(e.value)
    with pytest.raises(capi.MncError) as e:
        capi.Index.load(str(tmp_path / "missing.mmi"))
    assert e.value.code == capi.ERR_IO
    # bit rot that keeps the size: every field the device code would index with is checked at
    # load (header counts against the file size, offsets monotone, keys ascending and < 2^30,
    # occurrence words inside their contig) -- 'Damaged or empty index', never a GPU fault
    import struct
    hdr = struct.Struct("<8s4i3q")
    magic, k, w, n_contigs, mid_occ, n_keys, n_occ, names_bytes = hdr.unpack_from(raw)
    assert magic[:6] == b"MNCIDX" and n_contigs == len(names)
    body = hdr.size + names_bytes + n_contigs * 8
    def patched(off, fmt, value):
        b = bytearray(raw)
        struct.pack_into(fmt, b, off, value)
        return bytes(b)
    damaged = {
        "n_keys.mmi": hdr.pack(magic, k, w, n_contigs, mid_occ, n_keys + 1, n_occ, names_bytes) + raw[hdr.size:],
        "huge.mmi": hdr.pack(magic, k, w, n_contigs, mid_occ, 1 << 60, 1 << 61, names_bytes) + raw[hdr.size:],
        "contigs.mmi": hdr.pack(magic, k, w, n_contigs + 3, mid_occ, n_keys, n_occ, names_bytes) + raw[hdr.size:],
        "key_order.mmi": patched(body + 4 * 10, "<I", 0),
        "key_range.mmi": patched(body + 4 * (n_keys - 1), "<I", 1 << 31),
        "key_off.mmi": patched(body + 4 * n_keys + 8 * 7, "<Q", n_occ + 5),
        "pos_rid.mmi": patched(body + 4 * n_keys + 8 * (n_keys + 1) + 8 * 3, "<Q", (n_contigs + 9) << 32 | 100),
        "pos_off.mmi": patched(body + 4 * n_keys + 8 * (n_keys + 1) + 8 * 3, "<Q", 0x7ffffff0),
        "tail.mmi": raw + b"\0\0\0\0",
    }
    for name, blob in damaged.items():
        q = str(tmp_path / name)
        open(q, "wb").write(blob)
        with pytest.raises(capi.MncError) as e:
            capi.Index.load(q)
        assert e.value.code == capi.ERR_FORMAT, name
    fa = str(tmp_path / "empty.fna")
    open(fa, "w").write("")
    with pytest.raises(capi.MncError) as e:
        capi.Index.build(fa)
    assert e.value.code == capi.ERR_FORMAT
    with pytest.raises(capi.MncError) as e:
        capi.Index.from_seqs(["a:b"], [b"ACGT" * 100], k=21, w=11)
    assert e.value.code == capi.ERR_UNSUPPORTED


def test_best_hit_integer_restatement_equals_float_reference(capi, oracle):
    table = [([(1, 10), (1, 10)], -1), ([(1, 10), (2, 10)], 0), ([(2, 10), (1, 10)], 1),
             ([(2, 10), (1, 10), (1, 10)], -1), ([(1, 10), (1, 10), (1, 20)], 2), ([(5, 9)], 0),
             ([(2, 20), (1, 10)], -1)]
    for hits, want in table:
        assert capi.best_hit(hits) == want
    rng = np.random.default_rng(2)
    for _ in range(3000):
        n = int(rng.integers(1, 6))
        hi = int(rng.choice([4, 50, 5000, 1 << 20]))
        hits = [(int(rng.integers(0, hi)), int(rng.integers(1, hi + 1))) for _ in range(n)]
        assert capi.best_hit(hits) == oracle.best_hit(hits), hits


def test_host_counts(capi, world):
    names, seqs, idx, _ = world
    assign = np.array([0, 3, -1, -2, 2, 4, 0], dtype=np.int32)
    best = np.zeros(7, dtype=capi.HIT_DTYPE)
    best["mlen"] = [100, 200, 0, 0, 300, 400, 50]
    offsets = np.arange(8, dtype=np.int64) * 1000
    assert capi.counts(idx, assign, best, offsets, 1).tolist() == [2, 0, 2, 1]
    assert capi.counts(idx, assign, best, offsets, 2).tolist() == [2000, 0, 2000, 1000]
    assert capi.counts(idx, assign, best, offsets, 3).tolist() == [150, 0, 500, 400]


def test_synth_is_deterministic_and_sliceable(capi):
    g = synth.genome(5, 10_000)
    assert np.array_equal(g, synth.genome(5, 10_000)) and not np.array_equal(g, synth.genome(6, 10_000))
    assert set(np.unique(g)) == set(b"ACGT")
    d = synth.diverge(g, 9, 30_000)
    assert 0.02 < (d != g).mean() < 0.04
    seqs = [synth.genome(1, 50_000), synth.genome(2, 60_000)]
    b, o, t = synth.reads(seqs, 64, 1000, seed=4)
    b2, _, t2 = synth.reads(seqs, 32, 1000, seed=4, first=32)
    assert np.array_equal(b[32 * 1000:], b2) and np.array_equal(t[32:], t2)     # any slice, same bytes
    assert set(np.unique(t)) <= {-1, 0, 1}


def test_no_cpu_fallback(capi, world):
    """Without a gfx950 device the product refuses to classify (it never falls back to CPU)."""
    n = capi.device_count()
    if n > 0:
        pytest.skip("a GPU is present")
    names, seqs, idx, _ = world
    with pytest.raises(capi.MncError) as e:
        capi.Engine(idx, 0)
    assert e.value.code == capi.ERR_NODEVICE
    # the merge over index parts (C2) is a kernel of the library too: no host form behind the entry points
    with pytest.raises(capi.MncError) as e:
        capi.merge_summaries_device(0x1000, 2, 4, 0x2000)
    assert e.value.code == capi.ERR_NODEVICE
    with pytest.raises(capi.MncError) as e:
        capi.shard_summary_device(0x1000, 0x1000, 0x1000, 4, 0, 0x2000)
    assert e.value.code == capi.ERR_NODEVICE


def test_product_does_not_touch_the_oracle():
    """The product package must not import, link or read anything under oracle/."""
    pkg = os.path.join(ROOT, "monica_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in text and "liborc" not in text and "mm_oracle" not in text, f


# ---------------------------------------------------------------- minimap2's index file (.mmi)
def write_mmi_by_the_published_layout(path, names, seqs, h, y, b=14, descending=False):
    """A second statement of the format, independent of the library's writer: "MMI\\2", u32 w k b n_seq
    flag; per sequence u8 name length + name + u32 length; per bucket i32 n + n words, u32 size + size (key, value)
    pairs; 4-bit bases, 8 per u32.  Entries go out in ascending or descending key order: a loader must not care."""
    import struct
    code = np.full(256, 4, dtype=np.uint32)
    for k, c in enumerate("ACGT"):
        code[ord(c)] = code[ord(c.lower())] = k
    code[ord("U")] = code[ord("u")] = 3
    with open(path, "wb") as f:
        f.write(b"MMI\x02" + struct.pack("<5I", 10, 15, b, len(names), 0))
        for nm, s in zip(names, seqs):
            f.write(struct.pack("<B", len(nm)) + nm.encode() + struct.pack("<I", len(s)))
        bucket = (h & np.uint64((1 << b) - 1)).astype(np.int64)
        order = np.lexsort((y, h, bucket))
        hb, hh, yy = bucket[order], h[order], y[order]
        starts = np.searchsorted(hb, np.arange((1 << b) + 1))
        for bi in range(1 << b):
            lo, hi = starts[bi], starts[bi + 1]
            keys, first, counts = np.unique(hh[lo:hi], return_index=True, return_counts=True)
            p, kv = [], []
            for kx, fx, cx in zip(keys, first, counts):
                key = int(kx) >> b << 1
                if cx == 1:
                    kv.append((key | 1, int(yy[lo + fx])))
                else:
                    kv.append((key, len(p) << 32 | int(cx)))
                    p.extend(int(v) for v in yy[lo + fx: lo + fx + cx])
            if descending:
                kv.reverse()
            f.write(struct.pack("<i", len(p)) + struct.pack(f"<{len(p)}Q", *p))
            f.write(struct.pack("<I", len(kv)) + b"".join(struct.pack("<2Q", *e) for e in kv))
        cat = np.concatenate([code[np.frombuffer(s.tobytes(), dtype=np.uint8)] for s in seqs])
        cat = np.concatenate([cat, np.zeros(-len(cat) % 8, dtype=np.uint32)]).reshape(-1, 8)
        words = (cat << (np.arange(8, dtype=np.uint32) * 4)).sum(axis=1).astype(np.uint32)
        f.write(words.tobytes())


def test_mmi_files_load_and_round_trip(capi, world, tmp_path):
    """aligner.py:45-46 / 59: mappy writes and reads minimap2's .mmi.  An index saved in that format, and one laid out by
    the test from the published description, load as the same index (same native file, byte for byte)."""
    names, seqs, idx, oidx = world
    native = tmp_path / "a.idx"
    idx.save(str(native))
    mmi = tmp_path / "a.mmi"
    idx.save(str(mmi), mmi=True)
    assert open(mmi, "rb").read(4) == b"MMI\x02"
    back = capi.Index.load(str(mmi))
    assert back.info().mid_occ == idx.info().mid_occ and list(back.contig_names) == list(idx.contig_names)
    back.save(str(tmp_path / "b.idx"))
    assert open(tmp_path / "b.idx", "rb").read() == open(native, "rb").read()
    h, y = idx.dump()
    for desc in (False, True):
        other = tmp_path / f"by_hand_{int(desc)}.mmi"
        write_mmi_by_the_published_layout(str(other), names, seqs, h, y, descending=desc)
        if not desc:
            assert open(other, "rb").read() == open(mmi, "rb").read()         # the library's writer: ascending keys
        again = capi.Index.load(str(other))
        again.save(str(tmp_path / "c.idx"))
        assert open(tmp_path / "c.idx", "rb").read() == open(native, "rb").read()


def test_damaged_mmi_files_are_refused(capi, world, tmp_path):
    import struct
    names, seqs, idx, _ = world
    good = tmp_path / "g.mmi"
    idx.save(str(good), mmi=True)
    blob = open(good, "rb").read()
    cases = {"cut in the buckets": blob[: len(blob) // 2], "cut in the sequences": blob[:-8], "cut in the header": blob[:16],
             "hpc flag": blob[:20] + struct.pack("<I", 1) + blob[24:], "no sequences flag": blob[:20] + struct.pack("<I", 2) + blob[24:],
             "other k": blob[:8] + struct.pack("<I", 19) + blob[12:], "no sequence": blob[:16] + struct.pack("<I", 0) + blob[20:]}
    for what, data in cases.items():
        p = tmp_path / "bad.mmi"
        p.write_bytes(data)
        with pytest.raises(capi.MncError):
            capi.Index.load(str(p))
    # an occurrence word that points past its contig
    h, y = idx.dump()
    y2 = y.copy()
    y2[0] = np.uint64(int(y2[0]) & 0xffffffff00000000 | 0x7ffffffe)
    p = tmp_path / "bad2.mmi"
    write_mmi_by_the_published_layout(str(p), names, seqs, h, y2)
    with pytest.raises(capi.MncError):
        capi.Index.load(str(p))


def test_contigs_sketched_in_pieces_equal_the_whole_scan(capi, oracle):
    """aligner.py:45: the index builder sketches long contigs in pieces on all host threads (csrc/index.cpp:
    contig_minimizers).  Ambiguous bases, homopolymer runs and short tandem repeats -- everywhere, so also around the
    piece boundaries at multiples of 2^18 -- must give the minimizers of the oracle's one scan per contig."""
    rng = np.random.default_rng(41)
    seqs = []
    for n in (900_000, 262_144, 262_150, 300):
        s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n)
        for _ in range(n // 150):                                     # runs of N, of one base, of a 2- or 3-mer
            at, kind = int(rng.integers(0, n)), int(rng.integers(0, 4))
            ln = int(rng.integers(1, 60))
            if kind == 0:
                s[at:at + int(rng.integers(1, 4))] = ord("N")
            elif kind == 1:
                s[at:at + ln] = s[at]
            else:
                unit = s[at:at + kind].copy()
                reps = np.tile(unit, ln // kind + 1)[: len(s[at:at + ln])]
                s[at:at + ln] = reps
        for b in range(1 << 18, n, 1 << 18):                          # and the boundaries themselves
            s[b - 30:b - 28] = ord("N")
            s[b + 3:b + 40] = ord("A")
        seqs.append(s)
    names = [f"Genus{i}_species{i}:ACC{i:06d}.1" for i in range(len(seqs))]
    idx = capi.Index.from_seqs(names, seqs)
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    h, y = idx.dump()
    oh, oy = oidx.dump()
    assert len(h) == len(oh) and np.array_equal(h, oh) and np.array_equal(y, oy)
    assert idx.info().mid_occ == oidx.mid_occ
