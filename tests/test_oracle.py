"""CPU tests of the oracle itself: known-answer cases that can be derived by hand
(SURVEY.md section 8c), algebraic properties, the closed-form minimizer rule the GPU kernel
implements, and the committed golden fixture."""
import os

import numpy as np
import pytest

from monica_amd import synth
import util

HERE = os.path.dirname(os.path.abspath(__file__))
M64 = (1 << 64) - 1


# ------------------------------------------------------------------ hash
def py_hash64(key, mask):
    key = (~key + (key << 21)) & mask
    key = key ^ key >> 24
    key = ((key + (key << 3)) + (key << 8)) & mask
    key = key ^ key >> 14
    key = ((key + (key << 2)) + (key << 4)) & mask
    key = key ^ key >> 28
    key = (key + (key << 31)) & mask
    return key


def test_hash64_matches_python_and_is_a_permutation(oracle):
    mask = (1 << 30) - 1
    rng = np.random.default_rng(1)
    keys = [0, 1, mask, 0x155555555 & mask] + [int(x) for x in rng.integers(0, mask + 1, 200)]
    for k in keys:
        assert oracle.hash64(k, mask) == py_hash64(k, mask)
    small = (1 << 12) - 1                       # invertible mix => a permutation of the domain
    img = {oracle.hash64(k, small) for k in range(small + 1)}
    assert len(img) == small + 1


# ------------------------------------------------------------------ sketch
def np_hash30(key):
    mask = np.uint64((1 << 30) - 1)
    key = key.astype(np.uint64)
    key = (~key + (key << np.uint64(21))) & mask
    key = key ^ (key >> np.uint64(24))
    key = ((key + (key << np.uint64(3))) + (key << np.uint64(8))) & mask
    key = key ^ (key >> np.uint64(14))
    key = ((key + (key << np.uint64(2))) + (key << np.uint64(4))) & mask
    key = key ^ (key >> np.uint64(28))
    return key


def closed_form_minimizers(seq, k=15, w=10):
    """The rule the GPU sketch kernel evaluates per position, for ACGT-only sequences
    (DESIGN.md section 4, K1).  Returns (hash, pos<<1|strand) in increasing position."""
    code = np.zeros(256, dtype=np.int64)
    for c, v in zip(b"ACGT", range(4)):
        code[c] = v
    b = code[np.frombuffer(seq, dtype=np.uint8)]
    n = len(b) - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64)
    fw = np.zeros(n, dtype=np.uint64)
    rv = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        fw = (fw << np.uint64(2)) | b[j:j + n].astype(np.uint64)
        rv = rv | ((3 - b[j:j + n]).astype(np.uint64) << np.uint64(2 * j))
    strand = (fw >= rv).astype(np.uint64)
    assert (fw != rv).all()
    h = np_hash30(np.where(strand == 1, rv, fw)).astype(np.int64)
    emit = np.zeros(n, dtype=bool)
    if n < w:
        m = h.min()
        emit[np.nonzero(h == m)[0].max()] = True
    else:
        for p in range(n):
            L = 0
            q = p - 1
            while q >= 0 and p - q < w and h[q] >= h[p]:
                L += 1
                q -= 1
            R = 0
            q = p + 1
            while q < n and q - p < w and h[q] >= h[p]:
                R += 1
                q += 1
            emit[p] = L + R + 1 >= w
        m1 = h[:w - 1].min()
        P1 = np.nonzero(h[:w - 1] == m1)[0].max()
        for p in range(w - 1):
            if h[p] == m1 and p != P1:
                emit[p] = True
        if h[w - 1] == m1:
            emit[P1] = False
    pos = np.nonzero(emit)[0]
    return h[pos].astype(np.uint64), ((pos + k - 1).astype(np.uint64) << np.uint64(1)) | strand[pos]


def low_complexity(rng, length):
    """Sequences rich in repeated k-mers inside one window: tandem repeats with mutations."""
    parts, total = [], 0
    while total < length:
        kind = rng.integers(0, 4)
        if kind == 0:
            s = util.ACGT[rng.integers(0, 4, rng.integers(5, 60))]
        else:
            unit = util.ACGT[rng.integers(0, 4, rng.integers(1, 9))]
            s = np.tile(unit, rng.integers(3, 30))
            if kind == 3 and len(s) > 4:
                s = s.copy()
                s[rng.integers(0, len(s))] = util.ACGT[rng.integers(0, 4)]
        parts.append(s)
        total += len(s)
    return np.concatenate(parts)[:length].tobytes()


def test_closed_form_minimizer_rule_equals_state_machine(oracle):
    rng = np.random.default_rng(5)
    cases = [b"A" * 40, b"AC" * 30, b"ACG" * 25, b"ACGTTGCAGT" * 8, b"A" * 24, b"A" * 25, b"A" * 23,
             b"ACGTACGTACGTACGTACGTACGTAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAACGT"]
    for _ in range(300):
        cases.append(low_complexity(rng, int(rng.integers(15, 400))))
    for _ in range(30):
        cases.append(util.ACGT[rng.integers(0, 4, int(rng.integers(15, 3000)))].tobytes())
    n_ties = 0
    for s in cases:
        want = oracle.sketch(s)
        h, y = closed_form_minimizers(s)
        assert len(h) == len(want), (s[:80], len(h), len(want))
        assert np.array_equal(h, want["x"] >> np.uint64(8)), s[:80]
        assert np.array_equal(y, want["y"]), s[:80]
        assert (np.diff(want["y"].astype(np.int64) >> 1) > 0).all()     # increasing position
        n_ties += int(len(np.unique(h)) < len(h))
    assert n_ties > 50            # the corpus really exercises equal hashes


def test_minimizer_density_and_window_guarantee(oracle):
    rng = np.random.default_rng(9)
    s = util.ACGT[rng.integers(0, 4, 50_000)].tobytes()
    mz = oracle.sketch(s)
    assert abs(len(mz) / (len(s) - 14) - 2 / 11) < 0.01         # density ~ 2/(w+1)
    pos = (mz["y"].astype(np.int64) & 0xffffffff) >> 1
    assert np.diff(pos).max() <= 10                              # no window without a minimizer


def test_reverse_complement_gives_same_hash_multiset(oracle):
    rng = np.random.default_rng(3)
    a = util.ACGT[rng.integers(0, 4, 4000)]
    f = oracle.sketch(a.tobytes())
    r = oracle.sketch(util.revcomp(a).tobytes())
    # interior minimizers agree as a multiset (the two ends see different first windows)
    hf, hr = np.sort(f["x"]), np.sort(r["x"])
    common = np.intersect1d(hf, hr)
    assert len(common) >= len(hf) - 4 and len(common) >= len(hr) - 4


def test_ambiguous_base_resets_the_run(oracle):
    g = synth.genome(77, 300).tobytes()
    clean = oracle.sketch(g)
    s = bytearray(g)
    s[150] = ord("N")
    broken = oracle.sketch(bytes(s))
    pos = (broken["y"].astype(np.int64) & 0xffffffff) >> 1
    assert not ((pos >= 150) & (pos < 150 + 15)).any()           # no k-mer spans the N
    assert len(broken) < len(clean) + 2
    assert oracle.sketch(b"N" * 100).size == 0
    assert oracle.sketch(b"").size == 0
    lower = oracle.sketch(g.lower())
    assert np.array_equal(lower["x"], clean["x"])


# ------------------------------------------------------------------ best_hit truth table (SURVEY 8c i)
def test_best_hit_truth_table(oracle):
    c, d, e = 0, 1, 2
    assert oracle.best_hit([(1, 10), (1, 10)]) == -1
    assert oracle.best_hit([(1, 10), (2, 10)]) == c
    assert oracle.best_hit([(2, 10), (1, 10)]) == d
    assert oracle.best_hit([(2, 10), (1, 10), (1, 10)]) == -1
    assert oracle.best_hit([(1, 10), (1, 10), (1, 20)]) == e
    assert oracle.best_hit([(3, 7)]) == 0
    assert oracle.best_hit([(2, 20), (1, 10)]) == -1             # equal rationals, different terms
    assert oracle.best_hit([(0, 50), (0, 70)]) == -1


# ------------------------------------------------------------------ mapping properties
@pytest.fixture(scope="module")
def small_world(oracle):
    names, seqs = util.small_genomes(4, 120_000, 160_000)
    return names, seqs, oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])


def test_error_free_read_maps_to_its_contig(oracle, small_world):
    names, seqs, oidx = small_world
    for g in range(2):
        read = seqs[g][30_000:34_000]
        for s in (read.tobytes(), util.revcomp(read).tobytes()):
            regs = oidx.map(s)
            pri = regs[regs["id"] == regs["parent"]]
            assert len(pri) == 1 and pri[0]["rid"] == g and pri[0]["mapq"] == 60
            assert pri[0]["score"] > 3900 and pri[0]["qs"] <= 20 and pri[0]["qe"] >= 3980
            assert pri[0]["mlen"] <= pri[0]["blen"]
    rf = oidx.map(seqs[0][30_000:34_000].tobytes())
    rr = oidx.map(util.revcomp(seqs[0][30_000:34_000]).tobytes())
    assert rf[0]["rev"] == 0 and rr[0]["rev"] == 1 and rf[0]["score"] == rr[0]["score"]


def test_chimera_yields_two_primaries_and_long_deletion_joins(oracle, small_world):
    names, seqs, oidx = small_world
    chim = np.concatenate([seqs[0][10_000:12_500], seqs[1][50_000:52_500]]).tobytes()
    regs = oidx.map(chim)
    pri = regs[regs["id"] == regs["parent"]]
    assert sorted(pri["rid"].tolist()) == [0, 1]
    dele = np.concatenate([seqs[0][60_000:62_500], seqs[0][63_500:66_000]]).tobytes()
    regs = oidx.map(dele)
    pri = regs[regs["id"] == regs["parent"]]
    assert len(pri) == 1 and pri[0]["cnt"] > 500 and pri[0]["re"] - pri[0]["rs"] > 5_900   # joined


def test_counts_properties(oracle, small_world):
    names, seqs, oidx = small_world
    bases, offsets, truth = synth.reads(seqs, 150, 3000, seed=321)
    assign, best, nhits, flat = oidx.classify(bases, offsets, 60)
    mapped = assign >= 0
    assert (assign[truth < 0] == oracle.UNMAPPED).all()           # pure-random reads stay unmapped
    assert (assign[mapped] == truth[mapped]).mean() > 0.97
    assert flat.size == nhits.sum()
    assert (best["mlen"][mapped] > 0).all() and (best["mapq"][mapped] >= 60).all()
    a2, _, _, _ = oidx.classify(bases, offsets, 60, n_threads=4)  # threads do not change results
    assert np.array_equal(a2, assign)
    a0, _, nh0, _ = oidx.classify(bases, offsets, 0)
    assert (nh0 >= nhits).all()


# ------------------------------------------------------------------ golden fixture
def test_oracle_reproduces_golden_fixture(oracle):
    g = np.load(os.path.join(HERE, "golden", "small_case.npz"))
    lens = g["genome_lens"]
    seqs, o = [], 0
    for L in lens:
        seqs.append(g["genome_bytes"][o:o + L])
        o += L
    oidx = oracle.Index.from_seqs([str(x) for x in g["names"]], [s.tobytes() for s in seqs])
    assert oidx.mid_occ == int(g["mid_occ"]) and oidx.n_keys == int(g["n_keys"]) and oidx.n_minimizers == int(g["n_occ"])
    ih, iy = oidx.dump()
    assert int(ih.sum(dtype=np.uint64)) == int(g["index_hash_sum"]) and int(iy.sum(dtype=np.uint64)) == int(g["index_y_sum"])
    raw, offs = g["bases"].tobytes(), g["offsets"]
    # ---- with base-level alignment (the default, as in mappy)
    assign, best, nhits, flat = oidx.classify(g["bases"], g["offsets"], 60)
    assert np.array_equal(assign, g["dp_assign"]) and np.array_equal(nhits, g["dp_nhits"])
    for k in oracle.HIT_DTYPE.names:
        assert np.array_equal(best[k], g["dp_best"][k]) and np.array_equal(flat[k], g["dp_hits"][k])
    regs, cig, k, kc = g["dp_regs"], g["dp_cigars"], 0, 0
    for r in range(len(offs) - 1):
        got, cigs = oidx.map_cigar(raw[offs[r]:offs[r + 1]])
        assert len(got) == g["dp_reg_cnt"][r]
        for name in oracle.REG_DTYPE.names:
            assert np.array_equal(got[name], regs[name][k:k + len(got)]), (r, name)
        for c in cigs:
            want = [(int(x) >> 4, "MID"[int(x) & 0xf]) for x in cig[kc:kc + len(c)]]
            assert c == want, r
            kc += len(c)
        k += len(got)
    assert kc == len(cig)
    # ---- chain level
    oidx.opt.cigar = 0
    assign, best, nhits, flat = oidx.classify(g["bases"], g["offsets"], 60)
    assert np.array_equal(assign, g["assign"]) and np.array_equal(nhits, g["nhits"])
    for k in oracle.HIT_DTYPE.names:
        assert np.array_equal(best[k], g["best"][k]) and np.array_equal(flat[k], g["hits"][k])
    regs, k = g["regs"], 0
    for r in range(len(offs) - 1):
        got = oidx.map(raw[offs[r]:offs[r + 1]])
        assert len(got) == g["reg_cnt"][r]
        for name in g["regs"].dtype.names:
            assert np.array_equal(got[name], regs[name][k:k + len(got)]), (r, name)
        k += len(got)


def test_the_repeats_genome_model_is_a_function_of_its_seeds():
    """`synth.genome_set_repeats` (bench.py --genome-model repeats): same lengths as the i.i.d. set, a few per cent of the
    bases replaced by repeated elements, identical on every call, the diverged second half 3 % from the first."""
    from monica_amd import synth
    names, seqs = synth.genome_set_repeats(4, min_len=200_000, max_len=260_000)
    _, plain = synth.genome_set(4, min_len=200_000, max_len=260_000)
    assert [len(s) for s in seqs] == [len(s) for s in plain]
    assert all(0.02 < (a != b).mean() < 0.25 for a, b in zip(seqs[:2], plain[:2]))
    again = synth.genome_set_repeats(4, min_len=200_000, max_len=260_000)[1]
    assert all(np.array_equal(a, b) for a, b in zip(seqs, again))
    assert 0.02 < (seqs[2] != seqs[0]).mean() < 0.04                  # point-diverged copy of genome 0, elements included
    # an operon copy occurs more than once in a genome: some 31-mer of the genome is repeated
    g = seqs[0].tobytes()
    seen, repeated = set(), 0
    for i in range(0, len(g) - 31, 7):
        k = g[i:i + 31]
        repeated += k in seen
        seen.add(k)
    assert repeated > 100
