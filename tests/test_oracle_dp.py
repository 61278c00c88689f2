"""The oracle's base-level alignment stage (oracle/mm_ksw.c, oracle/mm_align.c): what mappy 2.17
always runs inside index.map() and monica reads as hit.mapq / hit.NM / hit.mlen
(monica/genomes/aligner.py:194-195, 216-217).

PARITY UNPINNED at the mappy boundary: these tests pin the restatement with (i) two independent
formulations of ksw2's recurrence that must agree cell for cell (the literal SSE-layout
simulation in difference form vs. a plain absolute-score DP), and (ii) hand-derivable known
answers (exact read: NM 0, mlen = len; one substitution: NM 1; a 10-base deletion: one 10D,
NM 10; ...)."""
import numpy as np
import pytest

from monica_amd import synth
import util


def _mutate(rng, t, rate):
    out = []
    for c in t:
        r = rng.random()
        if r < rate * 0.4:
            out.append((c + rng.integers(1, 4)) % 4)
        elif r < rate * 0.7:
            out.append(c)
            out.append(rng.integers(0, 4))
        elif r < rate:
            pass
        else:
            out.append(c)
    return np.array(out if out else [0], dtype=np.uint8)


def test_literal_ksw2_simulation_equals_plain_dp(oracle):
    """Whenever the band does not clip the matrix, the difference-form kernel (int8 lanes, 16-wide
    rounding, in-place updates) must compute the textbook two-piece affine DP: scores, maxima
    with their tie order, Z-drop decisions, CIGARs -- in every mode minimap2 uses it."""
    o = oracle
    rng = np.random.default_rng(7)
    modes = (o.EZ_APPROX_MAX, 0, o.EZ_EXTZ_ONLY, o.EZ_EXTZ_ONLY | o.EZ_RIGHT | o.EZ_REV_CIGAR, o.EZ_RIGHT)
    for it in range(600):
        L = int(rng.integers(1, 120)) if it % 3 else int(rng.integers(1, 12))
        kind = it % 5
        if kind == 0:
            t = rng.integers(0, 4, L)
        elif kind == 1:
            t = np.tile(rng.integers(0, 4, int(rng.integers(1, 4))), L)[:L]        # low complexity
        elif kind == 2:
            t = rng.integers(0, 2, L)
        else:
            t = rng.integers(0, 5, L) if it % 7 == 0 else rng.integers(0, 4, L)     # with ambiguous bases
        t = t.astype(np.uint8)
        q = _mutate(rng, t, rng.choice([0.0, 0.05, 0.15, 0.4])) if it % 4 else rng.integers(0, 4, int(rng.integers(1, 120))).astype(np.uint8)
        if it % 11 == 0:
            k = int(rng.integers(0, len(q) + 1))
            q = np.concatenate([q[:k], rng.integers(0, 4, int(rng.integers(5, 60))).astype(np.uint8), q[k:]])
        for flag in modes:
            for zd in (400, 20):
                a = o.ksw_extd2(q, t, w=751, zdrop=zd, flag=flag)
                b = o.dp_clean(q, t, zdrop=zd, flag=flag)
                assert a == b, (it, flag, zd, q.tolist(), t.tolist())


def test_ksw2_known_answers(oracle):
    o = oracle
    t = np.array([0, 1, 2, 3] * 10, dtype=np.uint8)
    r = o.ksw_extd2(t, t)
    assert r["score"] == 80 and r["cigar"] == [(40, "M")] and r["max"] == 80 and (r["max_t"], r["max_q"]) == (39, 39)
    q = t.copy(); q[20] ^= 1
    r = o.ksw_extd2(q, t)
    assert r["score"] == 80 - 2 - 4 and r["cigar"] == [(40, "M")]
    rng = np.random.default_rng(3)
    t = rng.integers(0, 4, 120).astype(np.uint8)
    r = o.ksw_extd2(np.delete(t, slice(50, 53)), t)                 # 3 target bases missing from the query
    assert r["score"] == 117 * 2 - (4 + 2 * 3) and sum(l for l, op in r["cigar"] if op == "D") == 3
    r = o.ksw_extd2(np.delete(t, slice(40, 80)), t)                 # a long gap takes the second cost: 24 + 1 * 40
    assert r["score"] == 80 * 2 - (24 + 40) and (40, "D") in r["cigar"]
    # extension: stops at the maximum; junk after the match costs nothing
    q = np.concatenate([t[:60], (t[60:] + 1) % 4])
    r = o.ksw_extd2(q, t, flag=o.EZ_EXTZ_ONLY)
    assert r["max"] == 120 and (r["max_t"], r["max_q"]) == (59, 59) and r["cigar"] == [(60, "M")] and not r["reach_end"]
    # Z-drop: a long stretch of mismatches ends an extension early
    junk = rng.integers(0, 4, 2000).astype(np.uint8)
    r = o.ksw_extd2(np.concatenate([t, junk]), np.concatenate([t, rng.integers(0, 4, 2000).astype(np.uint8)]), zdrop=100, flag=o.EZ_EXTZ_ONLY)
    assert r["zdropped"] == 1 and r["max"] >= 240 and r["max_t"] >= 119
    # the band: anti-diagonals beyond w from the main diagonal do not exist
    r = o.ksw_extd2(t[:100], np.concatenate([t[:50], rng.integers(0, 4, 300).astype(np.uint8), t[50:100]]), w=20)
    assert r["zdropped"] == 1                                       # the corner is outside the band
    assert o.lib().orc_local_score(40, t[:40].ctypes.data, 120, t.ctypes.data, o.simple_mat().ctypes.data, 4, 2) == 80


@pytest.fixture(scope="module")
def world(oracle):
    names, seqs = util.small_genomes()
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    return names, seqs, oidx


def _one(oidx, read):
    regs, cigs = oidx.map_cigar(np.asarray(read, dtype=np.uint8).tobytes())
    return regs, cigs


def test_pipeline_known_answers(oracle, world):
    names, seqs, oidx = world
    g0, g1 = seqs[0], seqs[1]
    for read, rev in ((g0[5000:8000], 0), (util.revcomp(g0[5000:8000]), 1)):
        regs, cigs = _one(oidx, read)                                # an error-free read aligns end to end
        assert len(regs) == 1 and cigs[0] == [(3000, "M")]
        r = regs[0]
        assert (r["rev"], r["qs"], r["qe"], r["rs"], r["re"]) == (rev, 0, 3000, 5000, 8000)
        assert (r["mlen"], r["blen"], r["n_ambi"], r["dp_max"], r["dp_max2"], r["mapq"]) == (3000, 3000, 0, 6000, 0, 60)
    read = g0[5000:8000].copy()
    read[1500] = ord("A") if read[1500] != ord("A") else ord("C")
    regs, cigs = _one(oidx, read)                                    # one substitution: NM 1
    assert cigs[0] == [(3000, "M")] and regs[0]["blen"] - regs[0]["mlen"] == 1 and regs[0]["dp_max"] == 6000 - 6
    regs, cigs = _one(oidx, np.concatenate([g0[5000:6500], g0[6510:8000]]))   # 10 bases deleted: one gap, NM 10
    assert [op for _, op in cigs[0]] == ["M", "D", "M"] and cigs[0][1] == (10, "D")
    assert regs[0]["blen"] - regs[0]["mlen"] == 10 and regs[0]["mlen"] == 2990
    assert regs[0]["dp_max"] == 2 * 2990 - (4 + 2 * 10)
    regs, cigs = _one(oidx, np.concatenate([g0[5000:6500], np.frombuffer(b"ACGTACGTAC", dtype=np.uint8), g0[6500:8000]]))
    assert cigs[0][1] == (10, "I") and regs[0]["blen"] - regs[0]["mlen"] == 10 and regs[0]["mlen"] == 3000
    read = g0[20000:24000].copy(); read[1500] = ord("N")
    regs, cigs = _one(oidx, read)                                    # an ambiguous base: in neither mlen nor blen, but in NM
    r = regs[0]
    assert (r["mlen"], r["blen"], r["n_ambi"]) == (3999, 3999, 1) and r["dp_max"] == 2 * 3999 - 1
    regs, cigs = _one(oidx, np.concatenate([g0[100000:102500], g0[103500:106000]]))   # 1 kb deletion: long-join, one region
    assert len(regs) == 1 and cigs[0] == [(2500, "M"), (1000, "D"), (2500, "M")]
    assert regs[0]["mlen"] == 5000 and regs[0]["dp_max"] == 2 * 5000 - (4 + 2 * 1000)
    regs, cigs = _one(oidx, np.concatenate([g0[70000:72500], g1[90000:92500]]))       # chimera: two primaries
    assert len(regs) == 2 and all(r["id"] == r["parent"] and r["mapq"] == 60 for r in regs)
    assert {int(r["rid"]) for r in regs} == {0, 1}
    regs, cigs = _one(oidx, util.ACGT[np.random.default_rng(5).integers(0, 4, 4000)])
    assert len(regs) == 0


def test_region_invariants_on_noisy_reads(oracle, world):
    """CIGAR consumes exactly [qs, qe) x [rs, re); mlen / blen / n_ambi are what a walk over it
    gives; a 10 %-error read keeps ~90 % identity; classification matches the truth."""
    names, seqs, oidx = world
    bases, offsets, truth = synth.reads(seqs, 60, 4000, seed=11)
    raw = bases.tobytes()
    n_primary = 0
    for r in range(60):
        s = raw[offsets[r]:offsets[r + 1]]
        regs, cigs = oidx.map_cigar(s)
        for g, c in zip(regs, cigs):
            assert g["flags"] & 1 and g["n_cigar"] == len(c) and len(c) > 0
            assert c[0][1] == "M" and c[-1][1] == "M"
            assert all(c[i][1] != c[i + 1][1] for i in range(len(c) - 1))
            qspan = sum(l for l, op in c if op in "MI")
            tspan = sum(l for l, op in c if op in "MD")
            assert qspan == g["qe"] - g["qs"] and tspan == g["re"] - g["rs"]
            q = np.frombuffer(s, dtype=np.uint8)
            q = util.revcomp(q)[len(q) - g["qe"]:len(q) - g["qs"]] if g["rev"] else q[g["qs"]:g["qe"]]
            t = seqs[g["rid"]][g["rs"]:g["re"]]
            qi = ti = m = b = 0
            for l, op in c:
                if op == "M":
                    m += int((q[qi:qi + l] == t[ti:ti + l]).sum()); b += l; qi += l; ti += l
                elif op == "I":
                    b += l; qi += l
                else:
                    b += l; ti += l
            assert (m, b) == (g["mlen"], g["blen"])
            if g["id"] == g["parent"]:
                n_primary += 1
                if truth[r] >= 0:
                    assert g["mlen"] / g["blen"] > 0.8 and g["dp_max"] > 2000
    assert n_primary >= 55
    assign, best, nhits, flat = oidx.classify(bases, offsets, 60)
    mapped = assign >= 0
    assert mapped.sum() >= 55 and (assign[mapped] == truth[mapped]).all()
    assert (best["nm"][mapped] < 0.2 * best["mlen"][mapped]).all()            # NM is an edit distance now


def test_chain_level_contract_is_still_available(oracle, world):
    names, seqs, oidx = world
    bases, offsets, truth = synth.reads(seqs, 40, 3000, seed=12)
    dp = oidx.classify(bases, offsets, 60)
    oidx.opt.cigar = 0
    try:
        ch = oidx.classify(bases, offsets, 60)
        regs = oidx.map(bases[:3000].tobytes())
    finally:
        oidx.opt.cigar = 1
    assert (regs["flags"] == 0).all() and (regs["dp_max"] == 0).all()
    assert (dp[0] == ch[0]).mean() > 0.9
    m = (dp[0] >= 0) & (ch[0] >= 0)
    assert (dp[1]["mlen"][m] > ch[1]["mlen"][m]).all()                      # exact matches vs. seed matches


def test_striped_local_alignment_equals_the_plain_recurrence(oracle):
    """ksw_ll_i16 literally (eight int16 lanes, striped query, lazy F, unsigned saturation as the floor) against plain
    Smith-Waterman on the padded query with the striped layout's tie rules restated: score, end of the query, end of
    the target -- the three numbers mm_align1_inv takes from it.  Random, related and low-complexity pairs."""
    import ctypes as C
    L = oracle.lib()
    mat = np.zeros(25, dtype=np.int8)
    L.orc_gen_simple_mat(5, mat.ctypes.data_as(C.c_void_p), C.c_int8(2), C.c_int8(4), C.c_int8(1))
    rng = np.random.default_rng(11)

    def both(q, t):
        out = []
        for f, extra in ((L.orc_ksw_ll_i16, (5,)), (L.orc_local_end, ())):
            qe, te = C.c_int(), C.c_int()
            args = [len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p)] + list(extra) + \
                   [mat.ctypes.data_as(C.c_void_p), 4, 2, C.byref(qe), C.byref(te)]
            out.append((f(*args), qe.value, te.value))
        return out

    in_padding = 0
    for it in range(6000):
        ql, tl = int(rng.integers(1, 150)), int(rng.integers(1, 150))
        base = rng.integers(0, 4, max(ql, tl) + 40).astype(np.uint8)
        if it % 4 == 0:
            q, t = rng.integers(0, 4, ql).astype(np.uint8), rng.integers(0, 4, tl).astype(np.uint8)
        else:
            o1, o2 = int(rng.integers(0, 20)), int(rng.integers(0, 20))
            q, t = base[o1:o1 + ql].copy(), base[o2:o2 + tl].copy()
            for arr in (q, t):
                k = rng.random(len(arr)) < 0.05 * (it % 4)
                arr[k] = rng.integers(0, 5, int(k.sum()))
            if it % 8 == 3:
                q, t = q % 2, t % 2                                  # two letters: ties everywhere
        a, b = both(np.ascontiguousarray(q), np.ascontiguousarray(t))
        assert a == b, (ql, tl, a, b)
        assert a[0] == L.orc_local_score(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p),
                                         mat.ctypes.data_as(C.c_void_p), 4, 2)
        in_padding += a[1] >= ql
    assert in_padding > 100                                       # the quirk of the layout is exercised: an end among the padding positions


def test_inversion_between_the_halves_of_a_split_region(oracle):
    """mm_align1_inv: a read whose middle block is inverted.  The chain runs over the block, the gap filling there
    Z-drops with a positive inversion test, the region splits -- and the stretch between the halves is aligned on the
    other strand: a third region, flagged, with MAPQ 0, covering the block on query and target."""
    import util
    names, seqs = util.small_genomes()
    idx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    idx.opt.cigar = 1
    g0 = seqs[0]
    for inv_len in (300, 700, 1500):
        read = np.concatenate([g0[40000:42000], util.revcomp(g0[42000:42000 + inv_len]), g0[42000 + inv_len:44000 + inv_len]])
        regs, cigs = idx.map_cigar(read.tobytes())
        inv = [i for i in range(len(regs)) if regs["flags"][i] & 16]
        assert len(regs) == 3 and len(inv) == 1
        i = inv[0]
        assert regs["rev"][i] == 1 and regs["mapq"][i] == 0 and regs["cnt"][i] == 0 and regs["score"][i] == 0
        assert abs(int(regs["qs"][i]) - 2000) <= 15 and abs(int(regs["qe"][i]) - (2000 + inv_len)) <= 15
        assert abs(int(regs["rs"][i]) - 42000) <= 15 and abs(int(regs["re"][i]) - (42000 + inv_len)) <= 15
        assert regs["dp_max"][i] >= 2 * inv_len - 40 and regs["mlen"][i] >= inv_len - 20
        assert sum(l for l, op in cigs[i] if op in "MI") == regs["qe"][i] - regs["qs"][i]
        others = [k for k in range(3) if k != i]
        assert all(regs["mapq"][k] == 60 and regs["flags"][k] & 6 for k in others)      # the two halves, split
    # the same block on the forward strand: no split, no inversion region
    read = np.concatenate([g0[40000:42000], g0[42000:42700], g0[42700:44700]])
    regs, _ = idx.map_cigar(read.tobytes())
    assert len(regs) == 1 and regs["flags"][0] == 1
