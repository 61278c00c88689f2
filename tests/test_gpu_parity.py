"""GPU parity: every stage of the HIP path against the CPU oracle, through the C-ABI.

Bit-exact is the bar: minimizers, anchors, chaining scores/back-pointers, regions (every
field), per-read decisions, gated hit lists, taxon counts."""
import numpy as np
import pytest

from monica_amd import synth
import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(capi, oracle):
    names, seqs = util.small_genomes()
    idx = capi.Index.from_seqs(names, seqs)
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    eng = capi.Engine(idx, 0)
    return dict(names=names, seqs=seqs, idx=idx, oidx=oidx, eng=eng)


def _compare_dp(capi, oracle, world, bases, offsets, min_mapq=60):
    """Base-level alignment on (the default contract, what mappy computes): every region field,
    every CIGAR, gated hits and decisions against the oracle."""
    eng, oidx = world["eng"], world["oidx"]
    eng.set_contract(capi.CONTRACT_DP)
    oidx.opt.cigar = 1
    n = len(offsets) - 1
    assign, best, nhits = eng.classify(bases, offsets, min_mapq)
    regs = eng.dump(capi.DUMP_REGS, capi.REG_DTYPE)
    reg_off = eng.dump(capi.DUMP_REG_OFFSETS, np.int64)
    cigs = eng.cigars()
    hit_off, hits = eng.fetch_hits()
    raw = bases.tobytes()
    for r in range(n):
        oregs, ocigs = oidx.map_cigar(raw[offsets[r]:offsets[r + 1]])
        gr = regs[reg_off[r]:reg_off[r + 1]]
        assert len(gr) == len(oregs), f"read {r}: region count {len(gr)} != {len(oregs)} after base-level alignment"
        for name in capi.REG_DTYPE.names:
            assert np.array_equal(gr[name], oregs[name]), f"read {r}: region field {name}: {gr[name]} != {oregs[name]}"
        assert cigs[reg_off[r]:reg_off[r + 1]] == ocigs, f"read {r}: CIGAR"
    oassign, obest, onh, oflat = oidx.classify(bases, offsets, min_mapq)
    assert np.array_equal(assign, oassign)
    assert np.array_equal(nhits, onh)
    for name in capi.HIT_DTYPE.names:
        assert np.array_equal(best[name], obest[name]), name
        assert np.array_equal(hits[name], oflat[name]), name
    assert np.array_equal(np.diff(hit_off), onh)
    return assign, best, nhits


def _compare_batch(capi, oracle, world, bases, offsets, min_mapq=60, dp=True):
    """Stage by stage at the chain level, then -- with `dp` -- the complete path with base-level
    alignment; returns the decisions of the latter (the default contract)."""
    out = _compare_chain(capi, oracle, world, bases, offsets, min_mapq)
    if dp:
        out = _compare_dp(capi, oracle, world, bases, offsets, min_mapq)
    return out


def _compare_chain(capi, oracle, world, bases, offsets, min_mapq=60):
    eng, oidx = world["eng"], world["oidx"]
    eng.set_contract(capi.CONTRACT_CHAIN)
    oidx.opt.cigar = 0
    n = len(offsets) - 1
    assign, best, nhits = eng.classify(bases, offsets, min_mapq)
    mz = eng.dump(capi.DUMP_MINIMIZERS, capi.MZ_DTYPE)
    mz_off = eng.dump(capi.DUMP_MZ_OFFSETS, np.int64)
    an = eng.dump(capi.DUMP_ANCHORS, capi.ANCHOR_DTYPE)
    an_off = eng.dump(capi.DUMP_AN_OFFSETS, np.int64)
    f = eng.dump(capi.DUMP_CHAIN_F, np.int32)
    p = eng.dump(capi.DUMP_CHAIN_P, np.int32)
    v = eng.dump(capi.DUMP_CHAIN_V, np.int32)
    regs = eng.dump(capi.DUMP_REGS, capi.REG_DTYPE)
    reg_off = eng.dump(capi.DUMP_REG_OFFSETS, np.int64)
    rep = eng.dump(capi.DUMP_REP_LEN, np.int32)
    hit_off, hits = eng.fetch_hits()
    raw = bases.tobytes()
    for r in range(n):
        s = raw[offsets[r]:offsets[r + 1]]
        # K1
        omz = oracle.sketch(s)
        got = mz[mz_off[r]:mz_off[r + 1]]
        assert len(got) == len(omz), f"read {r}: minimizer count {len(got)} != {len(omz)}"
        assert np.array_equal(got["hash"].astype(np.uint64), omz["x"] >> np.uint64(8)), f"read {r}: hashes"
        assert np.array_equal(got["pos_strand"].astype(np.uint64), omz["y"] & np.uint64(0xffffffff)), f"read {r}: positions"
        # K2/K3
        oa, orep = oidx.seeds(s)
        ga = an[an_off[r]:an_off[r + 1]]
        assert rep[r] == orep, f"read {r}: rep_len {rep[r]} != {orep}"
        assert len(ga) == len(oa), f"read {r}: anchor count {len(ga)} != {len(oa)}"
        assert np.array_equal(ga["x"], oa["x"]) and np.array_equal(ga["y"], oa["y"]), f"read {r}: anchors"
        # K4
        _, of, op, ov, ou, ob = oidx.chain(s)
        assert np.array_equal(f[an_off[r]:an_off[r + 1]], of), f"read {r}: chain f"
        assert np.array_equal(p[an_off[r]:an_off[r + 1]], op), f"read {r}: chain p"
        if len(ga) <= 2560 and len(s) < 65536:      # larger reads: the sequential backtrack reuses v[] as its list, as minimap2 does
            assert np.array_equal(v[an_off[r]:an_off[r + 1]], ov), f"read {r}: chain v"
        # K5/K6
        oregs = oidx.map(s)
        gr = regs[reg_off[r]:reg_off[r + 1]]
        assert len(gr) == len(oregs), f"read {r}: region count {len(gr)} != {len(oregs)}"
        for name in capi.REG_DTYPE.names:
            assert np.array_equal(gr[name], oregs[name]), f"read {r}: region field {name}: {gr[name]} != {oregs[name]}"
    oassign, obest, onh, oflat = oidx.classify(bases, offsets, min_mapq)
    assert np.array_equal(assign, oassign)
    assert np.array_equal(nhits, onh)
    for name in capi.HIT_DTYPE.names:
        assert np.array_equal(best[name], obest[name]), name
        assert np.array_equal(hits[name], oflat[name]), name
    assert np.array_equal(np.diff(hit_off), onh)
    return assign, best, nhits


def test_edge_cases(capi, oracle, world):
    rng = np.random.default_rng(7)
    reads = util.edge_reads(world["seqs"], rng)
    bases, offsets = util.pack_reads(reads)
    assign, best, nhits = _compare_batch(capi, oracle, world, bases, offsets)
    assert assign[0] == capi.UNMAPPED and assign[1] == capi.UNMAPPED
    assert assign[6] == 0 and assign[7] == 0            # error-free reads hit their contig


def test_synthetic_reads(capi, oracle, world):
    bases, offsets, truth = synth.reads(world["seqs"], 400, 5000, seed=0x5EED + 1)
    assign, best, nhits = _compare_batch(capi, oracle, world, bases, offsets)
    mapped = assign >= 0
    assert (assign[mapped] == truth[mapped]).mean() > 0.98
    assert (assign[truth < 0] == capi.UNMAPPED).all()


def test_ragged_lengths_and_low_mapq(capi, oracle, world):
    rng = np.random.default_rng(11)
    full, offs, _ = synth.reads(world["seqs"], 120, 6000, seed=99)
    reads = []
    for r in range(120):
        L = int(rng.integers(1, 6000))
        reads.append(full[offs[r]:offs[r] + L])
    bases, offsets = util.pack_reads(reads)
    _compare_batch(capi, oracle, world, bases, offsets, min_mapq=0)
    _compare_batch(capi, oracle, world, bases, offsets, min_mapq=30)


def test_empty_batch(capi, world):
    assign, best, nhits = world["eng"].classify(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64))
    assert len(assign) == 0


def test_device_counts_match_host_counts(capi, oracle, world):
    import torch
    bases, offsets, truth = synth.reads(world["seqs"], 300, 3000, seed=5)
    eng, idx = world["eng"], world["idx"]
    dev = torch.device("cuda:0")
    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(offsets).to(dev)
    d_assign = torch.empty(len(truth), dtype=torch.int32, device=dev)
    d_best = torch.zeros(len(truth) * 4, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(len(idx.genome_names) * 3, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.classify_device(d_bases.data_ptr(), d_off.data_ptr(), len(truth), int(offsets[-1]), 3000, 60,
                        d_assign.data_ptr(), d_best.data_ptr(), 0, d_counts.data_ptr())
    eng.sync()
    assign = d_assign.cpu().numpy()
    best = d_best.cpu().numpy().view(capi.HIT_DTYPE)
    got = d_counts.cpu().numpy().reshape(-1, 3)
    for mode in (1, 2, 3):
        want = capi.counts(idx, assign, best, offsets, mode)
        assert np.array_equal(got[:, mode - 1], want)
    oassign, obest, _, _ = world["oidx"].classify(bases, offsets, 60)
    assert np.array_equal(assign, oassign)
    assert got[:, 0].sum() == (oassign >= 0).sum()
    assert got[:, 1].sum() == 3000 * (oassign >= 0).sum()


def test_index_sharded_merge_equals_multi_part_reference(capi, oracle):
    """BASELINE config 4 on one GPU: two index parts (a genome and its diverged copy in
    different parts), per-part summaries merged like dist.gather_and_merge does across
    ranks, against the reference's multi-part loop evaluated with the oracle."""
    import torch
    from monica_amd import aligner, dist as mdist
    names, seqs = util.small_genomes(4, 120_000, 150_000)
    bases, offsets, truth = synth.reads(seqs, 300, 3000, seed=17)
    parts = [(names[:2], seqs[:2], 0), (names[2:], seqs[2:], 2)]
    summaries, lists = [], [[] for _ in range(300)]
    for pn, ps, base in parts:
        idx = capi.Index.from_seqs(pn, ps)
        eng = capi.Engine(idx, 0)
        assign, best, nhits = eng.classify(bases, offsets, 60)
        summaries.append(mdist.shard_summary(assign, best, nhits, rid_offset=base))
        oidx = oracle.Index.from_seqs(pn, [s.tobytes() for s in ps])
        oa, ob, onh, flat = oidx.classify(bases, offsets, 60)
        assert np.array_equal(assign, oa) and np.array_equal(nhits, onh)
        for k in capi.HIT_DTYPE.names:
            assert np.array_equal(best[k], ob[k]), k
        k = 0
        for r in range(300):
            for h in flat[k:k + onh[r]]:
                lists[r].append((int(h["rid"]) + base, int(h["nm"]), int(h["mlen"])))
            k += onh[r]
        eng.close()
    got, nm, ml, tot = mdist.merge_summaries(torch.stack(summaries))
    want = []
    for hits in lists:
        if not hits:
            want.append(mdist.UNMAPPED)
        else:
            b = hits[0] if len(hits) == 1 else aligner.best_hit(hits)
            want.append(b[0] if b else mdist.AMBIGUOUS)
    assert got.tolist() == want
    assert tot.tolist() == [len(h) for h in lists]
    assert (np.array(want) >= 0).sum() > 250


def _world_from(capi, oracle, names, seqs):
    idx = capi.Index.from_seqs(names, seqs)
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    eng = capi.Engine(idx, 0)
    return dict(names=names, seqs=seqs, idx=idx, oidx=oidx, eng=eng)


def test_long_reads_take_the_sequential_chain_path(capi, oracle):
    """Reads of 65 536 bases or more (and LDS-stage overflows in the partition kernel) leave
    the half-wave chain kernel for the sequential one; results must not change."""
    names, seqs = util.small_genomes(2, 400_000, 450_000)
    w = _world_from(capi, oracle, names, seqs)
    g0, g1 = seqs
    rng = np.random.default_rng(3)
    def noisy(a, rate=0.03):
        a = a.copy()
        m = rng.random(len(a)) < rate
        a[m] = util.ACGT[rng.integers(0, 4, int(m.sum()))]
        return a
    reads = [g0[1000:71_000], noisy(g1[5000:155_000]), util.revcomp(noisy(g0[100_000:370_000])),
             g0[2000:67_535], g0[2000:67_536], g1[300:5300], np.concatenate([g0[10_000:50_000], g1[10_000:50_000]])]
    bases, offsets = util.pack_reads(reads)
    assign, best, nhits = _compare_batch(capi, oracle, w, bases, offsets)
    assert assign.tolist()[:6] == [0, 1, 0, 0, 0, 1]
    assert nhits[6] == 2                                  # chimera of two contigs: two primaries


def test_repetitive_index_gives_many_anchors(capi, oracle):
    """Four diverged copies of one genome: most minimizers occur 2-4 times, so a read collects
    thousands of anchors (the largest chain / sort size classes and the HBM sort fallback)."""
    base = synth.genome(0x77, 150_000)
    seqs = [base] + [synth.diverge(base, 0x78 + i, 3000) for i in range(3)] + [synth.genome(0x99, 120_000)]
    names = [synth.contig_name(i) for i in range(5)]
    w = _world_from(capi, oracle, names, seqs)
    assert w["idx"].mid_occ >= 5
    b, o, truth = synth.reads(seqs, 60, 6000, seed=41, sub=100, ins=50, dele=50)
    long_one = seqs[1][10_000:50_000]
    reads = [b[o[i]:o[i + 1]] for i in range(60)] + [long_one, seqs[4][1000:9000]]
    bases, offsets = util.pack_reads(reads)
    assign, best, nhits = _compare_batch(capi, oracle, w, bases, offsets, min_mapq=0)
    c = w["eng"].counters()
    assert c["anchors"] / len(reads) > 2500               # the regime this test is for
    _compare_batch(capi, oracle, w, bases, offsets, min_mapq=60)
    w["eng"].set_debug(2)                                 # and with every look-back through HBM
    _compare_batch(capi, oracle, w, bases, offsets, min_mapq=0)
    w["eng"].set_debug(0)


def test_many_chain_ends_per_read(capi, oracle):
    """A hundred diverged copies of one short unit: a short read chains to most of them, i.e. far
    more than 64 chain ends with a few thousand anchors -- the sequential walk kept inside the
    LDS backtrack kernel -- next to reads with a few dozen ends, which take the parallel form,
    and ends that share a peak."""
    unit = synth.genome(0xA1, 2000)
    copies = [synth.diverge(unit, 0xB0 + i, 30_000) for i in range(100)]
    spacer = synth.genome(0xA2, 300)
    contig = np.concatenate([np.concatenate([c, spacer]) for c in copies])
    few = [synth.diverge(unit[:1500], 0xC0 + i, 40_000) for i in range(12)]
    seqs = [contig, np.concatenate(few), synth.genome(0xA3, 50_000)]
    names = [synth.contig_name(i) for i in range(3)]
    w = _world_from(capi, oracle, names, seqs)
    reads = [copies[7][300:450], copies[50][1000:1160], copies[93][40:190], unit[600:760],
             few[3][100:1400], few[8][:900], seqs[2][1000:4000]]
    bases, offsets = util.pack_reads(reads)
    _compare_batch(capi, oracle, w, bases, offsets, min_mapq=0)
    per_read = np.diff(w["eng"].dump(capi.DUMP_AN_OFFSETS, np.int64))
    assert w["eng"].counters()["chains"] > 64 * 4 and per_read[:4].max() <= 2560 and per_read[:4].min() > 640   # the regime this test is for
    _compare_batch(capi, oracle, w, bases, offsets, min_mapq=60)


def test_small_indexes_always_get_a_perfect_hash(capi, oracle):
    """Tiny table regions make it likely that two keys of one displacement bucket share base and
    step; the builder then re-salts that region (or grows the regions).  Every index must load
    and answer exactly."""
    for k in range(12):
        seqs = [synth.genome(0x5A17 + k, 20_000 + 7_000 * k), synth.genome(0x5B17 + k, 15_000)]
        names = [synth.contig_name(i) for i in range(2)]
        w = _world_from(capi, oracle, names, seqs)
        b, o, truth = synth.reads(seqs, 24, 1500, seed=100 + k)
        assign, best, nhits = w["eng"].classify(b, o, 60)
        oassign, obest, onh, oflat = w["oidx"].classify(b, o, 60)
        assert np.array_equal(assign, oassign) and np.array_equal(nhits, onh)
        an = w["eng"].dump(capi.DUMP_ANCHORS, capi.ANCHOR_DTYPE)
        an_off = w["eng"].dump(capi.DUMP_AN_OFFSETS, np.int64)
        raw = b.tobytes()
        for r in range(0, 24, 5):
            oa, _ = w["oidx"].seeds(raw[o[r]:o[r + 1]])
            assert np.array_equal(an[an_off[r]:an_off[r + 1]]["x"], oa["x"])


def test_displacement_table_read_in_place(capi, oracle, world):
    """An index with more displacement buckets per region than the probe kernel's LDS copy holds
    (hundreds of genomes) reads the displacements from HBM; the debug switch forces that path."""
    b, o, _ = synth.reads(world["seqs"], 200, 2000, seed=19)
    try:
        world["eng"].set_debug(4)
        _compare_batch(capi, oracle, world, b, o)
    finally:
        world["eng"].set_debug(0)


def test_dense_sketch_overflows_the_query_budget_and_is_redone(capi, oracle, world):
    """Low-complexity reads keep (almost) every k-mer as a minimizer: more query records than
    the one-per-three-bases budget, so the batch is redone with exact room."""
    reads = [np.full(4000, c, dtype=np.uint8) for c in b"ACGT"] * 6
    reads += [np.tile(np.frombuffer(b"AC", dtype=np.uint8), 2500), np.tile(np.frombuffer(b"ACG", dtype=np.uint8), 1500)]
    reads += [world["seqs"][0][5000:8000]]
    bases, offsets = util.pack_reads(reads)
    assign, best, nhits = _compare_batch(capi, oracle, world, bases, offsets)
    assert world["eng"].counters()["minimizers"] > int(offsets[-1]) // 3
    assert assign[-1] == 0
    # the engine stays usable and exact afterwards
    b2, o2, _ = synth.reads(world["seqs"], 50, 2000, seed=8)
    _compare_batch(capi, oracle, world, b2, o2)


def test_batch_shapes(capi, oracle, world):
    """Batch sizes around the tile / super-tile / workgroup granularities."""
    full, offs, _ = synth.reads(world["seqs"], 1100, 1200, seed=77)
    for n in (1, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1025):
        _compare_batch(capi, oracle, world, full[: offs[n]], offs[: n + 1])


def test_chain_ring_stress_build(capi, oracle, world):
    """The chaining kernel keeps only a ring of recent anchors in LDS and reads anything older
    from HBM.  The stress build (no completed block kept in the ring) sends every look-back
    through that fall-back; results must stay bit-identical."""
    eng = world["eng"]
    bases, offsets, truth = synth.reads(world["seqs"], 300, 5000, seed=0x5EED + 9)
    extra = util.edge_reads(world["seqs"], np.random.default_rng(2))
    eb, eo = util.pack_reads([bases[offsets[i]:offsets[i + 1]] for i in range(300)] + extra)
    try:
        eng.set_debug(2)
        _compare_batch(capi, oracle, world, eb, eo)
        _compare_batch(capi, oracle, world, eb, eo, min_mapq=0)
    finally:
        eng.set_debug(0)
    _compare_batch(capi, oracle, world, eb, eo)


@pytest.mark.gpu
def test_device_built_tables_equal_the_host_form(capi):
    """The device tables of an index (per-region hash-and-displace perfect hash, displacement bytes, salts,
    presence filter) are built on the device; the host form of the same construction is the reference."""
    for n, lo, hi in ((4, 200_000, 300_000), (3, 1_500_000, 2_500_000)):
        names, seqs = util.small_genomes(n, lo, hi)
        a = capi.Index.from_seqs(names, seqs)
        b = capi.Index.from_seqs(names, seqs)
        capi.check(capi.lib().mnc_index_set_host_tables(b._h, 1))
        ta, tb = capi.Engine(a, 0).dump_tables(), capi.Engine(b, 0).dump_tables()
        assert len(ta) == len(tb) and np.array_equal(ta, tb)


def test_index_built_on_the_device_equals_the_host_builder(capi, oracle, tmp_path):
    """mnc_index_build_mem_device / mnc_index_build_device (csrc/k_idxbuild.hip: contig pieces through the batch sketch
    kernel, radix sort, run lengths) against the host builder and the oracle's index: every (hash, occurrence) pair in
    order, mid_occ, the totals -- contigs whose lengths sit on the 256 kb piece boundaries, contigs shorter than a
    k-mer / than a window, ambiguous bases (runs of N across a piece boundary, periodic n), lower case, a
    low-complexity stretch, many contigs of one genome."""
    rng = np.random.default_rng(17)
    P = 1 << 18
    seqs = [synth.genome(900 + i, L) for i, L in enumerate((P - 1, P, P + 1, 2 * P + 17, 3 * P - 5, 700_001, 9, 14, 15, 24, 25, 40, 100, 5000))]
    n_run = seqs[3].copy()
    n_run[P - 30:P + 45] = ord("N")                                   # a run of N across a piece boundary
    n_run[2 * P - 3:2 * P + 2] = ord("n")
    n_run[1000::7919] = ord("N")
    seqs.append(n_run)
    lower = np.frombuffer(seqs[5].tobytes().lower(), dtype=np.uint8).copy()
    seqs.append(lower[:300_000])
    lowc = seqs[5][:400_000].copy()
    lowc[100_000:100_600] = np.tile(np.frombuffer(b"ACACACGT", dtype=np.uint8), 75)
    lowc[P - 200:P + 200] = ord("A")                                  # a homopolymer across the boundary
    seqs.append(lowc)
    names = [synth.contig_name(i // 3) for i in range(len(seqs))]     # several contigs per genome
    host = capi.Index.from_seqs(names, seqs)
    dev = capi.Index.from_seqs(names, seqs, device=0)
    hh, hy = host.dump()
    dh, dy = dev.dump()
    assert np.array_equal(hh, dh) and np.array_equal(hy, dy)
    hi, di = host.info(), dev.info()
    for f in ("k", "w", "n_contigs", "n_genomes", "mid_occ", "n_keys", "n_occ", "total_len"):
        assert getattr(hi, f) == getattr(di, f), f
    assert host.genome_names == dev.genome_names and host.genome_lens == dev.genome_lens
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    assert dev.mid_occ == oidx.mid_occ
    oh, oy = oidx.dump()
    assert np.array_equal(dh, oh) and np.array_equal(dy, oy)
    # the two indexes classify alike (the contig bases of the alignment stage included)
    good = [s for s in seqs if len(s) > 100_000]
    b, o, _ = synth.reads(good, 300, 3000, seed=5)
    e1, e2 = capi.Engine(host, 0), capi.Engine(dev, 0)
    r1, r2 = e1.classify(b, o, 0), e2.classify(b, o, 0)
    assert np.array_equal(r1[0], r2[0]) and np.array_equal(r1[2], r2[2])
    for k in capi.HIT_DTYPE.names:
        assert np.array_equal(r1[1][k], r2[1][k]), k
    e1.close(), e2.close()
    # from a FASTA file, written to an index file: the file loads to the same index
    fa = str(tmp_path / "db.fna.gz")
    synth.write_fasta(fa, names[:6], seqs[:6])
    out = str(tmp_path / "index1.mmi")
    built = capi.Index.build(fa, out, device=0)
    again = capi.Index.load(out)
    ref = capi.Index.from_seqs(names[:6], seqs[:6])
    for x in (built, again):
        a1, a2 = x.dump()
        b1, b2 = ref.dump()
        assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and x.mid_occ == ref.mid_occ


def test_device_builder_on_degenerate_and_repetitive_inputs(capi):
    """What the device builder's early exits and its mid_occ selection must still get right (mappy.Aligner on such a
    FASTA gives an index object too, monica/genomes/aligner.py:45-48): contigs all shorter than a k-mer, all ambiguous,
    one contig -- no minimizer at all, yet the contig / genome tables and mid_occ are the host builder's; and a satellite
    whose k-mers occur tens of thousands of times (counts beyond the device histogram's bins: the host's selection)."""
    def same(names, seqs):
        host, dev = capi.Index.from_seqs(names, seqs), capi.Index.from_seqs(names, seqs, device=0)
        (hh, hy), (dh, dy) = host.dump(), dev.dump()
        assert np.array_equal(hh, dh) and np.array_equal(hy, dy)
        hi, di = host.info(), dev.info()
        for f in ("k", "w", "n_contigs", "n_genomes", "mid_occ", "n_keys", "n_occ", "total_len"):
            assert getattr(hi, f) == getattr(di, f), f
        assert host.genome_names == dev.genome_names and host.genome_lens == dev.genome_lens
        assert np.array_equal(host.contig_genome, dev.contig_genome)
        return dev
    short = [synth.genome(7 + i, L) for i, L in enumerate((3, 9, 14, 1))]
    d = same(["Ga_a:A.1", "Ga_a:A.1", "Gb_b:B.1", "Gc_c:C.1"], short)
    assert d.info().n_keys == 0 and d.info().n_genomes == 3 and len(d.contig_genome) == 4
    same(["Gn_n:N.1"], [np.full(5000, ord("N"), dtype=np.uint8)])
    same(["Ge_e:E.1"], [np.zeros(0, dtype=np.uint8)])
    # 5 000 copies of a 2 kb unit next to ordinary sequence: ~360 minimizers with 5 000 occurrences each, more than the
    # 2e-4 of the distinct ones that lie above the mid_occ rank -- the count at that rank is beyond the histogram's bins
    unit = synth.genome(99, 2000)
    sat = np.concatenate([synth.genome(100, 50_000), np.tile(unit, 5_000), synth.genome(101, 50_000)])
    d = same(["Gs_s:S.1", "Gt_t:T.1"], [sat, synth.genome(102, 300_000)])
    assert d.mid_occ > 4096


def test_the_20_genome_index_is_built_on_the_device_in_a_fraction_of_a_second(capi):
    import time
    names, seqs = synth.genome_set(20)
    bs = [s.tobytes() for s in seqs]
    capi.Index.from_seqs(names[:1], bs[:1], device=0)             # the first call pays for module load and allocator warm-up
    t0 = time.perf_counter()
    dev = capi.Index.from_seqs(names, bs, device=0)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    host = capi.Index.from_seqs(names, bs)
    t_host = time.perf_counter() - t0
    a, b = dev.dump(), host.dump()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and dev.mid_occ == host.mid_occ
    print(f"index of 94 Mbp: device {t_dev:.3f} s, host {t_host:.3f} s")
    assert t_dev < 0.6 and t_dev < t_host


@pytest.mark.parametrize("bits", [9, 10])
def test_more_table_regions_change_nothing(capi, oracle, bits):
    """The device tables are cut into 256, 512 or 1 024 regions by the index's size (a region stays at 2 MiB, the hot set of
    an XCD's L2); the query partition, the probe and the collect kernels follow (region-major runs, super-tiles of 256 /
    512 / 1 024 reads).  Forced onto a small index: every stage against the oracle as with 256 regions, the device-built
    tables equal to the host form, batches around the super-tile boundaries."""
    names, seqs = util.small_genomes()
    idx = capi.Index.from_seqs(names, seqs)
    capi.check(capi.lib().mnc_index_set_region_bits(idx._h, bits))
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    w = dict(names=names, seqs=seqs, idx=idx, oidx=oidx, eng=capi.Engine(idx, 0))
    rng = np.random.default_rng(bits)
    bases, offsets = util.pack_reads(util.edge_reads(seqs, rng))
    _compare_batch(capi, oracle, w, bases, offsets, min_mapq=0)
    full, offs, _ = synth.reads(seqs, 2600, 1200, seed=400 + bits)
    for n in (1, 255, 256, 257, 1023, 1025, 2600):
        _compare_chain(capi, oracle, w, full[:offs[n]], offs[:n + 1])
    b, o, truth = synth.reads(seqs, 500, 5000, seed=7)
    assign, best, nhits = _compare_batch(capi, oracle, w, b, o)
    assert (assign >= 0).sum() > 450
    # the same index with 256 regions: equal decisions; host-form tables equal the device-built ones at this region count
    ref = capi.Engine(capi.Index.from_seqs(names, seqs), 0)
    r = ref.classify(b, o, 60)
    assert np.array_equal(r[0], assign) and np.array_equal(r[2], nhits)
    host_form = capi.Index.from_seqs(names, seqs)
    capi.check(capi.lib().mnc_index_set_region_bits(host_form._h, bits))
    capi.check(capi.lib().mnc_index_set_host_tables(host_form._h, 1))
    ta, tb = w["eng"].dump_tables(), capi.Engine(host_form, 0).dump_tables()
    assert len(ta) == len(tb) and np.array_equal(ta, tb)
    assert int(ta[:4].view(np.int32)[0]) >> 16 == bits


def test_bacteria_like_genomes_with_operons_insertion_sequences_and_shared_stretches(capi, oracle):
    """`synth.genome_set_repeats` (the sensitivity workload of BASELINE.md: what the databases monica builds hold,
    database.py:52-67 -- rRNA-like operons in 5-7 copies, insertion sequences in 15-30, stretches shared between
    neighbours at 85-95 %) at test size: reads that start inside an element, cross one, or lie in a shared stretch have
    secondaries, sub-optimal chains and MAPQ < 60; stage by stage at the chain level and region by region with
    base-level alignment against the oracle."""
    names, seqs = synth.genome_set_repeats(6, min_len=300_000, max_len=420_000)
    plain_names, plain = synth.genome_set(6, min_len=300_000, max_len=420_000)
    assert [len(s) for s in seqs] == [len(s) for s in plain] and 0.03 < (seqs[0] != plain[0]).mean() < 0.2
    again = synth.genome_set_repeats(6, min_len=300_000, max_len=420_000)[1]
    assert all(np.array_equal(a, b) for a, b in zip(seqs, again))                       # a function of the seeds
    w = _world_from(capi, oracle, names, seqs)
    assert w["idx"].mid_occ >= capi.Index.from_seqs(plain_names, plain).mid_occ          # the elements can only raise the cut-off
    b, o, truth = synth.reads(seqs, 160, 4000, seed=23)
    assign, best, nhits = _compare_batch(capi, oracle, w, b, o, min_mapq=0)
    regs = w["eng"].dump(capi.DUMP_REGS, capi.REG_DTYPE)
    assert (regs["mapq"] < 60).any() and (regs["parent"] != regs["id"]).any()           # secondaries and doubtful mappings exist
    _compare_batch(capi, oracle, w, b, o, min_mapq=60)
    hard, _, _ = synth.reads(seqs, 60, 4000, seed=24, sub=700, ins=450, dele=450)        # 16 % errors: the literal kernel's share
    _compare_dp(capi, oracle, w, hard, np.arange(61, dtype=np.int64) * 4000, min_mapq=0)
