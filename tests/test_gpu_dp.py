"""GPU parity of the base-level alignment stage (csrc/k_align.hip) -- minimap2's
mm_align_skeleton / ksw_extd2 as mappy always runs it (monica/genomes/aligner.py:193-195,
215-217 read hit.mapq / hit.NM / hit.mlen from its result): every region field, every CIGAR,
gated hits and decisions, bit for bit against the CPU oracle."""
import numpy as np
import pytest

from monica_amd import synth
import util
from test_gpu_parity import _compare_dp, _world_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world(capi, oracle):
    names, seqs = util.small_genomes()
    return _world_from(capi, oracle, names, seqs)


def test_known_answers_on_the_gpu(capi, oracle, world):
    g0, g1 = world["seqs"][0], world["seqs"][1]
    one_sub = g0[5000:8000].copy()
    one_sub[1500] = ord("A") if one_sub[1500] != ord("A") else ord("C")
    with_n = g0[20000:24000].copy()
    with_n[1500] = ord("N")
    reads = [g0[5000:8000], util.revcomp(g0[5000:8000]), one_sub,
             np.concatenate([g0[5000:6500], g0[6510:8000]]),                              # 10-base deletion
             np.concatenate([g0[5000:6500], np.frombuffer(b"ACGTACGTAC", dtype=np.uint8), g0[6500:8000]]),
             with_n,
             np.concatenate([g0[100000:102500], g0[103500:106000]]),                      # 1 kb deletion: long-join
             np.concatenate([g0[70000:72500], g1[90000:92500]])]                          # chimera
    bases, offsets = util.pack_reads(reads)
    assign, best, nhits = _compare_dp(capi, oracle, world, bases, offsets)
    eng = world["eng"]
    regs = eng.dump(capi.DUMP_REGS, capi.REG_DTYPE)
    cigs = eng.cigars()
    assert cigs[0] == [(3000, "M")] and cigs[1] == [(3000, "M")] and cigs[2] == [(3000, "M")]
    assert (regs["mlen"][0], regs["blen"][0], regs["dp_max"][0], regs["mapq"][0]) == (3000, 3000, 6000, 60)
    assert best["nm"][:6].tolist() == [0, 0, 1, 10, 10, 1] and best["mlen"][:3].tolist() == [3000, 3000, 2999]
    assert cigs[3] == [(1499, "M"), (10, "D"), (1491, "M")]
    assert cigs[6] == [(2500, "M"), (1000, "D"), (2500, "M")]
    assert nhits[7] == 2


def test_edge_reads(capi, oracle, world):
    reads = util.edge_reads(world["seqs"], np.random.default_rng(7))
    bases, offsets = util.pack_reads(reads)
    _compare_dp(capi, oracle, world, bases, offsets)
    _compare_dp(capi, oracle, world, bases, offsets, min_mapq=0)


def test_noisy_reads(capi, oracle, world):
    bases, offsets, truth = synth.reads(world["seqs"], 300, 5000, seed=0x5EED + 1)
    assign, best, nhits = _compare_dp(capi, oracle, world, bases, offsets)
    mapped = assign >= 0
    assert mapped.sum() > 280 and (assign[mapped] == truth[mapped]).all()
    assert (best["nm"][mapped] < 0.2 * best["mlen"][mapped]).all()
    for rate in ((100, 50, 50), (800, 600, 600)):                    # 2 % and 20 % errors
        b, o, _ = synth.reads(world["seqs"], 60, 3000, seed=31, sub=rate[0], ins=rate[1], dele=rate[2])
        _compare_dp(capi, oracle, world, b, o, min_mapq=0)


def test_zdrop_split_inversion_and_long_extensions(capi, oracle, world):
    """A block of junk inside a read drops the score by more than zdrop in a gap filling: the
    region is split there and its tail aligned as a region of its own.  An inverted block makes
    the test answer "inversion" (lower Z-drop on the second pass and on the tail's left
    extension).  Long junk flanks run the extension kernels until the Z-drop, past the band."""
    g0, g1 = world["seqs"][0], world["seqs"][1]
    rng = np.random.default_rng(21)
    junk = lambda n: util.ACGT[rng.integers(0, 4, n)]
    reads = [np.concatenate([g0[30000:32000], junk(700), g0[32700:34700]]),            # junk replaces 700 bases
             np.concatenate([g0[40000:42000], util.revcomp(g0[42000:42700]), g0[42700:44700]]),   # inversion
             np.concatenate([g1[50000:52000], junk(300), g1[52300:54000], junk(500), g1[54500:56500]]),
             np.concatenate([junk(3000), g0[60000:63000], junk(3000)]),                  # flanks far longer than the band reach
             np.concatenate([junk(1200), util.revcomp(g1[70000:72000])]),
             np.concatenate([g0[80000:82000], junk(1500), g0[82000:84000]])]             # 1.5 kb insertion of junk
    bases, offsets = util.pack_reads(reads)
    _compare_dp(capi, oracle, world, bases, offsets, min_mapq=0)
    regs = world["eng"].dump(capi.DUMP_REGS, capi.REG_DTYPE)
    reg_off = world["eng"].dump(capi.DUMP_REG_OFFSETS, np.int64)
    assert reg_off[1] - reg_off[0] >= 2 and (regs["flags"][reg_off[0]:reg_off[1]] & 6).any()   # really split
    _compare_dp(capi, oracle, world, bases, offsets, min_mapq=60)


def test_inversion_regions(capi, oracle, world):
    """mm_align1_inv on the GPU: reads with an inverted block (300 .. 1 500 bases, either strand, with errors, two
    inversions in one read, an inversion next to junk): the stretch between the two halves of the split region is aligned
    on the other strand -- a region of its own (flag 16, MAPQ 0, no seeds), every field and its CIGAR equal to the
    oracle's.  With a diverged copy of the contig the second mapping has its own inversion region, a secondary of the
    first's."""
    g0, g1 = world["seqs"][0], world["seqs"][1]
    rng = np.random.default_rng(5)

    def noisy(x, rate):
        x = x.copy()
        k = rng.random(len(x)) < rate
        x[k] = util.ACGT[rng.integers(0, 4, int(k.sum()))]
        return x
    reads = []
    for inv_len in (300, 500, 700, 1000, 1500):
        reads.append(np.concatenate([g0[40000:42000], util.revcomp(g0[42000:42000 + inv_len]), g0[42000 + inv_len:44000 + inv_len]]))
    reads.append(util.revcomp(reads[2]))                                                  # the whole read on the other strand
    reads.append(noisy(reads[2], 0.05))
    reads.append(noisy(reads[4], 0.08))
    reads.append(np.concatenate([g1[10000:12000], util.revcomp(g1[12000:12600]), g1[12600:14600], util.revcomp(g1[14600:15400]), g1[15400:17400]]))
    reads.append(np.concatenate([g1[30000:32000], util.revcomp(g1[32000:32700]), util.ACGT[rng.integers(0, 4, 300)], g1[33000:35000]]))
    bases, offsets = util.pack_reads(reads)
    for mq in (0, 60):
        _compare_dp(capi, oracle, world, bases, offsets, min_mapq=mq)
    regs = world["eng"].dump(capi.DUMP_REGS, capi.REG_DTYPE)
    reg_off = world["eng"].dump(capi.DUMP_REG_OFFSETS, np.int64)
    n_inv = [int(((regs["flags"][reg_off[r]:reg_off[r + 1]] & 16) != 0).sum()) for r in range(len(reads))]
    # (the read with two inverted blocks gets ONE inversion region: when the second tail is reached, what precedes it in
    # the skeleton's array is the first inversion's region, not a split head, and mm_align1_inv declines)
    assert n_inv[:6] == [1] * 6 and n_inv[8] == 1, str(n_inv)
    inv = regs[(regs["flags"] & 16) != 0]
    assert (inv["mapq"] == 0).all() and (inv["cnt"] == 0).all() and (inv["flags"] & 1).all() and (inv["dp_max"] > 400).all()
    # a diverged copy of the contig: both mappings split, both get an inversion region
    base = synth.genome(0x51, 150_000)
    seqs = [base, synth.diverge(base, 0x52, 20_000)]
    w = _world_from(capi, oracle, [synth.contig_name(i) for i in range(2)], seqs)
    rd = np.concatenate([base[60000:62000], util.revcomp(base[62000:62800]), base[62800:64800]])
    b, o = util.pack_reads([rd, noisy(rd, 0.04)])
    _compare_dp(capi, oracle, w, b, o, min_mapq=0)
    regs = w["eng"].dump(capi.DUMP_REGS, capi.REG_DTYPE)
    assert ((regs["flags"] & 16) != 0).sum() >= 2


def test_repetitive_index_with_secondaries(capi, oracle):
    """Diverged copies: every read has secondaries whose DP scores set dp_max2 / n_sub of the
    primary (the DP branch of the MAPQ formula), and some reads change hands after alignment."""
    base = synth.genome(0x77, 120_000)
    seqs = [base] + [synth.diverge(base, 0x78 + i, r) for i, r in enumerate((3000, 10_000, 30_000))] + [synth.genome(0x99, 80_000)]
    names = [synth.contig_name(i) for i in range(5)]
    w = _world_from(capi, oracle, names, seqs)
    b, o, truth = synth.reads(seqs, 80, 4000, seed=41)
    assign, best, nhits = _compare_dp(capi, oracle, w, b, o, min_mapq=0)
    regs = w["eng"].dump(capi.DUMP_REGS, capi.REG_DTYPE)
    assert (regs["dp_max2"] > 0).sum() >= 30
    _compare_dp(capi, oracle, w, b, o, min_mapq=60)


def test_batch_shapes_and_tiny_reads(capi, oracle, world):
    full, offs, _ = synth.reads(world["seqs"], 130, 1500, seed=77)
    for n in (1, 2, 63, 64, 65, 130):
        _compare_dp(capi, oracle, world, full[: offs[n]], offs[: n + 1])
    reads = [world["seqs"][0][1000:1000 + L] for L in (0, 14, 15, 40, 60, 100, 150, 250, 400)]
    bases, offsets = util.pack_reads(reads)
    _compare_dp(capi, oracle, world, bases, offsets, min_mapq=0)


def test_long_reads_and_mixed_lengths(capi, oracle, world):
    """Reads far longer than the usual 5 kb: more than 64 kernel calls per region (the join works through
    them 64 at a time), regions whose bases do not fit the stitch kernel's LDS (read in place from HBM),
    next to short reads in the same batch; low and high error rates."""
    seqs = world["seqs"]
    rng = np.random.default_rng(77)
    reads = []
    for L, (sub, ins, dele) in ((30_000, (400, 300, 300)), (70_000, (200, 150, 150)), (18_000, (600, 450, 450)), (45_000, (0, 0, 0))):
        b, o, _ = synth.reads(seqs, 1, L, seed=int(rng.integers(1, 1 << 30)), sub=sub, ins=ins, dele=dele)
        reads.append(b[o[0]:o[1]])
    short_b, short_o, _ = synth.reads(seqs, 6, 1200, seed=5)
    reads += [short_b[short_o[i]:short_o[i + 1]] for i in range(6)]
    reads.append(util.revcomp(seqs[1][10_000:60_000]))
    bases, offsets = util.pack_reads(reads)
    assign, best, nhits = _compare_dp(capi, oracle, world, bases, offsets, min_mapq=0)
    assert (assign[:4] >= 0).all() and assign[-1] == 1
    regs = world["eng"].dump(capi.DUMP_REGS, capi.REG_DTYPE)
    assert regs["n_cigar"].max() > 1024 and (regs["qe"] - regs["qs"]).max() > 40_000


def test_long_reads_regions_planned_by_a_wave(capi, oracle, world):
    """Regions of reads with 512 chained anchors or more are planned by mnc_dp_plan_long (a wave per region: the passes
    over all anchors as wave-wide steps) instead of mnc_dp_plan (a lane per region).  Long reads with long insertions,
    deletions that cancel them, junk in the middle and noisy stretches -- what the seed filters and the long-join seeds
    exist for -- must come out the same from both forms (debug bit 0x800000: the lane form for everything) and equal the
    oracle."""
    seqs = world["seqs"]
    rng = np.random.default_rng(4242)
    junk = lambda n: util.ACGT[rng.integers(0, 4, n)]
    g = seqs[0]
    reads = []
    for L, (sub, ins, dele) in ((40_000, (400, 300, 300)), (36_000, (150, 100, 100)), (50_000, (550, 400, 400))):
        b, o, _ = synth.reads(seqs, 1, L, seed=int(rng.integers(1, 1 << 30)), sub=sub, ins=ins, dele=dele)
        reads.append(b[o[0]:o[1]])
    # an insertion cancelled by a deletion 1.5 kb on; two long gaps close together; junk flanks; a 600-base deletion
    reads.append(np.concatenate([g[10_000:25_000], junk(60), g[25_000:26_500], g[26_560:45_000]]))
    reads.append(np.concatenate([g[50_000:70_000], junk(45), g[70_000:70_400], junk(50), g[70_400:90_000]]))
    reads.append(np.concatenate([junk(1500), g[100_000:118_000], g[118_600:140_000], junk(900)]))
    reads.append(util.revcomp(np.concatenate([g[5_000:30_000], junk(800), g[30_800:52_000]])))
    bases, offsets = util.pack_reads(reads)
    eng = world["eng"]
    assign, best, nhits = _compare_dp(capi, oracle, world, bases, offsets, min_mapq=0)
    regs = eng.dump(capi.DUMP_REGS, capi.REG_DTYPE).copy()
    cigs = eng.dump(capi.DUMP_CIGARS, np.uint32).copy()
    segs = eng.dump(capi.DUMP_SEGS, capi.SEG_DTYPE).copy()
    assert len(regs) >= len(reads) and (regs["cnt"] >= 1024).any()
    try:
        eng.set_debug(0x800000)
        a2, b2, n2 = eng.classify(bases, offsets, 0)
        regs2 = eng.dump(capi.DUMP_REGS, capi.REG_DTYPE)
        cigs2 = eng.dump(capi.DUMP_CIGARS, np.uint32)
        segs2 = eng.dump(capi.DUMP_SEGS, capi.SEG_DTYPE)
    finally:
        eng.set_debug(0)
    assert np.array_equal(assign, a2) and np.array_equal(nhits, n2)
    for k in capi.REG_DTYPE.names:
        assert np.array_equal(regs[k], regs2[k]), k
    assert np.array_equal(cigs, cigs2)
    # the kernel calls themselves: the same set (their order in the pool depends on which wave came first)
    def planned(t):
        cols = np.stack([t[k] for k in ("read", "kind", "rid", "rev", "ts", "tlen", "qs", "qlen", "w", "zdrop", "flag", "ai", "big")], axis=1)
        return cols[np.lexsort(cols.T[::-1])]
    assert np.array_equal(planned(segs), planned(segs2))


def test_engines_on_threads_share_the_device_workspace(capi, oracle, world):
    """monica's thread pool: one engine per thread on the same index; the alignment scratch of the device
    is lent to one engine at a time.  Every thread must get what a single engine gets."""
    import threading
    batches = [synth.reads(world["seqs"], 150, 3000, seed=900 + i) for i in range(4)]
    want = []
    for b, o, _ in batches:
        a, best, nh = world["eng"].classify(b, o, 60)
        want.append((a.copy(), best.copy(), nh.copy()))
    got, errs = [None] * 4, []

    def run(i):
        try:
            eng = capi.Engine(world["idx"], 0)
            for _ in range(3):
                a, best, nh = eng.classify(batches[i][0], batches[i][1], 60)
            got[i] = (a.copy(), best.copy(), nh.copy())
            eng.close()
        except Exception as e:                                   # noqa: BLE001
            errs.append(e)
    threads = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert np.array_equal(got[i][0], want[i][0]) and np.array_equal(got[i][2], want[i][2])
        for name in capi.HIT_DTYPE.names:
            assert np.array_equal(got[i][1][name], want[i][1][name])


def test_results_do_not_depend_on_scheduling(capi, world):
    """Which workgroup takes which kernel call differs from run to run (atomic work queues): the regions
    and CIGARs of a batch full of Z-drops, long gaps and hand-backs between kernels must not."""
    b, o, _ = synth.reads(world["seqs"], 60, 3000, seed=31, sub=800, ins=600, dele=600)
    eng = world["eng"]
    eng.set_contract(capi.CONTRACT_DP)
    first = None
    for _ in range(8):
        eng.classify(b, o, 0)
        now = (eng.dump(capi.DUMP_REGS, capi.REG_DTYPE), eng.dump(capi.DUMP_REG_OFFSETS, np.int64), eng.dump(capi.DUMP_CIGARS, np.uint32))
        if first is None:
            first = now
        else:
            assert np.array_equal(now[1], first[1]) and np.array_equal(now[2], first[2])
            for name in capi.REG_DTYPE.names:
                assert np.array_equal(now[0][name], first[0][name]), name


@pytest.mark.parametrize("route", [0x20000, 0x40000, 0x80000 | 0x100000, 0x20000 | 0x40000 | 0x80000 | 0x100000, 0x400000,
                                   20 << 8 | 20 << 24, 48 << 8 | 48 << 24, 0x10, 0x800000, 0x20, 0x8, 0x1, 0x40 | 0x8, 0x40 | 0x1, 0x80, 0x80000000,
                                   0x20 | 0x20000 | 0x40000 | 0x80000 | 0x100000, 0x8 | 0x20000 | 0x40000 | 0x80000 | 0x100000,
                                   0x1 | 0x20000 | 0x40000 | 0x80000 | 0x100000, 0x80 | 0x20000 | 0x40000 | 0x80000 | 0x100000])
def test_every_kernel_family_can_be_taken_out(capi, oracle, world, route):
    """The alignment stage sorts its kernel calls over several kernels (packed gap filling, packed extensions, the
    long-call kernels, the step-by-step extension kernel, ksw2's kernel literally).  With a family switched off its
    calls go to the next kernel in line -- in the end all of them to the literal one -- and nothing may change.  (0x400000:
    without the 42-cell tier of the banded kernel; the last two: the planner sends every gap filling to the widest / to
    the narrowest band first -- which tier proves a band is a matter of speed, never of the result; 0x10 / 0x800000: every
    region planned by the plan kernel's wave form / by its lane form -- a batch this small takes the wave form by itself;
    The literal kernel's long calls run on a workgroup with the cells in registers (ksw_wg): eight waves with two cells a
    thread when a pass has few calls, four waves with four cells when it has many; 0x8 / 0x1: always the one / the other form,
    0x40 with either: that launch shape with the cells in the workspace, 0x80: round 3's four waves on the workspace, 0x20: one
    wave each.  0x80000000: the packed gap-filling kernels in their plain frame instead of the drifting one (values of
    anti-diagonal r kept as value + e r, which makes a mismatch and a gap extension cost nothing in the recurrence).)"""
    reads = [synth.reads(world["seqs"], 40, 3000, seed=31, sub=800, ins=600, dele=600), synth.reads(world["seqs"], 60, 5000, seed=0x5EED + 9)]
    g0 = world["seqs"][0]
    rng = np.random.default_rng(3)
    junk = lambda n: util.ACGT[rng.integers(0, 4, n)]
    extra = [np.concatenate([junk(900), g0[60000:63000], junk(700)]), np.concatenate([g0[100000:102500], g0[103500:106000]]),
             np.concatenate([g0[30000:32000], junk(700), g0[32700:34700]])]
    eng = world["eng"]
    try:
        eng.set_debug(route)
        for b, o, _ in reads:
            _compare_dp(capi, oracle, world, b, o, min_mapq=0)
        b, o = util.pack_reads(extra)
        _compare_dp(capi, oracle, world, b, o, min_mapq=0)
    finally:
        eng.set_debug(0)


def test_tiers_planned_by_anchor_density_on_reads_of_mixed_quality(capi, oracle, world):
    """Round 5: the gap-filling tier a segment tries first comes from its region's anchor density (k_align.hip: fill_pred_of;
    rho = cnt (w + 1) / (2 x query span) = (1 - eps)^k), where rounds 3-4 compared with two thresholds fitted at 10 %
    errors (debug bits 8-15 / 24-30 still give those).  A batch of reads at 2, 7, 10, 14 and 18 % errors: against the oracle
    as planned by density, the same region for region with the fixed thresholds, and the plans do differ."""
    parts = [synth.reads(world["seqs"], 60, 4000, seed=960 + i, sub=s_, ins=i_, dele=d_)
             for i, (s_, i_, d_) in enumerate(((80, 60, 60), (280, 210, 210), (400, 300, 300), (560, 420, 420), (720, 540, 540)))]
    bases = np.concatenate([p[0] for p in parts])
    offsets = np.arange(len(bases) // 4000 + 1, dtype=np.int64) * 4000
    eng = world["eng"]
    _compare_dp(capi, oracle, world, bases, offsets, min_mapq=0)
    by_density = (eng.dump(capi.DUMP_REGS, capi.REG_DTYPE), eng.cigars(), eng.counters())
    try:
        eng.set_debug(32 << 8 | 34 << 24)
        _compare_dp(capi, oracle, world, bases, offsets, min_mapq=0)
        fixed = (eng.dump(capi.DUMP_REGS, capi.REG_DTYPE), eng.cigars(), eng.counters())
    finally:
        eng.set_debug(0)
    assert by_density[0].tobytes() == fixed[0].tobytes() and by_density[1] == fixed[1]
    tiers = lambda c: [c[k] for k in ("dp_fill_tier1", "dp_fill_tier_mid", "dp_fill_tier2", "dp_fill_tier3")]
    assert tiers(by_density[2]) != tiers(fixed[2]) and by_density[2]["dp_segments"] == fixed[2]["dp_segments"]


def test_reads_with_many_errors_use_every_workspace_class(capi, oracle, world):
    """16 % errors: seeds are sparse, so gaps between them are long, the banded kernels' proofs fail more often and
    read flanks without seeds become extensions of thousands of bases -- calls for the literal kernel's large
    workspace class and for what the banded kernels hand back to it.  (At 16 % the first form of the stage took 2 s
    per 30 000 reads on eight large slots; tools/err_profile.py.)  Also: the stitch kernel reading bases in place."""
    b, o, _ = synth.reads(world["seqs"], 500, 6000, seed=778, sub=700, ins=450, dele=450)
    _compare_dp(capi, oracle, world, b, o, min_mapq=0)
    c = world["eng"].counters()
    assert c["dp_literal_big"] > 0 and c["dp_long_gaps"] > 0 and c["dp_long_extensions"] > 0 and c["dp_fill_tier3"] > 0
    eng = world["eng"]
    try:
        eng.set_debug(0x200000)
        _compare_dp(capi, oracle, world, b[: o[120]], o[:121], min_mapq=0)
        for form in (0x20, 0x8, 0x1, 0x80):                    # the literal kernel's long calls on one wave each; always the wide / always the four-wave form with the cells in registers; round 3's form
            eng.set_debug(form)
            _compare_dp(capi, oracle, world, b, o, min_mapq=0)
    finally:
        eng.set_debug(0)


def test_a_batch_of_many_short_regions_fits_the_cigar_pools(capi, oracle, world):
    """20 000 reads of 600 bases: more than 16 384 regions, so the stitch kernel's waves reserve the region CIGAR pool
    a chunk at a time (4 096 workgroups x 4 096 words) although the batch's bases would budget far less -- the pool
    is sized with that slack, and an alignment-stage overflow is redone with more room until it fits instead of
    failing after one retry.  Decisions, gated hits and a sample of CIGARs equal the oracle."""
    seqs = world["seqs"]
    n = 20_000
    bases, offsets, truth = synth.reads(seqs, n, 600, seed=4242)
    eng, oidx = world["eng"], world["oidx"]
    eng.set_contract(capi.CONTRACT_DP)
    oidx.opt.cigar = 1
    assign, best, nhits = eng.classify(bases, offsets, 60)
    reg_off = eng.dump(capi.DUMP_REG_OFFSETS, np.int64)
    assert reg_off[-1] >= 16_384
    regs = eng.dump(capi.DUMP_REGS, capi.REG_DTYPE)
    cigs = eng.cigars()
    oassign, obest, onh, _ = oidx.classify(bases, offsets, 60, n_threads=8)
    assert np.array_equal(assign, oassign) and np.array_equal(nhits, onh)
    for name in capi.HIT_DTYPE.names:
        assert np.array_equal(best[name], obest[name]), name
    raw = bases.tobytes()
    for r in range(0, n, 97):
        oregs, ocigs = oidx.map_cigar(raw[offsets[r]:offsets[r + 1]])
        assert cigs[reg_off[r]:reg_off[r + 1]] == ocigs, f"read {r}: CIGAR"
        for name in capi.REG_DTYPE.names:
            assert np.array_equal(regs[reg_off[r]:reg_off[r + 1]][name], oregs[name]), f"read {r}: {name}"
    mapped = assign >= 0
    assert mapped.sum() > 0.7 * n and (assign[mapped] == truth[mapped]).mean() > 0.99      # 600-base reads: four in five reach MAPQ 60


def test_packed_sequence_loads_at_every_alignment(capi, oracle, world):
    """The gap-filling and the stitch kernels take a read's bases from the sketch stage's 2-bit words (sixteen a word) and a
    contig's from its 4-bit words (eight a word), whole words at a time, turned round or complemented by strand.  Reads on the
    reverse strand at the very start of the batch (the words in front of them do not exist), reads whose offset in the batch
    is any residue modulo sixteen, regions that start at any residue modulo eight of their contig, reads shorter than a word."""
    g0, g1 = world["seqs"][0], world["seqs"][1]
    rng = np.random.default_rng(99)

    def noisy(r):
        r = r.copy()
        pos = rng.choice(len(r), len(r) // 25, replace=False)
        r[pos] = util.ACGT[rng.integers(0, 4, len(pos))]
        return r

    for lead in range(0, 17, 3):
        reads = [util.revcomp(noisy(g0[50000 + lead:52500 + 2 * lead]))]                  # read 0 on the reverse strand, batch offset 0
        for k in range(18):                                                              # every residue of the read's offset and of the contig start
            s = 60000 + 1013 * k + k
            piece = noisy((g0 if k & 1 else g1)[s:s + 1800 + k])
            reads.append(util.revcomp(piece) if k % 3 == 0 else piece)
            reads.append(util.ACGT[rng.integers(0, 4, 1 + (k + lead) % 23)])              # shifts what follows by 1 .. 23 bases
        reads.append(util.revcomp(noisy(g1[70000:70300])))                               # the batch's last read: the words behind it are the slack
        bases, offsets = util.pack_reads(reads)
        _compare_dp(capi, oracle, world, bases, offsets, min_mapq=0)


def test_flanks_that_zdrop_after_the_query_has_ended(capi, oracle, world):
    """An extension whose best score comes early, whose query then declines by less than zdrop until it ends, and whose
    anti-diagonals go on declining through the rest of the target window: ksw2's Z-drop fires AFTER the query's last row has
    been passed, which clears reach_end.  The packed extension kernel may leave the end of the window out only when it can
    rule that out (k_fill.hip: the stop bound and its Z-drop condition; on these reads the bound alone already keeps it from
    stopping -- a best score far from the query's end leaves too much query to gain from --, the condition is the net below it)."""
    g0, g1 = world["seqs"][0], world["seqs"][1]
    rng = np.random.default_rng(123)
    junk = lambda n: util.ACGT[rng.integers(0, 4, n)]

    def sparse(piece):                    # ~90 % identity without a 15-mer in common: a base changed every ten
        p = piece.copy()
        for i in range(4, len(p), 10):
            p[i] = util.ACGT[(np.searchsorted(util.ACGT, p[i]) + 1 + rng.integers(0, 3)) % 4]
        return p

    reads = []
    for k, (nj, ns) in enumerate([(120, 40), (160, 60), (190, 60), (220, 30), (240, 10), (100, 90), (180, 70), (250, 0)]):
        s = 20000 + 5000 * k
        core = (g0 if k & 1 else g1)[s:s + 2500]
        left = np.concatenate([junk(nj), sparse((g0 if k & 1 else g1)[s - ns:s])]) if ns else junk(nj)
        right = np.concatenate([sparse((g0 if k & 1 else g1)[s + 2500:s + 2500 + ns]), junk(nj)]) if ns else junk(nj)
        r = np.concatenate([left, core, right])
        reads.append(r)
        reads.append(util.revcomp(r))
    bases, offsets = util.pack_reads(reads)
    _compare_dp(capi, oracle, world, bases, offsets, min_mapq=0)


# ---------------------------------------------------------------- reads of hundreds of kilobases; the length limit
def _noisy_indels(rng, seg, rate):
    """`rate` errors per base: 40 % substitutions, 30 % deletions, 30 % one-base insertions."""
    seg = seg.copy()
    r = rng.random(len(seg))
    sub = r < rate * 0.4
    seg[sub] = util.ACGT[rng.integers(0, 4, int(sub.sum()))]
    keep = ~((r >= rate * 0.4) & (r < rate * 0.7))
    ins = np.flatnonzero((r >= rate * 0.7) & (r < rate))
    pieces, last = [], 0
    for i in ins:
        pieces += [seg[last:i][keep[last:i]], util.ACGT[rng.integers(0, 4, 1)]]
        last = i
    pieces.append(seg[last:][keep[last:]])
    return np.concatenate(pieces)


@pytest.fixture(scope="module")
def world_2mbp(capi, oracle):
    g = synth.genome(0x2A2A, 2_000_000)
    other = synth.genome(0x2A2B, 300_000)
    return _world_from(capi, oracle, [synth.contig_name(0), synth.contig_name(1)], [g, other])


def test_reads_of_250_kb_and_900_kb(capi, oracle, world_2mbp):
    """index.map() takes a read of any length (aligner.py:193, 215).  Reads of 250 kb and 900 kb from a 2 Mbp contig --
    8 % errors, a 5 kb deletion in each, one on the reverse strand, short reads beside them -- region by region and
    CIGAR by CIGAR against the oracle: tens of thousands of anchors a read (the sort in HBM, the sequential backtrack),
    thousands of kernel calls a region, a region whose bases no LDS holds."""
    g = world_2mbp["seqs"][0]
    rng = np.random.default_rng(250)

    def long_read(start, length, indel_at, rate=0.08):
        return np.concatenate([_noisy_indels(rng, g[start:start + indel_at], rate),
                               _noisy_indels(rng, g[start + indel_at + 5000:start + length], rate)])

    short_b, short_o, _ = synth.reads(world_2mbp["seqs"], 5, 3000, seed=9)
    reads = [long_read(100_000, 250_000, 120_000), short_b[short_o[0]:short_o[1]],
             util.revcomp(long_read(600_000, 900_000, 400_000)), short_b[short_o[1]:short_o[2]]]
    assert 200_000 < len(reads[0]) < 260_000 and 800_000 < len(reads[2]) < (1 << 20)
    bases, offsets = util.pack_reads(reads)
    assign, best, nhits = _compare_dp(capi, oracle, world_2mbp, bases, offsets, min_mapq=0)
    assert assign[0] == 0 and assign[2] == 0
    regs = world_2mbp["eng"].dump(capi.DUMP_REGS, capi.REG_DTYPE)
    assert (regs["qe"] - regs["qs"]).max() > 800_000 and regs["n_cigar"].max() > 50_000
    _compare_dp(capi, oracle, world_2mbp, bases, offsets, min_mapq=60)


def test_a_read_beyond_the_length_limit_costs_only_itself(capi, oracle, world_2mbp):
    """A read of 2^20 bases or more is outside what the kernels hold: it alone comes back MNC_SKIPPED (no hits); every
    other read of its batch is classified exactly as without it (the oracle's answers)."""
    g = world_2mbp["seqs"][0]
    rng = np.random.default_rng(12)
    b, o, truth = synth.reads(world_2mbp["seqs"], 40, 4000, seed=3)
    normal = [b[o[i]:o[i + 1]] for i in range(40)]
    huge = _noisy_indels(rng, g[200_000:1_450_000], 0.05)
    exact = g[100_000:100_000 + (1 << 20)].copy()                        # exactly 2^20 bases: the first length refused
    just_below = g[300_000:300_000 + (1 << 20) - 1].copy()               # the longest length taken
    assert len(huge) > 1_150_000 and len(exact) == 1 << 20
    reads = normal[:20] + [huge] + normal[20:] + [exact, just_below]
    bases, offsets = util.pack_reads(reads)
    eng = world_2mbp["eng"]
    eng.set_contract(capi.CONTRACT_DP)
    assign, best, nhits = eng.classify(bases, offsets, 60)
    assert assign[20] == capi.SKIPPED and assign[41] == capi.SKIPPED and nhits[20] == 0 and nhits[41] == 0
    assert best[20].tolist() == (0, 0, 0, 0)
    keep = np.array([i for i in range(len(reads)) if i not in (20, 41)])
    kb, ko = util.pack_reads([reads[i] for i in keep])
    world_2mbp["oidx"].opt.cigar = 1
    oa, ob, onh, _ = world_2mbp["oidx"].classify(kb, ko, 60)
    assert np.array_equal(assign[keep], oa) and np.array_equal(nhits[keep], onh)
    for k in capi.HIT_DTYPE.names:
        assert np.array_equal(best[k][keep], ob[k]), k
    assert assign[42] == 0 and best["mlen"][42] == (1 << 20) - 1 and best["nm"][42] == 0     # an error-free read one base below the limit
    # a batch of nothing but such reads
    a2, b2, n2 = eng.classify(*util.pack_reads([exact, huge]), 60)
    assert a2.tolist() == [capi.SKIPPED] * 2 and n2.tolist() == [0, 0]
