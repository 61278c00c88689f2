"""mm_align1's seed filters, checked against their PUBLISHED DESCRIPTION rather than against a second copy of the loops.

Before any base is aligned minimap2 (2.17, align.c) cleans a region's seeds up:
  * mm_fix_bad_ends: a chain end whose first (last) seeds are off the diagonal of what follows is trimmed -- a seed
    step whose reference and query advance differ by more than half the length accumulated so far moves the start there;
  * mm_filter_bad_seeds: an insertion and a deletion near each other that cancel (2 x min(inserted, deleted) > 40, within
    max_gap / 2 = 2 500 bases and ten long gaps) are an artefact of a bad seed: the seeds between them are ignored;
  * mm_filter_bad_seeds_alt: two gaps of 30+ bases with less matching sequence between them than they are long become
    one long-join gap: the seeds between are ignored, the seed behind carries the long-join flag.
The oracle (oracle/mm_align.c) and the GPU's plan kernel (csrc/k_align.hip: mnc_dp_plan) hold the same statement of
these loops; here the oracle's is exercised through a test hook on hand-made seed layouts whose outcome follows from the
description alone, and on random layouts through invariants (GPU == oracle on such reads is tests/test_gpu_dp.py).
"""
import ctypes as C

import numpy as np
import pytest

SPAN = 15
LONG_JOIN, IGNORE = 1 << 40, 1 << 41


@pytest.fixture(scope="module")
def run(oracle):
    L = oracle.lib()
    opt = oracle.Opt()
    L.orc_opt_init(C.byref(opt))

    def f(seeds, mlen=None):
        """seeds: [(reference end position, query end position)], in chain order"""
        a = np.zeros(len(seeds), dtype=oracle.A128_DTYPE)
        a["x"] = [x for x, _ in seeds]
        a["y"] = [(SPAN << 32) | y for _, y in seeds]
        as1, cnt1 = C.c_int32(), C.c_int32()
        L.orc_test_seed_filters(C.byref(opt), len(seeds), a.ctypes.data_as(C.c_void_p), mlen if mlen is not None else SPAN * len(seeds),
                                C.byref(as1), C.byref(cnt1))
        flags = a["y"].astype(np.uint64)
        return as1.value, cnt1.value, (flags & np.uint64(IGNORE)) != 0, (flags & np.uint64(LONG_JOIN)) != 0
    return f


def diagonal(n, step=30, x0=1000, y0=100):
    return [(x0 + i * step, y0 + i * step) for i in range(n)]


def shift(seeds, frm, dq=0, dr=0):
    """an indel in front of seed `frm`: every seed from there on moves by dq on the query / dr on the reference"""
    return [(x + (dr if i >= frm else 0), y + (dq if i >= frm else 0)) for i, (x, y) in enumerate(seeds)]


def test_clean_chain_is_left_alone(run):
    as1, cnt1, ign, lj = run(diagonal(60))
    assert (as1, cnt1) == (0, 60) and not ign.any() and not lj.any()


def test_small_indels_are_not_long_gaps(run):
    s = diagonal(60)
    for frm, d in ((10, 5), (20, -8), (30, 10), (45, -10)):          # |gap| <= 10: below both filters' thresholds
        s = shift(s, frm, dq=d if d > 0 else 0, dr=-d if d < 0 else 0)
    _, _, ign, lj = run(s)
    assert not ign.any() and not lj.any()


def test_an_insertion_cancelled_by_a_deletion_marks_the_seeds_between(run):
    # a 50-base insertion in front of seed 20, a 50-base deletion in front of seed 26: 2 x min(50, 50) = 100 > 40
    s = shift(shift(diagonal(60), 20, dq=50), 26, dr=50)
    as1, cnt1, ign, lj = run(s)
    assert ign[20:26].all() and not ign[:20].any() and not ign[26:].any()
    # the same two gaps, smaller than the threshold together: 2 x min(15, 15) = 30 <= 40
    s = shift(shift(diagonal(60), 20, dq=15), 26, dr=15)
    assert not run(s)[2].any()
    # only the part that cancels counts: 100 inserted, 12 deleted -> 24 <= 40
    s = shift(shift(diagonal(60), 20, dq=100), 26, dr=12)
    assert not run(s)[2][20:26].all()


def test_an_insertion_without_a_partner_marks_nothing(run):
    _, _, ign, lj = run(shift(diagonal(60), 20, dq=80))
    assert not ign.any() and not lj.any()


def test_partners_further_apart_than_half_max_gap_do_not_pair(run):
    s = shift(shift(diagonal(200, step=40), 20, dq=60), 120, dr=60)    # 100 seeds x 40 = 4 000 bases apart > 2 500
    _, _, ign, _ = run(s)
    assert not ign.any()
    s = shift(shift(diagonal(200, step=40), 20, dq=60), 70, dr=60)     # 2 000 bases apart
    assert run(s)[2][20:70].all()


def test_two_long_gaps_with_little_between_them_become_one_long_join(run):
    # a 40-base and a 50-base insertion with 2 x 30 = 60 bases of seeds between them (< 40 + 50): mm_filter_bad_seeds_alt
    s = shift(shift(diagonal(60), 20, dq=40), 23, dq=50)
    _, _, ign, lj = run(s)
    assert ign[20:23].all() and lj[23] and lj.sum() == 1 and not ign[23:].any() and not ign[:20].any()
    # the same gaps far apart: 20 seeds x 30 = 600 bases of matches between them
    s = shift(shift(diagonal(80), 20, dq=40), 40, dq=50)
    _, _, ign, lj = run(s)
    assert not ign.any() and not lj.any()


def test_an_off_diagonal_chain_end_is_trimmed(run):
    # the first three seeds sit 40 bases off the diagonal of the other fifty: the step into the main diagonal differs by
    # 40 > (accumulated length) / 2 -> the region starts at the first seed of the main diagonal
    s = [(1000 + i * 25, 140 + i * 25) for i in range(3)] + [(1000 + (i + 3) * 25, 100 + (i + 3) * 25) for i in range(50)]
    as1, cnt1, _, _ = run(s)
    assert as1 == 3 and cnt1 == 50
    # the same at the far end
    s = [(1000 + i * 25, 100 + i * 25) for i in range(50)] + [(1000 + (i + 50) * 25, 150 + (i + 50) * 25) for i in range(2)]
    as1, cnt1, _, _ = run(s)
    assert as1 == 0 and cnt1 == 50
    # an indel deep inside the chain is not an end problem
    s = shift(diagonal(80), 40, dq=45)
    as1, cnt1, _, _ = run(s)
    assert (as1, cnt1) == (0, 80)


def test_random_layouts_keep_the_invariants(run):
    """Whatever the layout: the kept range is a sub-range of the chain; flags only fall strictly inside it, never on its
    first seed; a long-join flag is preceded by an ignored seed; without a gap of more than ten bases nothing is flagged
    and -- the chain being one diagonal then -- nothing is trimmed."""
    rng = np.random.default_rng(3)
    flagged = 0
    for it in range(600):
        n = int(rng.integers(5, 120))
        s = diagonal(n, step=int(rng.integers(16, 60)))
        big = False
        for _ in range(int(rng.integers(0, 6))):
            frm = int(rng.integers(1, n))
            d = int(rng.integers(-120, 121)) if it % 3 else int(rng.integers(-3, 4))
            s = shift(s, frm, dq=max(d, 0), dr=max(-d, 0))
            big = big or abs(d) > 3
        as1, cnt1, ign, lj = run(s)
        assert 0 <= as1 and cnt1 >= 1 and as1 + cnt1 <= n
        assert not ign[:as1 + 1].any() and not ign[as1 + cnt1:].any()
        assert not lj[:as1 + 1].any() and not lj[as1 + cnt1:].any()
        for i in np.flatnonzero(lj):
            assert ign[i - 1]
        if not big:                                                # every step within 3 x 5 = 15 < 16 bases of the diagonal's
            assert not lj.any() and (as1, cnt1) == (0, n)
        flagged += int(ign.any())
    assert flagged > 50
