"""CPU checks of two arguments the alignment kernels rest on (csrc/k_fill.hip, csrc/k_align.hip), stated here in a few
lines of Python and tried on random cases:

* the Z-drop pre-test of the gap-filling kernels: minimap2's `mm_test_zdrop` (the oracle's
  oracle/mm_align.c follows it) walks the alignment base by base; the kernels skip that walk when
  (largest sum over contiguous CIGAR operations of gap cost - a per M column) + (a M - S - two-piece gap costs)
  is at most the threshold.  The bound must never be below the drop the walk would find.
* `mm_update_extra`'s running score s := max(s + d, 0) as composed maps x -> max(x + A, Bv), cut into shares
  (the stitch kernel's lanes): composing the shares' maps must give the sequential result and maximum."""
import numpy as np

A_, B_, Q_, E_, Q2_, E2_, AMBI = 2, 4, 4, 2, 24, 1, 1


def gap2(l):
    return min(Q_ + E_ * l, Q2_ + E2_ * l)


def random_alignment(rng):
    """CIGAR as (op, len) in walk order (start -> end), per-column scores of the M runs (a / -b / -ambi)."""
    ops, cols = [], []
    for _ in range(int(rng.integers(1, 40))):
        if ops and ops[-1][0] == 0 or not ops and rng.random() < 0.5 or ops and ops[-1][0] != 0 and rng.random() < 0.2:
            op = int(rng.integers(1, 3))
            ln = int(rng.integers(1, 60)) if rng.random() < 0.1 else int(rng.integers(1, 4))
            if ops and ops[-1][0] == op:
                continue
            ops.append((op, ln)), cols.append(None)
        else:
            ln = int(rng.integers(1, 50))
            err = rng.choice([0.0, 0.05, 0.3, 0.8])
            sc = np.where(rng.random(ln) < err, np.where(rng.random(ln) < 0.15, -AMBI, -B_), A_)
            if ops and ops[-1][0] == 0:
                continue
            ops.append((0, ln)), cols.append(sc)
    if not any(o == 0 for o, _ in ops):
        ops.append((0, 5)), cols.append(np.full(5, A_))
    return ops, cols


def walk_max_zdrop(ops, cols):
    """mm_test_zdrop's walk: the largest (max so far - score - |diagonal change| e) over the alignment."""
    score, mx, mi, mj, i, j, worst = 0, -(1 << 30), -1, -1, 0, 0, 0
    for (op, ln), sc in zip(ops, cols):
        if op == 0:
            for l in range(ln):
                score += int(sc[l])
                if score < mx:
                    li, lj = i + l - mi, j + l - mj
                    worst = max(worst, mx - score - abs(li - lj) * E_)
                else:
                    mx, mi, mj = score, i + l, j + l
            i += ln
            j += ln
        else:
            score -= Q_ + E_ * ln
            if op == 1:
                j += ln
            else:
                i += ln
            if score < mx:
                li, lj = i - mi, j - mj
                worst = max(worst, mx - score - abs(li - lj) * E_)
            else:
                mx, mi, mj = score, i, j
    return worst


def kernel_bound(ops, cols):
    S = sum(int(sc.sum()) for sc in cols if sc is not None) - sum(gap2(ln) for op, ln in ops if op != 0)
    mcols = sum(ln for op, ln in ops if op == 0)
    g2 = sum(gap2(ln) for op, ln in ops if op != 0)
    kd = kbest = 0
    for op, ln in reversed(ops):                       # the kernels see the operations last one first
        if op == 0:
            kd = max(kd - A_ * ln, 0)
        else:
            kd += Q_ + E_ * ln
            kbest = max(kbest, kd)
    return kbest + (A_ * mcols - S - g2)


def test_zdrop_pretest_bound_is_never_below_the_walk():
    rng = np.random.default_rng(20261004)
    tight = 0
    for _ in range(3000):
        ops, cols = random_alignment(rng)
        drop, bound = walk_max_zdrop(ops, cols), kernel_bound(ops, cols)
        assert bound >= drop, (ops, drop, bound)
        tight += bound <= 200 < sum(Q_ + E_ * ln for op, ln in ops if op) + B_ * sum(int((sc < 0).sum()) for sc in cols if sc is not None)
    assert tight > 20            # and it does decide cases the sum of all negative steps could not


def test_score_maps_compose_to_the_sequential_scan():
    rng = np.random.default_rng(7)
    for _ in range(300):
        d = rng.choice([A_, -B_, -AMBI, -(Q_ + E_), -(Q_ + 3 * E_)], int(rng.integers(1, 400)), p=[0.8, 0.08, 0.02, 0.07, 0.03])
        s = best = 0
        for x in d:
            s = max(s + int(x), 0)
            best = max(best, s)
        cuts = np.sort(rng.integers(0, len(d) + 1, int(rng.integers(0, 64))))
        x_in, top = 0, 0
        for share in np.split(d, cuts):                # a lane's share: A, Bv and their running maxima
            NONE = -(1 << 28)
            A = 0
            Bv = MA = MB = NONE
            for x in share:
                A += int(x)
                Bv = max(Bv + int(x), 0)
                MA, MB = max(MA, A), max(MB, Bv)
            if len(share):
                top = max(top, x_in + MA, MB)
            x_in = max(x_in + A, Bv)
        assert x_in == s and top == best
