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


# ---------------------------------------------------------------- the band proof of the gap-filling kernels
def fill_band(n, m, cells):
    d = n - m
    b = (2 * cells - 2 - abs(d)) // 2
    kmin = min(d, 0) - b
    if kmin & 1:
        kmin -= 1
    return b, kmin, kmin + 2 * cells - 1


def band_bound(n, m, kmin, kmax):
    """csrc/k_fill.hip: dp_band_bound -- the best score of any global path that leaves the band of offsets kmin..kmax."""
    d, U = n - m, -(1 << 28)
    D0, I0 = kmax + 1, kmax + 1 - d
    if n - D0 >= 0 and m - I0 >= 0:
        U = max(U, A_ * (n - D0) - gap2(D0) - gap2(I0))
    I0 = 1 - kmin
    D0 = I0 + d
    if m - I0 >= 0 and n - D0 >= 0:
        U = max(U, A_ * (m - I0) - gap2(I0) - gap2(D0))
    return U


def global_score(t, q, lo=None, hi=None):
    """Two-piece affine global alignment score (ksw2's recurrence without its band), optionally with the cells whose
    offset i - j lies outside [lo, hi] forbidden."""
    NEG = -(1 << 28)
    n, m = len(t), len(q)
    inside = (lambda i, j: True) if lo is None else (lambda i, j: lo <= i - j <= hi)
    H = np.full((n + 1, m + 1), NEG, dtype=np.int64)
    E1, E2, F1, F2 = H.copy(), H.copy(), H.copy(), H.copy()
    H[0, 0] = 0
    for i in range(n + 1):
        for j in range(m + 1):
            if i == 0 and j == 0:
                continue
            if i and j and not inside(i - 1, j - 1):
                continue
            if i == 0:
                H[i, j] = -gap2(j)
                continue
            if j == 0:
                H[i, j] = -gap2(i)
                continue
            E1[i, j] = max(E1[i - 1, j], H[i - 1, j] - Q_) - E_
            E2[i, j] = max(E2[i - 1, j], H[i - 1, j] - Q2_) - E2_
            F1[i, j] = max(F1[i, j - 1], H[i, j - 1] - Q_) - E_
            F2[i, j] = max(F2[i, j - 1], H[i, j - 1] - Q2_) - E2_
            H[i, j] = max(H[i - 1, j - 1] + (A_ if t[i - 1] == q[j - 1] else -B_), E1[i, j], E2[i, j], F1[i, j], F2[i, j])
    return int(H[n, m])


def test_a_banded_score_above_the_bound_is_the_full_matrix_score():
    rng = np.random.default_rng(99)
    proved = refused = 0
    for _ in range(250):
        n = int(rng.integers(12, 70))
        t = rng.integers(0, 4, n)
        q = list(t)
        for _ in range(int(rng.integers(0, 8))):                        # a few substitutions and indels, sometimes a long one
            k = int(rng.integers(0, len(q) + 1))
            what = rng.random()
            if what < 0.4 and k < len(q):
                q[k] = int(rng.integers(0, 4))
            elif what < 0.7:
                q[k:k] = list(rng.integers(0, 4, int(rng.integers(1, 12 if rng.random() < 0.2 else 3))))
            else:
                del q[k:k + int(rng.integers(1, 12 if rng.random() < 0.2 else 3))]
        m = len(q)
        if m < 4:
            continue
        for cells in (4, 8, 16):
            b, kmin, kmax = fill_band(n, m, cells)
            if b < 1:
                continue
            S_band, U = global_score(t, q, kmin, kmax), band_bound(n, m, kmin, kmax)
            if S_band > U:
                assert S_band == global_score(t, q), (n, m, cells)
                proved += 1
            else:
                refused += 1
    assert proved > 100 and refused > 30


# ---------------------------------------------------------------- a sequence-aware bound (examined in round 3, not used)
def banded_matrix(t, q, lo, hi):
    """H of the banded two-piece affine DP, cells with offset i - j outside [lo, hi] forbidden (NEG)."""
    NEG = -(1 << 28)
    n, m = len(t), len(q)
    H = np.full((n + 1, m + 1), NEG, dtype=np.int64)
    E1, E2, F1, F2 = H.copy(), H.copy(), H.copy(), H.copy()
    H[0, 0] = 0
    for i in range(n + 1):
        for j in range(max(0, i - hi - 1), min(m, i - lo + 1) + 1):
            if i == 0 and j == 0:
                continue
            if i and j and not lo <= (i - 1) - (j - 1) <= hi:
                continue
            if i == 0:
                H[i, j] = -gap2(j)
                continue
            if j == 0:
                H[i, j] = -gap2(i)
                continue
            E1[i, j] = max(E1[i - 1, j], H[i - 1, j] - Q_) - E_
            E2[i, j] = max(E2[i - 1, j], H[i - 1, j] - Q2_) - E2_
            F1[i, j] = max(F1[i, j - 1], H[i, j - 1] - Q_) - E_
            F2[i, j] = max(F2[i, j - 1], H[i, j - 1] - Q2_) - E2_
            H[i, j] = max(H[i - 1, j - 1] + (A_ if t[i - 1] == q[j - 1] else -B_), E1[i, j], E2[i, j], F1[i, j], F2[i, j])
    return H


def edge_bound(H, n, m, lo, hi):
    """A path that leaves the band leaves it for the first time from a cell on one of its two edges; up to there it is a
    path inside the band, so it has scored at most the banded H of that cell (whatever gap state it is in: H is the
    largest).  Beyond: at least one more gap base to be outside (continuing a gap costs min(e, e2) = 1), a fresh gap of
    the other kind to come back to the final offset n - m, and at most a per base for what is left of the shorter
    sequence.  """
    NEG = -(1 << 28)
    d, best = n - m, NEG
    for i in range(n + 1):
        for j, out_i, out_j in ((i - hi, i + 1, i - hi), (i - lo, i, i - lo + 1)):      # upper edge leaves down, lower edge right
            if not 0 <= j <= m or H[i, j] <= NEG // 2 or out_i > n or out_j > m:
                continue
            back = abs((out_i - out_j) - d)                     # gap bases of the other kind needed to end on the corner
            rest = min(n - out_i - (back if out_i - out_j < d else 0), m - out_j - (back if out_i - out_j > d else 0))
            if rest < 0:
                continue
            best = max(best, int(H[i, j]) - min(E_, E2_) - gap2(back) + A_ * rest)
    return best


def test_a_bound_from_the_band_edge_cells_is_valid_and_hardly_tighter():
    rng = np.random.default_rng(7)
    by_u = by_edge = cases = 0
    for _ in range(70):
        n = int(rng.integers(30, 120))
        t = rng.integers(0, 4, n)
        q = []
        for x in t:                                             # ~10 % errors, as the benchmark's reads
            r = rng.random()
            if r < 0.04:
                q.append(int(rng.integers(0, 4)))
            elif r < 0.07:
                q.extend([int(x), int(rng.integers(0, 4))])
            elif r < 0.10:
                continue
            else:
                q.append(int(x))
        m = len(q)
        for cells in (4, 6, 8):
            b, kmin, kmax = fill_band(n, m, cells)
            if b < 1 or m < 8:
                continue
            H = banded_matrix(t, q, kmin, kmax)
            S_band, U, Ue = int(H[n, m]), band_bound(n, m, kmin, kmax), edge_bound(H, n, m, kmin, kmax)
            cases += 1
            if S_band > U:
                by_u += 1
            if S_band > Ue:                                     # the claim: then no path outside the band does as well
                assert S_band == global_score(t, q), (n, m, cells)
                by_edge += 1
    # valid -- and no lever: the largest edge-cell bound sits at the first anti-diagonals, where nothing has been
    # scored yet and all that follows is priced as perfect matches
    assert cases > 150 and by_edge <= by_u + cases // 10
    print(f"band proofs out of {cases}: {by_u} by the length bound, {by_edge} by the edge-cell bound")
