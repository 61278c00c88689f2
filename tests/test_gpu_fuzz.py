"""Randomised differential test: worlds and reads drawn from a seeded generator, every stage of
the HIP path compared with the oracle (the per-stage comparison of test_gpu_parity).  Covers
mixtures the hand-written cases do not: tandem repeats inside genomes, diverged copies at
several distances, low-complexity and ambiguous stretches inside reads, chimeric reads, very
short and fairly long reads, many error rates."""
import numpy as np
import pytest

from monica_amd import synth
import util
from test_gpu_parity import _compare_batch, _world_from

pytestmark = pytest.mark.gpu


def _random_world(rng, k):
    n_base = int(rng.integers(1, 4))
    seqs = []
    for g in range(n_base):
        length = int(rng.integers(30_000, 160_000))
        s = synth.genome(0xF000 + 97 * k + g, length).copy()
        # tandem repeats and a duplicated segment inside the genome
        for _ in range(int(rng.integers(0, 4))):
            unit = int(rng.integers(5, 400))
            reps = int(rng.integers(2, 40))
            at = int(rng.integers(0, length - unit * reps - 1)) if length > unit * reps + 1 else 0
            s[at:at + unit * reps] = np.tile(s[at:at + unit], reps)[: unit * reps]
        if rng.random() < 0.5 and length > 20_000:
            a, b = int(rng.integers(0, length // 2 - 5000)), int(rng.integers(length // 2, length - 5000))
            s[b:b + 4000] = s[a:a + 4000]
        seqs.append(s)
    for g in range(n_base):
        for _ in range(int(rng.integers(0, 3))):
            seqs.append(synth.diverge(seqs[g], int(rng.integers(1, 1 << 30)), int(rng.choice([2_000, 10_000, 30_000, 80_000]))))
    names = [synth.contig_name(i) for i in range(len(seqs))]
    if rng.random() < 0.3 and len(seqs) > 1:                       # two contigs of one genome
        names[-1] = names[0]
    return names, seqs


def _random_reads(rng, seqs, n):
    reads = []
    for _ in range(n):
        kind = rng.random()
        g = seqs[int(rng.integers(0, len(seqs)))]
        length = int(rng.choice([20, 60, 150, 400, 1200, 3000, 7000, 12000]))
        length = min(length, len(g) - 1)
        at = int(rng.integers(0, len(g) - length))
        r = g[at:at + length].copy()
        if rng.random() < 0.5:
            r = util.revcomp(r)
        rate = float(rng.choice([0.0, 0.01, 0.05, 0.12, 0.25]))
        m = rng.random(len(r)) < rate
        r[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(m.sum()))]
        if kind < 0.10 and len(r) > 50:                            # low-complexity stretch
            a = int(rng.integers(0, len(r) - 40))
            r[a:a + 40] = np.tile(np.frombuffer(rng.choice([b"A", b"AC", b"ACG", b"AATT"]), dtype=np.uint8), 40)[:40]
        elif kind < 0.20 and len(r) > 30:                          # ambiguous bases
            for _ in range(int(rng.integers(1, 4))):
                a = int(rng.integers(0, len(r) - 3))
                r[a:a + int(rng.integers(1, 3))] = ord("N")
        elif kind < 0.30:                                          # chimeric read
            g2 = seqs[int(rng.integers(0, len(seqs)))]
            l2 = min(int(rng.integers(200, 3000)), len(g2) - 1)
            a2 = int(rng.integers(0, len(g2) - l2))
            r = np.concatenate([r, g2[a2:a2 + l2]])
        elif kind < 0.35:                                          # pure noise
            r = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, len(r))]
        if rng.random() < 0.1:
            r = np.char.lower(r.tobytes().decode()).encode() if False else np.frombuffer(r.tobytes().lower(), dtype=np.uint8)
        reads.append(np.ascontiguousarray(r))
    return reads


@pytest.mark.parametrize("k", range(10))
def test_random_world(capi, oracle, k):
    rng = np.random.default_rng(0xF00D + k)
    names, seqs = _random_world(rng, k)
    w = _world_from(capi, oracle, names, seqs)
    reads = _random_reads(rng, seqs, 70)
    bases, offsets = util.pack_reads(reads)
    _compare_batch(capi, oracle, w, bases, offsets, min_mapq=int(rng.choice([0, 20, 60])))
    if k % 3 == 0:
        w["eng"].set_debug(2)                                      # every look-back through HBM
        _compare_batch(capi, oracle, w, bases, offsets, min_mapq=60)
        w["eng"].set_debug(0)
    # every region planned by the LANE form of the plan kernel (a batch this small takes the wave form, mnc_dp_plan_long,
    # by itself; large batches use the lane form for all but long reads' regions)
    w["eng"].set_debug(0x800000)
    _compare_batch(capi, oracle, w, bases, offsets, min_mapq=0)
    w["eng"].set_debug(0)
