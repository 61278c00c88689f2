"""Shared builders for the parity tests: small synthetic genome sets and read batches that
include the edge cases of the domain (empty / shorter-than-k reads, ambiguous bases, lower
case, low-complexity sequence, chimeras, long deletions)."""
import numpy as np

from monica_amd import synth

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, dtype=np.uint8)
COMP[:] = ord("N")
for a, b in zip(b"ACGTacgt", b"TGCAtgca"):
    COMP[a] = b


def revcomp(a):
    return COMP[np.asarray(a, dtype=np.uint8)][::-1].copy()


def small_genomes(n=4, min_len=200_000, max_len=300_000, seed=0x20):
    return synth.genome_set(n, seed=seed, min_len=min_len, max_len=max_len)


def pack_reads(read_list):
    """list of uint8 arrays / bytes -> (bases, offsets)"""
    arrs = [np.frombuffer(r, dtype=np.uint8) if isinstance(r, (bytes, bytearray)) else np.asarray(r, dtype=np.uint8)
            for r in read_list]
    offsets = np.zeros(len(arrs) + 1, dtype=np.int64)
    for i, a in enumerate(arrs):
        offsets[i + 1] = offsets[i] + len(a)
    bases = np.concatenate(arrs) if arrs and offsets[-1] > 0 else np.zeros(0, dtype=np.uint8)
    return bases, offsets


def edge_reads(seqs, rng):
    """Hand-made reads for the corner cases; returns a list of uint8 arrays."""
    g0, g1 = seqs[0], seqs[1]
    out = []
    out.append(np.zeros(0, dtype=np.uint8))                       # empty
    out.append(g0[1000:1010].copy())                              # shorter than k
    out.append(g0[1000:1015].copy())                              # exactly one k-mer
    out.append(g0[1000:1020].copy())                              # fewer k-mers than w
    out.append(g0[1000:1024].copy())                              # exactly w k-mers
    out.append(g0[1000:1025].copy())                              # w + 1 k-mers
    out.append(g0[5000:8000].copy())                              # error-free, forward
    out.append(revcomp(g0[5000:8000]))                            # error-free, reverse
    r = g0[20000:24000].copy(); r[1500] = ord("N"); out.append(r)  # one ambiguous base
    r = g0[30000:34000].copy(); r[100:130] = ord("N"); r[3990:] = ord("N"); out.append(r)
    r = g1[30000:33000].copy(); r[::500] = ord("n"); out.append(r)  # periodic ambiguity
    out.append(np.frombuffer(g0[40000:43000].tobytes().lower(), dtype=np.uint8).copy())   # lower case
    out.append(np.full(3000, ord("A"), dtype=np.uint8))            # homopolymer
    out.append(np.tile(np.frombuffer(b"AC", dtype=np.uint8), 1500))            # dinucleotide repeat
    out.append(np.tile(np.frombuffer(b"ACGTTGCAGT", dtype=np.uint8), 300))     # period-10 repeat
    out.append(np.concatenate([g0[50000:50400], np.tile(np.frombuffer(b"GATTACA", dtype=np.uint8), 60),
                               g0[50400:52000]]))                # repeat inside a real read
    out.append(np.concatenate([np.full(40, ord("T"), dtype=np.uint8), g1[60000:62000]]))  # homopolymer head
    out.append(np.concatenate([g0[70000:72500], g1[90000:92500]]))             # chimera: 2 primaries
    out.append(np.concatenate([g0[100000:102500], g0[103500:106000]]))         # 1 kb deletion: long-join
    out.append(np.concatenate([g0[110000:112000], ACGT[rng.integers(0, 4, 900)], g0[112000:114000]]))  # insertion
    out.append(ACGT[rng.integers(0, 4, 5000)])                    # random: unmapped
    out.append(np.frombuffer(b"N" * 500, dtype=np.uint8).copy())  # all ambiguous
    return out


def edge_reads_small(seqs, rng):
    """Corner cases scaled for the small golden fixture (contigs of 50-70 kb)."""
    g0, g1 = seqs[0], seqs[1]
    out = [np.zeros(0, dtype=np.uint8), g0[100:110].copy(), g0[100:115].copy(), g0[100:124].copy(), g0[100:125].copy()]
    out.append(g0[5000:7000].copy())
    out.append(revcomp(g0[5000:7000]))
    r = g0[9000:11000].copy(); r[700] = ord("N"); out.append(r)
    r = g1[9000:11000].copy(); r[::300] = ord("n"); out.append(r)
    out.append(np.frombuffer(g0[12000:14000].tobytes().lower(), dtype=np.uint8).copy())
    out.append(np.full(1500, ord("A"), dtype=np.uint8))
    out.append(np.tile(np.frombuffer(b"AC", dtype=np.uint8), 800))
    out.append(np.concatenate([g0[15000:15400], np.tile(np.frombuffer(b"GATTACA", dtype=np.uint8), 50), g0[15400:16600]]))
    out.append(np.concatenate([g0[20000:21500], g1[30000:31500]]))
    out.append(np.concatenate([g0[33000:35500], g0[36500:39000]]))
    out.append(ACGT[rng.integers(0, 4, 2000)])
    return out
