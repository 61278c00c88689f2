"""Deterministic synthetic genomes and nanopore-like reads.

No real genome or read set is available offline, so the workloads of BASELINE.json are
generated: i.i.d. uniform ACGT contigs, optionally with point-diverged copies, and reads of a
fixed length with an i.i.d. substitution / insertion / deletion model.  The generator is a
counter-based SplitMix64 (``csrc/synth.cpp``): a value depends only on (seed, index), so every
rank and every machine produces identical bytes for the same ordinal.
"""
import ctypes as C

import numpy as np

from . import _capi

SEED_ECOLI = 0xEC011
SEED_20 = 0x20
SEED_20_DIV = 0x2020
SEED_500 = 0x500
SEED_READS = 0x5EED


def _mix(z):
    z &= (1 << 64) - 1
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & ((1 << 64) - 1)
    return z ^ (z >> 31)


def genome(seed, length):
    out = np.empty(length, dtype=np.uint8)
    _capi.check(_capi.lib().mnc_synth_genome(seed, length, out.ctypes.data))
    return out


def diverge(src, seed, rate_ppm):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    out = np.empty_like(src)
    _capi.check(_capi.lib().mnc_synth_diverge(src.ctypes.data, len(src), seed, rate_ppm, out.ctypes.data))
    return out


def contig_name(i):
    """``tax_unit:accession`` header convention of database.py:59."""
    return f"Genus{i}_species{i}:ACC{i:06d}.1"


def genome_set(n_genomes, seed=SEED_20, div_seed=SEED_20_DIV, min_len=2_000_000, max_len=7_000_000,
               diverged_half=True, rate_ppm=30_000):
    """n contigs with lengths uniform in [min_len, max_len]; when ``diverged_half`` the second
    half are ``rate_ppm`` point-diverged copies of the first half (so secondaries, sub-optimal
    scores and ambiguous decisions are exercised)."""
    names, seqs = [], []
    base = n_genomes // 2 if diverged_half else n_genomes
    for i in range(n_genomes):
        if i < base or not diverged_half:
            length = min_len + _mix(seed * 1_000_003 + i) % (max_len - min_len + 1)
            seqs.append(genome(_mix(seed) + i, int(length)))
        else:
            seqs.append(diverge(seqs[i - base], _mix(div_seed) + i, rate_ppm))
        names.append(contig_name(i))
    return names, seqs


def genome_set_repeats(n_genomes, seed=SEED_20, div_seed=SEED_20_DIV, min_len=2_000_000, max_len=7_000_000,
                       diverged_half=True, rate_ppm=30_000, rep_seed=0x4E9):
    """`genome_set`, made bacteria-like (opt-in: a sensitivity workload, not the BASELINE's headline): what the databases
    monica builds hold (database.py:52-67: whole bacterial genomes) and i.i.d. contigs do not --
      * 5-7 copies per genome of a 5 kb operon (rRNA-like): one ancestral operon, each genome's own 3-10 % away from it,
        its copies within the genome at 99.5-100 % identity;
      * 15-30 copies per genome of 1.3 kb insertion sequences at 97-100 % identity, drawn from five families that all
        genomes share;
      * 5 % of each genome shared with its neighbour (genome i + 1) at 85-95 % identity, in blocks of 10-50 kb.
    The elements overwrite stretches of the i.i.d. contigs in place (lengths stay what `genome_set` gives them);
    the diverged copies of the second half are made AFTER that, so they carry the same elements 3 % away.  Everything is
    a function of the seeds (numpy's PCG64 + the library's counter-based generators)."""
    base = n_genomes // 2 if diverged_half else n_genomes
    names, seqs = genome_set(base, seed=seed, div_seed=div_seed, min_len=min_len, max_len=max_len, diverged_half=False)
    rng = np.random.default_rng(rep_seed)
    operon0 = genome(_mix(rep_seed) + 1, 5000)
    families = [genome(_mix(rep_seed) + 10 + f, 1300) for f in range(5)]
    seqs = [np.array(s, copy=True) for s in seqs]

    def place(dst, piece, taken):
        for _ in range(100):                                   # a free stretch (elements do not overwrite one another)
            at = int(rng.integers(0, len(dst) - len(piece)))
            if not any(at < b and a < at + len(piece) for a, b in taken):
                dst[at:at + len(piece)] = piece
                taken.append((at, at + len(piece)))
                return
    for i, g in enumerate(seqs):
        taken = []
        own = diverge(operon0, _mix(rep_seed) + 1000 + i, int(rng.integers(30_000, 100_001)))
        for c in range(int(rng.integers(5, 8))):
            place(g, diverge(own, _mix(rep_seed) + 2000 + 16 * i + c, int(rng.integers(0, 5001))), taken)
        for c in range(int(rng.integers(15, 31))):
            fam = families[int(rng.integers(0, 5))]
            place(g, diverge(fam, _mix(rep_seed) + 3000 + 64 * i + c, int(rng.integers(0, 30_001))), taken)
    originals = [s.copy() for s in seqs]
    for i, g in enumerate(seqs):                               # 5 % from the neighbour, 85-95 % identical
        nb = originals[(i + 1) % len(seqs)]
        left = len(g) // 20
        k = 0
        while left > 0 and len(seqs) > 1:
            blk = min(left, int(rng.integers(10_000, 50_001)), len(nb) - 1, len(g) - 1)
            src = int(rng.integers(0, len(nb) - blk))
            at = int(rng.integers(0, len(g) - blk))
            g[at:at + blk] = diverge(nb[src:src + blk], _mix(rep_seed) + 5000 + 256 * i + k, int(rng.integers(50_000, 150_001)))
            left -= blk
            k += 1
    for i in range(base, n_genomes):
        seqs.append(diverge(seqs[i - base], _mix(div_seed) + i, rate_ppm))
        names.append(contig_name(i))
    return names, seqs


def ecoli_like():
    return [contig_name(0)], [genome(SEED_ECOLI, 4_641_652)]


def reads(genomes, n_reads, read_len=5000, seed=SEED_READS, first=0, sub=400, ins=300, dele=300,
          random_frac=200):
    """Returns (bases uint8[n_reads*read_len], offsets int64[n_reads+1], truth int32[n_reads]).
    Rates are in 1e-4 units: 4 % substitutions, 3 % insertions, 3 % deletions, 2 % pure-random
    reads (truth -1) by default."""
    n = len(genomes)
    keep = [np.ascontiguousarray(g, dtype=np.uint8) for g in genomes]
    ptrs = (C.c_void_p * n)(*[g.ctypes.data for g in keep])
    lens = (C.c_int64 * n)(*[len(g) for g in keep])
    out = np.empty(n_reads * read_len, dtype=np.uint8)
    truth = np.empty(n_reads, dtype=np.int32)
    _capi.check(_capi.lib().mnc_synth_reads(n, ptrs, lens, seed, first, n_reads, read_len,
                                            sub, ins, dele, random_frac, out.ctypes.data, truth.ctypes.data))
    offsets = np.arange(n_reads + 1, dtype=np.int64) * read_len
    return out, offsets, truth


class DeviceReads:
    """The same reads made in HBM (`mnc_synth_reads_device`, one wave per read; byte for byte the
    host generator's).  Holds the genomes on the device; `make` fills caller-owned torch tensors."""

    def __init__(self, genomes, device):
        import torch
        self.n = len(genomes)
        lens = np.array([len(g) for g in genomes], dtype=np.int64)
        off = np.zeros(self.n + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        cat = torch.empty(int(off[-1]), dtype=torch.uint8, device=device)
        for i, g in enumerate(genomes):
            cat[off[i]:off[i + 1]] = torch.from_numpy(np.ascontiguousarray(g, dtype=np.uint8)).to(device)
        self.genomes, self.offsets, self.device = cat, torch.from_numpy(off).to(device), device

    def make(self, out_bases, out_truth, n_reads, read_len=5000, seed=SEED_READS, first=0, sub=400, ins=300,
             dele=300, random_frac=200, stream=0):
        """out_bases: uint8[>= n_reads * read_len], out_truth: int32[>= n_reads] (or None), both on the
        device.  Asynchronous on `stream` (a raw hipStream_t; 0 = the default stream)."""
        assert out_bases.numel() >= n_reads * read_len and (out_truth is None or out_truth.numel() >= n_reads)
        _capi.check(_capi.lib().mnc_synth_reads_device(
            self.n, self.genomes.data_ptr(), self.offsets.data_ptr(), seed, first, n_reads, read_len, sub, ins, dele,
            random_frac, out_bases.data_ptr(), out_truth.data_ptr() if out_truth is not None else None, stream or None))


def write_fasta(path, names, seqs, width=80, gz=None):
    """Write contigs as (optionally gzipped) FASTA."""
    import gzip
    gz = path.endswith(".gz") if gz is None else gz
    opener = gzip.open if gz else open
    with opener(path, "wb") as f:
        for name, s in zip(names, seqs):
            f.write(b">" + name.encode() + b"\n")
            b = np.ascontiguousarray(s, dtype=np.uint8).tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")


def write_fastq(path, bases, offsets, ids=None, qual=b"I"):
    with open(path, "wb") as f:
        b = np.ascontiguousarray(bases, dtype=np.uint8).tobytes()
        for r in range(len(offsets) - 1):
            s = b[offsets[r]:offsets[r + 1]]
            rid = ids[r] if ids is not None else f"read{r}"
            f.write(b"@" + rid.encode() + b"\n" + s + b"\n+\n" + qual * len(s) + b"\n")
