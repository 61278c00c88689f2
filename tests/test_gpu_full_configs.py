"""BASELINE.json configs 3 and 4 at their real shape, rehearsed on ONE MI355X (`pytest -m gpu`).

  config 3  10 000 000 reads of 5 kb vs the 20-genome index, read-sharded over 8 ranks with one
            count reduce: here the 8 rank shares run one after the other on the one GPU, each as
            10 blocks of 125 000 reads generated in HBM by ordinal, per-rank count tables summed
            (the all-reduce's arithmetic).  Checked: the sum equals the counts of the per-read
            decisions of all 10 M reads in all three counting modes; mapped reads agree with the
            generator's truth; two samples (ordinals 0.. and 7 300 000..) equal the oracle read for
            read; and the share boundaries are the ones dist.shard_bounds gives the ranks.
  config 4  the 500-genome recipe (seed 0x500) in 8 index parts of 62 / 63 genomes -- the split
            dist.shard_bounds gives 8 GPUs -- every part maps ALL reads, MAPQ and the gate per part
            (aligner.py:91-103), per-read summaries merged by best_hit's rule (aligner.py:219-233,
            328-339): all 1 000 000 reads of the config, resident in HBM; 4 000 of them against the oracle's
            multi-part loop, all of them against truth and the count table.
The reference's call sequence for both is multi_threaded_aligner's loop (aligner.py:89-103).
"""
import os

import numpy as np
import pytest

from monica_amd import synth

pytestmark = pytest.mark.gpu

N_THREADS = min(16, os.cpu_count() or 1)


def test_device_read_generator_equals_the_host_generator(capi):
    """mnc_synth_reads_device == mnc_synth_reads byte for byte, truth included: ordinals far from 0,
    contigs shorter than a read's span (the read runs off the end), ambiguous bases in the source."""
    import torch
    dev = torch.device("cuda:0")
    names, seqs = synth.genome_set(6, min_len=30_000, max_len=90_000)
    seqs.append(synth.genome(77, 3_000))                           # shorter than one read
    seqs.append(synth.genome(78, 5_600))                           # shorter than read + slack
    seqs[0] = seqs[0].copy()
    seqs[0][1000:1040] = ord("N")
    seqs[0][20_000::977] = ord("n")
    gen = synth.DeviceReads(seqs, dev)
    for n, read_len, first, seed, rates in ((3000, 5000, 0, synth.SEED_READS + 3, {}),
                                           (2000, 1000, 9_999_000, 12345, dict(sub=900, ins=800, dele=700, random_frac=1000)),
                                           (500, 333, 1 << 40, 7, dict(sub=0, ins=0, dele=0, random_frac=0))):
        hb, ho, ht = synth.reads(seqs, n, read_len, seed=seed, first=first, **rates)
        d_b = torch.zeros(n * read_len, dtype=torch.uint8, device=dev)
        d_t = torch.full((n,), -7, dtype=torch.int32, device=dev)
        gen.make(d_b, d_t, n, read_len, seed=seed, first=first, **rates)
        torch.cuda.synchronize()
        assert np.array_equal(d_t.cpu().numpy(), ht)
        assert np.array_equal(d_b.cpu().numpy(), hb)


def test_config3_ten_million_reads_in_eight_rank_shares(capi, oracle):
    import torch
    from monica_amd import dist as mdist
    dev = torch.device("cuda:0")
    names, seqs = synth.genome_set(20)
    idx = capi.Index.from_seqs(names, seqs)
    eng = capi.Engine(idx, 0)
    gen = synth.DeviceReads(seqs, dev)
    total, world, block, read_len, seed = 10_000_000, 8, 125_000, 5000, synth.SEED_READS + 3
    n_genomes = len(idx.genome_names)
    d_bases = torch.empty(block * read_len, dtype=torch.uint8, device=dev)
    d_truth = torch.empty(block, dtype=torch.int32, device=dev)
    d_off = torch.arange(block + 1, dtype=torch.int64, device=dev) * read_len
    d_assign = torch.empty(block, dtype=torch.int32, device=dev)
    d_best = torch.zeros(block * 4, dtype=torch.int32, device=dev)
    d_nhits = torch.zeros(block, dtype=torch.int32, device=dev)
    assign_all = np.empty(total, dtype=np.int32)
    mlen_all = np.empty(total, dtype=np.int32)
    truth_all = np.empty(total, dtype=np.int32)
    rank_counts = []
    covered = 0
    for rank in range(world):
        lo, hi = mdist.shard_bounds(total, rank, world)
        assert lo == covered and (hi - lo) % block == 0
        covered = hi
        d_counts = torch.zeros(n_genomes * 3, dtype=torch.int64, device=dev)     # this rank's table
        for first in range(lo, hi, block):
            gen.make(d_bases, d_truth, block, read_len, seed=seed, first=first)
            torch.cuda.synchronize()
            eng.classify_device(d_bases.data_ptr(), d_off.data_ptr(), block, block * read_len, read_len, 60,
                                d_assign.data_ptr(), d_best.data_ptr(), d_nhits.data_ptr(), d_counts.data_ptr())
            eng.sync()
            assign_all[first:first + block] = d_assign.cpu().numpy()
            mlen_all[first:first + block] = d_best.view(-1, 4)[:, 3].cpu().numpy()
            truth_all[first:first + block] = d_truth.cpu().numpy()
        rank_counts.append(d_counts.cpu().numpy().reshape(-1, 3))
    assert covered == total
    reduced = np.sum(rank_counts, axis=0)                       # what the RCCL all-reduce leaves on every rank
    # ---- the counts of one pass over the per-read decisions (aligner.py:247-263)
    gid = np.asarray(idx.contig_genome)
    mapped = assign_all >= 0
    g = gid[assign_all[mapped]]
    want = np.zeros((n_genomes, 3), dtype=np.int64)
    want[:, 0] = np.bincount(g, minlength=n_genomes)
    want[:, 1] = want[:, 0] * read_len
    want[:, 2] = np.bincount(g, weights=mlen_all[mapped].astype(np.float64), minlength=n_genomes).astype(np.int64)
    assert np.array_equal(reduced, want)
    assert reduced[:, 0].sum() == mapped.sum()
    # ---- truth: 2 % of the reads are random sequence and must stay unmapped; mapped reads sit on their source
    assert abs((truth_all < 0).mean() - 0.02) < 0.001
    assert (assign_all[truth_all < 0] == capi.UNMAPPED).all()
    assert (assign_all[mapped] == truth_all[mapped]).mean() >= 0.999
    assert mapped.mean() > 0.9
    # ---- the oracle on two samples of the same ordinals
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    for first in (0, 7_300_000):
        n = 1500
        hb, ho, ht = synth.reads(seqs, n, read_len, seed=seed, first=first)
        assert np.array_equal(ht, truth_all[first:first + n])
        oa, ob, onh, _ = oidx.classify(hb, ho, 60, n_threads=N_THREADS)
        assert np.array_equal(assign_all[first:first + n], oa)
        assert np.array_equal(mlen_all[first:first + n][oa >= 0], ob["mlen"][oa >= 0])
    eng.close()


def _genomes_500():
    """The 500-genome recipe of SURVEY.md section 8d: genome i + 250 is a 3 % diverged copy of genome i."""
    names, seqs = synth.genome_set(500, seed=synth.SEED_500, div_seed=synth.SEED_500 + 0x2000)
    return names, seqs


def test_config4_500_genomes_in_eight_parts(capi, oracle):
    """BASELINE config 4 at its read count: 1 000 000 reads (generated in HBM by ordinal, 5 GB resident) against every one
    of the 8 index parts, blocks of 100 000 per call, ONE engine rebound part by part; the per-read summaries of the parts
    merged by best_hit's rule.  The first 4 000 reads (the host generator's same ordinals) against the oracle's multi-part
    loop hit list by hit list; all 1 M against the generator's truth, and the count table that follows from them."""
    import torch
    from monica_amd import aligner, dist as mdist
    dev = torch.device("cuda:0")
    names, seqs = _genomes_500()
    assert sum(len(s) for s in seqs) > 2_000_000_000
    P, n_oracle, n_reads, block, L = 8, 4000, 1_000_000, 100_000, 5000
    seed = synth.SEED_READS + 4
    bounds = [mdist.shard_bounds(len(names), p, P) for p in range(P)]
    assert sorted(hi - lo for lo, hi in bounds) == [62] * 4 + [63] * 4 and bounds[-1][1] == 500
    gen = synth.DeviceReads(seqs, dev)
    d_bases = torch.empty(n_reads * L, dtype=torch.uint8, device=dev)
    d_truth = torch.empty(n_reads, dtype=torch.int32, device=dev)
    for b0 in range(0, n_reads, 250_000):
        gen.make(d_bases[b0 * L:], d_truth[b0:], 250_000, L, seed=seed, first=b0)
    torch.cuda.synchronize()
    truth = d_truth.cpu().numpy()
    ob_, oo_, ot_ = synth.reads(seqs, n_oracle, L, seed=seed)
    assert np.array_equal(ot_, truth[:n_oracle])
    assert np.array_equal(d_bases[:n_oracle * L].cpu().numpy(), ob_)          # the same reads as the host generator's
    d_off = torch.arange(block + 1, dtype=torch.int64, device=dev) * L
    d_assign = torch.empty(n_reads, dtype=torch.int32, device=dev)
    d_best = torch.zeros(n_reads * 4, dtype=torch.int32, device=dev)
    d_nhits = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    summaries, lists = [], [[] for _ in range(n_oracle)]
    eng = None
    for lo, hi in bounds:
        pn, ps = names[lo:hi], seqs[lo:hi]
        idx = capi.Index.from_seqs(pn, ps, device=0)
        if eng is None:
            eng = capi.Engine(idx, 0)
        else:
            eng.set_index(idx)                                    # `index = index_loader(part)`: the same engine, the next part
        torch.cuda.synchronize()
        for b0 in range(0, n_reads, block):
            eng.classify_device(d_bases.data_ptr() + b0 * L, d_off.data_ptr(), block, block * L, L, 60,
                                d_assign.data_ptr() + b0 * 4, d_best.data_ptr() + b0 * 16, d_nhits.data_ptr() + b0 * 4, 0)
            eng.sync()
        summaries.append(mdist.shard_summary(d_assign, d_best, d_nhits, rid_offset=lo))
        assign, nhits = d_assign[:n_oracle].cpu().numpy(), d_nhits[:n_oracle].cpu().numpy()
        best = d_best[:n_oracle * 4].cpu().numpy().view(capi.HIT_DTYPE)
        oidx = oracle.Index.from_seqs(pn, [s.tobytes() for s in ps])
        assert idx.mid_occ == oidx.mid_occ
        oa, ob, onh, flat = oidx.classify(ob_, oo_, 60, n_threads=N_THREADS)
        assert np.array_equal(assign, oa) and np.array_equal(nhits, onh)
        for k in capi.HIT_DTYPE.names:
            assert np.array_equal(best[k][onh > 0], ob[k][onh > 0]), k
        k = 0
        for r in range(n_oracle):
            for h in flat[k:k + onh[r]]:
                lists[r].append((int(h["rid"]) + lo, int(h["nm"]), int(h["mlen"])))
            k += onh[r]
        del idx, oidx
    eng.close()
    got, nm, ml, tot = mdist.merge_summaries(torch.stack(summaries))
    got, tot = got.cpu().numpy(), tot.cpu().numpy()
    # ---- the reference's multi-part loop on the oracle's hit lists (aligner.py:219-233)
    want = []
    for hits in lists:
        if not hits:
            want.append(mdist.UNMAPPED)
        else:
            b = hits[0] if len(hits) == 1 else aligner.best_hit(hits)
            want.append(b[0] if b else mdist.AMBIGUOUS)
    assert got[:n_oracle].tolist() == want
    assert tot[:n_oracle].tolist() == [len(h) for h in lists]
    # ---- all 1 M reads against the generator's truth (contig i of the concatenated parts = genome i)
    mapped = got >= 0
    assert abs((truth < 0).mean() - 0.02) < 0.001
    assert mapped.mean() > 0.9
    assert (got[mapped] == truth[mapped]).mean() > 0.999
    assert (got[truth < 0] == mdist.UNMAPPED).all()
    assert (got == mdist.AMBIGUOUS).sum() < 0.01 * n_reads
    # ---- the count table of the job ('basic' mode, aligner.py:247-250): one count per mapped read, on its source genome
    counts = np.bincount(got[mapped], minlength=len(names))
    assert counts.sum() == mapped.sum() and (counts > 0).sum() >= 499
    want_counts = np.bincount(truth[truth >= 0], minlength=len(names))
    assert np.abs(counts - want_counts).sum() < 0.1 * n_reads                # what differs: the reads left unmapped or ambiguous
