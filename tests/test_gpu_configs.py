"""BASELINE.json configs under `pytest -m gpu`: the HIP path against the CPU oracle, read for
read, on the workloads the metric is quoted on (SURVEY.md section 8d / BASELINE.md):

  config 1  1 000 x 5 kb reads vs one 4 641 652-bp contig (seed 0xEC011)
  config 2  100 000 x 5 kb reads vs the 20-genome index (seeds 0x20 / 0x2020)
  config 4  the per-GPU shape of the 500-genome index split over 8 GPUs: parts of 62 genomes,
            every part maps all reads, per-part summaries merged (dist.merge_summaries)
            against the reference's multi-part loop (aligner.py:91-103, 219-233)
  config 5  consecutive 400-read micro-batches through mnc_classify_batch: per-read results and
            the summed counts equal one big batch
plus the committed golden fixture (tests/golden/small_case.npz) on the HIP path, including
the chaining peak array v[].  The call sequence is test/test_aligner.py:11-14, 43-44.
"""
import os

import numpy as np
import pytest

from monica_amd import synth
import util

pytestmark = pytest.mark.gpu

N_THREADS = min(16, os.cpu_count() or 1)


def _counts_from(idx_names_to_gid, contig_gid, assign, best, offsets):
    """The three counting modes of aligner.py:247-263 from per-read decisions."""
    n_genomes = max(contig_gid) + 1
    out = np.zeros((n_genomes, 3), dtype=np.int64)
    lens = np.diff(offsets)
    for r in np.nonzero(assign >= 0)[0]:
        g = contig_gid[assign[r]]
        out[g, 0] += 1
        out[g, 1] += lens[r]
        out[g, 2] += best["mlen"][r]
    return out


def _check_whole_batch(capi, oracle, names, seqs, bases, offsets, truth, min_correct, dp_sample=None):
    """Both contracts: the chain level on every read; with base-level alignment (the default, what
    mappy computes) on every read too unless `dp_sample` bounds the oracle's share (its literal
    ksw2 simulation runs at a few hundred reads/s)."""
    idx = capi.Index.from_seqs(names, seqs)
    eng = capi.Engine(idx, 0)
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    assert idx.mid_occ == oidx.mid_occ
    gid = [int(x) for x in idx.contig_genome]
    out = None
    for contract, cigar in ((capi.CONTRACT_CHAIN, 0), (capi.CONTRACT_DP, 1)):
        eng.set_contract(contract)
        oidx.opt.cigar = cigar
        assign, best, nhits = eng.classify(bases, offsets, 60)
        hit_off, hits = eng.fetch_hits()
        n = len(truth) if (cigar == 0 or dp_sample is None) else min(dp_sample, len(truth))
        oassign, obest, onh, oflat = oidx.classify(bases[:offsets[n]], offsets[:n + 1], 60, n_threads=N_THREADS)
        assert np.array_equal(assign[:n], oassign)
        assert np.array_equal(nhits[:n], onh)
        assert np.array_equal(np.diff(hit_off)[:n], onh)
        for k in capi.HIT_DTYPE.names:
            assert np.array_equal(best[k][:n], obest[k]), k
            assert np.array_equal(hits[k][:hit_off[n]], oflat[k]), k
        want = _counts_from(None, gid, oassign, obest, offsets[:n + 1])
        for mode in (1, 2, 3):
            got = capi.counts(idx, assign[:n], best[:n], offsets[:n + 1], mode)
            assert np.array_equal(got, want[:, mode - 1]), f"counts, mode {mode}"
        mapped = assign >= 0
        assert (assign[mapped] == truth[mapped]).mean() >= min_correct
        assert (assign[truth < 0] == capi.UNMAPPED).all()
        out = (assign, best, nhits)
    eng.close()
    return out


def test_config1_1k_reads_vs_single_contig(capi, oracle):
    names, seqs = synth.ecoli_like()
    assert len(seqs[0]) == 4_641_652
    bases, offsets, truth = synth.reads(seqs, 1000, 5000, seed=synth.SEED_READS + 1)
    assign, best, nhits = _check_whole_batch(capi, oracle, names, seqs, bases, offsets, truth, 1.0)
    assert (assign >= 0).sum() >= 950


def test_config2_100k_reads_vs_20_genomes(capi, oracle):
    import torch
    names, seqs = synth.genome_set(20)
    assert sum(len(s) for s in seqs) == 94_031_982
    n = 100_000
    bases, offsets, truth = synth.reads(seqs, n, 5000, seed=synth.SEED_READS + 2)
    # every one of the 100 000 reads against the oracle under both contracts (its scalar ksw2 simulation does ~1 900 reads/s
    # on the box's 16 cores: about a minute)
    assign, best, nhits = _check_whole_batch(capi, oracle, names, seqs, bases, offsets, truth, 0.999)
    # the device-resident entry point with on-device taxon counts (what bench.py times)
    idx = capi.Index.from_seqs(names, seqs)
    eng = capi.Engine(idx, 0)
    dev = torch.device("cuda:0")
    d_bases, d_off = torch.from_numpy(bases).to(dev), torch.from_numpy(offsets).to(dev)
    d_assign = torch.empty(n, dtype=torch.int32, device=dev)
    d_best = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(len(idx.genome_names) * 3, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.classify_device(d_bases.data_ptr(), d_off.data_ptr(), n, int(offsets[-1]), 5000, 60,
                        d_assign.data_ptr(), d_best.data_ptr(), 0, d_counts.data_ptr())
    eng.sync()
    assert np.array_equal(d_assign.cpu().numpy(), assign)
    got = d_counts.cpu().numpy().reshape(-1, 3)
    for mode in (1, 2, 3):
        assert np.array_equal(got[:, mode - 1], capi.counts(idx, assign, best, offsets, mode))
    eng.close()


def test_config4_shard_shape_62_genome_parts(capi, oracle):
    """Two parts of 62 genomes each (one GPU's share of the 500-genome index on 8 GPUs), taken
    from the 500-genome recipe (seed 0x500: genome i and i + 250 are a diverged pair, so the
    two parts hold each other's near-copies); 4 000 reads."""
    import torch
    from monica_amd import aligner, dist as mdist
    n_reads = 4000
    names, seqs = [], []
    base = 250
    for i in list(range(0, 62)) + list(range(base, base + 62)):
        if i < base:
            length = 2_000_000 + synth._mix(synth.SEED_500 * 1_000_003 + i) % 5_000_001
            seqs.append(synth.genome(synth._mix(synth.SEED_500) + i, int(length)))
        else:
            seqs.append(synth.diverge(seqs[i - base], synth._mix(synth.SEED_500 + 0x2000) + i, 30_000))
        names.append(synth.contig_name(i))
    bases, offsets, truth = synth.reads(seqs, n_reads, 5000, seed=synth.SEED_READS + 4)
    parts = [(names[:62], seqs[:62], 0), (names[62:], seqs[62:], 62)]
    summaries, lists = [], [[] for _ in range(n_reads)]
    for pn, ps, rid0 in parts:
        idx = capi.Index.from_seqs(pn, ps)
        assert idx.info().total_len > 200_000_000                  # the per-GPU shard size
        eng = capi.Engine(idx, 0)
        oidx = oracle.Index.from_seqs(pn, [s.tobytes() for s in ps])
        assert idx.mid_occ == oidx.mid_occ
        assign, best, nhits = eng.classify(bases, offsets, 60)
        oa, ob, onh, flat = oidx.classify(bases, offsets, 60, n_threads=N_THREADS)
        assert np.array_equal(assign, oa) and np.array_equal(nhits, onh)
        for k in capi.HIT_DTYPE.names:
            assert np.array_equal(best[k], ob[k]), k
        summaries.append(mdist.shard_summary(assign, best, nhits, rid_offset=rid0))
        k = 0
        for r in range(n_reads):
            for h in flat[k:k + onh[r]]:
                lists[r].append((int(h["rid"]) + rid0, int(h["nm"]), int(h["mlen"])))
            k += onh[r]
        eng.close()
        del idx, oidx
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
    want = np.array(want)
    mapped = want >= 0
    assert mapped.sum() > 0.9 * n_reads
    assert (want[mapped] == truth[mapped]).mean() > 0.999


def test_config5_micro_batches_equal_one_batch(capi, oracle):
    """120 consecutive 400-read micro-batches (two minutes of the simulated MinION run) through
    the host-buffer entry point: per-read results, and the counts summed over the micro-batches,
    equal one 48 000-read batch and the oracle."""
    names, seqs = synth.genome_set(20)
    idx = capi.Index.from_seqs(names, seqs)
    eng = capi.Engine(idx, 0)
    n_batches, per = 120, 400
    n = n_batches * per
    bases, offsets, truth = synth.reads(seqs, n, 5000, seed=synth.SEED_READS + 5)
    whole = eng.classify(bases, offsets, 60)
    n_genomes = len(idx.genome_names)
    counts = np.zeros((n_genomes, 3), dtype=np.int64)
    for b in range(n_batches):
        lo, hi = b * per, (b + 1) * per
        bo = offsets[lo:hi + 1] - offsets[lo]
        bb = bases[offsets[lo]:offsets[hi]]
        assign, best, nhits = eng.classify(bb, bo, 60)
        assert np.array_equal(assign, whole[0][lo:hi]), f"micro-batch {b}"
        assert np.array_equal(nhits, whole[2][lo:hi])
        for k in capi.HIT_DTYPE.names:
            assert np.array_equal(best[k], whole[1][k][lo:hi]), k
        for mode in (1, 2, 3):
            counts[:, mode - 1] += capi.counts(idx, assign, best, bo, mode)
    for mode in (1, 2, 3):
        assert np.array_equal(counts[:, mode - 1], capi.counts(idx, whole[0], whole[1], offsets, mode))
    oidx = oracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    oa, ob, onh, _ = oidx.classify(bases[:offsets[4000]], offsets[:4001], 60, n_threads=N_THREADS)
    assert np.array_equal(whole[0][:4000], oa) and np.array_equal(whole[2][:4000], onh)
    eng.close()


def test_golden_fixture_on_the_hip_path(capi):
    """tests/golden/small_case.npz (made by tests/golden/make_golden.py with the oracle) against
    the HIP path at the chain level: index, minimizer / anchor counts, chaining f / p / v,
    every region field, gated hits, decisions; and the complete path with base-level alignment
    (regions, CIGARs, gated hits, decisions)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "small_case.npz"))
    names = [str(x) for x in g["names"]]
    seqs, o = [], 0
    for L in g["genome_lens"]:
        seqs.append(g["genome_bytes"][o:o + L])
        o += L
    idx = capi.Index.from_seqs(names, seqs)
    info = idx.info()
    assert idx.mid_occ == int(g["mid_occ"]) and info.n_keys == int(g["n_keys"]) and info.n_occ == int(g["n_occ"])
    eng = capi.Engine(idx, 0)
    bases, offsets = g["bases"], g["offsets"]
    # ---- with base-level alignment (the default)
    assign, best, nhits = eng.classify(bases, offsets, 60)
    hit_off, hits = eng.fetch_hits()
    assert np.array_equal(assign, g["dp_assign"]) and np.array_equal(nhits, g["dp_nhits"])
    for k in capi.HIT_DTYPE.names:
        assert np.array_equal(best[k], g["dp_best"][k]), k
        assert np.array_equal(hits[k], g["dp_hits"][k]), k
    regs = eng.dump(capi.DUMP_REGS, capi.REG_DTYPE)
    assert np.array_equal(np.diff(eng.dump(capi.DUMP_REG_OFFSETS, np.int64)), g["dp_reg_cnt"])
    for k in capi.REG_DTYPE.names:
        assert np.array_equal(regs[k], g["dp_regs"][k]), k
    assert np.array_equal(eng.dump(capi.DUMP_CIGARS, np.uint32), g["dp_cigars"])
    # ---- chain level
    eng.set_contract(capi.CONTRACT_CHAIN)
    assign, best, nhits = eng.classify(bases, offsets, 60)
    hit_off, hits = eng.fetch_hits()
    assert np.array_equal(assign, g["assign"]) and np.array_equal(nhits, g["nhits"])
    for k in capi.HIT_DTYPE.names:
        assert np.array_equal(best[k], g["best"][k]), k
        assert np.array_equal(hits[k], g["hits"][k]), k
    mz = eng.dump(capi.DUMP_MINIMIZERS, capi.MZ_DTYPE)
    mz_off = eng.dump(capi.DUMP_MZ_OFFSETS, np.int64)
    assert np.array_equal(np.diff(mz_off), g["mz_cnt"])
    first = np.array([int(mz["hash"][mz_off[r]]) << 8 | 15 if mz_off[r + 1] > mz_off[r] else 0
                      for r in range(len(offsets) - 1)], dtype=np.uint64)
    assert np.array_equal(first, g["mz_first"])
    assert np.array_equal(np.diff(eng.dump(capi.DUMP_AN_OFFSETS, np.int64)), g["an_cnt"])
    assert np.array_equal(eng.dump(capi.DUMP_CHAIN_F, np.int32), g["chain_f"])
    assert np.array_equal(eng.dump(capi.DUMP_CHAIN_P, np.int32), g["chain_p"])
    assert np.array_equal(eng.dump(capi.DUMP_CHAIN_V, np.int32), g["chain_v"])
    regs = eng.dump(capi.DUMP_REGS, capi.REG_DTYPE)
    assert np.array_equal(np.diff(eng.dump(capi.DUMP_REG_OFFSETS, np.int64)), g["reg_cnt"])
    for k in g["regs"].dtype.names:
        assert np.array_equal(regs[k], g["regs"][k]), k
    eng.close()


def test_prefetched_batches_equal_plain_calls(capi):
    """mnc_engine_prefetch: the next batch's bases copied behind the running batch's kernels.  Whatever the order of
    announcements -- the batch that is classified next, another one, none, an announcement refused because one is
    pending -- the results are those of plain mnc_classify_batch calls."""
    import threading
    names, seqs = synth.genome_set(4, min_len=150_000, max_len=250_000)
    idx = capi.Index.from_seqs(names, seqs)
    eng = capi.Engine(idx, 0)
    batches = []
    for k, (n, L) in enumerate(((3000, 2000), (500, 5000), (2500, 1500), (1, 3000))):
        b, o, _ = synth.reads(seqs, n, L, seed=900 + k)
        batches.append((capi.pinned_array(b), np.ascontiguousarray(o)))
    plain = [eng.classify(b, o, 60) for b, o in batches]

    def same(got, want):
        return all(np.array_equal(g, w) for g, w in zip(got[:1] + got[2:], want[:1] + want[2:])) and \
            all(np.array_equal(got[1][k], want[1][k]) for k in capi.HIT_DTYPE.names)

    # announced, then classified: in a row, and with the announcement of the next one while a call runs
    assert eng.prefetch_ptr(batches[0][0].ctypes.data, batches[0][1].ctypes.data, len(batches[0][1]) - 1)
    assert not eng.prefetch_ptr(batches[1][0].ctypes.data, batches[1][1].ctypes.data, len(batches[1][1]) - 1)   # one spare buffer
    for k, (b, o) in enumerate(batches):
        nxt = batches[k + 1] if k + 1 < len(batches) else None
        done = []

        def announce():
            while nxt is not None and not eng.prefetch_ptr(nxt[0].ctypes.data, nxt[1].ctypes.data, len(nxt[1]) - 1):
                pass
            done.append(1)
        t = threading.Thread(target=announce)
        t.start()
        got = eng.classify_ptr(b.ctypes.data, o.ctypes.data, len(o) - 1, 60)
        t.join()
        assert same(got, plain[k]), f"batch {k}"
    # an announced batch that is NOT the next one classified: dropped, the call copies its own
    assert eng.prefetch_ptr(batches[2][0].ctypes.data, batches[2][1].ctypes.data, len(batches[2][1]) - 1)
    assert same(eng.classify_ptr(batches[1][0].ctypes.data, batches[1][1].ctypes.data, len(batches[1][1]) - 1, 60), plain[1])
    assert same(eng.classify(batches[2][0], batches[2][1], 60), plain[2])
    eng.close()
