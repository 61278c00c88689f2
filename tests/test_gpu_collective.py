"""The C-ABI collectives on a real RCCL communicator (`pytest -m gpu`).

`mnc_comm_unique_id` / `mnc_comm_init_rank` / `mnc_allreduce_counts` / `mnc_allgather_summaries`
(csrc/collective.cpp) are what a host program without torch.distributed uses for monica's two
cross-process merges (aligner.py:286-298 count tables; aligner.py:196-203, 218-223 hits carried between
index parts).  One GPU is what the test box has, so the communicator has ONE rank: the symbols resolved
from librccl.so are called with their real signatures on device buffers, on the engine's own stream,
and a one-rank sum / gather must leave the values as they were.  (World size 2 is covered without RCCL
by tests/test_dist.py; RCCL refuses two ranks on one device.)"""
import numpy as np
import pytest

from monica_amd import synth

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_through_the_c_abi(capi):
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    uid = capi.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = capi.Comm(uid, 1, 0)
    # ---- count table: int64 sum over ranks, in place
    counts = (torch.arange(60, dtype=torch.int64, device=dev) * 7919 + 3) << 20
    want = counts.clone()
    torch.cuda.synchronize()
    comm.allreduce_counts(counts.data_ptr(), counts.numel())
    torch.cuda.synchronize()
    assert torch.equal(counts, want)
    # ---- per-read summaries: 20 bytes per read, gathered in rank order
    send = torch.randint(-5, 1 << 30, (4001, 5), dtype=torch.int32, device=dev)
    recv = torch.zeros_like(send)
    comm.allgather_summaries(send.data_ptr(), recv.data_ptr(), send.numel() * 4)
    torch.cuda.synchronize()
    assert torch.equal(recv, send)
    comm.close()


def test_count_allreduce_on_the_engine_stream_after_a_batch(capi):
    """The order bench.py --collective capi relies on: the all-reduce is queued on the engine's stream behind the
    batch's count kernel and sees its result."""
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    names, seqs = synth.genome_set(4, min_len=150_000, max_len=250_000)
    idx = capi.Index.from_seqs(names, seqs)
    eng = capi.Engine(idx, 0)
    comm = capi.Comm(capi.Comm.unique_id(), 1, 0)
    bases, offsets, truth = synth.reads(seqs, 2000, 3000, seed=99)
    assign, best, nhits = eng.classify(bases, offsets, 60)
    d_bases, d_off = torch.from_numpy(bases).to(dev), torch.from_numpy(offsets).to(dev)
    d_assign = torch.empty(2000, dtype=torch.int32, device=dev)
    d_best = torch.zeros(2000 * 4, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(len(idx.genome_names) * 3, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    eng.classify_device(d_bases.data_ptr(), d_off.data_ptr(), 2000, int(offsets[-1]), 3000, 60,
                        d_assign.data_ptr(), d_best.data_ptr(), 0, d_counts.data_ptr())
    comm.allreduce_counts(d_counts.data_ptr(), d_counts.numel(), eng.stream)
    eng.sync()
    got = d_counts.cpu().numpy().reshape(-1, 3)
    for mode in (1, 2, 3):
        assert np.array_equal(got[:, mode - 1], capi.counts(idx, assign, best, offsets, mode))
    assert got[:, 0].sum() == (assign >= 0).sum() > 1800
    comm.close()
    eng.close()


def _lists_to_parts(lists, P, rng):
    """Random hit lists of one read each, cut into P consecutive parts; per part the summary a part's engine would
    report: {hits, nm, mlen, contig of the part's best_hit minimum (the last of equal ones), tied inside the part}."""
    from monica_amd import aligner
    n = len(lists)
    parts = np.zeros((P, n, 5), dtype=np.int32)
    parts[:, :, 3] = -1
    for r, hits in enumerate(lists):
        cuts = sorted(rng.integers(0, len(hits) + 1, P - 1).tolist())
        for p, (a, b) in enumerate(zip([0] + cuts, cuts + [len(hits)])):
            sub = hits[a:b]
            if not sub:
                continue
            best, tied = sub[0], False
            for h in sub[1:]:                                      # the running minimum with `<=` (aligner.py:331-337)
                l, rr = h[1] * best[2], best[1] * h[2]
                if l < rr:
                    best, tied = h, False
                elif l == rr:
                    best, tied = h, True
            assert (aligner.best_hit(sub) == 0) == tied if len(sub) > 1 else not tied
            parts[p, r] = (len(sub), best[1], best[2], best[0], int(tied))
    return parts


def test_merge_kernels_equal_best_hit_over_the_union(capi):
    """C2 behind the C-ABI (`mnc_shard_summary`, `mnc_merge_summaries`): the device merge of per-part summaries against
    `best_hit` (aligner.py:328-339) over the concatenated lists, the host form of `dist.merge_summaries`, and the truth
    table of tests/test_dist.py -- lists with many equal ratios (1/10 = 2/20) and ties inside and across parts."""
    import torch
    from monica_amd import aligner, dist as mdist
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    lists = []
    for _ in range(20_000):
        k = int(rng.integers(0, 7))
        lists.append([(int(rng.integers(0, 50)), int(rng.integers(0, 4)) * int(m // 10), int(m))
                      for m in rng.choice([10, 20, 30, 40], k)])
    for P in (1, 2, 3, 8):
        parts = _lists_to_parts(lists, P, rng)
        want = []
        for hits in lists:
            if not hits:
                want.append(mdist.UNMAPPED)
            else:
                b = hits[0] if len(hits) == 1 else aligner.best_hit(hits)
                want.append(b[0] if b else mdist.AMBIGUOUS)
        host = mdist.merge_summaries(torch.from_numpy(parts))
        got = mdist.merge_summaries(torch.from_numpy(parts).to(dev))
        torch.cuda.synchronize()
        assert got[0].is_cuda and got[0].cpu().tolist() == want
        for g, h in zip(got, host):
            assert torch.equal(g.cpu(), h)
        assert got[3].cpu().tolist() == [len(h) for h in lists]
    # ---- mnc_shard_summary against the host form on engine-shaped outputs
    n = 10_001
    nhits = rng.integers(0, 3, n).astype(np.int32)
    assign = np.where(nhits == 0, -1, np.where(rng.random(n) < 0.2, -2, rng.integers(0, 40, n))).astype(np.int32)
    best = np.zeros(n, dtype=capi.HIT_DTYPE)
    for k in capi.HIT_DTYPE.names:
        best[k] = rng.integers(0, 5000, n)
    best[nhits == 0] = 0
    host = mdist.shard_summary(assign, best, nhits, rid_offset=123)
    d = mdist.shard_summary(torch.from_numpy(assign).to(dev), torch.from_numpy(best.view(np.int32)).to(dev), torch.from_numpy(nhits).to(dev), rid_offset=123)
    torch.cuda.synchronize()
    assert d.is_cuda and torch.equal(d.cpu(), host)
    # ---- no parts / no reads
    with pytest.raises(capi.MncError):
        capi.merge_summaries_device(d.data_ptr(), 0, n, d.data_ptr())
    capi.merge_summaries_device(0, 1, 0, 0)
