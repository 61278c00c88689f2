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
