"""CPU tests of the multi-GPU logic with world_size 2 over gloo: the count all-reduce of the
read-sharded mode and the cross-part best_hit merge of the index-sharded mode."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from monica_amd import aligner, dist as mdist, synth
import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100_000):
        for world in (1, 2, 3, 8):
            cuts = [mdist.shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1


def test_merge_truth_table():
    S = lambda rows: torch.tensor(rows, dtype=torch.int32).unsqueeze(1)          # [parts, 1, 5]
    one = lambda rows: [int(x[0]) for x in mdist.merge_summaries(S(rows))]
    assert one([[0, 0, 0, -1, 0], [0, 0, 0, -1, 0]])[0] == mdist.UNMAPPED
    assert one([[1, 1, 10, 3, 0], [0, 0, 0, -1, 0]])[0] == 3
    assert one([[1, 1, 10, 3, 0], [1, 1, 10, 7, 0]])[0] == mdist.AMBIGUOUS        # same ratio in two parts
    assert one([[1, 2, 10, 3, 0], [1, 1, 10, 7, 0]])[0] == 7
    assert one([[2, 1, 10, 3, 1], [1, 2, 10, 7, 0]])[0] == mdist.AMBIGUOUS        # the best part is tied inside
    assert one([[2, 2, 10, 3, 1], [1, 1, 10, 7, 0]])[0] == 7                      # a tie that loses does not matter
    assert one([[1, 2, 20, 3, 0], [1, 1, 10, 7, 0]])[0] == mdist.AMBIGUOUS        # 2/20 == 1/10


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import pyoracle
    r, w = mdist.init("gloo")
    assert (r, w) == (rank, world)
    names, seqs = util.small_genomes(4, 100_000, 130_000)
    bases, offsets, truth = synth.reads(seqs, 120, 2500, seed=31)
    # ---- index-sharded: genome i and its diverged copy i+2 sit in different parts
    part = [(names[:2], seqs[:2], 0), (names[2:], seqs[2:], 2)][rank]
    oidx = pyoracle.Index.from_seqs(part[0], [s.tobytes() for s in part[1]])
    assign, best, nhits, flat = oidx.classify(bases, offsets, 60)
    summ = mdist.shard_summary(assign, best, nhits, rid_offset=part[2])
    m_assign, m_nm, m_ml, m_tot = mdist.gather_and_merge(summ)
    # ---- read-sharded: every rank counts its block; all-reduce
    lo, hi = mdist.shard_bounds(120, rank, world)
    counts = torch.zeros(4, dtype=torch.int64)
    for g in m_assign[lo:hi].tolist():
        if g >= 0:
            counts[g] += 1
    mdist.allreduce_counts(counts)
    if rank == 0:
        q.put((m_assign.numpy(), m_tot.numpy(), counts.numpy()))
    # every rank must hold the same merged result
    chk = m_assign.clone()
    dist.broadcast(chk, 0)
    assert torch.equal(chk, m_assign)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got_assign, got_total, got_counts = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    # expectation: the reference's multi-part loop on one process (hit lists concatenated in
    # part order, then aligner.best_hit)
    names, seqs = util.small_genomes(4, 100_000, 130_000)
    bases, offsets, truth = synth.reads(seqs, 120, 2500, seed=31)
    lists = [[] for _ in range(120)]
    for pn, ps, base in [(names[:2], seqs[:2], 0), (names[2:], seqs[2:], 2)]:
        oidx = oracle.Index.from_seqs(pn, [s.tobytes() for s in ps])
        a, b, nh, flat = oidx.classify(bases, offsets, 60)
        k = 0
        for r in range(120):
            for h in flat[k:k + nh[r]]:
                lists[r].append((int(h["rid"]) + base, int(h["nm"]), int(h["mlen"])))
            k += nh[r]
    want = []
    for hits in lists:
        if not hits:
            want.append(mdist.UNMAPPED)
        else:
            b = hits[0] if len(hits) == 1 else aligner.best_hit(hits)
            want.append(b[0] if b else mdist.AMBIGUOUS)
    assert got_assign.tolist() == want
    assert got_total.tolist() == [len(h) for h in lists]
    assert got_counts.sum() == sum(1 for w in want if w >= 0)
    assert got_counts.tolist() == [sum(1 for w in want if w == g) for g in range(4)]
    assert sum(1 for w in want if w >= 0) > 90


def _pipe_worker(rank, world, conns, q):
    sys.path.insert(0, ROOT)
    g = mdist.PipeGroup(rank, world, conns)
    rng = np.random.default_rng(5)                       # every rank draws the same table, takes its rows
    n = 500
    summ = torch.zeros((world, n, 5), dtype=torch.int32)
    summ[:, :, 0] = torch.from_numpy(rng.integers(0, 3, (world, n)).astype(np.int32))
    summ[:, :, 1] = torch.from_numpy(rng.integers(0, 6, (world, n)).astype(np.int32))
    summ[:, :, 2] = torch.from_numpy(rng.integers(1, 12, (world, n)).astype(np.int32))
    summ[:, :, 3] = torch.where(summ[:, :, 0] > 0, torch.from_numpy(rng.integers(0, 50, (world, n)).astype(np.int32)), torch.tensor(-1, dtype=torch.int32))
    summ[:, :, 4] = torch.from_numpy((rng.random((world, n)) < 0.1).astype(np.int32)) * (summ[:, :, 0] > 1).to(torch.int32)
    merged = mdist.gather_and_merge(summ[rank], group=g)
    counts = torch.arange(12, dtype=torch.int64) * (rank + 1)
    mdist.allreduce_counts(counts, group=g)
    q.put((rank, [m.numpy() for m in merged], counts.numpy(), summ.numpy()))


def test_world_size_3_without_torch_distributed():
    """The same two reductions over plain pipes (no gloo, no RCCL): what a host that drives the
    C-ABI's mnc_allreduce_counts / mnc_allgather_summaries itself has to reproduce."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    conns = mdist.PipeGroup.make(world)
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, conns[r], q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = [m.numpy() for m in mdist.merge_summaries(torch.from_numpy(got[0][3]))]
    for rank, merged, counts, _ in got:
        for a, b in zip(merged, want):
            assert np.array_equal(a, b), rank
        assert counts.tolist() == [k * (1 + 2 + 3) for k in range(12)]
