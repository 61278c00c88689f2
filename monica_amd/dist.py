"""Multi-GPU helpers: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).

Two ways the path shards:

* **Reads sharded, index replicated** (BASELINE configs 2, 3, 5).  No data-path collective;
  one `all_reduce(sum)` of the per-genome count table per batch -- the analogue of
  `alignment_update` (monica/genomes/aligner.py:282-302).
* **Index sharded** (config 4): every rank maps all reads against its part of the genomes,
  exactly like one pass of the reference's multi-part loop (aligner.py:91-103): MAPQ and the
  gate are per part, and `best_hit` (aligner.py:328-339) runs over the union of the parts'
  gated hits.  Because `best_hit` only asks whether the smallest NM/mlen is unique, a part
  is summarised per read by five integers {hits, nm, mlen, contig, tied}; the summaries are
  all-gathered (20 B per read and rank) and reduced identically on every rank.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

UNMAPPED = -1
AMBIGUOUS = -2


def init(backend=None, device=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (set by
    torch.distributed.run); returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_bounds(n, rank, world):
    """Contiguous block of read ordinals [lo, hi) of `rank` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class PipeGroup:
    """A process group over `multiprocessing` pipes (a star around rank 0) with the two collectives
    this path needs.  For hosts without torch.distributed / RCCL -- the same reductions a C caller
    gets from `mnc_allreduce_counts` / `mnc_allgather_summaries` -- and for tests of the merge
    rules with real processes."""

    def __init__(self, rank, world, conns):
        """conns: on rank 0 a list of world - 1 connections (to ranks 1..), else [the connection to rank 0]."""
        self.rank, self.world, self.conns = rank, world, conns

    @staticmethod
    def make(world):
        """Connections for `world` processes: element r goes to rank r's constructor."""
        import multiprocessing as mp
        pairs = [mp.Pipe() for _ in range(world - 1)]
        return [[a for a, _ in pairs]] + [[b] for _, b in pairs]

    def all_gather(self, t):
        if self.rank == 0:
            parts = [t] + [torch.from_numpy(c.recv()) for c in self.conns]
            for c in self.conns:
                c.send([p.cpu().numpy() for p in parts])
            return parts
        self.conns[0].send(t.cpu().numpy())
        return [torch.from_numpy(a) for a in self.conns[0].recv()]

    def all_reduce_sum(self, t):
        total = sum(p.cpu() for p in self.all_gather(t))          # `t` may live on a GPU; the received parts do not
        t.copy_(total.to(t.device))
        return t


def allreduce_counts(counts, group=None):
    """Sum the per-genome count table over ranks, in place (torch tensor)."""
    if group is not None:
        return group.all_reduce_sum(counts)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def shard_summary(assign, best, nhits, rid_offset=0):
    """Per-read summary of one index part: int32[n, 5] = {hits, nm, mlen, global contig, tied}.
    `best` is the engine's minimal gated hit (also for AMBIGUOUS reads).  Device tensors (the outputs of
    `mnc_classify_device`) go through the library's kernel `mnc_shard_summary`; host arrays -- what
    `mnc_classify_batch` returned -- are summarised on the host."""
    t = torch.as_tensor
    assign, nhits = t(assign), t(nhits)
    best = t(np.ascontiguousarray(best).view(np.int32).reshape(-1, 4)) if isinstance(best, np.ndarray) else best.reshape(-1, 4)
    n = assign.shape[0]
    if assign.is_cuda:
        from monica_amd import _capi
        assign, best, nhits = assign.contiguous(), best.contiguous(), nhits.contiguous()
        out = torch.empty((n, 5), dtype=torch.int32, device=assign.device)
        _capi.shard_summary_device(assign.data_ptr(), best.data_ptr(), nhits.data_ptr(), n, rid_offset, out.data_ptr(), _stream(assign))
        return out
    out = torch.zeros((n, 5), dtype=torch.int32)
    has = nhits > 0
    out[:, 0] = nhits
    out[:, 1] = best[:, 2]                       # nm
    out[:, 2] = best[:, 3]                       # mlen
    out[:, 3] = torch.where(has, best[:, 0] + rid_offset, torch.full_like(best[:, 0], -1))
    out[:, 4] = (assign == AMBIGUOUS).to(torch.int32)
    return out


def merge_summaries(stacked):
    """best_hit over the union of the parts.  stacked: int32[parts, n, 5] in part order.
    Returns (assign int32[n] with global contig ids, nm, mlen, total hits).  On a device this is the library's
    kernel `mnc_merge_summaries` (C2 behind the C-ABI); host tensors (gloo, `PipeGroup`) take the same rule below."""
    if stacked.is_cuda:
        from monica_amd import _capi
        stacked = stacked.contiguous()
        if stacked.dtype != torch.int32:
            raise TypeError("summaries are int32 records")
        P, n = int(stacked.shape[0]), int(stacked.shape[1])
        out = torch.empty((4, n), dtype=torch.int32, device=stacked.device)
        _capi.merge_summaries_device(stacked.data_ptr(), P, n, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(),
                                     _stream(stacked))
        return out[0], out[1], out[2], out[3]
    s = stacked.to(torch.int64)
    n = s.shape[1]
    has = torch.zeros(n, dtype=torch.bool)
    nm = torch.zeros(n, dtype=torch.int64)
    ml = torch.ones(n, dtype=torch.int64)
    rid = torch.full((n,), -1, dtype=torch.int64)
    tied = torch.zeros(n, dtype=torch.bool)
    total = torch.zeros(n, dtype=torch.int64)
    for p in range(s.shape[0]):
        c_has = s[p, :, 0] > 0
        c_nm, c_ml, c_rid, c_tied = s[p, :, 1], torch.clamp(s[p, :, 2], min=1), s[p, :, 3], s[p, :, 4] > 0
        lhs, rhs = c_nm * ml, nm * c_ml              # c_nm/c_ml ? nm/ml, exact in int64
        better = c_has & (~has | (lhs < rhs))
        equal = c_has & has & (lhs == rhs)
        tied = torch.where(better, c_tied, tied | equal)
        nm = torch.where(better | equal, c_nm, nm)
        ml = torch.where(better | equal, c_ml, ml)
        rid = torch.where(better | equal, c_rid, rid)
        has = has | c_has
        total = total + s[p, :, 0]
    assign = torch.where(~has, torch.full_like(rid, UNMAPPED), torch.where(tied, torch.full_like(rid, AMBIGUOUS), rid))
    return assign.to(torch.int32), nm.to(torch.int32), ml.to(torch.int32), total.to(torch.int32)


def gather_and_merge(summary, group=None):
    """All-gather the per-part summaries (rank order = part order) and merge them."""
    if group is not None:
        stacked = torch.stack([p.to(summary.device) for p in group.all_gather(summary.contiguous())])
    elif dist.is_initialized() and dist.get_world_size() > 1:
        parts = [torch.empty_like(summary) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, summary.contiguous())
        stacked = torch.stack(parts)
    else:
        stacked = summary.unsqueeze(0)
    return merge_summaries(stacked)
