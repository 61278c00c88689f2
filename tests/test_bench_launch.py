"""`python bench.py --gpus N` alone must run N ranks.

Without WORLD_SIZE in the environment and with --gpus N > 1, bench.py starts `python -m torch.distributed.run
--nproc-per-node N bench.py ...` as a CHILD process (the parent never touches the GPU, nothing execs), passes rank
0's JSON line through and exits with the child's code.  The count merge the ranks then run is the multi-process form
of alignment_update (monica/genomes/aligner.py:286-298)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, cwd=ROOT,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_gpus_2_starts_two_child_ranks_and_forwards_their_exit_code():
    """No GPU here: a rank refuses with bench.py's own message (the launcher may end the other one before it gets
    that far); its failure report names both ranks = two ranks were started; the parent's exit code is the
    launcher's (non-zero), and nothing JSON-like reaches stdout.  (--one-device: the rehearsal switch skips the
    device count, so the ranks do start.)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the -m gpu test below covers the real run")
    r = run_bench("--gpus", "2", "--one-device", "--steps", "1", "--warmup", "0", "--reads", "100", timeout=300)
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]
    assert "local_rank: 0" in r.stderr and "local_rank: 1" in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ""


def test_more_ranks_than_devices_is_one_line_and_no_rank():
    """`--gpus N` with N above the visible devices exits non-zero with ONE line before any rank starts (an 8-GPU line on
    a 1-GPU box must not half-run): here there are no devices at all."""
    import torch
    n = torch.cuda.device_count()
    r = run_bench("--gpus", str(n + 2), "--steps", "1", "--warmup", "0", "--reads", "100", timeout=120)
    assert r.returncode == 2
    lines = [l for l in r.stderr.splitlines() if l.strip()]
    assert len(lines) == 1 and f"--gpus {n + 2} but {n} device(s) visible" in lines[0], r.stderr[-2000:]
    assert "local_rank" not in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_gpus_2_alone_reports_two_ranks():
    """Two ranks on the one GPU of the test box (gloo stands in for RCCL, which refuses two ranks on one device):
    the line must say n_gpus 2, give every rank's reads, and a value = both ranks' reads over the slower one's time."""
    r = run_bench("--gpus", "2", "--one-device", "--backend", "gloo", "--steps", "3", "--warmup", "1",
                  "--reads", "4000", "--cpu-sample", "0")
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert out["ranks"]["group_ranks"] == 2 and out["ranks"]["backend"] == "gloo"
    assert out["ranks"]["reads_per_rank"] == [12000, 12000]
    assert out["rccl_ranks"] is None and out["group_ranks"] == 2     # gloo rehearsal: no RCCL communicator was asked, and the line says so
    assert abs(out["value"] - 24000 / out["timed_region_s"]) / out["value"] < 2e-2      # (the region is rounded to 0.1 ms of ~20)
    # the all-reduced count table holds BOTH ranks' mapped reads (rank 0's own are `mapped_reads_last_step`)
    assert 1.8 * out["mapped_reads_last_step"] < out["counts_checksum"] < 2.2 * out["mapped_reads_last_step"]


@pytest.mark.gpu
def test_one_rank_line_carries_the_timed_region_and_both_cpu_baselines():
    r = run_bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "4000", "--cpu-sample", "300")
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] is None and out["ranks"]["reads_per_rank"] == [8000]
    assert abs(out["timed_region_s"] * 1e3 / 2 - out["ms_per_step"]) < 0.06     # the region is rounded to 0.1 ms
    cpu = out["cpu_baseline"]
    assert cpu["agrees_with_gpu"] and cpu["one_core"]["cores"] == 1 and cpu["one_core"]["agrees_with_gpu"]
    assert out["roofline"]["frac"] > 0 and out["roofline"]["bound"] in ("valu", "hbm")


@pytest.mark.gpu
def test_four_ranks_rehearse_the_sharded_modes_on_one_device():
    """The N > 2 bookkeeping of the two multi-GPU modes, run once with four ranks on the one GPU (gloo): `--mode shard
    --parts 8` (two consecutive parts a rank: rank order = part order, the all-gather of the 20-byte records, the merge
    kernel on every rank) and `--mode config3` (dist.shard_bounds over four ranks, one all-reduce of the count table)."""
    r = run_bench("--gpus", "4", "--one-device", "--backend", "gloo", "--mode", "shard", "--parts", "8", "--genomes", "16",
                  "--reads", "3000", "--block", "3000", "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["group_ranks"] == 4 and out["rccl_ranks"] is None
    assert out["ranks"]["reads_per_rank"] == [6000] * 4                     # every rank: all 3 000 reads against its two parts
    assert out["mapped_reads"] > 2700 and out["assigned_to_source_genome"] > 0.99
    # the same reads in one process (all eight parts on one rank): the merge must not depend on how the parts are dealt
    r1 = run_bench("--gpus", "1", "--mode", "shard", "--parts", "8", "--genomes", "16", "--reads", "3000", "--block", "3000",
                   "--steps", "1", "--warmup", "0")
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("mapped_reads", "ambiguous_reads", "assigned_to_source_genome"):
        assert out[k] == one[k], k
    r = run_bench("--gpus", "4", "--one-device", "--backend", "gloo", "--mode", "config3", "--total-reads", "20001", "--reads", "2500",
                  "--steps", "1", "--warmup", "0")
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 4 and out["group_ranks"] == 4
    assert out["ranks"]["reads_per_rank"] == [5001, 5000, 5000, 5000]      # dist.shard_bounds: the first rank takes the odd read
    assert out["counts_equal_mapped_reads"] and out["mapped_reads"] > 0.9 * 20001 and out["random_reads_mapped"] == 0
