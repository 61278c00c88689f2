#!/usr/bin/env python3
"""Headline benchmark: reads/s classified on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (2-bit pack -> minimizer sketch -> index probe ->
anchor sort -> chain -> regions -> base-level alignment of every region (what mappy runs for
monica) -> MAPQ -> decision + taxon counts) over one batch of synthetic
5 kb reads already resident in HBM, followed -- for N > 1 -- by the RCCL all-reduce of the
per-taxon count vector (the analogue of alignment_update, aligner.py:282-302).  Reads shard
across ranks with no other collective ("weak" scaling: every rank classifies its own
`--reads` reads per step).

Workload at N=1 = BASELINE.json configs[1]: 100 k synthetic 5 kb reads vs the 20-genome
index.  Rank 0 prints one JSON line with `roofline` (index-probe kernel, HIP-event timed on
the engine stream over the timed region) and, at N=1, `cpu_baseline` (the CPU oracle timed on
a bounded sample of the same reads on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=100_000, help="reads per rank per step")
    ap.add_argument("--read-len", type=int, default=5000)
    ap.add_argument("--genomes", type=int, default=20)
    ap.add_argument("--genome-model", choices=["iid", "repeats"], default="iid",
                    help="iid: SURVEY 8d's i.i.d. contigs (the BASELINE workload); repeats: the same contigs with rRNA-like operons, "
                         "insertion sequences and stretches shared between neighbours (synth.genome_set_repeats) -- a sensitivity row")
    ap.add_argument("--min-len", type=int, default=2_000_000)
    ap.add_argument("--max-len", type=int, default=7_000_000)
    ap.add_argument("--min-mapq", type=int, default=60)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="reads for the CPU baseline (-1 auto, 0 off)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank classifies --reads reads per step; strong: --total-reads reads per "
                         "step are split over the ranks (BASELINE config 3 is 10 M reads over 8 GPUs)")
    ap.add_argument("--total-reads", type=int, default=800_000, help="reads per step over all ranks with --scaling strong")
    ap.add_argument("--contract", choices=["dp", "chain"], default="dp",
                    help="dp (default): minimap2's base-level alignment of every region, as mappy runs it for monica "
                         "(mapq / NM / mlen from the CIGAR); chain: stop after chaining (the kernels north_star lists)")
    ap.add_argument("--mode", choices=["batch", "stream", "shard", "files", "config3"], default="batch",
                    help="batch: the headline metric; stream: BASELINE config 5 (400 reads/s arrival, 1-s "
                         "micro-batches); shard: config 4 (index parts spread over the ranks, every part sees "
                         "all reads, summaries all-gathered and merged); files: the Python aligner API end to "
                         "end on FASTQ files (parse, H2D, kernels, D2H, routed FASTQ output); config3: the whole 10 M-read "
                         "job of BASELINE config 3 (--total-reads, default 10 M there) split over the ranks, each rank's "
                         "share generated in its HBM by ordinal and classified in blocks of --reads, ONE count all-reduce "
                         "at the end of the job")
    ap.add_argument("--block", type=int, default=500_000,
                    help="reads per classify call in --mode shard (a part leaves most reads unmapped and aligns the rest -- many of "
                         "them against a diverged copy of their genome, with long extensions on single waves: the larger the call, "
                         "the more work runs beside those; ~200 KB of HBM per read)")
    ap.add_argument("--parts", type=int, default=8, help="index parts in --mode shard")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                                                      "the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--collective", choices=["torch", "capi"], default="torch",
                    help="who issues the count all-reduce: torch.distributed (default) or the library's own entry points "
                         "(mnc_comm_* / mnc_allreduce_counts on an RCCL communicator made through the C-ABI, on the engine's stream)")
    ap.add_argument("--stream-seconds", type=int, default=1800)
    ap.add_argument("--stream-rate", type=int, default=400)
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "probe_traffic.json"),
                    help="optional PMC-derived HBM bytes per launch of the probe kernel")
    return ap.parse_args()


def launch_ranks(n):
    """`python bench.py --gpus N` alone (no WORLD_SIZE in the environment): start the N ranks as a CHILD
    `python -m torch.distributed.run` job, pass its output through and exit with its code.  This parent
    never touches the GPU (no torch import, no HIP call) and never execs: the child processes own the devices."""
    import socket
    import subprocess
    if "--one-device" not in sys.argv:
        # more ranks than devices: say so in one line and fail BEFORE any rank starts.  The count comes from a child
        # process (the library's mnc_device_count through ctypes): this parent stays away from the HIP runtime.
        probe = subprocess.run([sys.executable, "-c", "from monica_amd import _capi; print(_capi.device_count())"],
                               cwd=ROOT, capture_output=True, text=True)
        try:
            have = int(probe.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            have = 0
        if n > have:
            print(f"[bench] --gpus {n} but {have} device(s) visible: not starting any rank", file=sys.stderr)
            sys.exit(2)
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    last_json = None
    for line in proc.stdout:                         # rank 0 prints ONE JSON line; anything else goes to stderr untouched
        if line.lstrip().startswith("{"):
            last_json = line
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if last_json is not None:
        sys.stdout.write(last_json)
        sys.stdout.flush()
    sys.exit(rc)


def rank_report(world, backend, reads_done, dev, comm=None):
    """What the collective backend saw, for the JSON line: `group_ranks` = the size of the torch.distributed group the
    collectives ran on; `rccl_ranks` = ncclCommCount of the library's own communicator (`--collective capi`), None when
    RCCL was not asked directly (torch's nccl backend does not hand out its communicator); `reads_per_rank` = the reads
    each rank's engine reported as classified inside the timed region (`reads_done`: a count the caller kept per
    successful classify call), gathered over that same group."""
    import torch
    import torch.distributed as dist
    rccl = comm.count() if comm is not None else None
    if world <= 1:
        return {"backend": None, "group_ranks": 1, "rccl_ranks": rccl, "reads_per_rank": [int(reads_done)]}
    mine = torch.tensor([int(reads_done)], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    got = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    return {"backend": dist.get_backend(), "group_ranks": dist.get_world_size(), "rccl_ranks": rccl,
            "reads_per_rank": [int(g.item()) for g in got]}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    import torch
    import torch.distributed as dist
    from monica_amd import _capi, synth
    from monica_amd import dist as mdist

    first_read = rank * args.reads
    if args.scaling == "strong":                 # a fixed job: this rank's contiguous block of its reads
        lo, hi = mdist.shard_bounds(args.total_reads, rank, max(world, 1))
        args.reads, first_read = hi - lo, lo
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    # ---------------------------------------------------------------- workload (deterministic)
    t0 = time.time()
    make_genomes = synth.genome_set_repeats if args.genome_model == "repeats" else synth.genome_set
    names, seqs = make_genomes(args.genomes, min_len=args.min_len, max_len=args.max_len)
    if args.mode == "shard":
        return shard_mode(args, names, seqs, rank, local_rank, world, dev)
    if args.mode == "config3":
        return config3_mode(args, names, seqs, rank, local_rank, world, dev)
    if args.mode == "files":
        return files_mode(args, names, seqs, local_rank)
    index = _capi.Index.from_seqs(names, seqs)
    if os.environ.get("MNC_REGION_BITS"):            # sweep only: table regions = 1 << bits (8..10) instead of the builder's choice
        _capi.check(_capi.lib().mnc_index_set_region_bits(index._h, int(os.environ["MNC_REGION_BITS"])))
    info = index.info()
    bases, offsets, truth = synth.reads(seqs, args.reads, args.read_len, seed=synth.SEED_READS + 2,
                                        first=first_read)
    engine = _capi.Engine(index, local_rank)
    engine.set_contract(_capi.CONTRACT_DP if args.contract == "dp" else _capi.CONTRACT_CHAIN)
    # tuning / profiling switches (mnc_engine_set_debug): MNC_FILL_PRED, MNC_FILL_PRED_MID = the planner's thresholds for the
    # 32- and the 42-cell tier; MNC_DP_SERIAL = the alignment kernels one at a time; MNC_DEBUG_BITS = anything else
    env_debug = (int(os.environ.get("MNC_FILL_PRED", "0")) << 8 | int(os.environ.get("MNC_FILL_PRED_MID", "0")) << 24 |
                 int(os.environ.get("MNC_DEBUG_BITS", "0"), 0))
    if env_debug or os.environ.get("MNC_DP_SERIAL"):
        engine.set_debug(env_debug | (0x10000 if os.environ.get("MNC_DP_SERIAL") else 0))
    n_genomes = info.n_genomes
    t_setup = time.time() - t0

    if args.mode == "stream":
        return stream_mode(args, engine, index, seqs, synth)

    comm = None
    if args.collective == "capi":
        # the communicator through the C-ABI alone: rank 0 makes the id, the other ranks get its 128 bytes
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(_capi.Comm.unique_id()), dtype=torch.uint8).clone()
        if world > 1:
            uid = uid.to(dev)
            dist.broadcast(uid, 0)
            uid = uid.cpu()
        comm = _capi.Comm(uid.numpy().tobytes(), world, rank)

    d_bases = torch.from_numpy(bases).to(dev)
    d_off = torch.from_numpy(offsets).to(dev)
    d_assign = torch.empty(args.reads, dtype=torch.int32, device=dev)
    d_best = torch.zeros(args.reads * 4, dtype=torch.int32, device=dev)
    d_nhits = torch.zeros(args.reads, dtype=torch.int32, device=dev)
    d_counts = torch.zeros(n_genomes * 3, dtype=torch.int64, device=dev)
    total_bases = int(offsets[-1])
    torch.cuda.synchronize()

    done = [0]                                 # reads whose classify call returned, counted as they happen

    def step():
        d_counts.zero_()
        torch.cuda.current_stream().synchronize()
        engine.classify_device(d_bases.data_ptr(), d_off.data_ptr(), args.reads, total_bases, args.read_len,
                               args.min_mapq, d_assign.data_ptr(), d_best.data_ptr(), d_nhits.data_ptr(),
                               d_counts.data_ptr())
        done[0] += engine.n_reads
        if comm is not None:                   # on the engine's stream, behind the batch's last kernel
            comm.allreduce_counts(d_counts.data_ptr(), n_genomes * 3, engine.stream)
        engine.sync()
        if world > 1 and comm is None:
            dist.all_reduce(d_counts)          # RCCL over xGMI: n_genomes*3 int64

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    engine.set_profiling(True)
    engine.timings(reset=True)
    torch.cuda.synchronize()
    done[0] = 0
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    reads_timed = done[0]
    engine.set_profiling(False)
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    timings = engine.timings()
    counters = engine.counters()
    assign = d_assign.cpu().numpy()
    counts = d_counts.cpu().numpy().reshape(-1, 3)

    # the same batch without base-level alignment (a secondary figure: the path of north_star's
    # kernel list; its NM / mlen / mapq are chain-level estimates, not what monica reads from mappy)
    chain_level = None
    if args.contract == "dp" and world == 1:
        engine.set_contract(_capi.CONTRACT_CHAIN)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        tcl = time.perf_counter()
        n_cl = max(3, min(args.steps, 10))
        for _ in range(n_cl):
            step()
        torch.cuda.synchronize()
        dcl = time.perf_counter() - tcl
        a_chain = d_assign.cpu().numpy()
        chain_level = {"value": round(args.reads * n_cl / dcl, 1), "unit": "reads/s", "ms_per_step": round(dcl / n_cl * 1e3, 3),
                       "steps": n_cl, "decisions_equal_to_dp": round(float((a_chain == assign).mean()), 5)}
        engine.set_contract(_capi.CONTRACT_DP)

    # the alignment kernels one at a time (they share the chip in the timed steps): HIP events around
    # each launch, on the stream it runs on -- the dominant kernel's own duration for `roofline`
    dp_kernel_ms = None
    if args.contract == "dp" and world == 1:
        base_debug = env_debug
        engine.set_debug(base_debug | 0x10000)
        engine.set_profiling(True)
        engine.timings(reset=True)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        tk = engine.timings()
        engine.set_profiling(False)
        engine.set_debug(base_debug | (0x10000 if os.environ.get("MNC_DP_SERIAL") else 0))
        dp_kernel_ms = {k: tk[k][0] / max(tk[k][1], 1) for k in ("dp_fill_t1", "dp_fill_tm", "dp_fill_t2", "dp_fill_t3", "dp_lfill", "dp_ext", "dp_stitch") if k in tk}

    ranks = rank_report(world, args.backend, reads_timed, dev, comm)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---------------------------------------------------------------- roofline of the dominant kernel
    # With base-level alignment the batch is bound by the gap-filling kernel mnc_dp_fillp<16> (32-cell
    # band): 16-bit pair arithmetic on the vector ALU, neither HBM nor MFMA.  Its roof is the vector
    # issue rate of the instructions it is made of (v_pk_max/add/sub_i16, v_perm, v_alignbit, DPP moves):
    # one wave64 instruction per 4 cycles and SIMD, measured with 8 waves per SIMD by
    # tools/micro/valu_rate.hip (profiles/r02_valu_issue_rates.json: 4.1-4.2 cycles; the guide's 2-cycle
    # figure is v_fma_f32's) -> 256 CUs x 4 SIMDs x 2.4 GHz / 4.  One step of the kernel updates the 128
    # band cells of a wave (4 segments x 32) in `FILLP_INSTR_PER_STEP` vector instructions (its ISA): the
    # roof is that many cell updates/s.  Algorithmic work per launch = the anti-diagonals of the gap
    # fillings given to this tier (counter dp_fill_steps_t1) x 32 cells.
    # Round 5: the roof in MEASURED cycles at the MEASURED clock -- tools/micro/valu_rate.hip reads s_memtime against the
    # 100 MHz counter in every launch: 2 365 MHz while it ran, 4.15 cycles per v_pk_max_i16 / v_perm_b32 / DPP move with 8
    # waves a SIMD (profiles/r05s_valu_issue_rates.json, r05s_valu_clock.json): 1 024 SIMDs x 2.365e9 / 4.15 = 5.84e11
    # wave64 instructions/s.  (Rounds 2-4 priced against 2.4 GHz / 4 cycles = 6.144e11, 5 % above what the chip does:
    # `frac` moved up by that much with no change in the kernel; `frac_at_r04_peak` is the old convention.)
    VALU_CLOCK_HZ, VALU_CYCLES_PER_INSTR = 2.3648e9, 4.15
    VALU_PEAK = 256 * 4 * VALU_CLOCK_HZ / VALU_CYCLES_PER_INSTR
    VALU_PEAK_R04 = 256 * 4 * 2.4e9 / 4
    FILLP_INSTR_PER_STEP = 464 / 16                      # vector instructions of the unrolled 16-step block of the main loop (ISA listing of the round's last build: profiles/README.md; 480 before the second gap piece got its own frame, 521 with the first form of the drifting frame, 568 before it)
    VALU_PEAK_GUIDE = 256 * 4 * 2.4e9 / 2                # the guide's nominal 2-cycle wave64 issue (MI355X_MICROARCH.md), for comparison
    # Round 5, by instruction class: not every opcode of the loop costs 4.15 cycles.  With 4 waves a SIMD (what the kernel runs
    # with) v_add / v_sub / v_and / v_or / v_xor / v_mov issue in 2.32 cycles, v_bitop3 in 3.8, everything else the loop uses --
    # v_pk_*, v_perm, v_alignbit, v_bfi, v_lshlrev, DPP moves -- in 4.22 (profiles/r05y_valu_issue_rates.json).  The 16-step
    # block's 464 vector instructions are 118 of the first kind (64 v_and, 32 v_xor, 16 v_or, 5 v_add, 1 v_mov), 16 v_bitop3
    # and 330 of the rest (128 v_pk_max_i16, 32 DPP moves, 32 v_alignbit, 32 v_pk_sub_i16, 24 v_perm, ...): 1 727 cycles,
    # 3.72 an instruction -- the roof `frac_by_instruction_class` is priced against (the loop alone: the walks and the
    # sequence loads of a launch are further instructions, which `issue_rate_frac` counts)
    FILLP_BLOCK_CYCLES = 118 * 2.32 + 16 * 3.8 + 330 * 4.22
    VALU_PEAK_FILLP_CELLS = 256 * 4 * VALU_CLOCK_HZ / (FILLP_BLOCK_CYCLES / 16 / 128)
    roofline_dp = None
    if dp_kernel_ms and dp_kernel_ms.get("dp_fill_t1", 0) > 0 and counters.get("dp_fill_steps_t1", 0) > 0:
        cells = counters["dp_fill_steps_t1"] * 32
        t_s = dp_kernel_ms["dp_fill_t1"] / 1e3
        instr_per_cell = FILLP_INSTR_PER_STEP / 128
        peak = VALU_PEAK / instr_per_cell / 1e9
        ach = cells / t_s / 1e9
        fill_traffic = fill_insts = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(args.traffic_json)), "fill_traffic.json")) as f:
                tj = json.load(f)
            if tj.get("reads") == args.reads and tj.get("read_len") == args.read_len:
                fill_traffic = tj.get("hbm_bytes_per_launch")
                fill_insts = tj.get("sq_insts_valu")               # SQ_INSTS_VALU of one launch (a separate --pmc pass, profiles/)
        except Exception:
            pass
        roofline_dp = {"bound": "valu", "kernel": "mnc_dp_fillp<16, true>", "achieved": round(ach, 2), "peak": round(peak, 1),
                       "unit": "Gcell/s", "frac": round(ach / peak, 4),
                       "frac_at_r04_peak": round(ach / (VALU_PEAK_R04 / instr_per_cell / 1e9), 4),
                       "peak_clock_mhz_measured": round(VALU_CLOCK_HZ / 1e6, 1), "peak_cycles_per_instruction_measured": VALU_CYCLES_PER_INSTR,
                       # the stricter roof: every opcode of the loop at its own measured rate (see FILLP_BLOCK_CYCLES above)
                       "frac_by_instruction_class": round(ach / (VALU_PEAK_FILLP_CELLS / 1e9), 4),
                       "peak_by_instruction_class": round(VALU_PEAK_FILLP_CELLS / 1e9, 1),
                       "cycles_per_16_steps_by_instruction_class": round(FILLP_BLOCK_CYCLES, 1),
                       # the same launch against the guide's nominal issue rate (one wave64 instruction per 2 cycles and SIMD):
                       # the packed-16-bit / perm / DPP instructions this kernel is made of issue at half that (measured)
                       "frac_guide_nominal": round(ach / (VALU_PEAK_GUIDE / instr_per_cell / 1e9), 4),
                       "peak_guide_nominal": round(VALU_PEAK_GUIDE / instr_per_cell / 1e9, 1),
                       "traffic": fill_traffic,
                       "algorithmic_cells_per_launch": int(cells), "avg_launch_ms": round(dp_kernel_ms["dp_fill_t1"], 4),
                       "vector_instructions_per_cell": round(instr_per_cell, 4),
                       # independent of the instruction count above: the vector instructions the launch issued (SQ counter of the
                       # profiled build, all of the kernel -- walks and loads too) over what the chip can issue in the launch's time
                       "issue_rate_frac": round(fill_insts / t_s / VALU_PEAK, 4) if fill_insts else None,
                       "vector_instructions_per_launch": int(fill_insts) if fill_insts else None,
                       "bound_note": "int16 pair arithmetic on the vector ALU: the roof is the VALU issue rate (5.84e11 wave64 "
                                     "instructions/s = 1 024 SIMDs x 2 365 MHz / 4.15 cycles, both measured: profiles/r05s_valu_issue_rates.json, r05s_valu_clock.json), not HBM (the kernel writes 1 byte per cell: "
                                     f"{cells / t_s / 1e9:.0f} GB/s) and not MFMA",
                       # the same launch against the HBM roof, for comparison with the contract's "hbm" bound: algorithmic
                       # bytes = one direction byte written per band cell
                       "as_hbm": {"bound": "hbm", "achieved": round(cells / t_s / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                  "frac": round(cells / t_s / 1e9 / 8000.0, 4), "traffic": fill_traffic},
                       "kernels_one_at_a_time_ms": {k: round(v, 4) for k, v in dp_kernel_ms.items()}}

    # ---------------------------------------------------------------- roofline of the probe kernel
    probe_ms, probe_n = timings["probe"]
    M, H = counters["minimizers"], counters["probe_hits"]
    # algorithmic bytes of one launch: queries 8 B + one 16-B table slot
    # per query + one 16-B hit record per probe hit + 3 per-read result words
    algo_bytes = M * 8 + M * 16 + H * 16 + args.reads * (4 + 4 + 8 + 4 + 16)
    roofline = None
    if probe_n > 0 and probe_ms > 0:
        avg_s = probe_ms / probe_n / 1e3
        achieved = algo_bytes / avg_s / 1e9
        traffic = None
        try:
            with open(args.traffic_json) as f:
                tj = json.load(f)
            tj = tj.get("by_genomes", {}).get(str(args.genomes), tj if args.genomes == tj.get("genomes", 20) else {})
            if tj.get("reads") == args.reads and tj.get("read_len") == args.read_len:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            pass
        roofline = {"bound": "hbm", "kernel": "mnc_probe_buckets", "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "algorithmic_bytes_per_launch": int(algo_bytes),
                    "avg_launch_ms": round(probe_ms / probe_n, 4), "launches": probe_n}

    # the probe STAGE: partition + probe + collect, the same algorithmic bytes
    # (the stage's useful work: queries in, one table slot per query, hits out) over the three
    # kernels' time; `traffic` = their HBM bytes from the PMC passes, when profiles/ holds them
    roofline_stage = None
    try:
        st_ms = sum(timings[k][0] / max(timings[k][1], 1) for k in ("partition", "probe", "collect"))
        if st_ms > 0:
            ach = algo_bytes / (st_ms / 1e3) / 1e9
            st_traffic = None
            try:
                with open(args.traffic_json) as f:
                    tj = json.load(f)
                tj = tj.get("by_genomes", {}).get(str(args.genomes), tj if args.genomes == tj.get("genomes", 20) else {})
                if tj.get("reads") == args.reads and tj.get("read_len") == args.read_len:
                    st_traffic = tj.get("stage_hbm_bytes_per_launch")
            except Exception:
                pass
            roofline_stage = {"bound": "hbm", "kernels": ["mnc_partition_queries (+ scans)", "mnc_probe_buckets", "mnc_collect_hits"],
                              "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                              "traffic": st_traffic, "algorithmic_bytes_per_launch": int(algo_bytes), "ms": round(st_ms, 4)}
    except Exception:
        pass

    # ---------------------------------------------------------------- CPU baseline (oracle, bounded sample)
    cpu = None
    if world == 1 and args.cpu_sample != 0:
        from oracle import pyoracle
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        cores = min(cores, 16)                       # the GPU box's CPU share for one GPU
        oidx = pyoracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
        oidx.opt.cigar = 1 if args.contract == "dp" else 0
        # the oracle's literal ksw2 simulation is scalar: a few hundred reads/s on 16 cores
        n_s = args.cpu_sample if args.cpu_sample > 0 else (min(args.reads, 4000) if args.contract == "dp" else args.reads)
        ob, oo = bases[: n_s * args.read_len], offsets[: n_s + 1]
        reps, dt = 0, 0.0
        while True:                                  # about 10-30 s of CPU work on the sample
            tc = time.perf_counter()
            oa, _, _, _ = oidx.classify(ob, oo, args.min_mapq, n_threads=cores)
            dt += time.perf_counter() - tc
            reps += 1
            if args.cpu_sample > 0 or dt >= 10.0 or reps >= 50:
                break
        agree = bool(np.array_equal(oa, assign[:n_s]))
        # SURVEY 8(d) also asks for the 1-core figure: the same oracle, one thread, a smaller slice of the same sample
        n_1 = max(1, min(n_s, 500 if args.contract == "dp" else 20_000))
        tc = time.perf_counter()
        o1, _, _, _ = oidx.classify(bases[: n_1 * args.read_len], offsets[: n_1 + 1], args.min_mapq, n_threads=1)
        dt1 = time.perf_counter() - tc
        one_core = {"value": round(n_1 / dt1, 1), "unit": "reads/s", "cores": 1,
                    "sample": f"first {n_1} reads of the same batch, one pass, one thread, {dt1:.1f} s",
                    "agrees_with_gpu": bool(np.array_equal(o1, assign[:n_1]))}
        cpu = {"value": round(n_s * reps / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": f"first {n_s} reads of the same batch x {reps} passes, CPU oracle (minimap2-2.17 restatement "
                         f"{'with base-level alignment, scalar ksw2 simulation' if args.contract == 'dp' else 'at the chain level'}"
                         f", OpenMP over reads), {dt:.1f} s", "agrees_with_gpu": agree, "one_core": one_core}

    # the PCIe-inclusive rate (never `value`): fresh batches from page-locked host memory through the
    # host-buffer entry point -- H2D of the bases, all kernels, D2H of the decisions
    end_to_end = None
    if world == 1:
        import threading
        hb = _capi.pinned_array(bases)
        engine.classify(hb, offsets, args.min_mapq)
        te = time.perf_counter()
        n_e2e = 3
        for _ in range(n_e2e):
            e_assign, _, _ = engine.classify(hb, offsets, args.min_mapq)
        de = time.perf_counter() - te
        end_to_end = {"value": round(args.reads * n_e2e / de, 1), "unit": "reads/s", "ms_per_batch": round(de / n_e2e * 1e3, 3),
                      "what": "mnc_classify_batch on page-locked host buffers: H2D of 5 000 ASCII bytes per read, kernels, D2H",
                      "equal_to_resident": bool(np.array_equal(e_assign, assign))}
        # ... with the next batch's copy started before this batch's call (mnc_engine_prefetch: a copy stream and a spare
        # device buffer inside the ONE engine), as the aligner's loop does with the batch its parser thread has ready
        hb2 = _capi.pinned_array(bases)
        bufs = [hb, hb2]
        engine.classify(bufs[0], offsets, args.min_mapq)
        n_pf = 6

        def announce(buf):                          # the spare device buffer is free once the running call has taken its own
            while not engine.prefetch_ptr(buf.ctypes.data, offsets.ctypes.data, args.reads):
                time.sleep(0.0005)

        announce(bufs[0])
        te = time.perf_counter()
        for k in range(n_pf):
            t = threading.Thread(target=announce, args=(bufs[(k + 1) & 1],)) if k + 1 < n_pf else None
            if t:
                t.start()
            p_assign, _, _ = engine.classify_ptr(bufs[k & 1].ctypes.data, offsets.ctypes.data, args.reads, args.min_mapq)
            if t:
                t.join()
        dp_ = time.perf_counter() - te
        end_to_end["prefetch"] = {"value": round(args.reads * n_pf / dp_, 1), "unit": "reads/s", "ms_per_batch": round(dp_ / n_pf * 1e3, 3),
                                  "equal_to_resident": bool(np.array_equal(p_assign, assign)),
                                  "what": "one engine; the next batch's H2D copy runs behind this batch's kernels (mnc_engine_prefetch)"}
        # ... and as monica's thread pool drives it (aligner.py:65-111: one sample per thread): two engines of the same
        # device on two host threads, one's copies behind the other's kernels
        engine_b = _capi.Engine(index, local_rank)
        engine_b.set_contract(_capi.CONTRACT_DP if args.contract == "dp" else _capi.CONTRACT_CHAIN)
        hb_b = _capi.pinned_array(bases)
        engine_b.classify(hb_b, offsets, args.min_mapq)
        outs = [None, None]

        def drive(k, eng, buf):
            for _ in range(n_e2e):
                outs[k] = eng.classify(buf, offsets, args.min_mapq)[0]
        th = [threading.Thread(target=drive, args=(0, engine, hb)), threading.Thread(target=drive, args=(1, engine_b, hb_b))]
        te = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        de2 = time.perf_counter() - te
        end_to_end["two_engines"] = {"value": round(2 * args.reads * n_e2e / de2, 1), "unit": "reads/s",
                                     "ms_per_batch": round(de2 / (2 * n_e2e) * 1e3, 3),
                                     "equal_to_resident": bool(np.array_equal(outs[0], assign) and np.array_equal(outs[1], assign))}
        del engine_b

    if args.scaling == "strong":
        n_total = args.total_reads * args.steps
    else:
        n_total = args.reads * args.steps * world
    mapped = int((assign >= 0).sum())
    out = {
        "metric": "reads/sec classified",
        "value": round(n_total / elapsed, 1),
        "unit": "reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "timed_region_s": round(elapsed, 4),
        "group_ranks": ranks["group_ranks"], "rccl_ranks": ranks["rccl_ranks"],
        "ranks": ranks,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u32/u64 integer; int16 pairs in the alignment kernels (int32 / int8 in the long-call and literal ones); float32 islands: overlap ratio, MAPQ",
        "data": "synthetic",
        "config": {"workload": f"{args.reads} synthetic {args.read_len} nt reads per GPU per step vs "
                               f"{args.genomes}-genome minimizer index ({info.total_len} bp, {info.n_keys} keys, "
                               f"mid_occ {info.mid_occ})",
                   "reads_per_gpu_per_step": args.reads, "read_len": args.read_len, "genomes": args.genomes,
                   "genome_model": args.genome_model,
                   "contract": "base-level alignment of every region (mappy's MM_F_CIGAR)" if args.contract == "dp" else "chain level",
                   "collective": "mnc_allreduce_counts (C-ABI, RCCL communicator from mnc_comm_init_rank)" if comm is not None else "torch.distributed all_reduce (RCCL)",
                   "parallelism": f"read-sharded x{world}, index replicated, RCCL all-reduce of "
                                  f"{n_genomes * 3} int64 counts per step"},
        "roofline": roofline_dp if roofline_dp else roofline,
        "roofline_probe": roofline,
        "roofline_stage": roofline_stage,
        "cpu_baseline": cpu,
        "chain_level": chain_level,
        "end_to_end": end_to_end,
        "stage_ms_per_step": {k: round(v[0] / max(v[1], 1), 4) for k, v in timings.items() if v[1]},
        "batch_counters": counters,
        "mapped_reads_last_step": mapped,
        "counts_checksum": int(counts[:, 0].sum()),
        "setup_s": round(t_setup, 1),
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def stream_mode(args, engine, index, seqs, synth):
    """BASELINE config 5: a MinION run delivering `--stream-rate` reads/s for `--stream-seconds`
    seconds, classified as 1-second micro-batches through the host-buffer entry point
    (`mnc_classify_batch`: H2D of the bases, all kernels, D2H of the results).  The simulated
    clock is compressed: batches are processed back to back and the per-batch wall time is the
    latency a read would see after its batch closes."""
    n_batches, rate = args.stream_seconds, args.stream_rate
    pool, offs, truth = synth.reads(seqs, rate * 64, args.read_len, seed=synth.SEED_READS + 5)
    lat, mapped, total = [], 0, 0
    counts = np.zeros(len(index.genome_names), dtype=np.int64)
    t_all = time.perf_counter()
    for b in range(n_batches):
        k = b % 64
        bb = pool[k * rate * args.read_len:(k + 1) * rate * args.read_len]
        bo = offs[: rate + 1]
        t0 = time.perf_counter()
        assign, best, nhits = engine.classify(bb, bo, args.min_mapq)
        lat.append(time.perf_counter() - t0)
        ok = assign >= 0
        mapped += int(ok.sum())
        total += rate
        np.add.at(counts, index.contig_genome[assign[ok]], 1)
    wall = time.perf_counter() - t_all
    lat = np.array(lat[5:]) * 1e3
    print(json.dumps({
        "metric": "streaming micro-batch latency (BASELINE config 5)", "unit": "ms",
        "p50_ms": round(float(np.percentile(lat, 50)), 3), "p99_ms": round(float(np.percentile(lat, 99)), 3),
        "max_ms": round(float(lat.max()), 3), "batches": n_batches, "reads_per_batch": rate,
        "arrival_reads_per_s": rate, "sustained_reads_per_s": round(total / wall, 1),
        "real_time_factor": round(total / wall / rate, 1), "mapped_fraction": round(mapped / total, 4),
        "path": "host buffers -> mnc_classify_batch (H2D + kernels + D2H), 1 MI355X", "data": "synthetic"}))


def shard_mode(args, names, seqs, rank, local_rank, world, dev):
    """BASELINE config 4: the genomes are split into `--parts` index parts, spread round-robin
    over the ranks; every part classifies ALL reads (MAPQ and the gate are per part, as in the
    reference's multi-part loop, aligner.py:91-103), and the per-read summaries {hits, nm, mlen,
    contig, tied} are all-gathered (20 B per read and part, RCCL) and reduced with best_hit's
    rule on every rank.  A genome and its diverged copy sit in different parts, so the merge
    decides real ties."""
    import torch
    import torch.distributed as dist
    from monica_amd import _capi, synth
    from monica_amd import dist as mdist

    t0 = time.time()
    P, G = args.parts, len(names)
    if P % world:
        raise SystemExit(f"--parts {P} must be a multiple of the number of ranks {world}")
    bounds = [mdist.shard_bounds(G, p, P) for p in range(P)]
    mine = list(range(rank * (P // world), (rank + 1) * (P // world)))      # consecutive parts: rank order = part order
    parts = []
    for p in mine:
        lo, hi = bounds[p]
        idx = _capi.Index.from_seqs(names[lo:hi], seqs[lo:hi])
        parts.append((lo, idx))
    # ONE engine per rank (stream + batch buffers), rebound to the part at hand -- `index = index_loader(part)` in the
    # reference's loop (aligner.py:91-103); every part of the rank is resident in HBM after its first use
    eng = _capi.Engine(parts[0][1], local_rank)
    eng.set_contract(_capi.CONTRACT_DP if args.contract == "dp" else _capi.CONTRACT_CHAIN)
    bases, offsets, truth = synth.reads(seqs, args.reads, args.read_len, seed=synth.SEED_READS + 4)
    n = args.reads
    d_bases, d_off = torch.from_numpy(bases).to(dev), torch.from_numpy(offsets).to(dev)
    d_assign = torch.empty(n, dtype=torch.int32, device=dev)
    d_best = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    d_nhits = torch.zeros(n, dtype=torch.int32, device=dev)
    total_bases = int(offsets[-1])
    t_setup = time.time() - t0
    result = {}

    # a classify call takes at most --block reads (its per-base scratch is ~40 B per base and every part has an
    # engine of its own); the reads stay resident, a block is a slice of them
    L = args.read_len
    blocks = [(b0, min(n, b0 + args.block)) for b0 in range(0, n, args.block)]
    d_off_blk = torch.arange(args.block + 1, dtype=torch.int64, device=dev) * L

    # C2 through the library's own entry points (include/monica_amd.h): a part's 20-byte records written by
    # mnc_shard_summary on the engine's stream, all parts' records in part order in ONE buffer (rank r holds parts
    # r * P / world ...: rank order is part order, so the all-gather leaves them in hit order), merged by mnc_merge_summaries
    n_local = len(parts)
    d_local = torch.empty((n_local, n, 5), dtype=torch.int32, device=dev)
    d_all = torch.empty((P, n, 5), dtype=torch.int32, device=dev) if world > 1 else d_local
    d_merged = torch.empty(n, dtype=torch.int32, device=dev)

    done = [0]                                                        # reads x parts whose classify call returned

    def step():
        for k, (lo, idx) in enumerate(parts):
            eng.set_index(idx)                                        # (waits for the engine's stream)
            for b0, b1 in blocks:
                eng.classify_device(d_bases.data_ptr() + b0 * L, d_off_blk.data_ptr(), b1 - b0, (b1 - b0) * L, L, args.min_mapq,
                                    d_assign.data_ptr() + b0 * 4, d_best.data_ptr() + b0 * 16, d_nhits.data_ptr() + b0 * 4, 0)
                done[0] += eng.n_reads
            _capi.shard_summary_device(d_assign.data_ptr(), d_best.data_ptr(), d_nhits.data_ptr(), n, lo, d_local[k].data_ptr(), eng.stream)
            eng.sync()
        if world > 1:
            dist.all_gather(list(d_all.view(world, n_local, n, 5).unbind(0)), d_local)   # RCCL over xGMI; rank order = part order
        _capi.merge_summaries_device(d_all.data_ptr(), P, n, d_merged.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        result["assign"] = d_merged

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    done[0] = 0
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    assign = result["assign"].cpu().numpy()
    ranks = rank_report(world, args.backend, done[0], dev)
    if rank == 0:
        ok = truth >= 0
        print(json.dumps({
            "metric": "reads/sec classified, index sharded (BASELINE config 4)", "value": round(n * args.steps / elapsed, 1),
            "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "timed_region_s": round(elapsed, 4),
            "group_ranks": ranks["group_ranks"], "rccl_ranks": ranks["rccl_ranks"], "ranks": ranks, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32/u64 integer", "data": "synthetic",
            "config": {"workload": f"{n} synthetic {args.read_len} nt reads vs {G} genomes "
                                   f"({sum(len(s) for s in seqs)} bp) in {P} index parts, {P // world} per GPU; "
                                   "every part maps all reads",
                       "parallelism": f"index-sharded x{world}, all-gather of {P}x{n}x20 B summaries + best_hit merge"},
            "mapped_reads": int((assign >= 0).sum()), "ambiguous_reads": int((assign == -2).sum()),
            "assigned_to_source_genome": round(float((assign[ok] == truth[ok]).mean()), 4),
            "setup_s": round(t_setup, 1)}))
    if world > 1:
        dist.destroy_process_group()


def config3_mode(args, names, seqs, rank, local_rank, world, dev):
    """BASELINE config 3 as one job: --total-reads reads (10 M) vs the 20-genome index, the read ordinals split over
    the ranks in contiguous shares (dist.shard_bounds), every share generated in its rank's HBM (50 GB at one GPU)
    and resident when the clock starts, classified in blocks of --reads with on-device count accumulation, ONE
    RCCL all-reduce of the count table at the end of the job (north_star; aligner.py:286-298 is the analogue).
    A step = the whole job."""
    import torch
    import torch.distributed as dist
    from monica_amd import _capi, synth
    from monica_amd import dist as mdist

    t0 = time.time()
    total = args.total_reads if args.total_reads != 800_000 else 10_000_000
    lo, hi = mdist.shard_bounds(total, rank, max(world, 1))
    n, L, blk = hi - lo, args.read_len, args.reads
    index = _capi.Index.from_seqs(names, seqs)
    info = index.info()
    engine = _capi.Engine(index, local_rank)
    engine.set_contract(_capi.CONTRACT_DP if args.contract == "dp" else _capi.CONTRACT_CHAIN)
    gen = synth.DeviceReads(seqs, dev)
    d_bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    d_truth = torch.empty(n, dtype=torch.int32, device=dev)
    for b0 in range(0, n, 1_000_000):
        b1 = min(n, b0 + 1_000_000)
        gen.make(d_bases[b0 * L:], d_truth[b0:], b1 - b0, L, seed=synth.SEED_READS + 3, first=lo + b0)
    torch.cuda.synchronize()
    d_off = torch.arange(blk + 1, dtype=torch.int64, device=dev) * L
    d_assign = torch.empty(n, dtype=torch.int32, device=dev)
    d_best = torch.zeros(n * 4, dtype=torch.int32, device=dev)
    n_genomes = info.n_genomes
    d_counts = torch.zeros(n_genomes * 3, dtype=torch.int64, device=dev)
    t_setup = time.time() - t0

    done = [0]                                                 # reads whose classify call returned

    def step():
        d_counts.zero_()
        torch.cuda.current_stream().synchronize()
        for b0 in range(0, n, blk):
            m = min(blk, n - b0)
            engine.classify_device(d_bases.data_ptr() + b0 * L, d_off.data_ptr(), m, m * L, L, args.min_mapq,
                                   d_assign.data_ptr() + b0 * 4, d_best.data_ptr() + b0 * 16, 0, d_counts.data_ptr())
            done[0] += engine.n_reads
        engine.sync()
        if world > 1:
            dist.all_reduce(d_counts)                          # the job's one collective

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    done[0] = 0
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    assign, truth = d_assign.cpu().numpy(), d_truth.cpu().numpy()
    counts = d_counts.cpu().numpy().reshape(-1, 3)
    mapped = assign >= 0
    local = torch.tensor([int(mapped.sum()), int((assign[mapped] == truth[mapped]).sum()), int((assign[truth < 0] != -1).sum())],
                         dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(local)
    ranks = rank_report(world, args.backend, done[0], dev)
    if rank == 0:
        mp, ok, bad = (int(x) for x in local.cpu())
        print(json.dumps({
            "metric": "reads/sec classified, 10 M-read job (BASELINE config 3)", "value": round(total * args.steps / elapsed, 1),
            "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "timed_region_s": round(elapsed, 4),
            "group_ranks": ranks["group_ranks"], "rccl_ranks": ranks["rccl_ranks"], "ranks": ranks, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u32/u64 integer; int16 pairs in the alignment kernels", "data": "synthetic",
            "config": {"workload": f"{total} synthetic {L} nt reads in all ({n} on rank 0, blocks of {blk}) vs {args.genomes}-genome "
                                   f"minimizer index ({info.total_len} bp)", "contract": args.contract,
                       "parallelism": f"read-sharded x{world}, index replicated, one RCCL all-reduce of {n_genomes * 3} int64 counts per job"},
            "mapped_reads": mp, "counts_checksum": int(counts[:, 0].sum()), "counts_equal_mapped_reads": bool(int(counts[:, 0].sum()) == mp),
            "assigned_to_source_genome": round(ok / max(mp, 1), 5), "random_reads_mapped": bad, "setup_s": round(t_setup, 1)}))
    if world > 1:
        dist.destroy_process_group()


def files_mode(args, names, seqs, local_rank):
    """The drop-in surface end to end (monica_amd.aligner.multi_threaded_aligner, the mirror of
    monica/genomes/aligner.py:65-111) on files: index file load, FASTQ parse, H2D, kernels, D2H,
    carried-hits update, routed FASTQ output, alignment.pkl.  PCIe- and disk-inclusive; never the
    headline `value`."""
    import shutil
    import tempfile
    from monica_amd import _capi, synth
    from monica_amd import aligner as al

    os.environ["MONICA_AMD_DEVICE"] = str(local_rank)
    work = tempfile.mkdtemp(prefix="mnc_files_")
    try:
        query, out = os.path.join(work, "query"), os.path.join(work, "out")
        os.makedirs(query), os.makedirs(out)
        idx_path = os.path.join(work, "index1.mmi")
        _capi.Index.from_seqs(names, seqs).save(idx_path)
        bases, offsets, truth = synth.reads(seqs, args.reads, args.read_len, seed=synth.SEED_READS + 2)
        fq = os.path.join(query, "sample.fastq")
        t0 = time.perf_counter()
        synth.write_fastq(fq, bases, offsets)
        t_write = time.perf_counter() - t0
        size = os.path.getsize(fq)
        cwd = os.getcwd()
        t0 = time.perf_counter()
        result = al.multi_threaded_aligner(query, [idx_path], mode="basic", n_threads=1, output_folder=out)
        wall = time.perf_counter() - t0
        os.chdir(cwd)
        first_phases = {k: round(v, 3) for k, v in al.TIMINGS.get("sample", {}).items()}
        # monica's real-time loop calls the same function again when new reads have arrived
        synth.write_fastq(fq, bases, offsets)
        al.TIMINGS.clear()
        import resource
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        t0 = time.perf_counter()
        al.multi_threaded_aligner(query, [idx_path], mode="basic", n_threads=1, output_folder=out)
        wall2 = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        os.chdir(cwd)
        counted = sum(sum(c.values()) for c in result["sample"].values())
        routed = {k: os.path.getsize(os.path.join(query, k, "sample.fastq")) for k in ("mapped", "unmapped", "ambiguous")}
        print(json.dumps({
            "metric": "reads/sec through the Python aligner API on FASTQ files", "value": round(args.reads / wall, 1),
            "unit": "reads/s", "n_gpus": 1, "reads": args.reads, "fastq_bytes": size, "wall_s": round(wall, 3),
            "includes": "index load + upload, FASTQ parse, H2D, kernels, D2H, carried-hits update, routed FASTQ "
                        "output, alignment.pkl", "mapped_reads_counted": int(counted), "routed_bytes": routed,
            "fastq_write_s_python": round(t_write, 2),
            "aligner_phase_s": first_phases,
            "second_call": {"value": round(args.reads / wall2, 1), "wall_s": round(wall2, 3),
                            "aligner_phase_s": {k: round(v, 3) for k, v in al.TIMINGS.get("sample", {}).items()},
                            # host threads' time inside the call, all of them: what the pipeline's stages cost together
                            "cpu_user_s": round(ru1.ru_utime - ru0.ru_utime, 3), "cpu_sys_s": round(ru1.ru_stime - ru0.ru_stime, 3),
                            "host_cores": os.cpu_count()},
            "index_load_s": round(al.TIMINGS.get("_index_loader", {}).get("load", 0.0), 3), "data": "synthetic"}))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
