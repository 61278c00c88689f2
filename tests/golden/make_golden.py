#!/usr/bin/env python3
"""Regenerate tests/golden/small_case.npz.

The reference (monica + mappy 2.17) cannot run in this image and its tests hold no golden
vectors, so these vectors are produced by the CPU oracle
(oracle/mm_oracle.c) from seeded synthetic data.  They pin the oracle against regressions
and give the GPU tests a fixture that does not depend on the oracle being importable.
PARITY UNPINNED with respect to mappy itself.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from monica_amd import synth          # noqa: E402
from oracle import pyoracle           # noqa: E402
import util                           # noqa: E402


def main():
    names, seqs = synth.genome_set(4, seed=0x601D, div_seed=0x601E, min_len=50_000, max_len=70_000)
    rng = np.random.default_rng(0x601D)
    b, o, truth = synth.reads(seqs, 40, 2000, seed=0x601F)
    reads = [b[o[i]:o[i + 1]] for i in range(40)] + util.edge_reads_small(seqs, rng)
    bases, offsets = util.pack_reads(reads)
    oidx = pyoracle.Index.from_seqs(names, [s.tobytes() for s in seqs])
    # ---- with base-level alignment (what mappy does): decisions, regions, CIGARs
    dp_assign, dp_best, dp_nhits, dp_flat = oidx.classify(bases, offsets, 60)
    dp_regs, dp_reg_cnt, dp_cig = [], [], []
    for r in range(len(reads)):
        g, cigs = oidx.map_cigar(bases[offsets[r]:offsets[r + 1]].tobytes())
        dp_regs.append(g), dp_reg_cnt.append(len(g))
        for c in cigs:
            dp_cig.extend(l << 4 | "MID".index(op) for l, op in c)
    # ---- chain level (base-level alignment off): every stage up to the chain-level regions
    oidx.opt.cigar = 0
    assign, best, nhits, flat = oidx.classify(bases, offsets, 60)
    raw = bases.tobytes()
    mz_cnt, an_cnt, reg_cnt, regs, mz_first = [], [], [], [], []
    cf, cp, cv = [], [], []
    for r in range(len(reads)):
        s = raw[offsets[r]:offsets[r + 1]]
        m = pyoracle.sketch(s)
        a, _ = oidx.seeds(s)
        g = oidx.map(s)
        _, f, p, v, _, _ = oidx.chain(s)
        cf.append(f), cp.append(p), cv.append(v)
        mz_cnt.append(len(m)), an_cnt.append(len(a)), reg_cnt.append(len(g)), regs.append(g)
        mz_first.append(int(m["x"][0]) if len(m) else 0)
    ih, iy = oidx.dump()
    np.savez_compressed(
        os.path.join(HERE, "small_case.npz"),
        names=np.array(names), genome_bytes=np.concatenate(seqs), genome_lens=np.array([len(s) for s in seqs]),
        bases=bases, offsets=offsets, truth=np.concatenate([truth, np.full(len(reads) - 40, -9, dtype=np.int32)]),
        mid_occ=np.int32(oidx.mid_occ), index_hash_sum=np.uint64(int(ih.sum(dtype=np.uint64))),
        index_y_sum=np.uint64(int(iy.sum(dtype=np.uint64))), n_keys=np.int64(oidx.n_keys), n_occ=np.int64(oidx.n_minimizers),
        assign=assign, best=best, nhits=nhits, hits=flat,
        mz_cnt=np.array(mz_cnt), an_cnt=np.array(an_cnt), reg_cnt=np.array(reg_cnt), mz_first=np.array(mz_first, dtype=np.uint64),
        dp_assign=dp_assign, dp_best=dp_best, dp_nhits=dp_nhits, dp_hits=dp_flat, dp_reg_cnt=np.array(dp_reg_cnt),
        dp_regs=np.concatenate(dp_regs), dp_cigars=np.array(dp_cig, dtype=np.uint32),
        chain_f=np.concatenate(cf), chain_p=np.concatenate(cp), chain_v=np.concatenate(cv),
        regs=np.concatenate(regs) if regs else np.zeros(0, dtype=pyoracle.REG_DTYPE))
    print("wrote small_case.npz:", len(reads), "reads,", int((assign >= 0).sum()), "classified at the chain level,",
          int((dp_assign >= 0).sum()), "with base-level alignment")


if __name__ == "__main__":
    main()
