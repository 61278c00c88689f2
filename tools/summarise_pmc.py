#!/usr/bin/env python3
"""Average rocprofv3 PMC counters per kernel launch (separate --pmc passes, as the MI355X guide
prescribes) and write profiles/<tag>_pmc_hbm_counters.json + profiles/probe_traffic.json."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}_*")):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "out_counter_collection.csv"), recursive=True):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])
            per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        for (_, k), cs in per.items():
            for c, v in cs.items():
                agg[k][c].append(v)
out = {k: {c: {"avg_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in agg.items()}
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_pmc_hbm_counters.json"), "w"), indent=1)


def hbm(kernel):          # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH doubled (gfx950 correction for coalesced streams)
    k = [x for x in out if x.endswith(kernel) or kernel + "<" in x]
    if not k and kernel == "mnc_partition_queries":        # 512 / 1 024 table regions: the form over several tiles a workgroup
        k = [x for x in out if "mnc_partition_group<" in x]
    if not k:
        return None
    c = out[k[0]]
    return (2 * c["FETCH_SIZE"]["avg_per_launch"] + c["WRITE_SIZE"]["avg_per_launch"]) * 1024


probe = hbm("mnc_probe_buckets")
stage = [hbm(k) for k in ("mnc_partition_queries", "mnc_probe_buckets", "mnc_collect_hits")]
if probe:
    import re
    m = re.search(r"g(\d+)$", tag)
    genomes = int(m.group(1)) if m else 20
    pk = [x for x in out if x.endswith("mnc_probe_buckets") or "mnc_probe_buckets<" in x][0]
    entry = {"kernel": pk, "genomes": genomes, "reads": 100000, "read_len": 5000, "hbm_bytes_per_launch": int(probe),
             "stage_hbm_bytes_per_launch": int(sum(x for x in stage if x)),
             "fetch_size_kib": out[pk]["FETCH_SIZE"]["avg_per_launch"], "write_size_kib": out[pk]["WRITE_SIZE"]["avg_per_launch"],
             "tcc_hit": out[pk].get("TCC_HIT_sum", {}).get("avg_per_launch"), "tcc_miss": out[pk].get("TCC_MISS_sum", {}).get("avg_per_launch"),
             "note": f"{tag}: separate rocprofv3 --pmc passes of bench.py --genomes {genomes} --steps 2 --warmup 1 --contract chain "
                     "(tools/prof_r03.sh); FETCH_SIZE/WRITE_SIZE in KiB; FETCH doubled per the guide's gfx950 correction for "
                     "coalesced streams; stage = partition + probe + collect"}
    path = os.path.join(root, "profiles", "probe_traffic.json")
    try:
        cur = json.load(open(path))
    except Exception:
        cur = {}
    by = cur.get("by_genomes", {})
    by[str(genomes)] = entry
    if genomes == 20:                                     # the default workload's figures stay at the top level
        cur = dict(entry)
    cur["by_genomes"] = by
    json.dump(cur, open(path, "w"), indent=1)
    print(tag, "probe HBM bytes/launch", int(probe), "stage", int(sum(x for x in stage if x)))
# the dominant kernel with base-level alignment: the 32-cell tier of the banded gap-filling kernel
fk = [x for x in out if "mnc_dp_fillp<16" in x]
if fk and "FETCH_SIZE" in out[fk[0]] and "WRITE_SIZE" in out[fk[0]]:
    c = out[fk[0]]
    json.dump({"kernel": fk[0].split("::")[-1], "reads": 100000, "read_len": 5000,
               "hbm_bytes_per_launch": int((2 * c["FETCH_SIZE"]["avg_per_launch"] + c["WRITE_SIZE"]["avg_per_launch"]) * 1024),
               "fetch_size_kib": c["FETCH_SIZE"]["avg_per_launch"], "write_size_kib": c["WRITE_SIZE"]["avg_per_launch"],
               "sq_insts_valu": c.get("SQ_INSTS_VALU", {}).get("avg_per_launch"), "sq_busy_cycles": c.get("SQ_BUSY_CYCLES", {}).get("avg_per_launch"),
               "note": f"{tag}: separate rocprofv3 --pmc passes of bench.py --steps 1 --warmup 1 with MNC_DP_SERIAL=1 (the alignment "
                       "kernels one at a time); FETCH doubled per the guide's gfx950 correction"},
              open(os.path.join(root, "profiles", "fill_traffic.json"), "w"), indent=1)
    print("fill kernel HBM bytes/launch", int((2 * c["FETCH_SIZE"]["avg_per_launch"] + c["WRITE_SIZE"]["avg_per_launch"]) * 1024))
ks = glob.glob(os.path.join(root, "gpurun_out", f"prof_{tag}", "**", "out_kernel_stats.csv"), recursive=True)
if ks:
    import shutil
    shutil.copy(ks[0], os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
b = os.path.join(root, "gpurun_out", f"prof_{tag}_bench_under_rocprof.json")
if os.path.exists(b):
    import shutil
    shutil.copy(b, os.path.join(root, "profiles", f"{tag}_bench_under_rocprof.json"))

# the chain DP kernel's issue counters (round 4: VERDICT r03 item 6)
ck = [x for x in out if "mnc_chain_dp_ring" in x]
if ck and "SQ_INSTS_VALU" in out[ck[0]]:
    c = out[ck[0]]
    g = lambda n: c.get(n, {}).get("avg_per_launch")
    json.dump({"kernel": ck[0].split("::")[-1], "reads": 100000, "read_len": 5000, "sq_insts_valu": g("SQ_INSTS_VALU"), "sq_busy_cycles": g("SQ_BUSY_CYCLES"),
               "sq_active_inst_valu": g("SQ_ACTIVE_INST_VALU"), "sq_wave_cycles": g("SQ_WAVE_CYCLES"), "sq_insts_salu": g("SQ_INSTS_SALU"), "sq_insts_lds": g("SQ_INSTS_LDS"),
               "note": f"{tag}: separate rocprofv3 --pmc passes of bench.py --steps 2 --warmup 1 --contract chain (tools/prof_r04.sh)"},
              open(os.path.join(root, "profiles", f"{tag}_sq_chain_dp.json"), "w"), indent=1)
    print("chain DP SQ:", {n: g(n) for n in ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")})
