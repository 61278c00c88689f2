cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02sq}
export MNC_DP_SERIAL=1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/sq_${TAG}_$tag -o out --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 > $R/gpurun_out/sq_${TAG}_$tag.log 2>&1
done
cd $R && python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_${TAG}_*/**/out_counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])
        per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (_, k), cs in per.items():
        for c, v in cs.items():
            agg[k][c].append(v)
out = {k: {c: {"avg_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in agg.items() if "mnc_dp" in k or "chain_dp" in k}
json.dump(out, open("gpurun_out/${TAG}_sq_counters.json", "w"), indent=1)
for k, cs in out.items():
    print(k[:40], {c: f"{v['avg_per_launch']:.3g}" for c, v in cs.items()})
PY
