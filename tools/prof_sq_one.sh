# one --pmc pass of the bench with the alignment kernels one at a time: per-kernel averages of the given counters
#   gpurun -- 'bash tools/prof_sq_one.sh "SQ_INSTS_VALU SQ_INSTS_SALU" stitch'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MNC_DP_SERIAL=1
rm -rf $R/gpurun_out/sq_one
timeout 300 rocprofv3 --kernel-trace --pmc $1 -d $R/gpurun_out/sq_one -o out --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 > $R/gpurun_out/sq_one.log 2>&1
cd $R && python3 - "$2" <<'PY'
import csv, glob, collections, sys
per = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/sq_one/**/out_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])
        per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (_, k), cs in per.items():
    for c, v in cs.items():
        agg[k][c].append(v)
for k, cs in agg.items():
    if sys.argv[1] in k:
        print(k[:50], {c: "%.4g x%d" % (sum(v) / len(v), len(v)) for c, v in cs.items()})
PY
