cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_err -o out --output-format csv -- python3 $R/tools/err_profile.py $1 $2 $3 30000 > $R/gpurun_out/prof_err.log 2>&1
cd $R && python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_err/**/out_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    agg.setdefault(k, []).append(d)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"{k:60s} n={len(v):4d} total={sum(v):9.2f} ms max={max(v):8.3f}")
PY
