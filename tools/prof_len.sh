# kernel durations of tools/len_profile.py (a batch with a long-tailed read length distribution)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_len
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_len -o out --output-format csv -- python3 $R/tools/len_profile.py ${1:-20000} > $R/gpurun_out/prof_len.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("gpurun_out/prof_len/**/out_kernel_trace.csv", recursive=True)[0])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    agg.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:22]:
    print(f"{k:60s} n={len(v):4d} total={sum(v):9.2f} ms max={max(v):8.3f}")
PY
