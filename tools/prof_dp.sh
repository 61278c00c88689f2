cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
[ -n "${SERIAL:-}" ] && export MNC_DP_SERIAL=1
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R && python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof_$TAG/**/out_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:60]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    agg.setdefault(k, []).append(d)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f"{k:60s} n={len(v):4d} total={sum(v):9.2f} ms avg={sum(v)/len(v):8.3f} max={max(v):8.3f}")
# the three launches of mnc_dp_align per batch, in order
al = [ (int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows if r["Kernel_Name"].startswith("mnc::mnc_dp_align") or "mnc_dp_align" in r["Kernel_Name"]]
al.sort()
print("dp_align launches (ms):", [round(d, 2) for _, d in al[-9:]])
fl = [ (int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"][:40]) for r in rows if "mnc_dp_fill" in r["Kernel_Name"]]
fl.sort()
print("dp_fill launches (ms):", [(round(d, 2), n) for _, d, n in fl[-6:]])
PY
