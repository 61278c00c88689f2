# the literal kernel's long-call forms one kernel at a time (debug 0x10000): 16 % errors, 30 000 reads
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dbg in 0x10000 0x10080 0x10008 0x10001; do
  echo "== debug $dbg"
  rm -rf $R/gpurun_out/prof_s
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_s -o out --output-format csv -- python3 $R/tools/err_profile.py 700 450 450 30000 $dbg > $R/gpurun_out/prof_s.log 2>&1
  tail -3 $R/gpurun_out/prof_s.log | head -1
  python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_s/**/out_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
plans = sorted(int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("mnc::mnc_dp_plan("))
ts = plans[-2] if len(plans) > 1 else plans[-1]
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= ts and ("dp_align" in r["Kernel_Name"] or "dp_ext<64, 8" in r["Kernel_Name"] or "dp_fill<64, 4" in r["Kernel_Name"]) and e - s > 100000:
        print(f'{(s-ts)/1e6:9.3f} {(e-s)/1e6:9.3f} {r["Kernel_Name"][:40]}')
PY
done
