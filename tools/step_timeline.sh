# kernel timeline of one timed step of the default bench (start / end relative to the first gap-filling kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/tl
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tl -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $R/gpurun_out/tl.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/tl/**/out_kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'mnc_dp_fillp<16>' in r['Kernel_Name']]
i0 = idx[2]
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0 - 4:i0 + 30]:
    s = (int(r['Start_Timestamp']) - t0) / 1e6; e = (int(r['End_Timestamp']) - t0) / 1e6
    n = r['Kernel_Name'].split('(')[0].replace('void mnc::', '').replace('mnc::', '')
    if e - s > 0.05: print("%-34s %8.2f -> %8.2f  (%6.2f)  q%s" % (n[:34], s, e, e - s, r['Queue_Id']))
PY
