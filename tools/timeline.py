"""Print the kernels of the last batch of a rocprofv3 kernel trace as a timeline (ms from the batch's first kernel)."""
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/out_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last batch: from the last mnc_pack_bases on
idx = max(i for i, r in enumerate(rows) if "mnc_pack_bases" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    if e - s > float(sys.argv[2] if len(sys.argv) > 2 else 0.15):
        print(f"{s:8.2f} {e:8.2f} {e - s:7.2f}  {r['Kernel_Name'].split('(')[0][:60]}  grid={r.get('Grid_Size','')} q={r.get('Queue_Id','')}")
