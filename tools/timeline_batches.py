"""The kernels of the slowest and of a median batch of a rocprofv3 kernel trace (a batch = from one mnc_pack_bases to the next):
python tools/timeline_batches.py <prof dir under gpurun_out/prof_*> [min ms]"""
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/**/out_kernel_trace.csv", recursive=True)[0]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "mnc_pack_bases" in r["Kernel_Name"]]
spans = []
for a, b in zip(starts, starts[1:] + [len(rows)]):
    t0 = int(rows[a]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows[a:b])
    spans.append(((t1 - t0) / 1e6, a, b))
spans = spans[5:]                                              # (warm-up)
order = sorted(spans)
for name, (dur, a, b) in (("median", order[len(order) // 2]), ("slowest", order[-1])):
    print(f"== {name} batch: {dur:.2f} ms of kernels from first start to last end")
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
        if e - s > thr:
            print(f"{s:8.2f} {e:8.2f} {e - s:7.2f}  {r['Kernel_Name'].split('(')[0][:64]}")
