# vector / scalar instruction counts per kernel of one batch at a given error rate (rocprofv3 --pmc: kernels run one at a time)
# usage: prof_sq_err.sh "<sub> <ins> <del>" <reads> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
E=${1:-"700 450 450"}; N=${2:-30000}; TAG=${3:-r05sq}
rm -rf $R/gpurun_out/sq_$TAG
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU -d $R/gpurun_out/sq_$TAG -o out --output-format csv -- python3 $R/tools/err_profile.py $E $N > $R/gpurun_out/sq_$TAG.log 2>&1
tail -3 $R/gpurun_out/sq_$TAG.log | head -1
cd $R && python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/sq_$TAG/**/out_counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); n[k] += 1
tot = sum(c["SQ_INSTS_VALU"] for c in per.values())
print("two batches (warm-up + timed); VALU instructions in all: %.3g = %.1f ms of issue at 6.144e11/s" % (tot, tot / 6.144e11 * 1e3))
for k, c in sorted(per.items(), key=lambda kc: -kc[1]["SQ_INSTS_VALU"])[:24]:
    print(f"{k:62s} launches {n[k]:3d}  VALU {c['SQ_INSTS_VALU']:.3g} ({c['SQ_INSTS_VALU'] / 6.144e11 * 1e3:7.2f} ms)  SALU {c['SQ_INSTS_SALU']:.3g}")
PY
