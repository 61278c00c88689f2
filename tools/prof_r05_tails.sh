# Round 5 (as round 4's prof_r04_tails.sh), the literal kernel's long calls: 13 / 16 % errors, one config-4 block, kernel stats of each (max launch durations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
summ() {
python3 - "$1" <<PY
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/out_kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:70]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    agg.setdefault(k, []).append(d)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{k:70s} n={len(v):4d} total={sum(v):9.2f} ms max={max(v):8.3f}")
PY
}
for e in "500 400 400" "700 450 450"; do
  echo "== errors $e"
  python3 $R/tools/err_profile.py $e 30000 2>&1 | tail -3
  rm -rf $R/gpurun_out/prof_err
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_err -o out --output-format csv -- python3 $R/tools/err_profile.py $e 30000 > $R/gpurun_out/prof_err.log 2>&1
  summ $R/gpurun_out/prof_err
done
echo "== config 4 block"
python3 $R/tools/shard_block_profile.py 2>&1 | tail -3
rm -rf $R/gpurun_out/prof_blk
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_blk -o out --output-format csv -- python3 $R/tools/shard_block_profile.py > $R/gpurun_out/prof_blk.log 2>&1
summ $R/gpurun_out/prof_blk
