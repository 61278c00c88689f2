# the repeats genome model (synth.genome_set_repeats: operons, insertion sequences, shared stretches) against the i.i.d. one:
# bench line + stage table, error profiles, parity sweep
cd $GRAFT_REPO_ROOT
echo "== parity sweep, i.i.d. genomes (3 / 10 / 16 / 20 % errors)"
python3 tools/parity_sweep.py 20000 iid 20 2>&1 | tail -6
echo "== parity sweep, repeats model"
python3 tools/parity_sweep.py 20000 repeats 20 2>&1 | tail -6
for m in iid repeats; do
  echo "== bench, genome model $m"
  python3 bench.py --genome-model $m --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | tail -1 > gpurun_out/r05_bench_$m.json
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r05_bench_$m.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "reads/s", d["value"], "chain level", d.get("chain_level", {}).get("value"))
print("stages", d["stage_ms_per_step"])
c = d["batch_counters"]
print("counters", {k: c[k] for k in ("minimizers", "probe_hits", "anchors", "chains", "regions", "gated_hits", "ambiguous_reads", "dp_segments", "dp_long_gaps", "dp_long_extensions", "dp_literal_big", "dp_literal_mid") if k in c})
print("workload", d["config"]["workload"], "| mapped", d.get("mapped_reads_last_step"))
print("probe", d.get("roofline_probe", {}).get("frac"), "stage", d.get("roofline_stage", {}).get("frac"), "dominant kernel ms", d.get("roofline", {}).get("kernel_ms"))
PY
done
for m in iid repeats; do
  echo "== error profiles, genome model $m"
  for e in "400 300 300" "500 400 400" "700 450 450"; do MNC_GENOME_MODEL=$m python3 tools/err_profile.py $e 30000 2>&1 | tail -3 | head -2; done
done
