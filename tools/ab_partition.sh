# the grouped partition kernel (G tiles a workgroup at 512 / 1 024 regions) against the one-tile form: parity test, then the
# chain-level bench at 62 and 120 genomes both ways
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "regions or configs or long" 2>&1 | tail -5
for g in 62 120; do
  for t in 0 1; do
    MNC_PARTITION_TILES=$t python bench.py --genomes $g --contract chain --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/abp_g${g}_t$t.json 2> gpurun_out/abp_g${g}_t$t.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/abp_g${g}_t$t.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print("genomes $g one_tile=$t reads/s %.0f partition %.3f probe %.3f collect %.3f stage_frac %.3f" % (d["value"], s["partition"], s["probe"], s["collect"], d["roofline_stage"]["frac"]))
PY
  done
done
