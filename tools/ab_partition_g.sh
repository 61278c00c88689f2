# the grouped partition kernel with two or four tiles a workgroup at 1 024 table regions (chain-level bench at 62 / 120 genomes)
cd $GRAFT_REPO_ROOT
for g in 62 120; do
  for G in 0 2; do
    MNC_PARTITION_G=$G python bench.py --genomes $g --contract chain --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/abg_g${g}_G$G.json 2> gpurun_out/abg_g${g}_G$G.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/abg_g${g}_G$G.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print("genomes $g G=$G (0: four tiles) reads/s %.0f partition %.3f probe %.3f collect %.3f" % (d["value"], s["partition"], s["probe"], s["collect"]))
PY
  done
done
