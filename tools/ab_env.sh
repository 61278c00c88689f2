# the default bench under different settings of one environment variable: tools/ab_env.sh NAME v1 v2 ... (stage times per step)
cd $GRAFT_REPO_ROOT
N=$1; shift
for v in "$@"; do
  env $N=$v python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/abe_$v.json 2> gpurun_out/abe_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/abe_$v.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print("$N=$v ms/step %.3f plan %.3f window %.3f stitch %.3f checksum %s" % (d["ms_per_step"], s["dp_plan"], s["dp_fill"], s["dp_stitch"], d.get("counts_checksum")))
PY
done
