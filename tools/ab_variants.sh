# the default bench with experimental builds of the library (variants/lib_<name>.so, same ABI): step, window and the tiers alone
cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ $v = base ]; then unset MONICA_AMD_LIB; else export MONICA_AMD_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so; fi
  python bench.py --steps 10 --warmup 2 --cpu-sample 0 > gpurun_out/abv_$v.json 2> gpurun_out/abv_$v.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/abv_$v.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]; k = d["roofline"]["kernels_one_at_a_time_ms"]
print("$v ms/step %.3f window %.3f stitch %.3f alone %s equal %s" % (d["ms_per_step"], s["dp_fill"], s["dp_stitch"], k, d.get("counts_checksum")))
PY
done
