R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd $R
for sh in 0 1 2; do for g in 62 120; do MNC_PROBE_SHAPE=$sh timeout 300 python3 bench.py --genomes $g --steps 10 --warmup 2 --cpu-sample 0 --contract chain 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('shape $sh genomes $g step ms', d['ms_per_step'], 'partition', s['partition'], 'probe', s['probe'], 'collect', s['collect'], 'stage', round(s['partition']+s['probe']+s['collect'],3), d['roofline_probe']['frac'])" >> $OUT/r03n_shapes.txt; done; done
