# Probe STAGE (partition + probe + collect) against the number of table regions at three index sizes
# (gpurun -- 'bash tools/sweep_regions.sh > gpurun_out/r03_region_sweep.txt')
for g in 20 62 120; do
  for b in 8 9 10; do
    MNC_REGION_BITS=$b timeout 300 python3 bench.py --genomes $g --steps 10 --warmup 2 --cpu-sample 0 --contract chain 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('genomes $g regions', 1<<$b, 'step ms', d['ms_per_step'], 'partition', s['partition'], 'probe', s['probe'], 'collect', s['collect'], 'stage', round(s['partition']+s['probe']+s['collect'],3))"
  done
done
