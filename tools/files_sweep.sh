# files mode (bench.py --mode files, second call) against host-thread settings
# (gpurun -- 'bash tools/files_sweep.sh > gpurun_out/r03_files_sweep.txt')
for pol in active passive; do
  for t in 16 12 8 6; do
    OMP_WAIT_POLICY=$pol MNC_IO_THREADS=$t timeout 200 python3 bench.py --mode files 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['second_call']
print('OMP_WAIT_POLICY=$pol MNC_IO_THREADS=$t second call', s['value'], 'reads/s wall', s['wall_s'], s['aligner_phase_s'])"
  done
done
