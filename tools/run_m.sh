cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" > $OUT/r03m_tests.txt
for g in 20 62 120; do timeout 300 python3 bench.py --genomes $g --steps 10 --warmup 2 --cpu-sample 0 --contract chain 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
print('genomes $g step ms', d['ms_per_step'], 'partition', s['partition'], 'probe', s['probe'], 'collect', s['collect'], 'stage', round(s['partition']+s['probe']+s['collect'],3), d['roofline_probe']['frac'])" >> $OUT/r03m_sizes.txt; done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_r03mg62_TCC -o out --output-format csv -- python3 $R/bench.py --genomes 62 --steps 2 --warmup 1 --cpu-sample 0 --contract chain > $OUT/pmc_r03mg62_TCC.log 2>&1
cd $R
bash tools/files_sweep.sh > $OUT/r03m_files_sweep.txt 2>&1
timeout 900 python3 tools/parity_sweep.py 20000 > $OUT/r03m_parity_sweep.txt 2>&1
