# SQ instruction counters of the compute-bound kernels (separate --pmc passes, as the guide asks)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/sq_$tag -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $R/gpurun_out/sq_$tag.log 2>&1
done
