cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r01k -o out --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $R/gpurun_out/prof_r01k.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmck_$tag -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $R/gpurun_out/pmck_$tag.log 2>&1
done
cd $R && timeout 300 python bench.py > gpurun_out/bench_r01k.json 2> gpurun_out/bench_r01k.err
