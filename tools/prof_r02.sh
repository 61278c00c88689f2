# Round-2 profile: kernel trace of the default bench (base-level alignment on), PMC passes of the
# chain-level stages (the HBM-bound kernels), summaries into profiles/.
#   gpurun -- 'bash tools/prof_r02.sh r02d'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=$R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 > $OUT/prof_${TAG}_bench_under_rocprof.json 2> $OUT/prof_$TAG.err
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${TAG}_$tag -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --contract chain > $OUT/pmc_${TAG}_$tag.log 2>&1
done
# the alignment kernels (DP contract), one kernel at a time on the chip: HBM bytes and vector instructions
export MNC_DP_SERIAL=1
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${TAG}_dp_$tag -o out --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 > $OUT/pmc_${TAG}_dp_$tag.log 2>&1
done
unset MNC_DP_SERIAL
timeout 200 python3 $R/bench.py --mode stream > $OUT/stream_$TAG.json 2> $OUT/stream_$TAG.err
cd $R && python3 tools/summarise_pmc.py $TAG && timeout 400 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 600 gpurun_out/bench_$TAG.json
