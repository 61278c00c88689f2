# kernel trace + stats of the default bench and of the same with the alignment kernels one at a time (the two traces of tools/prof_r04.sh alone)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r04w}
OUT=$R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 > $OUT/prof_${TAG}_bench_under_rocprof.json 2> $OUT/prof_$TAG.err
export MNC_DP_SERIAL=1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/serial_$TAG -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/serial_$TAG.json 2> $OUT/serial_$TAG.err
unset MNC_DP_SERIAL
cp $(find $OUT/prof_$TAG -name "out_kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
cp $(find $OUT/serial_$TAG -name "out_kernel_stats.csv" | head -1) $OUT/${TAG}_serial_kernel_stats.csv
head -8 $OUT/${TAG}_serial_kernel_stats.csv | cut -c1-150
