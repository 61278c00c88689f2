# plain kernel trace (no counters) of the bench with the alignment kernels one at a time: the per-kernel durations that
# bench.py's roofline.avg_launch_ms (HIP events in the same mode) has to agree with
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
export MNC_DP_SERIAL=1
rm -rf $R/gpurun_out/serial_$TAG
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/serial_$TAG -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $R/gpurun_out/serial_$TAG.json 2> $R/gpurun_out/serial_$TAG.err
cd $R && head -8 gpurun_out/serial_$TAG/out_kernel_stats.csv
