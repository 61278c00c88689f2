# Round-5 profile (gpurun -- 'bash tools/prof_r05.sh r05z'): the same passes as round 4's on the round's last build
#   1. kernel trace + stats of the default bench (base-level alignment on)
#   2. the same with the alignment kernels one at a time (the launches roofline.avg_launch_ms is taken from)
#   3. separate --pmc passes: HBM bytes of the chain-level stages at 20, 62 and 120 genomes (the probe kernel at every
#      index size), HBM bytes and vector instructions of the alignment kernels one at a time
#   4. stream mode, files mode, the plain bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
OUT=$R/gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-sample 0 > $OUT/prof_${TAG}_bench_under_rocprof.json 2> $OUT/prof_$TAG.err
export MNC_DP_SERIAL=1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/serial_$TAG -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/serial_$TAG.json 2> $OUT/serial_$TAG.err
unset MNC_DP_SERIAL
for g in 20 62 120; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | tr ' ' '_')
    sfx=$([ $g = 20 ] && echo "" || echo "g$g")
    timeout 400 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${TAG}${sfx}_$tag -o out --output-format csv -- python3 $R/bench.py --genomes $g --steps 2 --warmup 1 --cpu-sample 0 --contract chain > $OUT/pmc_${TAG}${sfx}_$tag.log 2>&1
  done
done
for c in "SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${TAG}_chain_$tag -o out --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --contract chain > $OUT/pmc_${TAG}_chain_$tag.log 2>&1
done
export MNC_DP_SERIAL=1
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${TAG}_dp_$tag -o out --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 > $OUT/pmc_${TAG}_dp_$tag.log 2>&1
done
unset MNC_DP_SERIAL
timeout 200 python3 $R/bench.py --mode stream > $OUT/stream_$TAG.json 2> $OUT/stream_$TAG.err
timeout 200 python3 $R/bench.py --mode files > $OUT/files_$TAG.json 2> $OUT/files_$TAG.err
for g in 62 120; do
  timeout 300 python3 $R/bench.py --genomes $g --steps 10 --warmup 2 --cpu-sample 0 --contract chain > $OUT/bench_${TAG}_g${g}_chain.json 2> /dev/null
done
cd $R && python3 tools/summarise_pmc.py $TAG && python3 tools/summarise_pmc.py ${TAG}g62 && python3 tools/summarise_pmc.py ${TAG}g120
timeout 400 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cp gpurun_out/serial_$TAG/*/out_kernel_stats.csv gpurun_out/serial_${TAG}_kernel_stats.csv 2>/dev/null || cp $(find gpurun_out/serial_$TAG -name "out_kernel_stats.csv" | head -1) gpurun_out/serial_${TAG}_kernel_stats.csv
tail -c 400 gpurun_out/bench_$TAG.json
