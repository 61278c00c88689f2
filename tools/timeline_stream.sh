# kernel timeline of one 400-read micro-batch (bench.py --mode stream), last batch of a short run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_stream -o out --output-format csv -- python3 $R/bench.py --mode stream --stream-seconds 20 > $R/gpurun_out/prof_stream.log 2>&1
cd $R && python3 tools/timeline.py stream 0.02
