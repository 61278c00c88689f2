# kernel timeline of the last 400-read (or $1-read) batch of tools/small_batches.py: every kernel of 10 us or more
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=${1:-400}
rm -rf $R/gpurun_out/prof_tl
timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o out --output-format csv -- python3 $R/tools/small_batches.py $n > $R/gpurun_out/prof_tl.log 2>&1
tail -1 $R/gpurun_out/prof_tl.log
(cd $R && python3 tools/timeline.py tl 0.01)
