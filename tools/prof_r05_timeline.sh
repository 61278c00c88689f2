# kernel timelines (last batch) of: 30 000 reads at 13 % / 16 % errors, one config-4 block
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in "500 400 400" "700 450 450"; do
  echo "== errors $e"
  rm -rf $R/gpurun_out/prof_tl
  timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o out --output-format csv -- python3 $R/tools/err_profile.py $e 30000 > $R/gpurun_out/prof_tl.log 2>&1
  tail -3 $R/gpurun_out/prof_tl.log | head -1
  (cd $R && python3 tools/timeline.py tl 0.4)
done
echo "== config 4 block"
rm -rf $R/gpurun_out/prof_tl
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_tl -o out --output-format csv -- python3 $R/tools/shard_block_profile.py > $R/gpurun_out/prof_tl.log 2>&1
tail -3 $R/gpurun_out/prof_tl.log | head -1
(cd $R && python3 tools/timeline.py tl 0.4)
