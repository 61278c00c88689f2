# kernel timeline of one config-4 block (tools/shard_block_profile.py) -- which kernels the alignment window waits for
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_shardblk
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_shardblk -o out --output-format csv -- python3 $R/tools/shard_block_profile.py > $R/gpurun_out/shardblk.log 2>&1
cd $R && python3 tools/timeline.py shardblk 0.4
