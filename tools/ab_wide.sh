# the literal kernel's form for few calls: sixteen waves x one cell (the build) against eight waves x two cells (libmonica_amd_82.so)
R=$GRAFT_REPO_ROOT
for lib in "" "$R/monica_amd/libmonica_amd_82.so"; do
  echo "== ${lib:-default (16 x 1)}"
  MONICA_AMD_LIB=$lib python $R/tools/wg_probe.py 1500 64 0 2>&1 | grep mode
  MONICA_AMD_LIB=$lib python $R/tools/shard_block_profile.py 2>&1 | grep "call ms"
  MONICA_AMD_LIB=$lib python $R/tools/err_profile.py 500 400 400 30000 2>&1 | grep "^error"
  MONICA_AMD_LIB=$lib python $R/bench.py --mode stream --stream-seconds 600 2>/dev/null | tail -1 | cut -c1-160
done
