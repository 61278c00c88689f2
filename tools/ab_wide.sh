# The literal kernel's form for few calls: eight waves x two cells a thread (the build) against sixteen waves x one cell
# (an alternative library built here from the same sources: -DMNC_WIDE_NW=16 -DMNC_WIDE_C=1), on the workloads where the
# long calls matter.  gpurun -- 'bash tools/ab_wide.sh'
R=$GRAFT_REPO_ROOT
ALT=$R/gpurun_out/libmonica_amd_16x1.so
rm -rf /tmp/alt && mkdir -p /tmp/alt/monica_amd && cp -r $R/monica_amd/csrc /tmp/alt/monica_amd/csrc && cp -r $R/include /tmp/alt/include
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-result -DMNC_WIDE_NW=16 -DMNC_WIDE_C=1"
make -C /tmp/alt/monica_amd/csrc -j16 HIPFLAGS="$FLAGS" LIB=$ALT > /tmp/alt/build.log 2>&1 || { tail -5 /tmp/alt/build.log; exit 1; }
for lib in "" "$ALT"; do
  echo "== ${lib:-the build (8 waves x 2 cells)}"
  MONICA_AMD_LIB=$lib python $R/tools/wg_probe.py 1500 64 0 2>&1 | grep mode
  MONICA_AMD_LIB=$lib python $R/tools/shard_block_profile.py 2>&1 | grep "call ms"
  MONICA_AMD_LIB=$lib python $R/tools/err_profile.py 500 400 400 30000 2>&1 | grep "^error"
  MONICA_AMD_LIB=$lib python $R/bench.py --mode stream --stream-seconds 600 2>/dev/null | tail -1 | cut -c1-160
done
