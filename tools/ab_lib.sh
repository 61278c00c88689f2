#!/bin/bash
# The library as built against another build of it (MONICA_AMD_LIB), same box: the bench line, batches of other sizes,
# 13 % / 16 % errors, a config-4 block.   bash tools/ab_lib.sh <tag> <other .so>
tag=${1:-ab}
other=$2
out=gpurun_out/${tag}_ab.txt
: > $out
for lib in "" "$other" "" "$other"; do
  echo "== ${lib:-library of the tree}" >> $out
  export MONICA_AMD_LIB=$lib
  [ -z "$lib" ] && unset MONICA_AMD_LIB
  python bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["stage_ms_per_step"]["dp_fill"], d["roofline"]["kernels_one_at_a_time_ms"])' >> $out
  python tools/small_batches.py 3000,12500,30000 2>/dev/null | cut -c1-70 >> $out
  python tools/err_profile.py 500 400 400 30000 2>/dev/null | head -1 >> $out
  python tools/err_profile.py 700 450 450 30000 2>/dev/null | head -1 >> $out
  python tools/shard_block_profile.py 2>/dev/null | grep -E "call ms|one kernel" | cut -c1-300 >> $out
done
cat $out
