#!/bin/bash
# Files mode's pipeline under other settings, three timed calls each (tools/files_trace.py).  bash tools/files_variants.sh <tag>
tag=${1:-r05y}
out=gpurun_out/${tag}_files_variants.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/files_trace.py 2>&1 | grep "^call" >> $out; }
run X=1
run X=2
run MNC_ROUTE_THREADS=8
run MONICA_AMD_BATCH_RAMP=0.2,0.5
cat $out
for i in 1 2 3 4 5; do python bench.py --mode files 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d["second_call"]))'; done
VERBOSE=1 python tools/files_trace.py 2>&1 | awk '/^call 3/,0' > gpurun_out/${tag}_files_trace.txt
