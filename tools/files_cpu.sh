#!/bin/bash
# Files mode's second call with the host threads' user and system time, as is and with the thread teams set other ways.
# bash tools/files_cpu.sh <tag>
tag=${1:-r05y}
out=gpurun_out/${tag}_files_cpu.txt
df -T /tmp | tail -1 > $out
nproc >> $out
run() { echo "== $*" >> $out; env "$@" python bench.py --mode files 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d["second_call"]))' >> $out; }
run X=1
run X=1
run X=1
run OMP_WAIT_POLICY=passive
run OMP_WAIT_POLICY=passive
run MNC_PARSE_THREADS=8 MNC_ROUTE_THREADS=8
run MNC_PARSE_THREADS=6 MNC_ROUTE_THREADS=10
run MNC_PARSE_THREADS=4 MNC_ROUTE_THREADS=12 OMP_WAIT_POLICY=passive
run GOMP_SPINCOUNT=0
cat $out
