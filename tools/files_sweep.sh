# files mode (bench.py --mode files, second call) against the write path, the team sizes and the batch size
run() { env "$@" python bench.py --mode files 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); s=d['second_call']; print('$*: second call', s['value'], s['wall_s'], s['aligner_phase_s'])"; }
run X=1
run MNC_ROUTE_TEXT=1
run MONICA_AMD_BATCH_READS=50000
run MONICA_AMD_BATCH_READS=16000
run MONICA_AMD_BATCH_READS=12500
run MONICA_AMD_BATCH_READS=8000
