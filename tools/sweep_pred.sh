#!/bin/bash
# planner thresholds of the banded tiers: ms per step for a grid of (32-cell, 42-cell) thresholds
for p in 32 33 34 35; do for m in 33 34 35 36; do
  MNC_FILL_PRED=$p MNC_FILL_PRED_MID=$m python bench.py --cpu-sample 0 --steps 6 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
k = d['roofline']['kernels_one_at_a_time_ms']
c = d['batch_counters']
print('pred', $p, 'mid', $m, 'ms_per_step', d['ms_per_step'], 'window', d['stage_ms_per_step'].get('dp_fill'), {x: round(k[x], 2) for x in ('dp_fill_t1', 'dp_fill_tm', 'dp_fill_t2', 'dp_fill_t3')}, c['dp_fill_tier1'], c['dp_fill_tier_mid'], c['dp_fill_tier2'])
"
done; done
