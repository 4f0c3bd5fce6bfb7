#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for q in 4 8 6 12; do
  GPU_MAX_HW_QUEUES=$q python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('queues $q: fps', d['value'], 'median', c['step_fps']['median'], '2nd', c['ordered_pass_second_passes'], c['lookahead_thread_hw_queue'])"
done
done
