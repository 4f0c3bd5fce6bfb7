#!/bin/bash
# three short runs of the default configuration (box-to-box and run-to-run spread is +-5 %): bash tools/bench3.sh [extra bench.py flags]
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  python $R/bench.py --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('fps', d['value'], 'step min/median/max', c['step_fps']['min'], c['step_fps']['median'], c['step_fps']['max'], '2nd', c['ordered_pass_second_passes'], c['frame_thread_host_us_per_picture'])"
done
