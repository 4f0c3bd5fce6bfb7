#!/bin/bash
# stream rate by the number of persistent workers of the I picture's ordered pass (bench.py --ipic-workers)
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in "$@"; do
  python $R/bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 --ipic-workers $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('workers', sys.argv[1], 'fps', d['value'], 'median step', d['config']['step_fps']['median'], 'second passes', d['config']['ordered_pass_second_passes'])" $w
done
