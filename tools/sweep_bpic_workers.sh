#!/bin/bash
# stream rate by the number of persistent workers of the B pictures' ordered pass (bench.py --bpic-workers), interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do for w in "$@"; do
  python $R/bench.py --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 --bpic-workers $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('B workers $w: fps', d['value'], 'median step', d['config']['step_fps']['median'], 'second passes', d['config']['ordered_pass_second_passes'])"
done; done
