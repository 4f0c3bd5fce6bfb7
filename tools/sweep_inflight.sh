#!/bin/bash
# frame threads per device (pictures in flight), interleaved repetitions: bash tools/sweep_inflight.sh 12 16 24 32
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for v in "$@"; do
    python $R/bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 --in-flight $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in flight $v: fps', d['value'], 'median step', d['config']['step_fps']['median'], 'second passes', d['config']['ordered_pass_second_passes'])"
  done
done
