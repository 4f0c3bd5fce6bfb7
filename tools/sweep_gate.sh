#!/bin/bash
# frame threads (pictures held) x execution slots (pictures running), interleaved: bash tools/sweep_gate.sh "16 0" "24 12" "32 12" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
CFGS=("$@")
for r in 1 2; do
  for cfg in "${CFGS[@]}"; do
    set -- $cfg
    python $R/bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 --in-flight $1 --exec-slots $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('held $1 running $2: fps', d['value'], 'median step', d['config']['step_fps']['median'], 'second passes', d['config']['ordered_pass_second_passes'])"
  done
done
