#!/bin/bash
# stream priorities by place in the dependency graph: bash tools/sweep_prio.sh "<readers> <leaf_low> <in-flight> <exec-slots>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
CFGS=("$@")
for r in 1 2; do
  for cfg in "${CFGS[@]}"; do
    set -- $cfg
    python $R/bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 --priority-readers $1 --leaf-low $2 --in-flight $3 --exec-slots $4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('readers>=$1 leaf_low $2 held $3 slots $4: fps', d['value'], 'median step', d['config']['step_fps']['median'], 'second passes', d['config']['ordered_pass_second_passes'])"
  done
done
