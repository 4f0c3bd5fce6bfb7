#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OVHIP_X_MV_D2H=3 python -m pytest $R/tests/test_gpu_pipe.py -x -q 2>&1 | tail -2
run() {
  env "$@" python $R/bench.py --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(' '.join(sys.argv[1:]), '| fps', d['value'], 'median', c['step_fps']['median'], '2nd', c['ordered_pass_second_passes'])" "$@"
}
for rep in 1 2 3; do
  run OVHIP_X_MV_D2H=1
  run OVHIP_X_MV_D2H=3
  run OVHIP_X_MV_D2H=0
done
