#!/bin/bash
# experiment (library built with EXTRA=-DOVHIP_TUNING): where do a picture's copies cost the stream?  OVHIP_X_H2D_HOSTWAIT=1: the frame
# thread waits for its uploads on the host before it enqueues the launches; OVHIP_X_MV_D2H=0 / 1 / 2: refined vectors not copied
# back / on the picture's stream between two kernels / on a side stream
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  env "$@" python $R/bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' '.join(sys.argv[1:]) or 'base', '| fps', d['value'], 'median step', d['config']['step_fps']['median'], 'second passes', d['config']['ordered_pass_second_passes'])" "$@"
}
for rep in 1 2; do
  run OVHIP_X_NONE=1
  run OVHIP_X_H2D_HOSTWAIT=1
  run OVHIP_X_MV_D2H=0
  run OVHIP_X_MV_D2H=2
  run OVHIP_X_H2D_HOSTWAIT=1 OVHIP_X_MV_D2H=2
done
