#!/bin/bash
# the recorder inside the run (every frame thread replays its picture's call log, then submits): frame threads x execution slots
R=${GRAFT_REPO_ROOT:-/root/repo}
for slots in 0 16 12; do
  python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-reference-stream --check 0 --record-threads 16,32,48,64 --exec-slots $slots 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('exec slots $slots: value', d['value'], {k: (v['fps'], v['record_ms_per_picture']) for k, v in c['variants']['recorded_in_run']['by_threads'].items()})"
done
