#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for m in normal; do
  extra=""; [ $m = resident ] && extra="--debug-resident"
  python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 --trace $R/gpurun_out/trace_$m.npy $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['config']['step_fps']['median'], d['config']['frame_thread_host_us_per_picture'])"
  python $R/tools/debug/dep_latency.py $R/gpurun_out/trace_$m.npy
done
