#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
for out in none digest; do
for v in 0 16; do
  python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 --output $out --upload-ahead $v --trace $R/gpurun_out/trace_${out}_$v.npy 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('output $out ahead $v:', d['value'], d['config']['step_fps']['median'], d['config']['frame_thread_host_us_per_picture'])"
  python $R/tools/debug/dep_latency.py $R/gpurun_out/trace_${out}_$v.npy | sed -n '2,9p;17,30p'
done
done
