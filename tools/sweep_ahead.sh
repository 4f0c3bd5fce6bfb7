#!/bin/bash
# uploader threads: pictures uploaded ahead of the frame threads (bench.py --upload-ahead N), interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for v in "$@"; do
    python $R/bench.py --steps ${STEPS:-12} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 --upload-ahead $v 2>$R/gpurun_out/ahead_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('upload ahead $v: fps', d['value'], 'median step', c['step_fps']['median'], 'second passes', c['ordered_pass_second_passes'], c['frame_thread_host_us_per_picture'])" || tail -3 $R/gpurun_out/ahead_err.txt
  done
done
