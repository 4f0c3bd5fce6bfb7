#!/bin/bash
# experiment (library built with EXTRA=-DOVHIP_TUNING): upload packing and shared upload lanes
#   OVHIP_X_PACK_LIMIT=bytes   arrays up to this size ride in the staging block (one host memcpy each) instead of a DMA of their own
#   OVHIP_X_UPLOAD_STREAMS=n   n shared copy streams per device instead of every picture's own stream
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  env "$@" python $R/bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print(' '.join(sys.argv[1:]), '| fps', d['value'], 'median', c['step_fps']['median'], 'copies', c['h2d_copies_per_step'], 'host us', c['frame_thread_host_us_per_picture']['class_split_and_parameter_block'], c['frame_thread_host_us_per_picture']['enqueue_copies'], '2nd', c['ordered_pass_second_passes'])" "$@"
}
for rep in 1 2; do
  run OVHIP_X_PACK_LIMIT=131072 OVHIP_X_UPLOAD_STREAMS=0
  run OVHIP_X_PACK_LIMIT=1300000 OVHIP_X_UPLOAD_STREAMS=0
  run OVHIP_X_PACK_LIMIT=16000000 OVHIP_X_UPLOAD_STREAMS=0
  run OVHIP_X_PACK_LIMIT=131072 OVHIP_X_UPLOAD_STREAMS=2
  run OVHIP_X_PACK_LIMIT=1300000 OVHIP_X_UPLOAD_STREAMS=2
  run OVHIP_X_PACK_LIMIT=1300000 OVHIP_X_UPLOAD_STREAMS=4
  run OVHIP_X_PACK_LIMIT=16000000 OVHIP_X_UPLOAD_STREAMS=1
  run OVHIP_X_PACK_LIMIT=16000000 OVHIP_X_UPLOAD_STREAMS=2
done
