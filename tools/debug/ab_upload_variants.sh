#!/bin/bash
# headline + variants (frames leaving the device, in order without look-ahead, recorder in the run) with the uploads on a picture's own stream (0) / on N shared streams
mkdir -p gpurun_out
reps=${1:-3}; shift; ns=${*:-0 2}
{
for i in $(seq $reps); do
  for n in $ns; do
    OVVC_HIP_UPLOAD_STREAMS=$n python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-stream --no-live-decoder --check 0 --record-threads 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; v=c['variants']
print('upload streams $n: headline', d['value'], 'none', v.get('output_none'), 'frame', v.get('output_frame'), 'in_order', v.get('in_order_no_lookahead'), 'resident', c.get('resident_replay_fps'), 'recorded16', v['recorded_in_run']['by_threads']['16']['fps'])"
  done
done
} > gpurun_out/ab_upload_variants.log 2>&1
cat gpurun_out/ab_upload_variants.log
