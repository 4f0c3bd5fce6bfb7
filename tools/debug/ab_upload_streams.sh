#!/bin/bash
# A / B of the shared upload streams on the headline stream (through gpurun): OVVC_HIP_UPLOAD_STREAMS = 0 (a picture's own stream) / N
mkdir -p gpurun_out
reps=${1:-8}; shift; ns=${*:-0 2}
{
for i in $(seq $reps); do
  for n in $ns; do
    OVVC_HIP_UPLOAD_STREAMS=$n python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-stream --no-live-decoder --no-isolated-survey --check 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('upload streams $n:', d['value'], 'h2d GB/s', c.get('h2d_GBps_in_the_timed_region'))"
  done
done
} > gpurun_out/ab_upload_streams.log 2>&1
python - <<'PY'
import re, statistics as st
v={}
for l in open('gpurun_out/ab_upload_streams.log'):
    m=re.match(r'upload streams (\d+): ([\d.]+)', l)
    if m: v.setdefault(int(m.group(1)),[]).append(float(m.group(2)))
for k in sorted(v): print(k, 'n', len(v[k]), 'mean %.0f' % st.mean(v[k]), 'median %.0f' % st.median(v[k]), 'sd %.0f' % (st.stdev(v[k]) if len(v[k])>1 else 0), [round(x) for x in v[k]])
PY
