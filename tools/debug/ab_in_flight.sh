#!/bin/bash
# frame threads (= pictures in flight) of the stream driver on the headline stream
mkdir -p gpurun_out
reps=${1:-4}; shift; ns=${*:-12 16 20 24}
{
for i in $(seq $reps); do
  for n in $ns; do
    python bench.py --in-flight $n --steps 10 --warmup 3 --no-cpu-baseline --no-reference-stream --no-live-decoder --no-isolated-survey --check 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('in flight $n:', d['value'], 'second passes', c.get('ordered_pass_second_passes'))"
  done
done
} > gpurun_out/ab_in_flight.log 2>&1
python - <<'PY'
import re, statistics as st
v={}
for l in open('gpurun_out/ab_in_flight.log'):
    m=re.match(r'in flight (\d+): ([\d.]+)', l)
    if m: v.setdefault(int(m.group(1)),[]).append(float(m.group(2)))
    elif l.strip(): print(l.strip()[:200])
for k in sorted(v): print(k, 'n', len(v[k]), 'mean %.0f' % st.mean(v[k]), 'sd %.0f' % (st.stdev(v[k]) if len(v[k])>1 else 0), [round(x) for x in v[k]])
PY
