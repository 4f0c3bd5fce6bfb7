#!/bin/bash
# sweep of the staging block's pack limit (library built -DOVHIP_TUNING as libovvc_hip_tuning.so) on the headline stream
mkdir -p gpurun_out
reps=${1:-6}; shift; ls_=${*:-131072 1048576 8388608}
{
for i in $(seq $reps); do
  for n in $ls_; do
    OVVC_HIP_LIB_NAME=libovvc_hip_tuning.so OVHIP_X_PACK_LIMIT=$n python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-stream --no-live-decoder --no-isolated-survey --check 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('pack limit $n:', d['value'], 'copies per step', c.get('h2d_copies_per_step'), 'host us', c['frame_thread_host_us_per_picture'].get('class_split_and_parameter_block'))"
  done
done
} > gpurun_out/ab_pack_limit.log 2>&1
python - <<'PY'
import re, statistics as st
v={}
for l in open('gpurun_out/ab_pack_limit.log'):
    m=re.match(r'pack limit (\d+): ([\d.]+)', l)
    if m: v.setdefault(int(m.group(1)),[]).append(float(m.group(2)))
    elif l.strip(): print(l.strip()[:200])
for k in sorted(v): print(k, 'n', len(v[k]), 'mean %.0f' % st.mean(v[k]), 'sd %.0f' % (st.stdev(v[k]) if len(v[k])>1 else 0), [round(x) for x in v[k]])
PY
tail -3 gpurun_out/ab_pack_limit.log
