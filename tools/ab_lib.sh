#!/bin/bash
# A / B of two builds of the library on the headline stream:  tools/ab_lib.sh libovvc_hip_x.so [repetitions]   (the other side: libovvc_hip.so)
other=${1:?library file name under openvvc_amd/}; n=${2:-3}
for i in $(seq $n); do
  for lib in libovvc_hip.so $other; do
    OVVC_HIP_LIB_NAME=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-stream --no-live-decoder --no-isolated-survey --check 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], 'in order', d['config']['variants'].get('in_order_no_lookahead'), 'output none', d['config']['variants'].get('output_none'))"
  done
done
