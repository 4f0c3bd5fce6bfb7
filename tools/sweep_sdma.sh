for v in base wait base wait; do
  if [ $v = wait ]; then export OVHIP_EXP_H2D_HOSTWAIT=1; else unset OVHIP_EXP_H2D_HOSTWAIT; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-isolated-survey --check 0 --output none > gpurun_out/exp_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/exp_$v.json')); print('$v', d['value'])"
done
