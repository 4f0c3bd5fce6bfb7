# frame threads per device (pictures in flight) with the ordered pass as persistent workers
for v in 12 16 24 32; do
  for r in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-isolated-survey --check 0 --output none --in-flight $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('in flight $v: stream', d['value'], d['config']['ordered_pass_second_passes'])"; done
done
