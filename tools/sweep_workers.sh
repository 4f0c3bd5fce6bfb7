# ordered pass as W persistent workers (OVHIP_FLOW_WORKERS; 0 = one workgroup per item): the B picture's pass and the I picture's alone, then the stream
python -m pytest tests/test_gpu_intra.py -x -q 2>&1 | tail -1
for v in ${WORKERS_LIST:-0 512 1024 2048 4096}; do
  export OVHIP_FLOW_WORKERS=$v
  echo "== OVHIP_FLOW_WORKERS=$v"
  python tools/kbench.py --ipic --no-check 2>&1 | grep -E "  intra|I picture"
  for r in 1 2; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-isolated-survey --check 0 --output none 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stream', d['value'], d['config']['ordered_pass_second_passes'])"
  done
done
