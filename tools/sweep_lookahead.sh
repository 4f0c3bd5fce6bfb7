run() { # label, env..., args
  label=$1; shift
  out=$(env "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], 'second_passes', d['config']['ordered_pass_second_passes'], d['config']['lookahead_thread_hw_queue'])" 2>&1)
  echo "$label: $out"
}
B="python bench.py --no-isolated-survey --no-cpu-baseline --check 0 --steps 10 --warmup 2 --output none"
run "shared queue, paced 16384        " $B --ahead-own-queue 0
run "own queue, paced 16384           " $B
run "own queue, one launch per 32768  " $B --ahead-chunk 0
run "own queue, unpaced, HWQ 5        " GPU_MAX_HW_QUEUES=5 $B --ahead-chunk 0
run "own queue, unpaced, HWQ 6        " GPU_MAX_HW_QUEUES=6 $B --ahead-chunk 0
run "own queue, unpaced, no cap       " OVHIP_FLOW_RESIDENT=0 $B --ahead-chunk 0
run "own queue, unpaced, cap 8        " OVHIP_FLOW_RESIDENT=8 $B --ahead-chunk 0
