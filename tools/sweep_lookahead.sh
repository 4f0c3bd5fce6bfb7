run() { # label, env..., args
  label=$1; shift
  out=$(env "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], 'second_passes', d['config']['ordered_pass_second_passes'])" 2>&1)
  echo "$label: $out"
}
B="python bench.py --no-isolated-survey --no-cpu-baseline --check 0 --steps 10 --warmup 2 --output none"
run "cap 2 levels   " OVHIP_FLOW_RESIDENT=2 $B
run "cap 3 levels   " OVHIP_FLOW_RESIDENT=3 $B
run "cap 4 levels   " OVHIP_FLOW_RESIDENT=4 $B
run "cap 4 levels b " OVHIP_FLOW_RESIDENT=4 $B
run "cap 6 levels   " OVHIP_FLOW_RESIDENT=6 $B
run "cap 10 levels  " OVHIP_FLOW_RESIDENT=10 $B
run "cap 16 levels  " OVHIP_FLOW_RESIDENT=16 $B
