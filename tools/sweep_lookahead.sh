run() { # label, env..., args
  label=$1; shift
  out=$(env "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], 'second_passes', d['config']['ordered_pass_second_passes'], 'dpb_alloc', d['config']['dpb']['device_pictures_allocated'])" 2>&1)
  echo "$label: $out"
}
B="python bench.py --no-isolated-survey --no-cpu-baseline --check 0 --steps 10 --warmup 2 --output none"
run "cap 4, paced 16384 (a)      " OVHIP_FLOW_RESIDENT=4 $B --ahead-chunk 16384
run "cap 4, paced 16384 (b)      " OVHIP_FLOW_RESIDENT=4 $B --ahead-chunk 16384
run "cap 4, paced 12288          " OVHIP_FLOW_RESIDENT=4 $B --ahead-chunk 12288
run "cap 4, paced 24576          " OVHIP_FLOW_RESIDENT=4 $B --ahead-chunk 24576
run "cap 4, paced 32768          " OVHIP_FLOW_RESIDENT=4 $B --ahead-chunk 32768
run "cap 3, paced 16384          " OVHIP_FLOW_RESIDENT=3 $B --ahead-chunk 16384
run "cap 6, paced 16384          " OVHIP_FLOW_RESIDENT=6 $B --ahead-chunk 16384
run "cap 8, paced 16384          " OVHIP_FLOW_RESIDENT=8 $B --ahead-chunk 16384
run "cap 4, paced 16384, 12 thr  " OVHIP_FLOW_RESIDENT=4 $B --ahead-chunk 16384 --in-flight 12
run "cap 4, paced 16384, 24 thr  " OVHIP_FLOW_RESIDENT=4 $B --ahead-chunk 16384 --in-flight 24
run "cap 4, paced 16384, digest  " OVHIP_FLOW_RESIDENT=4 python bench.py --no-isolated-survey --no-cpu-baseline --check 0 --steps 10 --warmup 2 --output digest --ahead-chunk 16384
