#!/bin/bash
# the driver's own command, timed, with the fields that matter printed
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
SECONDS=0; python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
echo "wall $SECONDS s"; tail -2 gpurun_out/final_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final_bench.json").read().strip().splitlines()[-1]); c = d["config"]
print("value", d["value"], "ms/step", d["ms_per_step"], "steps", c["step_fps"]["min"], c["step_fps"]["median"], c["step_fps"]["max"])
print("none", c["variants"]["output_none"], "in-order", c["variants"]["in_order_no_lookahead"], "resident", c["resident_replay_fps"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], "ref stream differ", c["reference_stream"]["samples_differing_from_the_reference"],
      "check", c["check"]["differ"], c["check"]["differ_in_flight"], "second passes", c["ordered_pass_second_passes"])
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["largest_streaming_kernel"]["kernel"], d["roofline"]["largest_streaming_kernel"]["frac"])
PY
