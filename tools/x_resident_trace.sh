#!/bin/bash
# kernel trace of the timed configuration with and without the per-picture copies (bench.py --debug-resident): do the kernels run
# longer next to the copies, or does the stream wait between them?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for m in normal resident; do
  extra=""; [ $m = resident ] && extra="--debug-resident"
  rm -rf $O/prof_x_$m
  rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_x_$m -o x -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 $extra > $O/x_$m.log 2>&1
  db=$(find $O/prof_x_$m -name "*_results.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/x_trace_$m.txt 2>&1
  [ -n "$db" ] && python $R/tools/trace_gaps.py $db > $O/x_gaps_$m.txt 2>&1; cat $O/x_gaps_$m.txt
  tail -1 $O/x_$m.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['value'], d['config']['step_fps'])"
  rm -rf $O/prof_x_$m
done
