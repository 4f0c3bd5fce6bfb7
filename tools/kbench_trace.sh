#!/bin/bash
# rocprofv3 kernel trace of tools/kbench.py: every kernel's average duration with ONE picture in flight and the device to itself
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; tag=${1:-kb}
rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o $tag -- python $R/tools/kbench.py --no-check --reps 20 ${@:2} > $O/${tag}_kbench.log 2>&1
db=$(find $O/prof_$tag -name "*_results.db" | head -1)
[ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/${tag}_isolated_kernel_trace.txt 2>&1
rm -rf $O/prof_$tag
grep -E "^ +[0-9]+ " $O/${tag}_isolated_kernel_trace.txt | head -24 | cut -c1-110
