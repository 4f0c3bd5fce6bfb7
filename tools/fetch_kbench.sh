#!/bin/bash
# HBM traffic counters of the picture's kernels over tools/kbench.py (one 4K B picture in flight): FETCH_SIZE / WRITE_SIZE per launch
# (separate passes; KiB as reported -- FETCH_SIZE is doubled when converted to bytes on gfx950, MI355X_MICROARCH.md)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; tag=${1:-fetch}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_${tag}_$c
  rocprofv3 --pmc $c -d /tmp/p_${tag}_$c -o x -- python $R/tools/kbench.py --no-check --reps 5 > /tmp/p_${tag}_$c.log 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/p_${tag}_$c -name "*_results.db" | head -1) > $O/${tag}_pmc_$c.txt 2>&1
  grep -E "k_itx|k_alf|k_sao|k_mc2|k_mcxa|k_dbf|k_lmcs" $O/${tag}_pmc_$c.txt | cut -c1-160 | head -14
done
