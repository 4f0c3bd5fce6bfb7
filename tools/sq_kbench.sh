#!/bin/bash
# SQ counters of the picture's kernels over tools/kbench.py (one 4K B picture in flight): instruction mix per launch
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; tag=${1:-sq}
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES -d /tmp/p_$tag -o x -- python $R/tools/kbench.py --no-check --reps 5 > /tmp/p_$tag.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/p_$tag -name "*_results.db" | head -1) > $O/${tag}_pmc_sq.txt 2>&1
grep -E "k_itx|k_alf|k_sao|k_mc2|k_mcxa|k_dbf|k_intra" $O/${tag}_pmc_sq.txt | cut -c1-200 | head -40
