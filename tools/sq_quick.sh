cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d /tmp/p1 -o x -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/p1.log 2>&1
python $R/tools/rocprof_summary.py $(find /tmp/p1 -name "*_results.db" | head -1) | grep "k_alf\|k_sao\|k_mc2"
