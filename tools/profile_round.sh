#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01h
# -> gpurun_out/<tag>_{trace,fetch,write,sq1,sq2}.txt + bench json.  Counter passes are separate runs, never combined with traces.
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cmd="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-isolated-survey"   # every launch in the timed configuration: the averages below are of that configuration
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_trace -o ${tag} -- $cmd > $O/${tag}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/prof_${tag}_f -o ${tag} -- $cmd > $O/${tag}_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/prof_${tag}_w -o ${tag} -- $cmd > $O/${tag}_w.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY -d $O/prof_${tag}_sq1 -o ${tag} -- $cmd > $O/${tag}_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/prof_${tag}_sq2 -o ${tag} -- $cmd > $O/${tag}_sq2.log 2>&1
for k in trace f w sq1 sq2; do
  db=$(find $O/prof_${tag}_$k -name "*_results.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/${tag}_$k.txt 2>&1
done
python $R/bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
rm -rf $O/prof_${tag}_*
ls -la $O | tail -12
