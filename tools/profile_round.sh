#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01h
# -> gpurun_out/<tag>_{trace,fetch,write,sq1,sq2}.txt + bench json.  Counter passes are separate runs, never combined with traces.
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cmd="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-reference-stream --no-live-decoder --no-isolated-survey --check 0"   # every launch in the timed configuration: the averages below are of that configuration
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_trace -o ${tag} -- $cmd > $O/${tag}_trace.log 2>&1
# counter passes: rocprofv3's counter service sometimes dies (SIGSEGV inside the launch) when 16 frame threads launch at once;
# a pass is retried, then repeated with fewer frame threads (bytes and instructions per dispatch do not depend on them)
pmc() {   # pmc <suffix> <counters...>
  local k=$1; shift
  for extra in "" "" "--in-flight 4" "--in-flight 1"; do
    rm -rf $O/prof_${tag}_$k
    rocprofv3 --pmc "$@" -d $O/prof_${tag}_$k -o ${tag} -- $cmd $extra > $O/${tag}_$k.log 2>&1 && [ -n "$(find $O/prof_${tag}_$k -name '*_results.db' | head -1)" ] && { echo "pmc $k ok ($extra)"; return; }
    echo "pmc $k failed ($extra), again"
  done
}
pmc f FETCH_SIZE
pmc w WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
for k in trace f w sq1 sq2; do
  db=$(find $O/prof_${tag}_$k -name "*_results.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/${tag}_$k.txt 2>&1
done
# the reference decoder's stream on the device, alone (its kernels' averages are not those of the synthetic stream above)
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_ref -o ${tag} -- python $R/tools/ref_stream_trace.py > $O/${tag}_ref.log 2>&1
db=$(find $O/prof_${tag}_ref -name "*_results.db" | head -1)
[ -n "$db" ] && python $R/tools/rocprof_summary.py $db > $O/${tag}_reference_stream_kernel_trace.txt 2>&1
python $R/bench.py --steps 20 --warmup 5 > $O/${tag}_bench.json 2> $O/${tag}_bench.err
# the names the summaries are committed under (profiles/)
mv $O/${tag}_trace.txt $O/${tag}_full_pipeline_kernel_trace.txt
mv $O/${tag}_f.txt $O/${tag}_pmc_fetch.txt
mv $O/${tag}_w.txt $O/${tag}_pmc_write.txt
cat $O/${tag}_sq1.txt $O/${tag}_sq2.txt > $O/${tag}_pmc_sq.txt && rm -f $O/${tag}_sq1.txt $O/${tag}_sq2.txt
# traffic per kernel for bench.py's roofline.traffic (profiles/traffic.json: copy it there with the summaries)
python $R/tools/traffic_from_pmc.py $O/${tag}_pmc_fetch.txt $O/${tag}_pmc_write.txt 3840 2160 614 "$cmd (tools/profile_round.sh $tag)" $O/${tag}_full_pipeline_kernel_trace.txt > $O/${tag}_traffic.json 2> $O/${tag}_traffic.err
rm -rf $O/prof_${tag}_*
ls -la $O | tail -12
