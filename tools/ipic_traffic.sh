#!/bin/bash
# I picture's ordered pass alone: time per level (kbench --ipic) and its HBM traffic (FETCH_SIZE / WRITE_SIZE in separate PMC passes)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; tag=${1:-ip}
# per-dispatch FETCH_SIZE of the I picture's pass = (sum over the run - what the B picture's dispatches take) is what DESIGN quotes; the
# summary below is the SUM over all k_intra_flow dispatches of the run (the B picture's 15 and the I picture's 11)
python $R/tools/kbench.py --ipic --reps 5 2>&1 | grep -E "I picture|bit-exact|intra" | tee $O/${tag}_ipic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pp_$c; rocprofv3 --pmc $c -d /tmp/pp_$c -o x -- python $R/tools/kbench.py --ipic --no-check --reps 1 > /tmp/pp_$c.log 2>&1
  python $R/tools/rocprof_summary.py $(find /tmp/pp_$c -name "*_results.db" | head -1) 2>/dev/null | grep -E "k_intra_flow\b|k_intra_flow\(" | head -3 | cut -c1-120 | tee -a $O/${tag}_ipic.txt
done
