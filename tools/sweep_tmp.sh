for v in 0 1 2 3; do
  export OVHIP_ITX_ABLATE=$v
  echo "== ablate $v"; bash tools/kbench_trace.sh kbv 2>&1 | grep -E "k_itx_all"
done
