for v in "-DLMCS_NOUNTAG" ""; do
  touch openvvc_amd/csrc/kernels_lmcs.hip; make -C openvvc_amd/csrc -j16 EXTRA="$v" > /dev/null 2>&1 || { echo build failed; continue; }
  echo "== $v"; bash tools/kbench_trace.sh kbv 2>&1 | grep -E "k_lmcs_inverse"
done
