for v in 2 3 4; do
  touch openvvc_amd/csrc/kernels_intra.hip; make -C openvvc_amd/csrc -j16 EXTRA="-DFLOW_POLLS=$v" > /dev/null 2>&1 || { echo build failed; continue; }
  echo "== polls $v"; python -m pytest tests/test_gpu_intra.py -x -q 2>&1 | tail -1; python tools/kbench.py --ipic --no-check 2>&1 | grep -E "  intra|I picture"
done
