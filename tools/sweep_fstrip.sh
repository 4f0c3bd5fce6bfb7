# ordered pass: samples per item (FSTRIP) -- the 4K I picture alone and the B picture's intra stage, then the stream
for v in 256 128 64; do
  touch openvvc_amd/csrc/kernels_intra.hip; make -C openvvc_amd/csrc -j16 EXTRA="-DFSTRIP=$v" > /dev/null 2>&1 || { echo build failed $v; continue; }
  echo "== FSTRIP=$v"
  python -m pytest tests/test_gpu_intra.py -x -q 2>&1 | tail -1
  python tools/kbench.py --ipic --no-check 2>&1 | grep -E "intra|I picture|level"
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-isolated-survey --check 0 --output none 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stream', d['value'], d['config']['ordered_pass_second_passes'])"
done
touch openvvc_amd/csrc/kernels_intra.hip; make -C openvvc_amd/csrc -j16 > /dev/null 2>&1
