set -o pipefail
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_intra.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
  for lib in libovvc_hip_base.so libovvc_hip.so; do
    echo "== $lib"; OVVC_HIP_LIB_NAME=$lib python tools/debug/ipic_time.py 2>&1 | grep "rep [345]"
  done
done
for lib in libovvc_hip_base.so libovvc_hip.so; do echo "== kbench $lib"; OVVC_HIP_LIB_NAME=$lib python tools/kbench.py --no-check --reps 20 2>&1 | tail -15; done
} > gpurun_out/ab_ipic.log 2>&1
tail -60 gpurun_out/ab_ipic.log
