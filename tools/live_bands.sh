#!/bin/bash
# live decode at 4K with band-wise submission on / off: tools/live_bands.sh [pics] [threads] [gop] [extra gen_pipe words]   (run through gpurun)
pics=${1:-65}; thr=${2:-16}; gop=${3:-32}; shift 3 2>/dev/null
for exe in oracle/_ref/patched/gen_pipe oracle/_ref/gen_pipe; do
  for bands in 0 1 2; do
    for seed in 31337 4242 777 1234 99; do
      out=$(timeout 900 $exe /tmp live threads $thr size 3840 2160 gop $gop pics $pics reps 2 profile bands $bands seed $seed "$@" 2>/tmp/live_bands.err); rc=$?
      if echo "$out" | grep -q "^{"; then echo "== $exe bands $bands seed $seed rc $rc"; echo "$out" | grep "^{"; break; fi
      tail -1 /tmp/live_bands.err
    done
  done
done
