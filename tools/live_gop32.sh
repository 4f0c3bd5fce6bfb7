#!/bin/bash
# live decode of a GOP-32 random-access stream at 4K on the GPU box: tools/live_gop32.sh [pics] [threads]   (run through gpurun)
# (the parse is a random walk: a seed whose stream holds a 64x2 ISP partition -- reference result undefined -- is skipped)
pics=${1:-65}; thr=${2:-8,16,32}
for exe in oracle/_ref/gen_pipe oracle/_ref/patched/gen_pipe; do
  for seed in 31337 4242 777 1234 99; do
    out=$(timeout 900 $exe /tmp live threads $thr size 3840 2160 gop 32 pics $pics reps 2 profile noisp seed $seed 2>/tmp/live_gop32.err); rc=$?
    if echo "$out" | grep -q "^{"; then echo "== $exe seed $seed rc $rc"; echo "$out" | grep "^{"; break; fi
    tail -1 /tmp/live_gop32.err
  done
done
