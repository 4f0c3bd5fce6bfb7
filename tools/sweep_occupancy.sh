#!/bin/bash
# Rebuilds one kernel file per configuration with -DOV_WPE_<TAG>=n and prints the survey launch times.
# Usage (on the GPU box): tools/sweep_occupancy.sh "MCX:kernels_mcx:0 4 5" "DBF:kernels_dbf:0 4 5" ...
cd "$(dirname "$0")/.."
for spec in "$@"; do
  IFS=: read tag file vals <<<"$spec"
  for v in $vals; do
    rm -f openvvc_amd/csrc/build/$file.o
    make -s -C openvvc_amd/csrc EXTRA="-DOV_WPE_$tag=$v" >/dev/null 2>&1 || { echo "$tag=$v build failed"; continue; }
    python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['roofline']['survey_launch_us']
print('$tag=$v', d['value'], {k: s[k] for k in s})"
  done
  rm -f openvvc_amd/csrc/build/$file.o
  make -s -C openvvc_amd/csrc >/dev/null 2>&1
done
