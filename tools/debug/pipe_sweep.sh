#!/bin/bash
# build container only (needs oracle/_ref/gen_pipe): chained streams of the reference slice decoder for seeds seed0..seed1, each decoded by the
# oracle chain and compared with the reference frames.  usage: tools/debug/pipe_sweep.sh variant w h seed0 seed1
v=$1; w=$2; h=$3
for s in $(seq $4 $5); do
  d=/tmp/sw_$v_$s; mkdir -p $d
  if ! oracle/_ref/gen_pipe $d seed $s variant $v size $w $h >/dev/null 2>$d/log; then echo "seed $s: ref pass failed: $(tail -1 $d/log)"; continue; fi
  if ! oracle/_ref/gen_pipe $d shim seed $s variant $v size $w $h >/dev/null 2>$d/log; then echo "seed $s: shim pass failed: $(grep -v 'picture\|pass' $d/log | tail -2 | tr '\n' ' ')"; continue; fi
  r=$(python tools/debug/pipe_compare.py $d chain 2>&1 | awk '{ if ($0 !~ /Y 0 \| Cb 0 \| Cr 0/ || $0 ~ /equal False/) print }' | tr '\n' ';')
  echo "seed $s: ${r:-ok} $(tail -1 $d/log | cut -c1-150)"
  rm -rf $d
done
