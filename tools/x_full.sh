#!/bin/bash
# the driver's configuration (--steps 20 --warmup 5) with and without uploads ahead, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
for r in 1 2 3; do
  for v in 16 0; do
    python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-stream --upload-ahead $v 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('upload ahead $v: fps', d['value'], c['step_fps']['min'], c['step_fps']['median'], c['step_fps']['max'], 'resident', c['resident_replay_fps'], 'none', c['variants']['output_none'], 'in-order', c['variants']['in_order_no_lookahead'])"
  done
done
