#!/bin/bash
# the stream with the fingerprint output and without any output, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for o in digest none "none --debug-digests-only"; do
  python $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-reference-stream --no-isolated-survey --check 0 --output $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('output $o:', d['value'], 'median step', d['config']['step_fps']['median'])"
done; done
