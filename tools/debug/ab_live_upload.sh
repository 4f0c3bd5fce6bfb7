#!/bin/bash
# the live decoder with the uploads on a picture's own stream (0) / on shared upload streams (2): 257 pictures, 16 and 32 frame threads
for i in 1 2 3; do for n in 0 2; do
  OVVC_HIP_UPLOAD_STREAMS=$n oracle/_ref/patched/gen_pipe /tmp live threads 16,32 size 3840 2160 pics 257 gop 32 noisp seed 31337 reps 2 2>/dev/null | grep "^{" | python3 -c "
import sys,json
print('upload streams $n:', ' / '.join('%d thr %.1f (err %d diff %d)' % (r['frame_threads'], r['pictures_per_second'], r['shim_error'], r['samples_differing']) for r in map(json.loads, sys.stdin)))"
done; done
