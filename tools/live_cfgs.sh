#!/bin/bash
# the three live configurations of bench.py / DESIGN with a given environment: tools/live_cfgs.sh <label> [gen_pipe words ...]   (through gpurun)
label=$1; shift
exe=oracle/_ref/patched/gen_pipe
for cfg in "threads 16 size 3840 2160 pics 33 reps 3" "threads 16,32 size 3840 2160 pics 65 gop 32 noisp seed 31337 cont 4 reps 2" "threads 16,32 size 3840 2160 pics 257 gop 32 noisp seed 31337 reps 2"; do
  $exe /tmp live $cfg profile "$@" 2>/dev/null | grep "^{" | python3 -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); n=r['pictures']
    print('$label | %-3d pics %2d thr | fps %6.1f | err %d diff %d | hold %.1f wait %.1f hooks %.1f dev %.1f | bands %d/%d' % (n, r['frame_threads'], r['pictures_per_second'], r['shim_error'], r['samples_differing']+r['collocated_motion_entries_differing'], 1e3*r['thread_seconds_with_a_picture']/n, 1e3*r['thread_seconds_waiting_for_collocated_rows']/n, 1e3*r['thread_seconds_in_shim_hooks']/n, 1e3*r['thread_seconds_in_shim_device_half']/n, r['bands_sent'], r['bands_left_to_a_later_hook']))"
done
