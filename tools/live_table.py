"""prints one line per `gen_pipe ... live` JSON line on stdin: rate, differences, a frame thread's milliseconds per picture by what they went to"""
import json
import sys

for l in sys.stdin:
    if not l.startswith("{"):
        print(l.rstrip())
        continue
    d = json.loads(l)
    n = d["pictures"]
    ms = lambda k: round(1e3 * d[k] / n, 2)
    busy, sync, hooks, dev = ms("thread_seconds_with_a_picture"), ms("thread_seconds_waiting_for_collocated_rows"), ms("thread_seconds_in_shim_hooks"), ms("thread_seconds_in_shim_device_half")
    print(f'{d["width"]}x{d["height"]} out={d["output"][:6]} threads {d["frame_threads"]:2d}: {d["pictures_per_second"]:7.1f} pictures/s, differing {d["samples_differing"]} / {d["collocated_motion_entries_differing"]}; '
          f'ms per picture: held {busy}, waiting for collocated rows {sync}, device half {dev}, recording {round(hooks - dev, 2)}, parse {round(busy - sync - hooks, 2)} ({d["shim_hook_calls"] // n} hook calls)')
