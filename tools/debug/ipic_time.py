"""Debug: one 4K intra picture alone on the device -- wall time of flush + wait, time of the ordered pass, levels."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from openvvc_amd import capi, engine, synth
w, h = 3840, 2160
if len(sys.argv) > 1: synth.LM_FRAC = float(sys.argv[1])       # e.g. 0: no cross-component chroma blocks
ctx = engine.Context(0)
wl = synth.make_workload(w, h, 0x266, tools=synth.INTRA_TOOLS, intra_frac=1.0)
print("tasks", wl.stats["n_itasks"], "levels", wl.stats["n_ilevels"])
job = engine.Job(ctx, w, h)
refs = [ctx.upload_pic(*r) for r in wl.refs]
dst = ctx.new_pic(w, h)
job.load_workload(wl)
for rep in range(6):
    job.time_stage("intra" if rep >= 3 else None)
    t0 = time.perf_counter()
    job.flush(dst, refs, None); job.wait()
    t1 = time.perf_counter()
    print("rep", rep, "flush+wait %.3f ms" % ((t1 - t0) * 1e3), ("intra stage %.3f ms" % (job.stage_time()[0] * 1e3 / max(1, job.stage_time()[1]))) if rep >= 3 else "")
    job.begin(); job.load_workload(wl)
