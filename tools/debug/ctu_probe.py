"""Debug (needs a library built with -DOVHIP_CTU_PROBE): phase times of the first CTUs of the one-launch ordered pass on a 4K I picture."""
import sys, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from openvvc_amd import capi, engine, synth
W, H = 3840, 2160
ctx = engine.Context(0)
wl = synth.make_workload(W, H, 0x266 + 7777, tools=synth.INTRA_TOOLS, intra_frac=1.0)
job = engine.Job(ctx, W, H)
dst = ctx.new_pic(W, H)
job.load_workload(wl)
probe = ctx.upload(np.zeros(64 * 256, np.uint64))
ctx.lib.ovhip_debug_set_ctu_probe.argtypes = [C.c_void_p]
assert ctx.lib.ovhip_debug_set_ctu_probe(probe.ptr) == 0
job.params.stages = capi.STAGE_ALL | capi.STAGE_INTRA_CTU
for _ in range(2):
    job.flush(dst, [], None); job.wait()
p = probe.download(np.uint64).reshape(64, 256).astype(np.int64)
rec = capi.Recorder(W, H); rec.append_raw(capi.REC_ITASK, wl.itasks); t, cs = rec.itasks_by_ctu(7)
for b in (0, 1, 2, 5, 31, 40):
    r = p[b]
    runs = r[4:]; runs = runs[runs > 0]
    d = np.diff(np.concatenate([[r[1]], runs])) / 100.0
    tt = t[int(cs[b]["first"]):int(cs[b]["first"]) + int(cs[b]["n"])]
    print(f"CTU {b} ({cs[b]['cx']},{cs[b]['cy']}) tasks {cs[b]['n']} runs {len(runs)}: wait+load {(r[1]-r[0])/100:.1f} us, tasks {(r[2]-r[1])/100:.1f} us, publish {(r[3]-r[2])/100:.1f} us; per run us: median {np.median(d):.2f} mean {d.mean():.2f} max {d.max():.2f}")
    print("   first runs:", np.round(d[:16], 2).tolist())
