"""Debug: duration of a k_intra_level launch holding ONE task, per task class (run under rocprofv3 --kernel-trace; the
companion tools/debug/item_latency_read.py prints the medians)."""
import sys, json
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import golden_io
from openvvc_amd import capi, engine
ctx = engine.Context(0)
g = golden_io.load("intra.ovg")
tasks = np.frombuffer(g["task"].tobytes(), dtype=capi.ITASK_DTYPE)
H, W = g["pic_y"].shape
pic = ctx.upload_pic(g["pic_y"], g["pic_cb"], g["pic_cr"])
res = ctx.new_pic(W, H)
def pick(cond):
    i = np.flatnonzero(cond)
    return tasks[i[len(i) // 2]:i[len(i) // 2] + 1].copy() if len(i) else None
L, C_ = tasks["kind"] == 0, tasks["kind"] == 1
mip = (tasks["flags"] & capi.IF_MIP) != 0
sz = lambda a, b: (tasks["log2_w"] == a) & (tasks["log2_h"] == b)
null = tasks[:1].copy(); null["kind"] = capi.IT_RES_C; null["flags"] = 0
classes = [("null", null),
           ("dc4x4", pick(L & ~mip & (tasks["mode"] == 1) & sz(2, 2))),
           ("planar8x8", pick(L & ~mip & (tasks["mode"] == 0) & sz(3, 3))),
           ("ang50_8x8", pick(L & ~mip & (tasks["mode"] == 50) & sz(3, 3))),
           ("ang45_8x8", pick(L & ~mip & (tasks["mode"] == 45) & sz(3, 3))),
           ("ang30_8x8", pick(L & ~mip & (tasks["mode"] == 30) & sz(3, 3))),
           ("ang45_16x16", pick(L & ~mip & (tasks["mode"] == 45) & sz(4, 4))),
           ("ang45_32x32", pick(L & ~mip & (tasks["mode"] == 45) & sz(5, 5))),
           ("ang45_64x64", pick(L & ~mip & (tasks["mode"] >= 40) & sz(6, 6))),
           ("mip4x4", pick(L & mip & sz(2, 2))), ("mip8x8", pick(L & mip & sz(3, 3))), ("mip16x16", pick(L & mip & sz(4, 4))),
           ("c_dc4x4", pick(C_ & (tasks["mode"] == 1) & sz(2, 2))), ("c_ang8x8", pick(C_ & (tasks["mode"] > 1) & (tasks["mode"] < 67) & sz(3, 3))),
           ("cclm8x8", pick(C_ & (tasks["mode"] == 67) & sz(3, 3))), ("mdlm8x8", pick(C_ & (tasks["mode"] == 69) & sz(3, 3)))]
classes = [(n, t) for n, t in classes if t is not None]
N = 40
for n, t in classes:
    d = ctx.upload(t)
    geom = int(ctx.lib.ovhip_intra_level_geom(t.ctypes.data, 1))
    for _ in range(N):
        ctx.intra_level(pic, res, d, 0, 1, geom=geom)
    ctx.sync()
json.dump({"names": [n for n, _ in classes], "N": N}, open("gpurun_out/item_classes.json", "w"))
