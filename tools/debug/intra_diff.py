"""Debug: classify the intra.ovg cases that differ on the GPU."""
import sys, collections
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import golden_io
from openvvc_amd import capi, engine
BAND = 384
ctx = engine.Context(0)
g = golden_io.load("intra.ovg")
tasks = np.frombuffer(g["task"].tobytes(), dtype=capi.ITASK_DTYPE)
H, W = g["pic_y"].shape
base = [np.zeros((BAND, W), np.uint16), np.zeros((BAND // 2, W // 2), np.uint16), np.zeros((BAND // 2, W // 2), np.uint16)]
base[0][:H] = g["pic_y"]; base[1][:H // 2] = g["pic_cb"]; base[2][:H // 2] = g["pic_cr"]
NB = 160
tall = [np.tile(p, (NB, 1)) for p in base]
res = ctx.new_pic(W, BAND * NB)
cnt = collections.Counter(); tot = collections.Counter(); shown = 0
for b0 in range(0, len(tasks), NB):
    t = tasks[b0:b0 + NB].copy(); k = np.arange(len(t))
    t["y"] += np.where(t["kind"] == capi.IT_LUMA, k * BAND, k * (BAND // 2)).astype(np.uint16)
    pic = ctx.upload_pic(*tall); ctx.intra_level(pic, res, ctx.upload(t), 0, len(t)); ctx.sync()
    y, cb, cr = pic.download(); pic.free()
    for i in range(len(t)):
        tt = t[i]; w, h, x, yy = 1 << int(tt["log2_w"]), 1 << int(tt["log2_h"]), int(tt["x"]), int(tt["y"])
        eo = g["exp_off"][b0 + i]
        m = int(tt["mode"]); fl = int(tt["flags"])
        cls = "mip" if fl & capi.IF_MIP else ("pl" if m == 0 else "dc" if m == 1 else "lm" if (tt["kind"] == 1 and m >= 67) else "ang_h" if m < 34 else "ang_v")
        key = (int(tt["kind"]), cls, int(tt["mrl_idx"]))
        tot[key] += 1
        e0 = g["exp"][eo[0]:eo[0] + w * h].reshape(h, w)
        got = (y if tt["kind"] == 0 else cb)[yy:yy + h, x:x + w]
        ok = np.array_equal(got, e0)
        if ok and tt["kind"] == 1:
            ok = np.array_equal(cr[yy:yy + h, x:x + w], g["exp"][eo[1]:eo[1] + w * h].reshape(h, w))
        if not ok:
            cnt[key] += 1
            if shown < 6 and cnt[key] == 1:
                shown += 1
                print("case", b0 + i, {n: int(tt[n]) for n in tt.dtype.names if n != "pad"})
                print("got\n", got[:8, :8]); print("exp\n", e0[:8, :8])
for k in sorted(tot): print(k, cnt[k], "/", tot[k])
