import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from openvvc_amd import capi, engine
from shim_cases import ShimStream
from test_shim_cpu import isp_cases
import oracle_lib
from oracle_lib import HostPic
ctx = engine.Context(0)
base, info, exp = isp_cases()
s = ShimStream("shim_isp.ovg")
h, w = base.shape
cb = np.full((h // 2, w // 2), 512, np.uint16)
job = engine.Job(ctx, w, h)
dst = ctx.new_pic(w, h)
shown = 0
for i, (x, y, l2w, l2h, vertical, mode, bits, off) in enumerate(info):
    c = s.case(i)
    dst.upload(base, cb, cb)
    job.begin()
    job.rec.append_raw(capi.REC_COEF, c["coef"]); job.rec.append_raw(capi.REC_TB, c["tb"]); job.rec.append_raw(capi.REC_ITASK, c["itask"])
    p = capi.JobParams(); p.log2_ctu_s = 7; p.stages = capi.STAGE_ITX | capi.STAGE_INTRA
    job.flush(dst, [], None, params=p); job.wait()
    yy = dst.download()[0]
    bw, bh = 1 << int(l2w), 1 << int(l2h)
    got = yy[y:y + bh, x:x + bw]; want = exp[int(off):int(off) + bw * bh].reshape(bh, bw)
    if not np.array_equal(got, want) and shown < 4:
        shown += 1
        print("case", i, bw, bh, "vertical", vertical, "mode", mode, hex(bits))
        print(" tb:", [(int(t["x"]), int(t["y"]), int(t["log2_w"]), int(t["log2_h"]), hex(int(t["kind"])), int(t["tr_h"]), int(t["tr_v"])) for t in c["tb"]])
        d = np.argwhere(got != want)
        print(" diff cols", sorted(set(d[:, 1].tolist())), "rows", sorted(set(d[:, 0].tolist()))[:10])
        # residual picture from the oracle for comparison
        hp = HostPic(w, h, base.copy()); res = HostPic(w, h)
        oracle_lib.itx_res(hp, c["tb"], c["coef"], None, res)
        print(" oracle residual cols:\n", res.y.view(np.int16)[y:y + min(bh, 8), x:x + bw])
        print(" got-want:\n", (got.astype(int) - want.astype(int))[:8])
