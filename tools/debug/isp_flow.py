import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from openvvc_amd import capi, engine
from shim_cases import ShimStream
from test_shim_cpu import isp_cases
ctx = engine.Context(0)
base, info, exp = isp_cases()
s = ShimStream("shim_isp.ovg")
h, w = base.shape
cb = np.full((h // 2, w // 2), 512, np.uint16)
job = engine.Job(ctx, w, h)
dst = ctx.new_pic(w, h)
bad = {}
for rep in range(6):
    for i, (x, y, l2w, l2h, vertical, mode, bits, off) in enumerate(info):
        c = s.case(i)
        dst.upload(base, cb, cb)
        job.begin()
        job.rec.append_raw(capi.REC_COEF, c["coef"]); job.rec.append_raw(capi.REC_TB, c["tb"]); job.rec.append_raw(capi.REC_ITASK, c["itask"])
        p = capi.JobParams(); p.log2_ctu_s = 7; p.stages = capi.STAGE_ITX | capi.STAGE_INTRA | int(sys.argv[1], 0)
        job.flush(dst, [], None, params=p); job.wait()
        yy = dst.download()[0]
        bw, bh = 1 << int(l2w), 1 << int(l2h)
        got = yy[y:y + bh, x:x + bw]; want = exp[int(off):int(off) + bw * bh].reshape(bh, bw)
        if not np.array_equal(got, want):
            d = np.argwhere(got != want)
            bad.setdefault(i, []).append((rep, len(d), sorted(set(d[:, 1].tolist()))[:8], sorted(set(d[:, 0].tolist()))[:6]))
for i, v in bad.items():
    print("case", i, info[i].tolist(), v)
    print("  tasks", [(int(t["x"]), int(t["y"]), int(t["log2_w"]), int(t["log2_h"]), int(t["level"]), int(t["avl_abv"]), int(t["avl_lft"]), hex(int(t["flags"]))) for t in s.case(i)["itask"]])
print("bad cases", len(bad))
