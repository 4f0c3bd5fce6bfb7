import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from openvvc_amd import capi, engine
from shim_cases import ShimStream
from test_shim_cpu import intra_ctu_cases
ctx = engine.Context(0)
base, cases = intra_ctu_cases()
s = ShimStream("shim_intra_ctu.ovg")
h, w = base[0].shape
job = engine.Job(ctx, w, h)
dst = ctx.new_pic(w, h)
for one_launch in (False, True):
    for i in range(3):
        c = s.case(i)
        print("case", i, one_launch, len(c["tb"]), len(c["coef"]), len(c["itask"]), flush=True)
        dst.upload(*base)
        job.begin()
        job.rec.append_raw(capi.REC_COEF, c["coef"])
        job.rec.append_raw(capi.REC_TB, c["tb"])
        job.rec.append_raw(capi.REC_ITASK, c["itask"])
        p = capi.JobParams(); p.log2_ctu_s = 7
        p.stages = capi.STAGE_ITX | capi.STAGE_INTRA | (capi.STAGE_INTRA_CTU if one_launch else 0)
        print(" flush", flush=True)
        job.flush(dst, [], None, params=p)
        print(" wait", flush=True)
        job.wait()
        y, cb, cr = dst.download()
        print(" ok", np.array_equal(y[128:256,128:256], cases[i][2]), flush=True)
