import sys, hashlib; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from openvvc_amd import engine, synth
dev = torch.device("cuda", 0)
wls = [synth.make_workload(3840, 2160, 0x266), synth.make_workload(3840, 2160, 77)]
md5 = lambda planes: hashlib.md5(b"".join(p.tobytes() for p in planes)).hexdigest()
# single stream references
ref = []
c = engine.Context(0)
for wl in wls:
    rp = engine.ResidentPicture(c, wl); rp.decode(); ref.append(md5(rp.result())); rp.free()
c.close()
streams = [torch.cuda.Stream(dev) for _ in range(2)]
ctxs = [engine.Context(0, stream=s.cuda_stream) for s in streams]
rps = [engine.ResidentPicture(cx, wl) for cx, wl in zip(ctxs, wls)]
bad = 0
for it in range(30):
    for rp in rps: rp.decode()
    if it % 10 == 9:
        torch.cuda.synchronize()
        for i, rp in enumerate(rps):
            if md5(rp.result()) != ref[i]: bad += 1
print("mismatches:", bad, "refs", ref)
