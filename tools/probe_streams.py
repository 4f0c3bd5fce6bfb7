"""How much does keeping S independent pictures in flight (S streams, no cross-stream dependency) buy?"""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from openvvc_amd import engine, synth
dev = torch.device("cuda", 0)
wl = synth.make_workload(3840, 2160, 0x266)
for S in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(dev) for _ in range(S)]
    ctxs = [engine.Context(0, stream=s.cuda_stream) for s in streams]
    rps = [engine.ResidentPicture(c, wl) for c in ctxs]
    for i in range(4 * S):
        rps[i % S].decode()
    torch.cuda.synchronize()
    N = 120
    t0 = time.perf_counter()
    for i in range(N):
        rps[i % S].decode()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("streams %d: %.1f frames/s  (%.1f us/frame)" % (S, N / dt, dt / N * 1e6))
    for r in rps: r.free()
    for c in ctxs: c.close()
