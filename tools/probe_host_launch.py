"""Host time per picture (9 launches through ctypes) vs GPU time per picture: is the launch path a bottleneck?"""
import sys, time; sys.path.insert(0, '/root/repo')
import torch
from openvvc_amd import engine, synth
dev = torch.device("cuda", 0)
wl = synth.make_workload(3840, 2160, 0x266)
streams = [torch.cuda.Stream(dev) for _ in range(2)]
ctxs = [engine.Context(0, stream=s.cuda_stream) for s in streams]
rps = [engine.ResidentPicture(c, wl) for c in ctxs]
for i in range(8): rps[i % 2].decode()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for i in range(N): rps[i % 2].decode()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f us/picture, total %.1f us/picture" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
