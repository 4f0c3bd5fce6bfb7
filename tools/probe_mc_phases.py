"""Debug probe: per-phase shader-clock breakdown of k_mc2 (library built with EXTRA=-DOV_MC_PHASES)."""
import sys, ctypes; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from openvvc_amd import engine, synth, capi
dev = torch.device("cuda", 0); stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = engine.Context(0, stream=stream.cuda_stream)
lib = capi.load()
wl = synth.make_workload(3840, 2160, 0x266)
rp = engine.ResidentPicture(ctx, wl)
def probe(stage, fn, names):
    if not hasattr(lib, fn):
        return
    for _ in range(3): rp.run_stage(stage)
    torch.cuda.synchronize()
    N = 10
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(N): rp.run_stage(stage)
    e1.record(stream); torch.cuda.synchronize()
    buf = np.zeros((65536, 8), np.uint32)
    getattr(lib, fn)(buf.ctypes.data_as(ctypes.c_void_p))
    m = buf[buf[:, 7] == 1][:, :len(names)].astype(np.float64)
    print(stage, "units", len(m), "launch us %.1f" % (e0.elapsed_time(e1) / N * 1000))
    tot = m.sum()
    for i, nm in enumerate(names):
        print("  %-28s mean %8.0f  median %8.0f clk/unit  %5.1f%%" % (nm, m[:, i].mean(), np.median(m[:, i]), 100.0 * m[:, i].sum() / tot))
    t = m.sum(axis=1)
    print("  total %.0f clk/unit   p50 %.0f  p99 %.0f  max %.0f   units > 2 x median: %d" %
          (tot / len(m), np.median(t), np.percentile(t, 99), t.max(), int((t > 2 * np.median(t)).sum())))
    slow = np.argsort(t)[-5:]
    print("  slowest units:", [(int(np.flatnonzero(buf[:, 7] == 1)[i]), int(t[i])) for i in slow])

probe("mcp", "ovhip_debug_mc_phases",
      ["unit fetch", "window issue", "window wait+park+taps", "H passes", "V+combine (stores issued)", "store drain"])
probe("mca", "ovhip_debug_mca_phases",
      ["unit fetch", "sub-block MVs", "windows", "H passes", "V + PROF tiles", "PROF + luma store", "chroma + drain"])
