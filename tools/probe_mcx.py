import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from openvvc_amd import engine, synth, capi
dev=torch.device("cuda",0); stream=torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx=engine.Context(0, stream=stream.cuda_stream)
def timeit(rp, name, n=50):
    for _ in range(5): rp.run_stage(name)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(n): rp.run_stage(name)
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1000
for tools in (("bdof",),("dmvr",),("bdof","dmvr")):
    wl=synth.make_workload(3840,2160,0x266,tools=tools)
    rp=engine.ResidentPicture(ctx,wl)
    ux=wl.mcx_units
    nd=int(((ux["flags"]&64)!=0).sum()); nb=len(ux)-nd
    print(tools, "units", len(ux), "dmvr", nd, "bdof-only", nb, "mcx us %.1f"%timeit(rp,"mcx"), "mcp us %.1f"%timeit(rp,"mcp"))
    rp.free()
