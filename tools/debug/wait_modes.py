"""Debug: picture 2 predicted from picture 1 on another stream, ordered by an event: host wait vs stream wait."""
import sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import oracle_pipeline
from openvvc_amd import capi, engine, synth
w, h = 832, 480
dev = torch.device("cuda", 0)
wl1 = synth.make_workload(w, h, 0x51, tools=synth.INTRA_TOOLS, intra_frac=0.2)
wl2 = synth.make_workload(w, h, 0x52, tools=synth.INTRA_TOOLS, intra_frac=0.2)
ref1 = oracle_pipeline.decode(wl1)
wl2.refs[0] = (ref1.y.copy(), ref1.cb.copy(), ref1.cr.copy())
ref2 = oracle_pipeline.decode(wl2)
def run(mode):
    c1, c2 = engine.Context(0), engine.Context(0)
    s1 = torch.cuda.ExternalStream(c1.stream, device=dev)
    refs1 = [c1.upload_pic(*r) for r in wl1.refs]
    dst1, dst2 = c1.new_pic(w, h), c2.new_pic(w, h)
    refs2 = [dst1] + [c2.upload_pic(*r) for r in wl2.refs[1:]]
    j1, j2 = engine.Job(c1, w, h), engine.Job(c2, w, h)
    for rep in range(2):
        j1.load_workload(wl1); j2.load_workload(wl2)
        j1.flush(dst1, refs1, None)
        ev = torch.cuda.Event(); s1.record_event(ev)
        handles = (C.c_void_p * 1)(ev.cuda_event)
        j2.params.wait_events = C.cast(handles, C.POINTER(C.c_void_p))
        j2.params.n_wait_events = 1
        j2.params.wait_on_host = 1 if mode == "host" else 0
        j2.flush(dst2, refs2, None); j2.wait(); j1.wait()
        got = dst2.download(); g1 = dst1.download()
        print(mode, rep, "pic2 diff", int((got[0] != ref2.y).sum()), "pic1 diff", int((g1[0] != ref1.y).sum()), "event", hex(ev.cuda_event),
              "streams", hex(c1.stream), hex(c2.stream), "retries", j1.stats().n_ordered_retries, j2.stats().n_ordered_retries)
        j1.begin(); j2.begin()
    j1.close(); j2.close(); c1.close(); c2.close()
for m in (sys.argv[1:] or ["host", "stream", "stream", "host"]):
    run(m)
