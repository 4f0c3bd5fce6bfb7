"""Debug: deblocking time against the order of the edge lists (as the recorder emits them CTU by CTU / component-major / raster)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from openvvc_amd import capi, engine, synth
w, h = 3840, 2160
ctx = engine.Context(0)
st = torch.cuda.ExternalStream(ctx.stream, device=torch.device("cuda", 0))
wl = synth.make_workload(w, h, 0x266)
rs = np.random.RandomState(1)
pic = ctx.upload_pic(*wl.refs[0])
def order(e, how):
    if how == "emitted": return e
    if how == "comp-major": return e[np.argsort(e["comp"], kind="stable")]
    if how == "raster": return np.sort(e, order=["comp", "uy", "ux"])
    if how == "ctu-rows":                       # inside a CTU row by row, components apart
        key = (e["comp"].astype(np.int64) << 40) | ((e["uy"] >> 5).astype(np.int64) << 30) | ((e["ux"] >> 5).astype(np.int64) << 20) | ((e["uy"] & 31).astype(np.int64) << 8) | (e["ux"] & 31)
        return e[np.argsort(key, kind="stable")]
    if how == "ctu-rows-mixed":                 # the same, components inside the CTU
        key = ((e["uy"] >> 5).astype(np.int64) << 40) | ((e["ux"] >> 5).astype(np.int64) << 30) | (e["comp"].astype(np.int64) << 20) | ((e["uy"] & 31).astype(np.int64) << 8) | (e["ux"] & 31)
        return e[np.argsort(key, kind="stable")]
for how in ("emitted", "comp-major", "ctu-rows", "ctu-rows-mixed", "raster"):
    ev, eh = (ctx.upload(np.ascontiguousarray(order(e, how))) for e in wl.dbf_edges)
    for _ in range(5): ctx.dbf_edges(pic, ev, eh, 0, 0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record_event(a)
    for _ in range(50): ctx.dbf_edges(pic, ev, eh, 0, 0)
    st.record_event(b); b.synchronize()
    print(f"{how:16s} {a.elapsed_time(b) * 1000 / 50:7.2f} us per picture (V + H)   edges {len(wl.dbf_edges[0])} + {len(wl.dbf_edges[1])}")
