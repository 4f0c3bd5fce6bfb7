"""Debug: the inverse luma mapping alone on the device (4K luma plane, in place)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
from openvvc_amd import capi, engine, synth
w, h = 3840, 2160
ctx = engine.Context(0)
st = torch.cuda.ExternalStream(ctx.stream, device=torch.device("cuda", 0))
rs = np.random.RandomState(3)
pics = [ctx.upload_pic(*synth.random_picture(rs, w, h)) for _ in range(24)]        # 24 x 25 MB: not cache-resident
lut = ctx.upload(rs.randint(0, 1024, 1024).astype(np.uint16))
for _ in range(2):
    for p in pics: ctx.lmcs_inverse(p, lut)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record_event(a)
for _ in range(4):
    for p in pics: ctx.lmcs_inverse(p, lut)
st.record_event(b); b.synchronize()
us = a.elapsed_time(b) * 1000 / (4 * len(pics))
print(f"k_lmcs_inverse {us:6.2f} us per 4K luma plane, {2 * w * h * 2 / us / 1e6:5.2f} TB/s of 2 x {w * h * 2 / 1e6:.1f} MB")
