"""Debug (needs openvvc_amd/libovvc_hip_probe.so = the library with kernels_intra.hip built -DOVHIP_CTU_PROBE): where the time of
one hop of the flow launch goes on a lone 4K intra picture.  Stamps per item (100 MHz): 0 entry, 1 poll begins, 2 inputs ready,
3 references in LDS (luma), 4 prediction done, 5 stores issued, 6 stores acknowledged, 7 units marked."""
import sys, ctypes as C, shutil, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from openvvc_amd import capi
import os
os.environ["OVHIP_FLOW_CHUNK"] = "100000000"          # one flow launch per picture: the stamps are indexed from the launch's first item
capi.LIB_PATH = capi.LIB_PATH.with_name("libovvc_hip_probe.so")
from openvvc_amd import engine, synth
W, H = 3840, 2160
ctx = engine.Context(0)
wl = synth.make_workload(W, H, 0x266, tools=synth.INTRA_TOOLS, intra_frac=1.0)
job = engine.Job(ctx, W, H)
dst = ctx.new_pic(W, H)
job.load_workload(wl)
t = wl.itasks
order = np.argsort(t["level"], kind="stable")
items = []
for i in order:
    tt = t[i]
    if tt["kind"] == capi.IT_REGION: items.append((i, 0, 0)); continue
    npx = 1 << (int(tt["log2_w"]) + int(tt["log2_h"])); strips = (npx + 255) // 256; comps = 1 if tt["kind"] == capi.IT_LUMA else 2
    for s in range(strips):
        for c in range(comps): items.append((i, s, c))
n = len(items)
probe = ctx.upload(np.zeros(n * 8, np.uint64))
ctx.lib.ovhip_debug_set_ctu_probe.argtypes = [C.c_void_p]
assert ctx.lib.ovhip_debug_set_ctu_probe(probe.ptr) == 0
for _ in range(3):
    job.flush(dst, [], None); job.wait()
    job.begin(); job.load_workload(wl)
job.flush(dst, [], None); job.wait()
p = probe.download(np.uint64).reshape(n, 8).astype(np.int64)
t0 = p[:, 0][p[:, 0] > 0].min()
us = lambda a: a / 100.0
print("items", n, "span first entry -> last mark: %.1f us" % us(p[:, 7].max() - t0))
# luma items of single-strip blocks: phases
uw = (W + 3) // 4
owner = np.full(((H + 3) // 4, uw), -1, np.int64)
luma = []
for k, (i, s, c) in enumerate(items):
    tt = t[i]
    if tt["kind"] != capi.IT_LUMA or (int(tt["log2_w"]) + int(tt["log2_h"])) > 8 or int(tt["flags"]) & capi.IF_ISP: continue
    x, y, w, h = int(tt["x"]), int(tt["y"]), 1 << int(tt["log2_w"]), 1 << int(tt["log2_h"])
    owner[y >> 2:(y + h + 3) >> 2, x >> 2:(x + w + 3) >> 2] = k
    luma.append(k)
area = {}
ph = {"poll begin->ready": [], "ready->refs": [], "refs->pred": [], "pred->issued": [], "issued->acked": [], "acked->marked": [], "producer marked->ready": [], "producer acked->ready": []}
waited = 0
for k in luma:
    i, s, c = items[k]; tt = t[i]
    r = p[k]
    if r[7] == 0: continue
    x, y = int(tt["x"]) >> 2, int(tt["y"]) >> 2
    prods = set()
    for j in range(int(tt["avl_abv"])):
        if y - 1 >= 0 and x + j < uw: prods.add(owner[y - 1, x + j])
    for j in range(int(tt["avl_lft"])):
        if x - 1 >= 0 and y + j < owner.shape[0]: prods.add(owner[y + j, x - 1])
    if int(tt["flags"]) & capi.IF_CORNER: prods.add(owner[y - 1, x - 1])
    prods.discard(-1)
    if not prods: continue
    last7 = max(p[q][7] for q in prods); last6 = max(p[q][6] for q in prods); last5 = max(p[q][5] for q in prods)
    if last5 < r[1]: continue                      # inputs were on their way before this item looked: not on a critical hop
    waited += 1
    area.setdefault((int(tt["log2_w"]) + int(tt["log2_h"]), "mip" if int(tt["flags"]) & capi.IF_MIP else ("pl/dc" if int(tt["mode"]) < 2 else "ang")), []).append((r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], r[7] - r[6], r[2] - last7))
    ph["poll begin->ready"].append(r[2] - r[1]); ph["ready->refs"].append(r[3] - r[2]); ph["refs->pred"].append(r[4] - r[3])
    ph["pred->issued"].append(r[5] - r[4]); ph["issued->acked"].append(r[6] - r[5]); ph["acked->marked"].append(r[7] - r[6])
    ph["producer marked->ready"].append(r[2] - last7); ph["producer acked->ready"].append(r[2] - last6)
    ph.setdefault("producer stores issued->refs in LDS (tagged hand-over)", []).append(r[3] - last5)
print("luma items that waited for their last producer:", waited, "of", len(luma))
for name, v in ph.items():
    v = us(np.array(v))
    print(f"  {name:28s} median {np.median(v):6.2f}  mean {v.mean():6.2f}  p90 {np.percentile(v, 90):6.2f} us")
hop = us(np.array(ph["producer marked->ready"]) + np.array(ph["ready->refs"]) + np.array(ph["refs->pred"]) + np.array(ph["pred->issued"]) + np.array(ph["issued->acked"]) + np.array(ph["acked->marked"]))
print("  hop (producer marked -> this item marked): median %.2f mean %.2f us" % (np.median(hop), hop.mean()))

print("by log2 area, kind: n, ready->refs, refs->pred, pred->issued, issued->acked, acked->marked, detect (median us)")
for k in sorted(area):
    a = np.array(area[k]) / 100.0
    print(" ", k, len(a), np.round(np.median(a, axis=0), 2).tolist())

# when does each plane finish?  (is the chroma chain behind the luma chain?)
kinds = np.array([int(t[i]["kind"]) for i, s_, c_ in items])
comp = np.array([c_ for i, s_, c_ in items])
end = p[:, 7]
for name, m in (("luma", kinds == capi.IT_LUMA), ("chroma Cb", (kinds == capi.IT_CHROMA) & (comp == 0)), ("chroma Cr", (kinds == capi.IT_CHROMA) & (comp == 1))):
    e = end[m & (end > 0)]
    print(f"last {name} item marked at {us(e.max() - t0):9.1f} us; 99 % of them by {us(np.percentile(e, 99) - t0):9.1f} us")
