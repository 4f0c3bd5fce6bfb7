"""Rough host cost of the recorder (ovhip_rec_*) for one 4K picture: time spent inside the C calls as seen through
ctypes, minus the ctypes call overhead calibrated on a trivial entry point.  (No GPU needed.)"""
import sys, time; sys.path.insert(0, '/root/repo')
import ctypes as C
from openvvc_amd import capi, synth
lib = capi.load()
acc = {}
def wrap(name):
    f = getattr(capi.Recorder, name)
    def g(self, *a, **k):
        t0 = time.perf_counter_ns()
        r = f(self, *a, **k)
        d = acc.setdefault(name, [0, 0]); d[0] += time.perf_counter_ns() - t0; d[1] += 1
        return r
    setattr(capi.Recorder, name, g)
for n in ("pu", "tu", "affine_cu", "lmcs_region", "tb_cmds_split"):
    if hasattr(capi.Recorder, n): wrap(n)
t0 = time.perf_counter()
wl = synth.make_workload(3840, 2160, 0x266)
print("make_workload (python generator + recorder): %.2f s" % (time.perf_counter() - t0))
# ctypes overhead of a call with one byref argument
N = 200000
x = C.c_int(0)
t0 = time.perf_counter_ns()
for _ in range(N): lib.ovhip_abi_version()
ovh = (time.perf_counter_ns() - t0) / N
print("ctypes call overhead ~%.0f ns" % ovh)
tot = 0
for n, (ns, cnt) in acc.items():
    net = max(ns - cnt * (ovh + 250), 0)      # + python wrapper frames
    tot += net
    print("%-14s %7d calls  %7.2f ms gross  ~%6.2f ms net  (%4.0f ns/call net)" % (n, cnt, ns / 1e6, net / 1e6, net / max(cnt, 1)))
print("recorder, one 4K picture (%d CUs): ~%.1f ms of host C time on one core -> ~%.0f pictures/s per host thread" % (wl.stats["n_cu"], tot / 1e6, 1e9 / max(tot, 1)))
