"""bench.py --trace T: what is a picture's latency made of once it COULD run?  For every timed picture: ready = max(taken by a frame
thread, last of its reference pictures published); latency after ready = published - ready.  If the device were the bound, pictures
would queue (ready long before they run) and the sum of "after ready" latencies over the pictures in flight would explain the rate;
if dependencies are, pictures run the moment they are ready and "after ready" is the bare pipeline latency of one picture.
python tools/debug/dep_latency.py gpurun_out/trace.npy"""
import sys
import numpy as np
t = np.load(sys.argv[1]); refs = np.load(sys.argv[1] + ".refs.npy")
take, sub, pub, thr, inhand, launched, returned, done, nrefs, poc, idx = (t[:, i] for i in range(11))
n = len(t)
first = int(idx[0])
pub_of = {int(idx[i]): pub[i] for i in range(n)}
ready_refs = np.array([max([pub_of.get(int(r), 0.0) for r in refs[i] if r >= 0] + [0.0]) for i in range(n)])
ready = np.maximum(take, ready_refs)
after = pub - ready
waited = ready_refs > take                      # the picture was in a thread's hands before its references were done
total = pub.max()
I = nrefs == 0
print(f"{n} pictures, {n / total:.0f} pictures/s, {int(thr.max()) + 1} threads")
print(f"B pictures taken BEFORE their references were done: {100 * waited[~I].mean():.0f} % (they wait {1e3 * (ready_refs - take)[waited & ~I].mean():.2f} ms on average)")
for name, m in (("B, waited for references", waited & ~I), ("B, references already done when taken", ~waited & ~I), ("I", I)):
    if m.any():
        a = after[m]
        print(f"  {name}: {m.sum()} pictures, latency after ready: mean {1e3 * a.mean():.2f} ms, median {1e3 * np.median(a):.2f}, p10 {1e3 * np.percentile(a, 10):.2f}, p90 {1e3 * np.percentile(a, 90):.2f}")
# pictures in flight that are READY (running or queued on the device) over time
ts = np.linspace(0, total, 2000)
inflight = np.array([((take <= x) & (pub > x)).sum() for x in ts])
runnable = np.array([((ready <= x) & (pub > x)).sum() for x in ts])
print(f"in flight (taken, not published): mean {inflight.mean():.1f}; of them READY (references done): mean {runnable.mean():.1f}, "
      f"share of time with <= 2 ready: {100 * (runnable <= 2).mean():.0f} %, <= 4: {100 * (runnable <= 4).mean():.0f} %, >= 8: {100 * (runnable >= 8).mean():.0f} %")
# by temporal layer of the GOP (poc mod 32)
g = 32
layer = np.array([0 if p % g == 0 else int(np.log2(g)) - int(np.log2(np.gcd(int(p) % g, g))) for p in poc])
for l in sorted(set(layer)):
    m = (layer == l) & ~I
    if m.any():
        print(f"  layer {l}: {m.sum():4d} pictures, waited for refs {100 * waited[m].mean():3.0f} %, latency after ready {1e3 * after[m].mean():.2f} ms (median {1e3 * np.median(after[m]):.2f})")

# ---- where the latency after ready goes (pictures that waited for their references) ----
m = waited & ~I
print("pictures that waited for their references, mean ms:")
print(f"  last reference published -> this thread has its references (wake-up)   {1e3 * (inhand - ready_refs)[m].mean():.3f}  (median {1e3 * np.median((inhand - ready_refs)[m]):.3f})")
print(f"  -> launches enqueued                                                    {1e3 * (launched - inhand)[m].mean():.3f}")
print(f"  -> complete on the device (ovhip_job_wait returned)                     {1e3 * (done - launched)[m].mean():.3f}  (median {1e3 * np.median((done - launched)[m]):.3f})")
print(f"  -> published                                                            {1e3 * (pub - done)[m].mean():.3f}")
print(f"  -> output done, thread free                                             {1e3 * (returned - pub)[m].mean():.3f}")
m = ~waited & ~I
if m.any():
    print("pictures whose references were done when they were taken, mean ms:")
    print(f"  taken -> references in hand (prepare, uploads enqueued)                 {1e3 * (inhand - take)[m].mean():.3f}")
    print(f"  -> launches enqueued                                                    {1e3 * (launched - inhand)[m].mean():.3f}")
    print(f"  -> complete on the device (uploads + kernels)                           {1e3 * (done - launched)[m].mean():.3f}")
    print(f"  -> published                                                            {1e3 * (pub - done)[m].mean():.3f}")
