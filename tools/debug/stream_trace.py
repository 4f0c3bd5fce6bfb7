"""Reads the timeline bench.py --trace wrote: where do the frame threads wait?  python tools/debug/stream_trace.py gpurun_out/trace.npy"""
import sys
import numpy as np
t = np.load(sys.argv[1])
take, sub, pub, thr, nrefs, poc, idx = t[:, 0], t[:, 1], t[:, 2], t[:, 3], t[:, 8], t[:, 9], t[:, 10]
n = len(t)
dur = pub - sub
total = pub.max()
print(f"{n} pictures in {total * 1e3:.1f} ms = {n / total:.0f} pictures/s; threads {int(thr.max()) + 1}")
I = nrefs == 0
print(f"I pictures: {I.sum()}, submit->published {dur[I].mean() * 1e3:.2f} ms avg (max {dur[I].max() * 1e3:.2f}); B pictures: {dur[~I].mean() * 1e3:.3f} ms avg, median {np.median(dur[~I]) * 1e3:.3f}, p90 {np.percentile(dur[~I], 90) * 1e3:.3f}, max {dur[~I].max() * 1e3:.2f}")
# how early was each I picture done relative to the first picture that needed it (the next picture in decoding order)?
order = np.argsort(idx)
for k in np.where(I)[0]:
    later = (idx > idx[k])
    first_b = np.where(later)[0]
    if len(first_b):
        j = first_b[np.argmin(idx[first_b])]
        print(f"  I picture idx {int(idx[k])}: taken {take[k] * 1e3:7.2f} ms, published {pub[k] * 1e3:7.2f} ms; next picture in decoding order (idx {int(idx[j])}) taken {take[j] * 1e3:7.2f} -> slack {1e3 * (take[j] - pub[k]):6.2f} ms")
# publication rate over time: 2 ms bins
bins = np.arange(0, total + 2e-3, 2e-3)
h, _ = np.histogram(pub, bins)
print("pictures published per 2 ms:", " ".join(str(int(x)) for x in h))
# slowest B pictures
slow = np.argsort(-dur * (~I))[:8]
for k in slow:
    print(f"  slow B idx {int(idx[k])} poc {int(poc[k])} thread {int(thr[k])}: submit {sub[k] * 1e3:.2f} -> published {pub[k] * 1e3:.2f} ({dur[k] * 1e3:.2f} ms)")
