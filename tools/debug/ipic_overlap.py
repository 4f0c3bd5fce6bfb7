"""Debug: what else runs while an intra picture's flow launches run (rocprofv3 --kernel-trace results db)."""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = list(c.execute("select start, end, grid_x, name, queue_id, stream_id from kernels order by start"))
flow = [(s, e, g, q, st) for s, e, g, n, q, st in rows if "k_intra_flow(" in n]
# an I picture: consecutive flow launches of >= 8192 workgroups on one stream
big = [f for f in flow if f[2] >= 8192 * 64]
if not big:
    print("no big flow launches"); sys.exit()
# group by stream, split on gaps > 20 ms
groups = []
for f in big:
    if groups and groups[-1][-1][4] == f[4] and f[0] - groups[-1][-1][1] < 5e6:
        groups[-1].append(f)
    else:
        groups.append([f])
groups = [g for g in groups if len(g) >= 8]
print("I-picture ordered passes found:", len(groups))
for g in groups[2:8]:
    t0, t1 = g[0][0], g[-1][1]
    others = [(s, e, n, q, st) for s, e, gx, n, q, st in rows if e > t0 and s < t1 and not (st == g[0][4])]
    busy = sum(min(e, t1) - max(s, t0) for s, e, n, q, st in others) / 1e3
    byq = {}
    for s, e, n, q, st in others:
        byq[q] = byq.get(q, 0) + (min(e, t1) - max(s, t0)) / 1e3
    # time covered by at least one other kernel
    ev = sorted([(max(s, t0), 1) for s, e, n, q, st in others] + [(min(e, t1), -1) for s, e, n, q, st in others])
    cov, depth, last = 0, 0, t0
    for t, d in ev:
        if depth > 0: cov += t - last
        depth += d; last = t
    print(f"I pass {(t1 - t0) / 1e3:8.1f} us on queue {g[0][3]} stream {g[0][4]}: {len(g)} launches; other kernels: {len(others)} launches, {busy:8.1f} us of kernel time, "
          f"covering {cov / 1e3:8.1f} us of the interval; per queue {dict((k, round(v)) for k, v in byq.items())}; streams {len(set(o[4] for o in others))}")
