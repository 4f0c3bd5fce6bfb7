#!/usr/bin/env python
"""How busy was the device?  From a rocprofv3 kernel trace (rocpd SQLite): over the last `frac` of the dispatches (the timed steps of
a bench.py run) -- span, time with at least one kernel running, mean number of kernels running, per hardware queue: busy share and
the gaps between consecutive kernels; memory copies in the same window when they were traced.
usage: tools/trace_gaps.py <results.db> [frac]"""
import sqlite3
import sys

import numpy as np


def cols(c, t):
    return [r[1] for r in c.execute(f"pragma table_info({t})")]


def main(path, frac=0.55):
    c = sqlite3.connect(path)
    kc = cols(c, "kernels")
    qcol = "queue_id" if "queue_id" in kc else None
    scol = "stream_id" if "stream_id" in kc else None
    sel = "start, end, name" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
    rows = [r for r in c.execute(f"select {sel} from kernels order by start")]
    rows = [r for r in rows if "k_spin" not in r[2] and "k_nop" not in r[2]]
    n0 = int(len(rows) * (1 - frac))
    rows = rows[n0:]
    st = np.array([r[0] for r in rows], float); en = np.array([r[1] for r in rows], float)
    t0, t1 = st.min(), en.max()
    span = t1 - t0
    # union of the intervals
    order = np.argsort(st)
    busy, cur_s, cur_e = 0.0, st[order[0]], en[order[0]]
    for i in order[1:]:
        if st[i] > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = st[i], en[i]
        else:
            cur_e = max(cur_e, en[i])
    busy += cur_e - cur_s
    print(f"{len(rows)} dispatches over {span / 1e6:.1f} ms: a kernel running {100 * busy / span:.1f} % of the time, "
          f"{(en - st).sum() / span:.2f} kernels running on average")
    flow = np.array(["k_intra_flow" in r[2] for r in rows])
    print(f"  without k_intra_flow: {((en - st)[~flow]).sum() / span:.2f} kernels running on average; k_intra_flow alone {((en - st)[flow]).sum() / span:.2f}")
    for label, idx in (("queue", 3), ("stream", 4)):
        ids = sorted(set(r[idx] for r in rows))
        if len(ids) <= 1:
            continue
        print(f"  by {label} ({len(ids)}):")
        for q in ids:
            m = np.array([r[idx] == q for r in rows])
            s, e = st[m], en[m]
            o = np.argsort(s); s, e = s[o], e[o]
            gaps = s[1:] - np.maximum.accumulate(e)[:-1]
            gaps = gaps[gaps > 0]
            if label == "stream" and len(ids) > 8:
                continue
            print(f"    {label} {q}: {m.sum():6d} dispatches, busy {100 * (e - s).sum() / span:5.1f} % (sum of durations), gaps: "
                  f"median {np.median(gaps) / 1e3 if len(gaps) else 0:.1f} us, mean {gaps.mean() / 1e3 if len(gaps) else 0:.1f} us, total {100 * gaps.sum() / span:.1f} %")
    try:
        mc = cols(c, "memory_copies")
        if mc:
            cp = [r for r in c.execute("select start, end, size, name from memory_copies") if r[0] >= t0 and r[1] <= t1]
            if cp:
                d = np.array([r[1] - r[0] for r in cp], float); b = np.array([r[2] for r in cp], float)
                print(f"  memory copies in the window: {len(cp)}, {b.sum() / 1e6:.0f} MB, sum of durations {100 * d.sum() / span:.1f} % of the span, "
                      f"mean {d.mean() / 1e3:.1f} us, {b.sum() / d.sum():.2f} GB/s while copying")
                names = sorted(set(r[3] for r in cp))
                for nm in names:
                    m = np.array([r[3] == nm for r in cp])
                    print(f"    {nm}: {m.sum()} copies, {b[m].sum() / 1e6:.0f} MB, mean {d[m].mean() / 1e3:.1f} us")
    except sqlite3.Error as ex:
        print("  (no memory copy table:", ex, ")")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.55)
