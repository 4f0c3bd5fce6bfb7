"""Debug: duration and start-to-start spacing of consecutive k_intra_level dispatches in a rocprofv3 results db."""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
rows = list(c.execute("select start, end, grid_x, grid_y, grid_z from kernels where name like '%k_intra_level%' order by start"))
a = np.array(rows, dtype=np.int64)
dur = (a[:, 1] - a[:, 0]) / 1e3
gap = (a[1:, 0] - a[:-1, 1]) / 1e3
pitch = (a[1:, 0] - a[:-1, 0]) / 1e3
ok = gap < 50
print("n", len(a), "dur mean/median/p90", dur.mean(), np.median(dur), np.percentile(dur, 90))
print("gap (end->next start) median", np.median(gap[ok]), "mean", gap[ok].mean(), " pitch median", np.median(pitch[ok]), "mean", pitch[ok].mean())
g = a[:, 2] // 64
for lo, hi in ((0, 8), (8, 32), (32, 128), (128, 1024), (1024, 1 << 30)):
    m = (g >= lo) & (g < hi)
    if m.any(): print(f"tasks [{lo},{hi}): n={m.sum()} dur median {np.median(dur[m]):.2f} mean {dur[m].mean():.2f}")
