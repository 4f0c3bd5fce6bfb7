"""Debug: a kernel's durations in launch order from a rocprofv3 results db, averaged over blocks of launches."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); pat = sys.argv[2]; blk = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print("columns:", cols)
rows = list(c.execute("select start, duration from kernels where name like ? order by start", (f"%{pat}%",)))
print("launches", len(rows))
for i in range(0, len(rows), blk):
    b = rows[i:i + blk]
    print(f"{i:6d} t={(b[0][0] - rows[0][0]) / 1e6:9.2f} ms  avg {sum(r[1] for r in b) / len(b) / 1e3:8.2f} us  max {max(r[1] for r in b) / 1e3:8.2f}")
