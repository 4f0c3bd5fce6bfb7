import sqlite3, sys, json
import numpy as np
c = sqlite3.connect(sys.argv[1])
meta = json.load(open(sys.argv[2]))
d = np.array([r[0] for r in c.execute("select duration from kernels where name like '%k_intra_level%' order by start")]) / 1e3
N = meta["N"]
for i, n in enumerate(meta["names"]):
    x = d[i * N:(i + 1) * N]
    print(f"{n:14s} median {np.median(x):6.2f} us  min {x.min():6.2f}")
