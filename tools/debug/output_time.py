"""Debug: time of the device output path at 4K (pack + D2H, row digests)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from openvvc_amd import engine
ctx = engine.Context(0)
w, h = 3840, 2160
rs = np.random.RandomState(1)
pic = ctx.upload_pic(rs.randint(0, 1024, (h, w)).astype(np.uint16), rs.randint(0, 1024, (h // 2, w // 2)).astype(np.uint16), rs.randint(0, 1024, (h // 2, w // 2)).astype(np.uint16))
for name, fn in (("output (pack + D2H to pageable numpy)", lambda: pic.output((0, 0, 0, 0))), ("digest (MD5 tree on device + 8.6 KB D2H + host MD5)", lambda: pic.digest((0, 0, 0, 0))),
                 ("download (3 pitched plane copies)", lambda: pic.download())):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(10): fn()
    print(name, "%.3f ms" % ((time.perf_counter() - t0) * 100))
