"""pictures/s of every step (intra period) of one timed region, from the timeline bench.py --trace wrote: is the rate steady within a run?
python tools/debug/step_rates.py gpurun_out/trace.npy [pictures per step]"""
import sys
import numpy as np
t = np.load(sys.argv[1])
pps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pub = np.sort(t[:, 2])
n = len(pub) // pps
edges = pub[pps - 1::pps][:n]
dur = np.diff(np.concatenate([[0.0], edges]))
print("steps", n, "rate per step:", " ".join(f"{pps / d:.0f}" for d in dur))
print(f"mean {len(pub) / pub[-1]:.0f}  min {pps / dur.max():.0f}  max {pps / dur.min():.0f}")
