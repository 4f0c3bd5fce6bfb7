"""Parity sweep over seeds (GPU): I pictures and B pictures with intra CUs at several sizes through ovhip_job_flush vs the oracle.
python tools/debug/seed_sweep.py [first seed] [count]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np                                   # noqa: E402
from openvvc_amd import engine, synth                # noqa: E402
import oracle_pipeline                               # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ctx = engine.Context(0)
bad = 0
for k in range(count):
    seed = first + k
    w, h = [(416, 240), (832, 480), (1280, 720), (1920, 1080)][k % 4]
    frac = [1.0, 0.15, 0.4, 0.05][(k // 4) % 4] if k % 2 else 1.0
    tools = synth.INTRA_TOOLS if k % 3 else tuple(t for t in synth.INTRA_TOOLS if t != "lmcs")
    wl = synth.make_workload(w, h, seed, tools=tools, intra_frac=frac)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    if k % 5 == 4:
        job.params.flow_workers = 5
    for rep in range(2):
        job.flush(dst, refs, None); job.wait()
    got = dst.download()
    ref = oracle_pipeline.decode(wl)
    ok = all(np.array_equal(a, b) for a, b in zip(got, (ref.y, ref.cb, ref.cr)))
    print(f"seed {seed} {w}x{h} intra {frac} lmcs {'lmcs' in tools}: {'ok' if ok else 'DIFFERS'}; tasks {wl.stats['n_itasks']}, retries {job.stats().n_ordered_retries}", flush=True)
    bad += not ok
    job.close()
print("sweep:", "all identical" if not bad else f"{bad} DIFFER")
sys.exit(1 if bad else 0)
