#!/usr/bin/env python
"""More chained streams of the reference's slice decoder through the HIP engine than the committed fixtures hold: for every (seed,
parameter-set variant, geometry, tiling) the prebuilt harness (oracle/_ref/gen_pipe: the reference compiled in the build container)
decodes the stream and records it through the installed shim slots ON THE GPU BOX, the device decodes it picture by picture from its own
earlier pictures, and every frame and every DMVR vector is compared (bench.py: reference_stream_on_device).  A stream the harness
refuses (64x2 ISP partitions: the reference's own result is undefined) is counted, not compared.
usage (through gpurun): python tools/gpu_pipe_sweep.py [first seed] [seeds per configuration]"""
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from openvvc_amd import capi, engine                                                    # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
CONFIGS = [  # (w, h, pictures, extra gen_pipe arguments)
    (416, 240, 5, ()), (416, 240, 5, ("variant", 1)), (264, 392, 5, ("variant", 1)), (832, 480, 5, ()), (416, 240, 9, ("qp", -6)),
    (264, 392, 5, ("tiles", 2, 2)), (416, 240, 5, ("tiles", 2, 2)), (416, 240, 5, ("tiles", 4, 1, "variant", 1)), (832, 480, 5, ("tiles", 3, 2)),
    (1920, 1080, 5, ()), (1920, 1080, 3, ("tiles", 2, 2)),
]
ctx = engine.Context(0)
tot = dict(streams=0, refused=0, pictures=0, samples_differing=0, vectors_differing=0, dmvr_calls=0)
bad = []
for w, h, n, extra in CONFIGS:
    for seed in range(first, first + per):
        r = bench.reference_stream_on_device(engine, capi, ctx, w, h, n, 0, extra=tuple(extra) + ("seed", seed))
        if r is None:
            tot["refused"] += 1
            continue
        tot["streams"] += 1; tot["pictures"] += r["pictures"]; tot["dmvr_calls"] += r["units"]["dmvr_calls"]
        tot["samples_differing"] += r["samples_differing_from_the_reference"]; tot["vectors_differing"] += r["refined_vectors_differing"]
        if r["samples_differing_from_the_reference"] or r["refined_vectors_differing"]:
            bad.append((w, h, n, extra, seed, r["samples_differing_from_the_reference"], r["refined_vectors_differing"]))
    print(f"{w}x{h} x{n} {' '.join(str(a) for a in extra) or '-'}: done, totals so far {tot}", flush=True)
ctx.close()
print(json.dumps({"totals": tot, "differing_streams": bad}))
sys.exit(1 if bad else 0)
