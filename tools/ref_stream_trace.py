#!/usr/bin/env python
"""What tools/profile_round.sh traces as the SECOND kernel trace of a round: the reference decoder's own stream (oracle/_ref/gen_pipe: 17
chained 3840x2160 pictures parsed by the reference's slicedec.c, recorded through the installed shim slots) decoded on the device -- one
picture at a time, then 16 in flight through the C stream driver -- and nothing else, so that its per-kernel averages are not mixed into
those of the synthetic headline stream."""
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
spec = importlib.util.spec_from_file_location("bench", ROOT / "bench.py")
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from openvvc_amd import capi, engine  # noqa: E402

ctx = engine.Context(0)
r = bench.reference_stream_on_device(engine, capi, ctx, 3840, 2160, 17, 4)
print(json.dumps({k: v for k, v in (r or {}).items() if k != "what"}))
