#!/usr/bin/env python
"""Isolated launch times of the picture's kernels on one 4K B picture (and the I picture's ordered pass), one picture in flight,
device to itself, plus a bit-exactness check against the oracle -- the loop for kernel work:  gpurun -- python tools/kbench.py [--no-check]
Prints one line per launch group: microseconds per launch and the HBM-roofline fraction of its algorithmic bytes."""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np                                   # noqa: E402
import torch                                         # noqa: E402
from openvvc_amd import capi, engine, synth          # noqa: E402
import bench                                         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--ipic", action="store_true", help="also the 4K I picture (ordered pass)")
    ap.add_argument("--isp-frac", type=float, default=None, help="share of the eligible intra CUs coded with ISP (default: synth.ISP_FRAC)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    args = ap.parse_args()
    W, H = args.width, args.height
    if args.isp_frac is not None:
        synth.ISP_FRAC = args.isp_frac
    ctx = engine.Context(0)
    wl = synth.make_workload(W, H, 0x266, tools=synth.INTRA_TOOLS, intra_frac=0.12)
    alg = bench.algorithmic_bytes(wl, wl.frame_bytes)
    job = engine.Job(ctx, W, H)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(W, H)
    job.load_workload(wl)
    for _ in range(3):
        job.flush(dst, refs, None); job.wait()
    out = {}
    for name in capi.TIME_STAGES:
        job.time_stage(name)
        for _ in range(args.reps):
            job.flush(dst, refs, None); job.wait()
        s, n = job.stage_time()
        out[name] = s / max(n, 1) * 1e6
    job.time_stage(None)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        job.flush(dst, refs, None); job.wait()
    whole = (time.perf_counter() - t0) / args.reps * 1e6
    print(f"{W}x{H} B picture, one in flight: {whole:.1f} us per picture (flush + wait), launch groups:")
    tot = 0.0
    for name in capi.TIME_STAGES:
        a = alg.get(name)
        frac = f"{a / (out[name] * 1e-6) / 8e12:.3f}" if a and out[name] > 0 else "  -  "
        print(f"  {name:11s} {out[name]:8.1f} us   frac_isolated {frac}")
        tot += out[name] if name != "h2d" else 0
    print(f"  sum of kernels {tot:.1f} us")
    if not args.no_check:
        import oracle_pipeline
        import ovvc_oracle_output as oo
        ref = oracle_pipeline.decode(wl)
        got = dst.download()
        bad = [n for n, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)) if not np.array_equal(a, b)]
        print("bit-exact vs oracle:", "YES" if not bad else f"NO ({bad})")
    if args.ipic:
        wi = synth.make_workload(W, H, 0x266 + 7777, tools=synth.INTRA_TOOLS, intra_frac=1.0)
        ji = engine.Job(ctx, W, H)
        ji.load_workload(wi)
        for _ in range(2):
            ji.flush(dst, [], None); ji.wait()
        ji.time_stage("intra")
        for _ in range(10):
            ji.flush(dst, [], None); ji.wait()
        s, n = ji.stage_time()
        lv = wi.stats["n_ilevels"]
        print(f"I picture ({wi.stats.get('n_isp_cus', 0)} ISP CUs): ordered pass {s / n * 1e3:.3f} ms, {lv} levels -> {s / n * 1e6 / lv:.2f} us per level; retries {ji.stats().n_ordered_retries}")
        if not args.no_check:
            ref = oracle_pipeline.decode(wi)
            got = dst.download()
            print("I picture bit-exact vs oracle:", "YES" if all(np.array_equal(a, b) for a, b in zip(got, (ref.y, ref.cb, ref.cr))) else "NO")


if __name__ == "__main__":
    main()
