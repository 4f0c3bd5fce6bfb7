"""How many pictures' worth of uploads per second does this box take?  A picture = 8.7 MB out of page-locked memory in 10 pieces (the
flush of bench.py's 4K picture) / in 3 pieces / in 1 piece, on 1, 4 and 16 streams at once (one host thread per stream, as the
frame threads do it); and the same with 5.7 MB."""
import sys, time, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

dev = torch.device("cuda", 0)
SIZES = {10: [2_600_000, 1_120_000, 1_060_000, 1_550_000, 1_550_000, 510_000, 64_000, 130_000, 60_000, 56_000]}


def run(total, pieces, streams, pics=60):
    frac = [s / sum(SIZES[10]) for s in SIZES[10]] if pieces == 10 else [1.0 / pieces] * pieces
    hs = [[torch.empty(int(total * f), dtype=torch.uint8).pin_memory() for f in frac] for _ in range(streams)]
    ds = [[torch.empty(int(total * f), dtype=torch.uint8, device=dev) for f in frac] for _ in range(streams)]
    ss = [torch.cuda.Stream(dev) for _ in range(streams)]

    def worker(k):
        with torch.cuda.stream(ss[k]):
            for _ in range(pics):
                for h, d in zip(hs[k], ds[k]):
                    d.copy_(h, non_blocking=True)
                ss[k].synchronize()
    for k in range(streams):
        worker.__call__  # noqa
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(streams)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    n = pics * streams
    return n / dt, total * n / dt / 1e9, dt / pics * 1e6


for total in (8_700_000, 5_700_000):
    for pieces in (10, 3, 1):
        for streams in (1, 4, 16):
            run(total, pieces, streams, 5)
            fps, gbs, us = run(total, pieces, streams)
            print(f"{total / 1e6:.1f} MB in {pieces:2d} piece(s), {streams:2d} stream(s): {fps:8.0f} pictures/s  {gbs:6.1f} GB/s  {us:7.0f} us per picture and stream")
