"""Host -> device copy rate out of page-locked memory on this box (what bounds the per-picture flush): one copy of
N bytes, and the same bytes as 11 copies, on one stream and on two streams at once."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch

dev = torch.device("cuda", 0)
def rate(nbytes, pieces=1, streams=1, reps=50):
    hs = [[torch.empty(nbytes // pieces, dtype=torch.uint8).pin_memory() for _ in range(pieces)] for _ in range(streams)]
    ds = [[torch.empty(nbytes // pieces, dtype=torch.uint8, device=dev) for _ in range(pieces)] for _ in range(streams)]
    ss = [torch.cuda.Stream(dev) for _ in range(streams)]
    def go():
        for k in range(streams):
            with torch.cuda.stream(ss[k]):
                for h, d in zip(hs[k], ds[k]):
                    d.copy_(h, non_blocking=True)
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt * 1e6, nbytes * streams / dt / 1e9

for nb in (64 << 10, 1 << 20, 5_600_000, 64 << 20):
    for pieces, streams in ((1, 1), (11, 1), (1, 2), (11, 2)):
        us, gbs = rate(nb, pieces, streams)
        print(f"{nb:>10} B x{streams} stream(s), {pieces:2d} piece(s): {us:9.1f} us  {gbs:7.2f} GB/s")
