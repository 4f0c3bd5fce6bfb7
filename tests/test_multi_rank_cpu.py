"""CPU, 2 ranks over gloo: the frame-sharded N > 1 path of bench.py (openvvc_amd/frames.py).  Every rank
decodes its own recorded picture (here with the oracle standing in for the HIP engine -- this test is about
the exchange protocol, not the kernels), pushes the result to the next rank where it becomes reference
picture 1 of that rank's next step, and after two steps every rank must hold exactly what a single-process
simulation of the same schedule produces."""
import hashlib
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
W, H, STEPS = 192, 128, 2


def _decode(seed, ref1_planes):
    """One step of one rank: recorded picture `seed`, reference 1 replaced by `ref1_planes` if given."""
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import oracle_pipeline
    from openvvc_amd import synth
    wl = synth.make_workload(W, H, seed)
    if ref1_planes is not None:
        wl.refs[1] = ref1_planes
    out = oracle_pipeline.decode(wl)
    return out.y.copy(), out.cb.copy(), out.cr.copy()


def _pack(planes):
    return torch.from_numpy(np.concatenate([p.ravel() for p in planes]).astype(np.int16))


def _unpack(t):
    a = t.numpy().astype(np.uint16)
    ys, cs = W * H, (W // 2) * (H // 2)
    return a[:ys].reshape(H, W), a[ys:ys + cs].reshape(H // 2, W // 2), a[ys + cs:].reshape(H // 2, W // 2)


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    from openvvc_amd import frames
    ref1 = None
    recv = torch.empty(W * H * 3 // 2, dtype=torch.int16)
    for step in range(STEPS):
        out = _decode(100 + rank, ref1)
        frames.ring_exchange(dist, _pack(out), recv, rank, world)
        ref1 = tuple(p.copy() for p in _unpack(recv))
    dist.barrier()
    q.put((rank, hashlib.md5(b"".join(p.tobytes() for p in out)).hexdigest()))
    dist.destroy_process_group()


def test_two_rank_reference_exchange(built_lib):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process simulation of the same schedule
    outs = [None] * world
    ref1 = [None] * world
    for step in range(STEPS):
        outs = [_decode(100 + r, ref1[r]) for r in range(world)]
        ref1 = [outs[(r - 1) % world] for r in range(world)]
    want = {r: hashlib.md5(b"".join(p.tobytes() for p in outs[r])).hexdigest() for r in range(world)}
    assert got == want
    assert got[0] != got[1]


def test_frame_owner():
    sys.path.insert(0, str(ROOT))
    from openvvc_amd import frames
    assert [frames.frame_owner(k, 4) for k in range(6)] == [0, 1, 2, 3, 0, 1]
