"""CPU: the random-access GOP schedule of the N > 1 path (openvvc_amd/gop.py): static properties for 1..8 ranks, and the schedule
EXECUTED by 2 ranks over gloo -- every rank runs its program (receive / decode / send), decoding with the oracle standing in
for the HIP engine (this test is about the schedule and the exchange, not the kernels), and every picture must come out exactly
as a single-process decode of the same stream in decoding order produces it."""
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
sys.path.insert(0, str(ROOT))
W, H = 192, 128
N_GOPS, GOP, IP = 4, 4, 8          # 17 pictures: I, then four GOPs of four, every second key picture intra


def test_gop_decode_order_and_references():
    from openvvc_amd import gop
    assert [p for p, _ in gop.gop_decode_order(16)] == [16, 8, 4, 2, 1, 3, 6, 5, 7, 12, 10, 9, 11, 14, 13, 15]
    assert [l for _, l in gop.gop_decode_order(8)] == [0, 1, 2, 3, 3, 2, 3, 3]
    pics = gop.build_stream(3, 16, 32, world=1)
    assert len(pics) == 49 and pics[0].intra and pics[0].poc == 0
    for p in pics:
        assert all(r < p.idx for r in p.refs)                       # references are decoded before
        if p.intra:
            assert not p.refs
        else:
            pocs = [pics[r].poc for r in p.refs]
            assert any(q < p.poc for q in pocs) and (p.layer == 0 or any(q > p.poc for q in pocs))
            assert all(pics[r].layer < p.layer or p.layer == 0 for r in p.refs)          # references come from lower temporal layers
    depth = {}
    for p in pics:
        depth[p.idx] = 1 + max([depth[r] for r in p.refs if pics[r].gop == p.gop], default=0)
    assert max(depth.values()) == 5                                                     # key + four B layers
    keys = [p for p in pics if p.layer == 0 and p.gop >= 0]
    assert [k.poc for k in keys] == [16, 32, 48] and [k.intra for k in keys] == [False, True, False]


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("gop_size,intra_period", [(16, 32), (32, 32), (32, 64), (4, 8)])
def test_schedule_is_consistent(world, gop_size, intra_period):
    """Only key pictures cross GPUs, one transfer per GOP; sends and receives pair up in one global order; the programs run to
    completion under rendezvous semantics."""
    from openvvc_amd import gop
    pics = gop.build_stream(3 * world + 1, gop_size, intra_period, world)
    gop.check_programs(pics, world)
    tr = gop.transfers(pics)
    assert all(pics[i].layer == 0 for i, _, _ in tr)
    if world > 1:
        assert len(tr) == 3 * world + 1 - 1 + (1 if world > 1 else 0) or len(tr) <= 3 * world + 1
        # a key picture goes to the owner of the NEXT GOP, and only there (none when that is its own rank)
        assert all(pics[i].gop == -1 or (d == gop.gop_owner(pics[i].gop + 1, world, intra_period // gop_size) and d != s) for i, s, d in tr)
        # the GOPs with an I picture are spread evenly: no rank decodes more than one of them more than another
        n_i = [sum(1 for p in pics if p.intra and p.gop >= 0 and p.owner == r) for r in range(world)]
        assert max(n_i) - min(n_i) <= 1, n_i
    else:
        assert not tr
    # frame-level parallelism: with a GOP per GPU the dependency chain is the key pictures only where they are not intra
    seq = float(len(pics))
    cp = gop.critical_path(pics)
    assert cp <= seq / min(world, 2) + gop_size or world == 1


def _decode(pics, p, planes_of):
    """Picture p of the stream: recorded picture seeded by its POC, reference pictures = the decoded pictures the schedule names."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_pipeline
    from openvvc_amd import synth
    wl = synth.make_workload(W, H, 100 + p.poc)
    if p.refs:
        for k in range(len(wl.refs)):
            wl.refs[k] = planes_of[p.refs[k % len(p.refs)]]
    out = oracle_pipeline.decode(wl)
    return out.y.copy(), out.cb.copy(), out.cr.copy()


def _pack(planes):
    return torch.from_numpy(np.concatenate([p.ravel() for p in planes]).astype(np.int16))


def _unpack(t):
    a = t.numpy().astype(np.uint16)
    ys, cs = W * H, (W // 2) * (H // 2)
    return a[:ys].reshape(H, W).copy(), a[ys:ys + cs].reshape(H // 2, W // 2).copy(), a[ys + cs:].reshape(H // 2, W // 2).copy()


def _md5(planes):
    return hashlib.md5(b"".join(p.tobytes() for p in planes)).hexdigest()


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, str(ROOT))
    from openvvc_amd import gop
    pics = gop.build_stream(N_GOPS, GOP, IP, world)
    planes, digests = {}, {}
    for op in gop.rank_program(pics, rank):
        if op[0] == "recv":
            t = torch.empty(W * H * 3 // 2, dtype=torch.int16)
            dist.recv(t, src=op[2])
            planes[op[1]] = _unpack(t)
        elif op[0] == "send":
            dist.send(_pack(planes[op[1]]), dst=op[2])
        else:
            planes[op[1]] = _decode(pics, pics[op[1]], planes)
            digests[op[1]] = _md5(planes[op[1]])
    dist.barrier()
    q.put((rank, digests))
    dist.destroy_process_group()


def test_two_rank_gop_schedule_executes(built_lib):
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, d = q.get(timeout=900)
        assert not set(d) & set(got)
        got.update(d)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, decoding order
    from openvvc_amd import gop
    pics = gop.build_stream(N_GOPS, GOP, IP, 1)
    planes, want = {}, {}
    for p in pics:
        planes[p.idx] = _decode(pics, p, planes)
        want[p.idx] = _md5(planes[p.idx])
    assert got == want and len(set(want.values())) == len(pics)
