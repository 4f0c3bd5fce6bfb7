"""Random-access GOP schedule of the rcn back-end across pictures in flight and across GPUs.

The reference decodes pictures on frame threads (ovdec.c:188-248, `--framethr`); a picture may start once the pictures of
its reference picture lists are decoded (ovdpb_frame_synchro, dpb.c).  Here the same dependencies are edges between
per-picture device jobs: inside one GPU they become stream events, between GPUs a point-to-point transfer of the decoded
picture (RCCL over xGMI on GPUs, gloo in the CPU tests).  There is no collective on the data path.

Layout across GPUs: GOP g is decoded by rank g mod world ("a GOP per GPU").  In a hierarchical-B GOP every picture references
pictures of its own GOP and the key picture of the previous one, so exactly ONE picture per GOP crosses GPUs (the key picture,
to the owner of the next GOP) -- 25 MB per 32 pictures at 4K instead of one transfer per picture with a picture-interleaved
layout -- and the only cross-GPU dependency chain is key picture -> next key picture, which an intra key picture (every
`intra_period`) cuts.

Every rank derives the WHOLE schedule from (n_gops, gop_size, intra_period, world): nothing is negotiated at run time.
Transfers are issued in one global order (by producer picture), each rank enqueuing its sends and receives in that order on its
communication stream, so that paired operations match and no cycle of waits can form.
"""
from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class Picture:
    idx: int                       # position in decoding order over the whole stream
    gop: int                       # -1: the stream's first picture
    poc: int
    layer: int                     # temporal layer, 0 = key picture
    intra: bool
    refs: list = field(default_factory=list)        # idx of the reference pictures (nearest first, list 0 then list 1)
    owner: int = 0
    sends: list = field(default_factory=list)       # ranks that need this picture (other than the owner)


def gop_decode_order(gop_size: int) -> list:
    """[(poc offset in 1..gop_size, temporal layer)] in decoding order: the key picture, then the pre-order bisection the
    JVET random-access configurations use (GOP 16: 16 8 4 2 1 3 6 5 7 12 10 9 11 14 13 15)."""
    assert gop_size >= 2 and gop_size & (gop_size - 1) == 0
    out = [(gop_size, 0)]

    def rec(lo, hi, layer):
        mid = (lo + hi) // 2
        if mid == lo:
            return
        out.append((mid, layer))
        rec(lo, mid, layer + 1)
        rec(mid, hi, layer + 1)
    rec(0, gop_size, 1)
    return out


def gop_owner(g: int, world: int, gops_per_intra_period: int = 1) -> int:
    """GOP g -> rank.  Every block of `world` consecutive GOPs goes once around the ranks.  The GOPs whose key picture is an I
    picture come every gops_per_intra_period-th (every second one in the CTC setting) and an I picture costs as much as twenty B
    pictures: when that period and `world` share a factor, g % world would hand all of them to the same ranks, so the start of
    the round moves by one rank per block."""
    from math import gcd
    shift = g // world if gcd(world, max(1, gops_per_intra_period)) > 1 else 0
    return (g + shift) % world


def build_stream(n_gops: int, gop_size: int = 32, intra_period: int = 32, world: int = 1, refs_per_list: int = 2) -> list:
    """The pictures of an RA stream in decoding order with their reference pictures, owners and cross-GPU sends."""
    assert intra_period % gop_size == 0
    pics = [Picture(0, -1, 0, 0, True, [], 0)]
    by_poc = {0: 0}
    for g in range(n_gops):
        base = g * gop_size
        decoded = {base: by_poc[base]}                       # POC -> idx of what this GOP may reference: the previous key ...
        layer_of = {base: 0}
        for off, layer in gop_decode_order(gop_size):
            poc = base + off
            intra = layer == 0 and poc % intra_period == 0
            refs = []
            if not intra:
                # only pictures of a LOWER temporal layer are reference pictures (the highest layer is never referenced):
                # what makes the pictures of one layer independent of each other
                ok = [p for p in decoded if layer_of[p] < layer or layer == 0]
                before = sorted((p for p in ok if p < poc), reverse=True)[:refs_per_list]
                after = sorted(p for p in ok if p > poc)[:refs_per_list]
                refs = [decoded[p] for p in before] + [decoded[p] for p in after]
            idx = len(pics)
            pics.append(Picture(idx, g, poc, layer, intra, refs, gop_owner(g, world, intra_period // gop_size)))
            by_poc[poc] = idx
            decoded[poc] = idx                               # ... and its own pictures decoded so far
            layer_of[poc] = layer
    for p in pics:
        for r in p.refs:
            q = pics[r]
            if q.owner != p.owner and p.owner not in q.sends:
                q.sends.append(p.owner)
    return pics


def transfers(pics: list) -> list:
    """[(picture idx, src rank, dst rank)] in the global issue order (producer picture, then destination)."""
    return [(p.idx, p.owner, d) for p in pics for d in sorted(p.sends)]


def rank_program(pics: list, rank: int) -> list:
    """What `rank` enqueues, in order: ("recv", idx, src) | ("decode", idx) | ("send", idx, dst).  The communication
    operations of a rank are its part of the global transfer list, in that order; between the decodes they are issued as early
    as they can be (a receive at once, a send as soon as its picture is decoded), so a receive is always posted before the
    peer's matching send can block anything behind it."""
    comm = [("send", i, d) if s == rank else ("recv", i, s) for i, s, d in transfers(pics) if rank in (s, d)]
    prog, k, done = [], 0, set()

    def drain(until_recv=None):
        nonlocal k
        while k < len(comm):
            op = comm[k]
            if op[0] == "send" and op[1] not in done:
                assert until_recv is None, "a send of a later picture precedes a receive this picture needs"
                return
            prog.append(op)
            k += 1
            if until_recv is not None and op == until_recv:
                return
    for p in pics:
        if p.owner != rank:
            continue
        for r in sorted(p.refs):
            if pics[r].owner != rank:
                want = next(c for c in comm if c[0] == "recv" and c[1] == r)
                if want not in prog:
                    drain(want)
        prog.append(("decode", p.idx))
        done.add(p.idx)
        drain()
    assert k == len(comm)
    return prog


def check_programs(pics: list, world: int) -> None:
    """Static validation: every reference is local or received before use; per ordered rank pair, sends and receives list
    the same pictures in the same order (what point-to-point matching requires); executing the programs with blocking
    receives cannot deadlock (simulated)."""
    progs = [rank_program(pics, r) for r in range(world)]
    for r, prog in enumerate(progs):
        have = set()
        done = set()
        for op in prog:
            if op[0] == "recv":
                have.add(op[1])
            elif op[0] == "decode":
                p = pics[op[1]]
                assert all((pics[q].owner == r and q in done) or q in have for q in p.refs), f"rank {r}: picture {p.idx} decodes before its references"
                done.add(p.idx)
    for a in range(world):
        for b in range(world):
            if a == b:
                continue
            s = [op[1] for op in progs[a] if op[0] == "send" and op[2] == b]
            t = [op[1] for op in progs[b] if op[0] == "recv" and op[2] == a]
            assert s == t, f"rank pair {a}->{b}: sends {s[:8]} vs receives {t[:8]}"
    # simulation with rendezvous semantics (a send completes only together with its receive)
    pc = [0] * world
    progress = True
    while progress:
        progress = False
        for r in range(world):
            while pc[r] < len(progs[r]):
                op = progs[r][pc[r]]
                if op[0] == "decode":
                    pc[r] += 1; progress = True
                    continue
                peer = op[2]
                want = ("recv" if op[0] == "send" else "send", op[1], r)
                if pc[peer] < len(progs[peer]) and progs[peer][pc[peer]] == want:
                    pc[r] += 1; pc[peer] += 1; progress = True
                    continue
                break
    assert all(pc[r] == len(progs[r]) for r in range(world)), f"schedule deadlocks at {[progs[r][pc[r]] if pc[r] < len(progs[r]) else None for r in range(world)]}"


def critical_path(pics: list, t_decode=lambda p: 1.0, t_xfer: float = 0.0) -> float:
    """Length of the longest dependency chain (decode times + transfers between owners): the bound frame-level parallelism
    cannot beat, whatever the number of pictures in flight."""
    end = {}
    for p in pics:
        start = 0.0
        for r in p.refs:
            start = max(start, end[r] + (t_xfer if pics[r].owner != p.owner else 0.0))
        end[p.idx] = start + t_decode(p)
    return max(end.values())
