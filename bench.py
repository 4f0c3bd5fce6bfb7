#!/usr/bin/env python
"""bench.py -- decoded frames/s of the MI355X rcn back-end, one DECODE STEP per picture.

A "step" is what the reference-side shim does at the end of a parsed picture (shim/rcn_hip.c flush_picture ->
ovhip_job_flush, all in C): asynchronous H2D of the picture's recorded command buffers + coefficient arena + deblocking
edge lists + filter parameters out of page-locked memory, the launch chain of the whole rcn path (prediction incl.
BDOF / DMVR / affine-PROF / GPM / CIIP + LMCS, inverse quantisation / LFNST / transforms + residual, deblocking, SAO,
ALF / CC-ALF; intra prediction of the picture's intra CUs as a dependency-ordered pass), and the D2H of the DMVR-refined
motion vectors.  The workload is a synthetic recorded 3840x2160 10-bit 4:2:0 random-access stream (BASELINE.json
configs[3]): B pictures with `--intra-frac` of their CUs intra, and `--i-sets` of the picture sets an I picture.

The steps are the pictures of a random-access stream in decoding order (openvvc_amd/gop.py: GOP `--gop`, hierarchical B, an I
picture every `--intra-period`): a picture's reference pictures ARE the decoded pictures its reference lists name, so a picture
starts when they are done (stream events) -- the dependency structure a decoder's frame threads live with.  Nothing is
replayed out of cache: every position of the GOP has its own command buffers, job and destination picture (distinct addresses,
`--contents` distinct recorded pictures; working set several times the 256 MiB Infinity Cache).  `--in-flight S` pictures are
in flight per GPU (one HIP stream + one host thread each: the reference's frame threads, ovdec.c:188-248).

N > 1: one process per GPU, a GOP per GPU; the only picture of a GOP another GPU needs is its key picture, sent to the owner
of the next GOP with RCCL point-to-point on a communication stream (no collective on the data path, nothing waited for on the
host).  A step = one intra period of the stream (--intra-period pictures: every step is the same work, so any --steps measures
the steady state; a picture count that is not a multiple of it would over- or under-represent the I picture, which the stream
fully exposes).  Every rank decodes --steps intra periods: the stream grows with N ("weak").

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

KNAME = {"mc": "k_mc2", "mcxa": "k_mcxa", "itx_luma": "k_itx_all (luma commands)", "lmcs_scale": "k_lmcs_scale",
         "itx_chroma": "k_itx_all (chroma commands + inverse-LMCS rider)", "dbf": "k_dbf_list<0> + k_dbf_list<1>",
         "sao": "k_sao", "alf": "k_alf", "intra": "k_intra_flow", "h2d": "H2D copies"}


def algorithmic_bytes(wl, S):
    """SURVEY 8d terms split per launch group, from the actual command buffers of one recorded picture."""
    from openvvc_amd import capi
    tb = wl.tb_cmds
    area = lambda a: int((a["w"].astype(np.int64) * a["h"]).sum()) if a is not None and len(a) else 0
    nref_area = lambda a: int((a["w"].astype(np.int64) * a["h"] * np.where(a["dir"] == 3, 2, 1)).sum()) if a is not None and len(a) else 0

    def itx_bytes(c):
        n_samples = 1 << (c["log2_w"].astype(np.int64) + c["log2_h"])
        raster = (c["kind"] & 0x80) != 0
        sbs = np.array([bin(int(m)).count("1") for m in c["sig_sb_map"]], np.int64)
        coef = np.where(raster, 2 * n_samples, 32 * sbs).sum()
        covered = n_samples.sum() + n_samples[c["plane2"] != 0xff].sum()
        return int(coef + c.nbytes + 2 * 2 * covered)

    def intra_bytes(t):
        """per ordered task: the block written once per plane, its residual read, the two reference arms (2w + 2h + 1
        samples) read, the task itself; a cross-component task also reads the co-located luma block (4x the area)."""
        if t is None or not len(t):
            return 0
        a = (1 << (t["log2_w"].astype(np.int64) + t["log2_h"]))
        arms = 2 * ((1 << t["log2_w"].astype(np.int64)) + (1 << t["log2_h"].astype(np.int64))) + 1
        luma = t["kind"] == capi.IT_LUMA
        pred = (t["kind"] == capi.IT_LUMA) | (t["kind"] == capi.IT_CHROMA)
        lm = (t["kind"] == capi.IT_CHROMA) & (t["mode"] >= 67)
        planes = np.where(luma, 1, 2)
        return int((planes * a * (2 + 2) + np.where(pred, planes * arms * 2, 0) + np.where(lm, 8 * a, 0)).sum() + t.nbytes)

    fused = wl.mc_units[((wl.mc_units["flags"] & 128) == 0) & (wl.mc_units["aux"] != 0)]
    ev, eh = wl.dbf_edges
    alf_tables = sum(np.asarray(wl.alf[k]).nbytes for k, _ in capi.ALF_TABLES)
    return {
        "mc": 3 * (nref_area(wl.mc_units) + area(wl.mc_units) + area(fused)) + wl.mc_units.nbytes,
        "mcxa": 3 * 3 * area(wl.mcx_units) + wl.mcx_units.nbytes + 16 * len(wl.mcx_units)
                + 3 * (nref_area(wl.aff_units) + area(wl.aff_units)) + wl.aff_units.nbytes + wl.aff_side.nbytes,
        "itx_luma": itx_bytes(tb[:wl.n_luma_cmds]),
        "lmcs_scale": len(wl.lmcs_regions) * (128 * 2 + 8 + 2) if wl.lmcs_regions is not None else 0,
        "itx_chroma": itx_bytes(tb[wl.n_luma_cmds:]) + (2 * 2 * wl.w * wl.h if wl.lmcs is not None else 0),
        "dbf": 2 * S + ev.nbytes + eh.nbytes,
        "sao": 2 * S + wl.sao_params.nbytes,
        "alf": 2 * S + alf_tables,
        "intra": intra_bytes(wl.itasks),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4,
                    help="timed steps; a step = one intra period of the stream (--intra-period pictures: its I picture and the B "
                         "pictures of its GOPs), so that every step is the same work and any count measures the steady state")
    ap.add_argument("--warmup", type=int, default=1, help="untimed steps before (intra periods)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--seed", type=int, default=0x266)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated-survey", action="store_true",
                    help="skip the one-picture-in-flight survey: every launch of the process then runs in the timed configuration "
                         "(what tools/profile_round.sh traces, so that rocprofv3's per-kernel averages are of that configuration)")
    ap.add_argument("--in-flight", type=int, default=16, help="pictures in flight per GPU (one HIP stream + one host thread each)")
    ap.add_argument("--contents", type=int, default=2, help="distinct recorded pictures (seeds) among the sets")
    ap.add_argument("--intra-frac", type=float, default=0.12, help="share of intra CUs in the B pictures")
    ap.add_argument("--gop", type=int, default=32, help="GOP size (hierarchical B, JVET random-access decoding order) = picture sets of the rotation")
    ap.add_argument("--intra-period", type=int, default=64, help="every key picture at a multiple of this POC is an I picture (a multiple of --gop; JVET CTC: about one second, 64 at 50 / 60 Hz)")
    ap.add_argument("--intra-ctu", action="store_true", help="ordered pass as the one-launch CTU wavefront instead of one launch per level")
    ap.add_argument("--intra-levels", action="store_true", help="ordered pass as one launch per level instead of one launch with per-unit dependency flags")
    ap.add_argument("--device-waits", action="store_true", help="reference pictures as stream waits (barrier packets) instead of host waits before the launches")
    ap.add_argument("--trace-gop", action="store_true", help="debug: host timeline of the pictures of the last run on stderr")
    ap.add_argument("--gop-rotation", type=int, default=1,
                    help="GOPs of picture sets / destination buffers in the rotation: with 1 a picture of GOP g + 1 overwrites the buffer of the same "
                         "position of GOP g and waits for its readers, which bounds the look-ahead to one GOP whatever --in-flight says")
    ap.add_argument("--check", type=int, default=0, metavar="N",
                    help="after the measurement: decode the first N pictures of the stream twice from the same start -- --in-flight pictures at a "
                         "time, then one at a time -- and compare the device digests (ovhip_pic_digest) of every picture")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = every rank decodes --steps intra periods (the stream grows with N); strong = the stream is --steps intra "
                         "periods in total, its GOPs dealt to the ranks (each rank times 1/N of the pictures)")
    ap.add_argument("--host-threads", type=int, default=-1, help="host threads issuing the flushes (-1: one per picture in flight)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from openvvc_amd import capi, engine, gop, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    # OVVC_BENCH_DEBUG_GLOO=1 (development only): all ranks on GPU 0, pictures exchanged through host memory over gloo -- runs
    # the N > 1 control flow (schedule, communication thread, events) on a one-GPU box.  Never set by the driver.
    debug_gloo = os.environ.get("OVVC_BENCH_DEBUG_GLOO") == "1"
    if debug_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if debug_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    W, H = args.width, args.height
    S = max(1, args.in_flight)
    G = args.gop                                   # pictures per GOP = picture sets of the rotation
    IP = args.intra_period
    PPS = IP if IP > 0 else G                      # pictures per step: one intra period (its I picture + the B pictures of its GOPs)
    K = G
    tools = synth.INTRA_TOOLS if args.intra_frac > 0 else synth.ALL_TOOLS
    wls = [synth.make_workload(W, H, args.seed + 1000 * c + rank, tools=tools, intra_frac=args.intra_frac)
           for c in range(max(1, min(args.contents, K)))]
    n_b = len(wls)
    wls.append(synth.make_workload(W, H, args.seed + 7777 + rank, tools=synth.INTRA_TOOLS, intra_frac=1.0))      # the I picture
    FB = wls[0].frame_bytes
    n_ref_slots = len(wls[0].refs)

    # ---- the stream: an RA sequence of GOPs (openvvc_amd/gop.py), GOP g decoded by rank g mod world.  Position j of the
    # GOP's decoding order <-> picture set j (own command buffers, job, destination picture at its own address); the key
    # picture (j = 0) rotates over three buffers because the previous GOP's pictures still read the previous key.
    ctxs = [engine.Context(local_rank) for _ in range(S)]
    ext = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None

    def torch_pic(ctx, planes=None):
        t = torch.zeros(H * W * 3 // 2, dtype=torch.int16, device=dev)     # zero-filled like ovhip_pic_alloc's: no sample may carry bit 15 in
        ysz, csz = H * W, (H // 2) * (W // 2)
        s = capi.Pic(t.data_ptr(), t.data_ptr() + 2 * ysz, t.data_ptr() + 2 * (ysz + csz), W, H, W, W // 2)
        p = engine.DevPic(ctx, s, owns=False)
        if planes is not None:
            p.upload(*planes)
        return t, p

    class Set:
        pass

    def new_job(wl):
        st = Set()
        st.wl = wl
        st.job = engine.Job(ctxs[0], W, H)
        st.job.load_workload(wl)
        st.lock = threading.Lock()
        return st

    order = gop.gop_decode_order(G)
    R = max(1, args.gop_rotation)                                                            # GOPs in the rotation
    NK = R + 2                                                                               # key pictures alive at a time
    sets = [new_job(wls[n_b]) if j % K == 0 else new_job(wls[j % n_b]) for j in range(R * K)]     # position 0: the key picture as I picture
    key_i = [sets[0]] + [new_job(wls[n_b]) for _ in range(NK - 1)]                           # consecutive key pictures overlap: own jobs
    key_b = [new_job(wls[0]) for _ in range(NK)] if IP > G else None                         # ... and as inter key pictures
    all_jobs = sets + key_i[1:] + (key_b if key_b else [])
    bufs_b = [torch_pic(ctxs[0]) for _ in range(R * K)]                # destination of GOP position j (j >= 1) of rotation slot r: [r * K + j]
    bufs_key = [torch_pic(ctxs[0], wls[0].refs[k % n_ref_slots]) for k in range(NK)]
    bufs_recv = [torch_pic(ctxs[0], wls[0].refs[k % n_ref_slots]) for k in range(2)] if world > 1 else []
    torch.cuda.synchronize(dev)
    working_set = (R * K + NK + len(bufs_recv)) * FB + len(all_jobs) * FB            # destinations + each job's SAO picture

    lv = capi.STAGE_INTRA_CTU if args.intra_ctu else (capi.STAGE_INTRA_LEVELS if args.intra_levels else 0)
    nthreads = S if args.host_threads < 0 else max(1, min(args.host_threads, S))
    second_passes = [0]            # pictures ovhip_job_wait had to decode a second time (flow launch gave up), whole run
    one_at_a_time = [False]        # survey of the kernels alone on the device: one host thread, one picture in flight
    gop_base = [0]                                 # GOPs this rank has decoded so far (keeps the key-buffer rotation going)
    keep_work = []

    def run_steps(n, resident=False, digests=None):
        """The next n pictures of this rank's share of the stream, in decoding order.  One host thread per picture in flight,
        each with its own HIP stream; a thread that became free takes the next picture (the reference's frame threads,
        ovdec.c:188-248) and, like the shim's flush_picture, waits for it before it takes another.  A picture starts when the
        pictures of its reference lists are decoded (stream events; RCCL point-to-point for a key picture decoded on another
        GPU), and not before the readers of the buffer it overwrites are done."""
        n_gops_rank = (n + G - 1) // G
        pics = gop.build_stream(n_gops_rank * world, G, IP, world)
        mine = [p for p in pics if p.owner == rank and p.gop >= 0][:n]
        first_gop = gop_base[0]
        gop_base[0] += n_gops_rank
        # buffers: picture idx -> (tensor, DevPic)
        buf = {}
        local_gop = lambda p: first_gop + p.gop // world
        for p in pics:
            if p.gop < 0:
                # the picture before the first GOP: for rank 0 the key buffer its previous GOP left, else whatever is there
                if rank == 0:
                    buf[p.idx] = bufs_key[(first_gop - 1) % NK]
            elif p.owner == rank:
                buf[p.idx] = bufs_key[local_gop(p) % NK] if p.layer == 0 else bufs_b[(local_gop(p) % R) * K + order.index((p.poc - p.gop * G, p.layer))]
            elif rank in p.sends:
                buf[p.idx] = bufs_recv[(first_gop + (p.gop + 1) // world) % 2]
        # who reads what (on this rank), who occupied a buffer before
        readers = {}
        for p in mine:
            for r in p.refs:
                readers.setdefault(r, []).append(p.idx)
        prog = gop.rank_program(pics, rank)
        issued = {p.idx: threading.Event() for p in pics}
        done_ev = {}
        prev_occ, occ = {}, {}
        for op in prog:
            if op[0] in ("decode", "recv") and op[1] in buf:
                key = buf[op[1]][0].data_ptr()
                if key in occ:
                    prev_occ[op[1]] = occ[key]
                occ[key] = op[1]
        for p in pics:
            if p.gop < 0 or (p.owner != rank and rank not in p.sends) or p.idx not in [q.idx for q in mine] and p.owner == rank:
                issued[p.idx].set()                   # not part of this run: final already
        mine_idx = {p.idx for p in mine}
        errs = []

        def wait_for(stream, idxs, collect=None):
            """Host: until the producers' flushes have been enqueued.  Device: `stream` behind their completion events -- or,
            with `collect`, the events are handed to the flush, which waits for them after its uploads."""
            for q in idxs:
                if q in mine_idx or (pics[q].owner != rank and q in buf):
                    issued[q].wait()
                    ev = done_ev.get(q)
                    if ev is not None:
                        if collect is not None:
                            collect.append(ev)
                        else:
                            stream.wait_event(ev)

        def decode(p, slot):
            t_pull = time.perf_counter()
            j = order.index((p.poc - p.gop * G, p.layer))
            st = sets[(local_gop(p) % R) * K + j] if j else (key_i[local_gop(p) % NK] if (p.intra or key_b is None) else key_b[local_gop(p) % NK])
            stream = ext[slot]
            evs = []
            wait_for(stream, p.refs, evs)
            if p.idx in prev_occ:
                wait_for(stream, [prev_occ[p.idx]] + readers.get(prev_occ[p.idx], []), evs)
            refs = [buf[p.refs[k % len(p.refs)]][1] for k in range(n_ref_slots)] if p.refs else []
            t_dep = time.perf_counter()
            with st.lock:
                st.job.bind(ctxs[slot])
                t_bind = time.perf_counter()
                st.job.params.stages = ((capi.STAGE_ALL | capi.STAGE_RESIDENT) if resident else capi.STAGE_ALL) | lv
                # the frame thread waits for its reference pictures itself (ovdpb_frame_synchro), after its uploads are under way:
                # on the host, in the flush (wait_on_host) -- or, --device-waits, as barriers in the stream
                handles = (C.c_void_p * max(1, len(evs)))(*[e.cuda_event for e in evs])
                st.job.params.wait_events = C.cast(handles, C.POINTER(C.c_void_p))
                st.job.params.n_wait_events = len(evs)
                st.job.params.wait_on_host = 0 if args.device_waits else 1
                st.job.params.before_launch = None
                st.job.flush(buf[p.idx][1], refs, None)
                ev = torch.cuda.Event()
                stream.record_event(ev)
                done_ev[p.idx] = ev
                issued[p.idx].set()
                t_iss = time.perf_counter()
                st.job.wait()
                second_passes[0] += int(st.job.stats().n_ordered_retries)
                if digests is not None:
                    out = (C.c_uint8 * 16)()
                    win = capi.Window(0, 0, 0, 0)
                    ctxs[slot]._chk(ctxs[slot].lib.ovhip_pic_digest(ctxs[slot].h, C.byref(buf[p.idx][1].s), C.byref(win), out), "pic_digest")
                    digests[p.idx] = bytes(out)
            if args.trace_gop:
                trace.append((p.idx, p.poc, p.layer, slot, t_pull, t_dep, t_bind, t_iss, time.perf_counter()))

        nxt, nlock = [0], threading.Lock()

        def worker(slot):
            try:
                while True:
                    with nlock:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(mine):
                        return
                    decode(mine[i], slot)
            except Exception as e:          # noqa: BLE001
                errs.append(e)
                for evt in issued.values():
                    evt.set()

        def comm():
            """This rank's sends and receives in the global transfer order, on the communication stream; nothing is waited
            for on the host except that the producing flush has been enqueued."""
            try:
                with torch.cuda.stream(comm_stream):
                    for op in prog:
                        if op[0] == "send" and op[1] in mine_idx:
                            wait_for(comm_stream, [op[1]])
                            if debug_gloo:
                                comm_stream.synchronize()
                                dist.send(buf[op[1]][0].cpu(), op[2])
                                continue
                            keep_work.append(dist.isend(buf[op[1]][0], op[2]))
                        elif op[0] == "recv" and op[1] in buf and any(op[1] in q.refs for q in mine):
                            if op[1] in prev_occ:
                                wait_for(comm_stream, [prev_occ[op[1]]] + readers.get(prev_occ[op[1]], []))
                            if debug_gloo:
                                comm_stream.synchronize()
                                t_host = torch.empty(buf[op[1]][0].shape, dtype=torch.int16)
                                dist.recv(t_host, op[2])
                                buf[op[1]][0].copy_(t_host)
                            else:
                                w = dist.irecv(buf[op[1]][0], op[2])
                                w.wait()      # NCCL: orders the communication stream behind the transfer, the host does not block
                                keep_work.append(w)
                            ev = torch.cuda.Event()
                            comm_stream.record_event(ev)
                            done_ev[op[1]] = ev
                            issued[op[1]].set()
            except Exception as e:          # noqa: BLE001
                errs.append(e)
                for evt in issued.values():
                    evt.set()

        trace = []
        th = [threading.Thread(target=worker, args=(s,)) for s in range(1 if one_at_a_time[0] else nthreads)]
        if world > 1:
            th.append(threading.Thread(target=comm))
        [t.start() for t in th]
        [t.join() for t in th]
        del keep_work[:-64]
        if errs:
            raise errs[0]
        if args.trace_gop and rank == 0:
            t0 = min(t[4] for t in trace)
            for t in sorted(trace)[:2 * G]:
                print("pic %3d poc %3d L%d slot %2d  pull %7.2f  deps %7.2f  bound %7.2f  issued %7.2f  done %7.2f ms" %
                      (t[0], t[1], t[2], t[3], *[(x - t0) * 1e3 for x in t[4:]]), file=sys.stderr)

    def barrier():
        if world > 1:
            dist.barrier()
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize(dev)

    def timed(n, resident=False):
        barrier()
        t0 = time.perf_counter()
        run_steps(n, resident)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=None if debug_gloo else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def set_timer(name):
        for st in all_jobs:
            st.job.time_stage(name)

    def read_timer():
        tot, cnt = 0.0, 0
        for st in all_jobs:
            s, n = st.job.stage_time()
            tot += s; cnt += n
        return tot / max(cnt, 1)

    run_steps(max(args.warmup * PPS, (2 if key_b else 1) * K * max(R, NK if key_b else 1)))      # every picture set flushed at least once
    barrier()
    all_stats = [st.job.stats() for st in sets]          # of full (non-resident) flushes
    flush_stats = all_stats[-1]                          # a B picture
    mean_stat = lambda f: float(np.mean([getattr(a, f) for a in all_stats]))

    # ---- untimed survey IN THE TIMED CONFIGURATION (same rotation, same pictures in flight): each launch group bracketed
    # in turn by a HIP-event pair on its stream (bracketing all of them at once would cost ~80 us of stream time per picture)
    stats0 = flush_stats
    present = ["mc", "mcxa", "itx_luma", "lmcs_scale", "itx_chroma", "intra", "dbf", "sao", "alf", "h2d"]
    if not stats0.n_regions:
        present.remove("lmcs_scale")
    if not any(len(w.itasks) for w in wls):
        present.remove("intra")
    survey = {}
    for name in present:
        set_timer(name)
        run_steps(3 * PPS)               # three intra periods per group: the pick is steadier than with one
        barrier()
        survey[name] = read_timer()
    kern = {k: v for k, v in survey.items() if k != "h2d"}
    # The roofline kernel = the largest launch group among the kernels the HBM roofline applies to.  The ordered pass is a
    # dependency chain: hops x latency per hop, neither bytes nor flops bound it; it is reported beside the roofline
    # ("ordered_pass"), not as its subject.
    dom = max((k for k in kern if k != "intra"), key=kern.get)
    # the same groups with ONE picture in flight (same rotation, nothing resident): what a launch takes when it has the device
    # to itself -- the timed configuration stretches every launch by the 15 other pictures it shares the device with
    isolated = {}
    if (rank == 0 or world > 1) and not args.no_isolated_survey:
        one_at_a_time[0] = True
        for name in present:
            if name == "h2d":
                continue
            set_timer(name)
            run_steps(K)
            barrier()
            isolated[name] = read_timer()
        one_at_a_time[0] = False
    if world > 1:
        pick = torch.tensor([present.index(dom)], dtype=torch.int64, device=None if debug_gloo else dev)
        dist.broadcast(pick, 0)
        dom = present[int(pick.item())]

    # ---- timed region: EXACTLY --steps decode steps, only the dominant launch group bracketed
    set_timer(dom)
    # a step = PPS pictures = one intra period.  weak: --steps intra periods per rank, each rank its own stream; strong: --steps
    # intra periods of ONE stream in total, its GOPs dealt to the ranks (the rank's share is rounded down to whole pictures and
    # the total says what was decoded)
    strong = world > 1 and args.scaling == "strong"
    steps_rank = args.steps * PPS if not strong else max(1, args.steps * PPS // world)          # pictures this rank decodes
    dt = timed(steps_rank)
    dom_avg = read_timer()
    set_timer(None)
    ms_per_step = dt * 1e3 / args.steps
    fps = world * steps_rank / dt

    # secondary figure: the round-1 measurement (device-resident replay of the same command buffers, no H2D / D2H)
    dt_res = timed(min(steps_rank, 120), resident=True)
    fps_res = world * min(steps_rank, 120) / dt_res

    check = None
    if args.check > 0 and world == 1:
        def from_the_start():
            gop_base[0] = 0
            for k, (_t, pk) in enumerate(bufs_key):
                pk.upload(*wls[0].refs[k % n_ref_slots])
            torch.cuda.synchronize(dev)
        da, db = {}, {}
        from_the_start()
        run_steps(args.check, digests=da)
        barrier()
        from_the_start()
        one_at_a_time[0] = True
        run_steps(args.check, digests=db)
        one_at_a_time[0] = False
        barrier()
        bad = [i for i in da if da[i] != db.get(i)]
        check = {"pictures": len(da), "differ": len(bad), "distinct_digests": len(set(da.values())),
                 "what": f"{S} pictures in flight vs one at a time, same stream from the same start, ovhip_pic_digest of every picture"}
        if bad:
            raise SystemExit(f"bench --check: pictures {sorted(bad)[:16]} decode differently with {S} pictures in flight")

    if rank == 0:
        algs = [algorithmic_bytes(wl, FB) for wl in wls]
        use = np.bincount([len(wls) - 1 if (j == 0 and G % IP == 0) else j % n_b for j in range(K)], minlength=len(wls)).astype(np.float64)
        if IP > G:
            use[len(wls) - 1] *= G / IP; use[0] += 1.0 - G / IP
        use /= use.sum()
        alg = {k: float(sum(u * a[k] for u, a in zip(use, algs))) for k in algs[0]}
        alg = {k: v for k, v in alg.items() if k in kern}
        achieved = alg[dom] / dom_avg / 1e9
        traffic = rocprof_avg = None
        try:
            tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
            if tj["workload"] == {"width": W, "height": H, "seed": args.seed}:
                names = [n.strip() for n in KNAME[dom].split("(")[0].split("+")]
                ks = [tj["kernels"][n] for n in names]
                # a launch group = one dispatch, except the ordered pass: one per level, averaged over the intra period
                per_group = (wls[-1].stats["n_ilevels"] + (IP - 1) * wls[0].stats["n_ilevels"]) / IP if dom == "intra" else 1
                traffic = int(sum(2 * k["fetch_kib"] + k["write_kib"] for k in ks) * 1024 * per_group)
                if dom != "intra" and all("trace_avg_us" in k for k in ks):
                    rocprof_avg = round(sum(k["trace_avg_us"] for k in ks), 2)
        except (OSError, KeyError, ValueError):
            traffic = None
        roofline = {"bound": "hbm", "kernel": KNAME[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "avg_launch_us": round(dom_avg * 1e6, 2),
                    # the committed rocprofv3 --kernel-trace average of the same command (profiles/, every launch in the timed
                    # configuration): execution only -- the HIP events of avg_launch_us also see the dispatch waiting in its
                    # hardware queue behind the other streams' packets
                    "rocprof_avg_launch_us": rocprof_avg,
                    "frac_at_rocprof_duration": round(alg[dom] / rocprof_avg / 1e3 / HBM_PEAK_GBPS, 5) if rocprof_avg else None,
                    "picked_from": "survey in the timed configuration (same rotation and pictures in flight)",
                    "survey_launch_us": {k: round(v * 1e6, 2) for k, v in survey.items()},
                    "frac_per_kernel": {k: round(alg[k] / kern[k] / 1e9 / HBM_PEAK_GBPS, 5) for k in alg},
                    "isolated_launch_us": {k: round(v * 1e6, 2) for k, v in isolated.items()},
                    "frac_isolated_per_kernel": {k: round(alg[k] / isolated[k] / 1e9 / HBM_PEAK_GBPS, 5) for k in alg if k in isolated and k != "intra"},
                    "frac_isolated": round(alg[dom] / isolated[dom] / 1e9 / HBM_PEAK_GBPS, 5) if dom in isolated else None,
                    "ordered_pass": ({"kernel": KNAME["intra"], "bound": "latency (dependency chain, DESIGN 4.1)",
                                      "avg_us_per_picture_timed": round(kern["intra"] * 1e6, 2),
                                      "avg_us_per_picture_isolated": round(isolated.get("intra", 0.0) * 1e6, 2),
                                      "levels_per_i_picture": int(wls[-1].stats["n_ilevels"]), "levels_per_b_picture": int(wls[0].stats["n_ilevels"])}
                                     if "intra" in kern else None),
                    "algorithmic_bytes": {k: int(v) for k, v in alg.items()},
                    "frame_frac": round(sum(alg.values()) * fps / world / 1e9 / HBM_PEAK_GBPS, 5)}

        cpu = None
        if not args.no_cpu_baseline:
            import oracle_pipeline
            wl0 = wls[0]
            t1 = time.perf_counter()
            oracle_pipeline.decode(wl0)
            t_one = time.perf_counter() - t1
            # all host cores: frame-level parallelism (one picture per thread, the reference's --framethr), bounded sample
            ncpu = os.cpu_count() or 1
            nthr = max(1, min(ncpu, 64))
            t1 = time.perf_counter()
            th = [threading.Thread(target=oracle_pipeline.decode, args=(wl0,)) for _ in range(nthr)]
            [t.start() for t in th]
            [t.join() for t in th]
            t_all = time.perf_counter() - t1
            cpu = {"value": round(nthr / t_all, 3), "unit": "frames/s", "cores": nthr, "kind": "port",
                   "value_1_thread": round(1.0 / t_one, 4),
                   "sample": f"oracle/liboracle.so (scalar C restatement of the rcn path) decoding the same {W}x{H} recorded "
                             f"picture: once on 1 thread ({t_one:.2f} s), then {nthr} pictures on {nthr} threads, one picture "
                             f"per thread as the reference's frame threads do ({t_all:.2f} s); {ncpu} logical cores present",
                   "calibration": _calibration()}

        st = wls[0].stats
        js = flush_stats
        out = {
            "metric": "decoded frames/sec, full rcn back-end decode step (H2D of the recorded picture + MC incl. BDOF/DMVR/"
                      "affine-PROF/GPM/CIIP + LMCS + inverse transform + ordered intra pass + deblocking + SAO + ALF/CC-ALF + D2H of refined MVs), "
                      "4K 10-bit RA recorded picture, bit-exact vs oracle",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "u16 samples / int16 coefficients / int32 accumulate", "data": "synthetic",
            "config": {"pictures_per_step": PPS, "pictures_per_rank": steps_rank,
                       "workload": f"{W}x{H} 10-bit 4:2:0 synthetic recorded random-access stream (BASELINE configs[3]): GOP {G} "
                                   f"(hierarchical B, JVET decoding order), intra period {IP}: per GOP {G - 1} B pictures with "
                                   f"{args.intra_frac:.0%} intra CUs + the key picture ({'I' if G % IP == 0 else 'I every ' + str(IP // G) + ' GOPs, else B'}); "
                                   f"reference pictures = the decoded pictures of the GOP structure; seeds "
                                   f"{[hex(w.seed) for w in wls]}, per-picture flush in C (ovhip_job_flush)",
                       "gop_size": G, "intra_period": IP,
                       "dependency_critical_path_pictures": round(gop.critical_path(gop.build_stream(4 * world, G, IP, world)), 1),
                       "intra_tasks_per_b_picture": st["n_itasks"], "intra_levels_per_b_picture": st["n_ilevels"],
                       "intra_levels_per_i_picture": wls[-1].stats["n_ilevels"],
                       "h2d_bytes_per_step": int(mean_stat("h2d_bytes")), "d2h_bytes_per_step": int(mean_stat("d2h_bytes")),
                       "ordered_pass_second_passes": second_passes[0],
                       "check": check,
                       "launches_per_step": round((int(all_stats[0].n_launches) + (IP - 1) * int(js.n_launches)) / IP, 1),
                       "h2d_copies_per_step": round(mean_stat("n_h2d"), 1),
                       "launches_per_b_picture": int(js.n_launches),
                       "launches_per_i_picture": int(all_stats[0].n_launches),
                       "distinct_pictures": K, "distinct_contents": len(wls), "working_set_bytes": int(working_set),
                       "pictures_in_flight_per_gpu": S,
                       "host_threads": nthreads,
                       "picture_assignment": "decoding order; a free host thread takes the next picture, waits (stream events) for "
                                             "its reference pictures, flushes it and waits for it (as the shim does)",
                       "recorder_in_timed_region": False,
                       "n_cu": st["n_cu"], "cu_modes": st["cu_modes"], "n_mc_units": st["n_mc_units"],
                       "n_mcx_units": st["n_mcx_units"], "n_aff_units": st["n_aff_units"], "n_tb_cmds": st["n_tb_cmds"],
                       "r_bar": round(st["r_bar"], 3), "coef_bytes": st["coef_bytes"],
                       "frame_algorithmic_bytes": int(sum(alg.values())),
                       "resident_replay_fps": round(fps_res, 2),
                       "parallelism": f"{S} pictures in flight per GPU" + (f"; a GOP per GPU over {world} GPUs, the key picture of a GOP "
                                      "sent to the owner of the next GOP with RCCL point-to-point on a communication stream (no "
                                      "collective, nothing waited for on the host)" if world > 1 else "")},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _calibration():
    """Reference scalar C vs the oracle port on identical slot-level cases, timed in the build container (committed; the
    reference does not travel to the GPU box)."""
    try:
        return json.loads((ROOT / "profiles" / "cpu_calibration.json").read_text())
    except (OSError, ValueError):
        return None


if __name__ == "__main__":
    main()
