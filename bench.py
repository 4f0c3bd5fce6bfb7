#!/usr/bin/env python
"""bench.py -- decoded frames/s of the MI355X rcn back-end on a random-access stream, timed region in C.

A "step" is one intra period of a synthetic recorded 3840x2160 10-bit 4:2:0 random-access stream (BASELINE.json configs[3]):
its I picture and the B pictures of its GOPs (openvvc_amd/gop.py: GOP `--gop`, hierarchical B in the JVET decoding order, an I
picture every `--intra-period`), every picture decoded the way the reference-side shim decodes it (shim/rcn_hip.c ->
ovhip_frame_submit): H2D of the picture's recorded command buffers out of page-locked memory, the launch chain of the whole rcn
path (prediction incl. BDOF / DMVR / affine-PROF / GPM / CIIP + LMCS, inverse quantisation / LFNST / transforms + residual, the
ordered intra pass, deblocking, SAO, ALF / CC-ALF), D2H of the DMVR-refined vectors, ovhip_job_wait, the picture's digest
(`--output`), and only THEN its publication to the pictures that reference it (device DPB, ovvc_dpb.c).

The timed region is ONE call into the library: ovhip_stream_run (openvvc_amd/csrc/ovvc_stream.c) -- `--in-flight` frame threads
(pthreads, one HIP stream each: the reference's frame threads, ovdec.c:188-248) take the pictures in decoding order; a picture's
reference pictures ARE the decoded pictures its reference lists name; an output thread takes finished pictures in POC order.
Python builds the stream description before and reads the result after; it is not in the loop.

Variants measured beside the headline figure (same stream, same library call, `config.variants`):
  output none / digest / frame   nothing leaves the device / per-picture MD5 fingerprint (16 B) / crop + pack + D2H of the 24.9 MB frame
  recorded_in_run                the recorder inside the timed region: every frame thread replays the picture's call log
                                 (ovhip_calllog_replay = every ovhip_rec_* call a parse thread makes) into its own job first

N > 1 (torchrun, one process per GPU): a GOP per rank; the key picture of a GOP goes to the owner of the next GOP -- the driver's
comm thread calls back into torch.distributed (RCCL point-to-point), one transfer per GOP, no collective on the data path.
`--local-devices N`: ONE process drives N logical devices instead (device DPB with event-ordered hipMemcpyPeerAsync), the
reference's --framethr model.

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
    ap.add_argument("--numa-bind", action="store_true",
                    help="restrict the process to the CPUs of the GPU's NUMA node (one-GPU runs).  Measured, no gain: 2916 vs 2847 pictures/s (noise), the "
                         "recorded_in_run variant with 32 threads 2080 bound vs 2677 unbound -- the recorder's slow-down with the thread count is not "
                         "remote memory")
    ap.add_argument("--no-reference-stream", action="store_true",
                    help="skip the leg on the stream the REFERENCE's own slice decoder parses and decodes (oracle/_ref/gen_pipe, prebuilt from "
                         "/root/reference in the build container): its scalar / SIMD decode rate on this host's cores = cpu_baseline kind "
                         "\"reference\", and the same nine 3840x2160 pictures through the HIP engine compared byte for byte with the reference's frames")
    ap.add_argument("--no-live-decoder", action="store_true",
                    help="skip the leg in which the reference's slice decoder + parser drive the installed shim on this GPU on 1 .. 16 frame threads "
                         "(oracle/_ref/gen_pipe live): decoded pictures/s with parsing inside, compared with the reference pass in process")
    ap.add_argument("--no-isolated-survey", action="store_true",
                    help="skip the one-picture-in-flight survey and the variants: every launch of the process then runs in the timed configuration "
                         "(what tools/profile_round.sh traces, so that rocprofv3's per-kernel averages are of that configuration)")
    ap.add_argument("--in-flight", type=int, default=16, help="frame threads (= pictures in flight) per device: pthreads of the C stream driver, one HIP stream each")
    ap.add_argument("--contents", type=int, default=2, help="distinct recorded B pictures (seeds)")
    ap.add_argument("--intra-frac", type=float, default=0.12, help="share of intra CUs in the B pictures")
    ap.add_argument("--gop", type=int, default=32, help="GOP size (hierarchical B, JVET random-access decoding order)")
    ap.add_argument("--intra-period", type=int, default=64, help="every key picture at a multiple of this POC is an I picture (a multiple of --gop; JVET CTC: about one second, 64 at 50 / 60 Hz)")
    ap.add_argument("--intra-ctu", action="store_true", help="ordered pass as the one-launch CTU wavefront")
    ap.add_argument("--intra-levels", action="store_true", help="ordered pass as one launch per level instead of one launch with per-unit dependency flags")
    ap.add_argument("--job-rotation", type=int, default=3,
                    help="GOPs of pre-recorded picture jobs in the rotation (a job = the page-locked command buffers of one stream position + its device "
                         "copies, in flight once at a time): bounds how far ahead of the oldest picture in flight the frame threads can run")
    ap.add_argument("--output", choices=("none", "digest", "frame"), default="digest",
                    help="what leaves the device per picture INSIDE the timed region: nothing / its MD5-tree fingerprint (computed on the device, 16 bytes "
                         "out) / the cropped, packed frame (one D2H of 24.9 MB at 4K into page-locked memory)")
    ap.add_argument("--check", type=int, default=6, metavar="N",
                    help="after the measurement: the first N pictures of the stream (I picture first) decoded by the ORACLE one at a time, each from the "
                         "oracle's own reference pictures, against the device's digests of the same pictures decoded --in-flight at a time; and the "
                         "first two GOPs in flight vs one at a time (0: skip)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = every rank decodes --steps intra periods (the stream grows with N); strong = the stream is --steps intra "
                         "periods in total, its GOPs dealt to the ranks")
    ap.add_argument("--dealing", choices=("gop", "picture"), default="gop",
                    help="N > 1 / --local-devices: a GOP per device (one transfer per GOP) or picture k -> device k mod N (SURVEY 8e as written: one transfer "
                         "per reference edge that crosses devices)")
    ap.add_argument("--local-devices", type=int, default=1,
                    help="ONE process driving this many logical devices (device DPB + hipMemcpyPeerAsync); with --same-gpu all of them on GPU 0")
    ap.add_argument("--same-gpu", action="store_true")
    ap.add_argument("--both-dealings", action="store_true", help="N > 1: also measure the other dealing (config.other_dealing)")
    ap.add_argument("--record-threads", type=str, default="1,4,16,32", help="frame-thread counts of the recorded_in_run variant")
    ap.add_argument("--debug-digests-only", action="store_true", help="debug (with --output none): the frame threads compute every picture's fingerprint, but no output thread takes them in POC order and no picture is held for it -- what of the digest mode's cost is the fingerprint, what the output order")
    ap.add_argument("--debug-resident", action="store_true", help="debug: the timed run replays the device copies of the warm-up's flushes (no H2D / D2H): for profiling what the copies cost; the line is NOT a measurement of the path")
    ap.add_argument("--trace", type=str, default="", help="debug: write the per-picture timeline of the timed region (taken / submitted / published, thread) to this file")
    ap.add_argument("--intra-lookahead", type=int, default=64,
                    help="pictures: one more frame thread per device starts pictures WITHOUT reference pictures (I pictures) up to this many pictures "
                         "before their turn in decoding order (0: strictly in order)")
    ap.add_argument("--ipic-workers", type=int, default=0, help="persistent workers of an I picture's ordered pass (0: the library's default, 4 x compute units)")
    ap.add_argument("--bpic-workers", type=int, default=0, help="persistent workers of a B picture's ordered pass (0: the library's default)")
    ap.add_argument("--ahead-own-queue", type=int, default=1, help="1: the look-ahead thread's stream gets a hardware queue no in-order thread's stream shares (probed at start)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from openvvc_amd import capi, engine, gop, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    if world == 1 and args.gpus > 1 and args.local_devices == 1:
        # `python bench.py --gpus N` without torch.distributed.run: ONE process drives N devices through the C stream driver (device
        # DPB, reference pictures by hipMemcpyPeerAsync) -- never a silent one-GPU run that prints n_gpus: 1
        args.local_devices = args.gpus
    if world > 1 and args.gpus not in (1, world):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if world == 1 and args.local_devices > 1 and not args.same_gpu and torch.cuda.device_count() < args.local_devices:
        raise SystemExit(f"bench.py: {args.local_devices} devices asked for, {torch.cuda.device_count()} visible (--same-gpu puts the logical devices on GPU 0)")
    # OVVC_BENCH_DEBUG_GLOO=1 (development only): all ranks on GPU 0, pictures exchanged through host memory over gloo -- runs
    # the N > 1 control flow (schedule, comm thread, callbacks) on a one-GPU box.  Never set by the driver.
    debug_gloo = os.environ.get("OVVC_BENCH_DEBUG_GLOO") == "1"
    if debug_gloo:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    all_cpus = os.sched_getaffinity(0)
    numa = _bind_to_gpu_numa_node(torch, local_rank) if args.numa_bind and world * max(1, args.local_devices) == 1 else None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if debug_gloo else "nccl", **({} if debug_gloo else {"device_id": dev}))

    W, H = args.width, args.height
    S = max(1, args.in_flight)
    G, IP = args.gop, args.intra_period
    assert IP % G == 0
    PPS = IP
    L = max(1, args.local_devices) if world == 1 else 1
    hip_devices = [local_rank if (args.same_gpu or world > 1) else k for k in range(L)] if L > 1 else [local_rank]

    tools = synth.INTRA_TOOLS if args.intra_frac > 0 else synth.ALL_TOOLS
    want_log = not args.no_isolated_survey
    wls = [synth.make_workload(W, H, args.seed + 1000 * c + rank, tools=tools, intra_frac=args.intra_frac, calllog=want_log)
           for c in range(max(1, args.contents))]
    n_b = len(wls)
    wls.append(synth.make_workload(W, H, args.seed + 7777 + rank, tools=synth.INTRA_TOOLS, intra_frac=1.0, calllog=want_log))      # the I picture
    FB = wls[0].frame_bytes

    # ---- library objects: device DPB, pre-recorded jobs (R GOPs of stream positions, + the I / key pictures), stream drivers ----
    ctx0 = engine.Context(hip_devices[0])
    dpb = engine.Dpb(tuple(hip_devices))
    # (with several devices each may sit on a GOP of its own: a job shared by two GOPs that are in flight at once could be held by the
    # later one while the earlier one, which it waits for through the key pictures, needs it)
    R = max(2, args.job_rotation, L + 2)
    n_jobs = R * G
    keep = []

    class K:
        def __init__(self):
            self._keep = {}

    def params_of(wl, workers=0):
        k = K(); keep.append(k)
        p = engine.Job.make_params(k, wl)
        p.flow_workers = workers
        return p

    # the I picture's ordered pass has ~100 items per level: --ipic-workers persistent workers instead of the default 4 x CUs leave the wave
    # slots, registers and LDS of the others to the B pictures' kernels beside it for the 5 ms it runs
    contents = [{"params": params_of(wl, args.ipic_workers if i == len(wls) - 1 else args.bpic_workers), "calllog": wl.calllog, "n_ref_slots": len(wl.refs)} for i, wl in enumerate(wls)]
    lv = capi.STAGE_INTRA_CTU if args.intra_ctu else (capi.STAGE_INTRA_LEVELS if args.intra_levels else 0)

    def content_of(p):
        """stream picture -> recorded content: I pictures show the I content; B pictures rotate over the B contents by stream position"""
        return n_b if p.intra else (p.idx % n_b)

    # a job per stream position modulo the rotation.  The job of position j holds ONE content for ever (its page-locked arrays are
    # loaded once): position -> content must be periodic with the rotation.  I pictures sit at idx = 1 + k * PPS: give them jobs of their own.
    n_ijobs = 3
    job_content = [j % n_b for j in range(n_jobs)] + [n_b] * n_ijobs
    # a job's device buffers live on the device of the context it was created on (ovhip_job_bind refuses another device): one set of
    # pre-recorded jobs per logical device, device d's set at [d * JPD, (d + 1) * JPD)
    JPD = n_jobs + n_ijobs
    ctxs = [ctx0] + [engine.Context(hip_devices[d]) for d in range(1, L)]
    jobs = []
    for d in range(L):
        for c in job_content:
            j = engine.Job(ctxs[d], W, H)
            j.load_workload(wls[c])
            jobs.append(j)

    def build(n_gops_total, world_=1, dealing="gop", n_dev=1):
        pics = gop.build_stream(n_gops_total, G, IP, world_)
        if dealing == "picture" and max(world_, n_dev) > 1:
            m = max(world_, n_dev)
            for p in pics:
                p.owner = p.idx % m if world_ > 1 else 0
            for p in pics:
                p.sends = []
            for p in pics:
                for r in p.refs:
                    q = pics[r]
                    if q.owner != p.owner and p.owner not in q.sends:
                        q.sends.append(p.owner)
        out, n_i = [], 0
        for p in pics:
            device = 0
            if n_dev > 1:
                device = (p.idx % n_dev) if dealing == "picture" else (gop.gop_owner(max(p.gop, 0), n_dev, IP // G))
            if p.intra:
                job = n_jobs + n_i % n_ijobs
                n_i += 1
                content = n_b
            else:
                job = p.idx % n_jobs
                content = job_content[job]
            job += device * JPD
            out.append({"content": content, "job": job, "poc": p.poc, "refs": p.refs[:capi.STREAM_MAX_REFS], "device": device,
                        "owner": p.owner, "send_mask": sum(1 << d for d in p.sends)})
        return pics, out

    OUT = {"none": capi.OUT_NONE, "digest": capi.OUT_DIGEST, "frame": capi.OUT_PACKED}
    lookahead = max(0, args.intra_lookahead)

    def new_stream(threads, output="none", flags=0, xfer=None, use_jobs=True, ahead=None):
        return engine.Stream(dpb, W, H, contents, jobs if use_jobs else [], threads_per_device=threads, flags=flags, output=OUT[output],
                             extra_stages=lv, rank=rank, xfer=xfer, intra_lookahead=(lookahead if ahead is None else ahead) if threads > 1 else 0,
                             ahead_own_queue=args.ahead_own_queue if threads > 1 else 0)

    # ---- multi-process exchange: the driver's comm thread calls back here, one call per transferred picture ----
    xfer = None
    rccl = None
    rank_accounts = None
    n_xfer_bytes = [0]
    if world > 1:
        stage = torch.empty(FB // 2 + 512, dtype=torch.int16, device=dev)       # one allocation = the picture's three planes
        pic_bytes = lambda p: ((p.w * p.h * 2 + 255) & ~255) + 2 * (((p.w // 2) * (p.h // 2) * 2 + 255) & ~255)
        assert pic_bytes(type("P", (), {"w": W, "h": H})) <= stage.numel() * 2

        def _send(user, idx, pic, dst):
            try:
                nb = pic_bytes(pic[0])
                ctx0._chk(ctx0.lib.ovhip_d2d(ctx0.h, stage.data_ptr(), pic[0].y, nb), "d2d")
                t = stage[:nb // 2]
                dist.send(t.cpu() if debug_gloo else t, dst)
                torch.cuda.synchronize(dev)
                n_xfer_bytes[0] += nb
                return 0
            except Exception as e:          # noqa: BLE001
                print(f"rank {rank}: send of picture {idx} failed: {e}", file=sys.stderr)
                return -4

        def _recv(user, idx, pic, src):
            try:
                nb = pic_bytes(pic[0])
                t = stage[:nb // 2]
                if debug_gloo:
                    th = torch.empty(nb // 2, dtype=torch.int16)
                    dist.recv(th, src)
                    t.copy_(th)
                else:
                    dist.recv(t, src)
                torch.cuda.synchronize(dev)
                ctx0._chk(ctx0.lib.ovhip_d2d(ctx0.h, pic[0].y, stage.data_ptr(), nb), "d2d")
                return 0
            except Exception as e:          # noqa: BLE001
                print(f"rank {rank}: receive of picture {idx} failed: {e}", file=sys.stderr)
                return -4

        xfer = capi.StreamXfer(None, capi.XFER_FN(_send), capi.XFER_FN(_recv))
        # The product path: RCCL point-to-point inside the C library (ovvc_rccl.hip: ncclSend / ncclRecv of the three planes in one
        # group per picture, issued by the driver's comm thread on a stream of its own); bench.py only carries rank 0's ncclUniqueId to
        # the other ranks.  The Python callbacks above stay as the fall-back (and are what OVVC_BENCH_DEBUG_GLOO runs on one GPU).
        rccl = None
        if not debug_gloo and os.environ.get("OVVC_BENCH_PY_XFER") != "1":
            try:
                uid = torch.zeros(128, dtype=torch.uint8, device=dev)
                if rank == 0:
                    uid.copy_(torch.frombuffer(bytearray(engine.RcclTransport.unique_id()), dtype=torch.uint8))
                dist.broadcast(uid, 0)
                rccl = engine.RcclTransport(bytes(uid.cpu().numpy().tobytes()), rank, world, local_rank)
                xfer = rccl.xfer
            except Exception as e:          # noqa: BLE001
                print(f"rank {rank}: RCCL transport not available ({e}): pictures go through torch.distributed", file=sys.stderr)
                rccl = None
        ok = torch.tensor([1 if rccl is not None else 0], dtype=torch.int32, device=None if debug_gloo else dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not int(ok.item()) and rccl is not None:          # every rank or none
            rccl.close(); rccl = None
            xfer = capi.StreamXfer(None, capi.XFER_FN(_send), capi.XFER_FN(_recv))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # ---- the local stream (this process alone): warm-up, surveys, variants, check ----
    n_loc_gops = (args.warmup + 3 * 11 + 8) * (IP // G) + 8 + (args.steps + max(1, args.warmup)) * (IP // G) * (L - 1)
    lpics, lspics = build(n_loc_gops, 1, args.dealing, L)
    larr = engine.Stream.pics_array(lspics)
    NL = len(lspics)
    st_main = new_stream(S)
    cur = [0]

    def run_local(st, n, flags=0, digests=False):
        """the next n pictures of the local stream on stream driver st"""
        assert cur[0] + n <= NL, "local stream exhausted"
        res, dg = st.run(larr, NL, cur[0], n, flags=flags | capi.STREAM_KEEP, digests=digests)
        cur[0] += n
        return res, dg

    second_passes = [0]

    def count(res):
        second_passes[0] += int(res.n_second_passes)
        return res

    # warm-up: every job flushed at least once, every buffer of the DPB pool allocated
    count(run_local(st_main, 1 + max(args.warmup * PPS, n_jobs + G))[0])
    barrier()
    all_stats = [j.stats() for j in jobs]
    flush_stats = all_stats[0]
    mean_stat = lambda f: float(np.mean([getattr(a, f) for a in all_stats[:n_jobs]]))

    def set_timer(name):
        for j in jobs:
            j.time_stage(name)

    last_ipic_time = [None]

    def read_timer():
        tot, cnt, itot, icnt = 0.0, 0, 0.0, 0
        for k, j in enumerate(jobs):
            s, n = j.stage_time()
            tot += s; cnt += n
            if k % JPD >= n_jobs:                     # the I pictures' jobs
                itot += s; icnt += n
        last_ipic_time[0] = itot / icnt if icnt else None
        return tot / max(cnt, 1)

    # ---- untimed survey IN THE TIMED CONFIGURATION (same jobs, same pictures in flight): each launch group bracketed in turn
    # by a HIP-event pair on its stream (bracketing all of them at once would cost ~80 us of stream time per picture)
    present = ["mc", "mcxa", "itx_luma", "lmcs_scale", "itx_chroma", "intra", "dbf", "sao", "alf", "h2d"]
    if not flush_stats.n_regions:
        present.remove("lmcs_scale")
    if not any(w.itasks is not None and len(w.itasks) for w in wls):
        present.remove("intra")
    survey = {}
    for name in present:
        set_timer(name)
        count(run_local(st_main, 2 * PPS)[0])
        survey[name] = read_timer()
    kern = {k: v for k, v in survey.items() if k != "h2d"}
    # the roofline's subject = the launch group with the largest stream time, whatever it is (r2 left the ordered pass out)
    dom = max(kern, key=kern.get)
    stream_dom = max((k for k in kern if k != "intra"), key=kern.get)
    isolated = {}
    ipic_isolated_s = None
    variants = {}
    if not args.no_isolated_survey:
        # the same groups with ONE picture in flight: what a launch takes when it has the device to itself
        st_one = new_stream(1)
        for name in present:
            if name == "h2d":
                continue
            set_timer(name)
            r1, _ = st_one.run(larr, NL, 0, 1 + 2 * G, flags=0)
            isolated[name] = read_timer()
            if name == "intra":
                ipic_isolated_s = last_ipic_time[0]
        set_timer(None)
        st_one.close()
    if world > 1:
        pick = torch.tensor([present.index(dom)], dtype=torch.int64, device=None if debug_gloo else dev)
        dist.broadcast(pick, 0)
        dom = present[int(pick.item())]

    # ---- timed region: EXACTLY --steps steps, one library call, only the dominant launch group bracketed ----
    set_timer(dom)
    strong = world > 1 and args.scaling == "strong"
    if world > 1:
        st_main.close()
        gops_rank = args.steps * (IP // G) if not strong else max(1, args.steps * (IP // G) // world)
        warm_gops = max(1, args.warmup) * (IP // G)
        tpics, tspics = build((warm_gops + gops_rank) * world, world, args.dealing, 1)
        tarr = engine.Stream.pics_array(tspics)
        NT = len(tspics)
        step_fps = []
        st_t = new_stream(S, output=args.output, xfer=xfer)
        n_warm = 1 + warm_gops * world * G
        res, _ = st_t.run(tarr, NT, 0, n_warm, flags=capi.STREAM_KEEP)
        barrier()
        t0 = time.perf_counter()
        res, _ = st_t.run(tarr, NT, n_warm, NT - n_warm, flags=capi.STREAM_KEEP)
        barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=None if debug_gloo else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        n_timed_rank = int(res.n_decoded)
        n_timed_all = NT - n_warm
        xfers = [int(res.n_sent), int(res.n_received)]
        count(res)
        # ---- every rank's own account of the timed run, gathered so that ONE line checks itself: what each rank decoded,
        #      sent and received (the driver's counters and, with RCCL, the transport's own byte counts), and what the picture dealing says it
        #      should have -- the first hardware record of the multi-GPU path must not need a second run to be believed
        exp_sent = sum(bin(int(p["send_mask"])).count("1") for p in tspics[n_warm:] if p["owner"] == rank)
        mine = {"rank": rank, "hip_device": local_rank if not args.same_gpu else 0, "pictures_decoded": int(res.n_decoded),
                "pictures_sent": int(res.n_sent), "pictures_received": int(res.n_received), "pictures_sent_expected_from_the_dealing": exp_sent,
                "transport": "rccl" if rccl is not None else "torch.distributed callbacks",
                "rccl": rccl.stats() if rccl is not None else None, "rccl_ranks_in_communicator": world if rccl is not None else None,
                "seconds_inside_the_c_driver": round(float(res.seconds), 6), "second_passes": int(res.n_second_passes), "status": int(res.status)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        rank_accounts = gathered
    else:
        st_t = new_stream(S, output=args.output) if args.output != "none" else st_main
        if st_t is not st_main:
            # continue the SAME stream on the driver that has the output thread: the pictures st_main kept are in the DPB under its
            # keys, so this driver starts a stream of its own -- one I picture, then warm-up of its own
            cur_t = [0]
            tpics, tspics = build((max(1, args.warmup) + args.steps) * (IP // G) * L, 1, args.dealing, L)
            tarr = engine.Stream.pics_array(tspics)
            NT = len(tspics)
            n_warm = 1 + max(1, args.warmup) * PPS * L
            st_t.run(tarr, NT, 0, n_warm, flags=capi.STREAM_KEEP)
        else:
            tarr, NT, n_warm, tspics = larr, NL, cur[0], lspics
        # (L logical devices: weak scaling -- a step is one intra period PER DEVICE, L x PPS pictures)
        barrier()
        t0 = time.perf_counter()
        trace = np.zeros((args.steps * PPS * L, 8))
        res, _ = st_t.run(tarr, NT, n_warm, args.steps * PPS * L, flags=capi.STREAM_KEEP | (capi.STREAM_RESIDENT if args.debug_resident else 0), trace=trace, digests=args.debug_digests_only)
        barrier()
        dt = time.perf_counter() - t0
        if args.trace:
            np.save(args.trace, np.concatenate([trace, np.array([[len(tspics[n_warm + i]["refs"]), tspics[n_warm + i]["poc"], n_warm + i, 0] for i in range(len(trace))], float)], axis=1))     # columns 8..11
            # + the reference pictures (indices in decoding order, -1 padded): tools/debug/dep_latency.py
            np.save(args.trace + ".refs.npy", np.array([(list(tspics[n_warm + i]["refs"]) + [-1] * 8)[:8] for i in range(len(trace))], np.int64))
        # the rate of every step of the timed region (publication times of the driver's timeline; tools/debug/step_rates.py)
        pub = np.sort(trace[:, 2])
        edges = pub[PPS * L - 1::PPS * L][:args.steps]
        sdur = np.diff(np.concatenate([[0.0], edges]))
        step_fps = [PPS * L / d for d in sdur if d > 0]
        if st_t is st_main:
            cur[0] += args.steps * PPS * L
        n_timed_rank = n_timed_all = args.steps * PPS * L
        assert int(res.n_decoded) == n_timed_rank
        xfers = [0, 0]
        count(res)
    dom_avg = read_timer()
    set_timer(None)
    q_moved, q_sharing = st_t.queue_info()
    ms_per_step = dt * 1e3 / args.steps
    fps = n_timed_all / dt
    lib_seconds = float(res.seconds)
    # host wall time per picture inside the frame threads (summed over threads / pictures): where a frame thread spends its time
    nd = max(1, int(res.n_decoded))
    host_us = {k: round(1e6 * float(res.host_seconds[i]) / nd, 1)
               for i, k in enumerate(("class_split_and_parameter_block", "enqueue_copies", "wait_for_reference_pictures", "enqueue_launches",
                                      "job_wait_publish_output"))}
    dpb_stats = dpb.stats()
    xfer_bytes_main = n_xfer_bytes[0]
    # ---- the OTHER dealing of pictures to devices on the same kind of stream (both are implemented in the C driver):
    #      two intra periods per device, warmed by one ----
    other_dealing = None
    # (one process per GPU: only on request or with --scaling strong -- the driver's weak-scaling runs are the first execution of the
    #  RCCL exchange on real devices, and the picture-interleaved dealing sends a picture per picture: the headline run is not put at
    #  the mercy of a second, much heavier exchange)
    if L > 1 or (world > 1 and (args.both_dealings or args.scaling == "strong")):
        od = "picture" if args.dealing == "gop" else "gop"
        ndev = world if world > 1 else L
        og_warm, og = (IP // G) * ndev, 2 * (IP // G) * ndev
        opics, ospics = build(og_warm + og, world, od, 1 if world > 1 else L)
        oarr = engine.Stream.pics_array(ospics)
        st_o = new_stream(S, output=args.output, xfer=xfer)
        on_warm = 1 + og_warm * G
        st_o.run(oarr, len(ospics), 0, on_warm, flags=capi.STREAM_KEEP)
        barrier()
        t0 = time.perf_counter()
        ores, _ = st_o.run(oarr, len(ospics), on_warm, len(ospics) - on_warm, flags=capi.STREAM_KEEP)
        barrier()
        odt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([odt], dtype=torch.float64, device=None if debug_gloo else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            odt = float(t.item())
        other_dealing = {"dealing": od, "fps": round((len(ospics) - on_warm) / odt, 1), "pictures": len(ospics) - on_warm,
                         "pictures_sent_by_this_rank": int(ores.n_sent), "peer_copies": int(dpb.stats().n_copies) - int(dpb_stats.n_copies)}
        count(ores)
        st_o.close()
    if world > 1:
        st_main = new_stream(S)
        cur[0] = 0
        lpics, lspics = build(n_loc_gops, 1, args.dealing, 1)
        larr = engine.Stream.pics_array(lspics); NL = len(lspics)
        count(run_local(st_main, 1 + G)[0])

    # ---- variants: the same stream through the same call with other things inside the timed region ----
    def timed_variant(st, n_pics, flags=0, warm=True):
        arr, n = larr, NL
        # warm-up: the I picture + one whole intra period, as the headline run has it -- the timed pictures then start on an intra
        # period's boundary with the look-ahead thread a whole period ahead (33 pictures of warm-up put the first timed I picture 31
        # pictures away and made "output none" read 10-15 % below "digest", which does strictly more)
        nw = 1 + PPS
        assert nw + n_pics <= n, "local stream too short for a variant"
        st.run(arr, n, 0, nw, flags=flags | capi.STREAM_KEEP)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        r, _ = st.run(arr, n, nw, n_pics, flags=flags | capi.STREAM_KEEP)
        torch.cuda.synchronize(dev)
        return n_pics / (time.perf_counter() - t0), r

    fps_res = None
    if rank == 0 and not args.no_isolated_survey:
        nv = 4 * PPS
        for mode in ("none", "digest", "frame"):
            if mode == args.output and world == 1:
                variants["output_" + mode] = round(fps, 1)
                continue
            stv = new_stream(S, output=mode)
            f, r = timed_variant(stv, nv)
            variants["output_" + mode] = round(f, 1)
            if mode == "frame":
                variants["output_frame_d2h_GBps"] = round(f * FB / 1e9, 2)
            stv.close()
        # strictly in decoding order (no look-ahead thread: the reference's frame threads take NAL units in order, ovdec.c:188-248): every
        # I picture's dependency chain is then in the way of the pictures behind it
        stv = new_stream(S, output=args.output, ahead=0)
        f, r = timed_variant(stv, nv)
        variants["in_order_no_lookahead"] = round(f, 1)
        stv.close()
        # the device-resident replay of the same command buffers (no H2D / D2H): r1's measurement, inputs resident in HBM
        try:
            f, r = timed_variant(st_main, nv, flags=capi.STREAM_RESIDENT)
            fps_res = f
        except engine.EngineError as err:             # a diagnostic variant: its failure is reported, it does not take the line with it
            variants["resident_replay_error"] = str(err)[:300]
        rec = {}
        for t in [int(x) for x in args.record_threads.split(",") if x]:
            stv = new_stream(t, output=args.output, flags=capi.STREAM_RECORD, use_jobs=False)
            n = max(G, min(nv, 24 * t))
            f, r = timed_variant(stv, n)
            rec[str(t)] = {"fps": round(f, 1), "record_ms_per_picture": round(1e3 * r.record_seconds / max(1, n + 1 + G), 3)}
            stv.close()
        variants["recorded_in_run"] = {"what": "recorder_in_timed_region: true -- every frame thread replays the picture's call log (all ovhip_rec_* "
                                               "calls of the picture: what a parse thread does minus CABAC) into its own job, then submits; fps by number of "
                                               "frame threads", "by_threads": rec}
    barrier()

    # ---- check: the device against the oracle on the first pictures of the stream; in flight vs one at a time on two GOPs ----
    check = None
    if args.check > 0 and rank == 0:
        sys.path.insert(0, str(ROOT / "oracle"))
        import oracle_pipeline
        import ovvc_oracle_output as oo
        nck = min(args.check, 1 + G)
        stc = new_stream(S)
        n2 = 1 + 2 * G
        _, dga = stc.run(larr, NL, 0, n2, digests=True)
        st1 = new_stream(1)
        _, dgb = st1.run(larr, NL, 0, n2, digests=True)
        st1.close(); stc.close()
        differ_self = int((dga != dgb).any(axis=1).sum())
        planes, differ = [], 0
        t0 = time.perf_counter()
        for i in range(nck):
            p = lspics[i]
            wl = wls[p["content"]]
            saved = list(wl.refs)
            if p["refs"]:
                for k in range(len(wl.refs)):
                    wl.refs[k] = planes[p["refs"][k % len(p["refs"])]] if p["refs"][k % len(p["refs"])] < len(planes) else None
            if any(r is None for r in wl.refs):
                wl.refs[:] = saved
                break
            o = oracle_pipeline.decode(wl)
            wl.refs[:] = saved
            planes.append((o.y, o.cb, o.cr))
            differ += bytes(dga[i]) != oo.picture_digest(o.y, o.cb, o.cr)
        check = {"pictures_vs_oracle": len(planes), "differ": differ + differ_self,
                 "oracle_seconds": round(time.perf_counter() - t0, 1),
                 "pictures_in_flight_vs_one_at_a_time": n2, "differ_in_flight": differ_self,
                 "what": f"the first {len(planes)} pictures of the stream (decoding order: I picture, then the first GOP's top layers) decoded by the "
                         f"oracle one at a time from its OWN reference pictures vs the device's digests with {S} pictures in flight; and {n2} pictures "
                         f"{S} in flight vs one at a time"}
        if differ or differ_self:
            raise SystemExit(f"bench --check: {differ} pictures differ from the oracle, {differ_self} differ between in-flight counts")

    # the legs beside the headline (the reference's stream on the device, the live decoder, the CPU baseline): at N = 1 only -- at N > 1
    # the other ranks would sit idle behind rank 0 for minutes, and the figures would be the N = 1 run's again
    one_gpu = world == 1 and L == 1
    ref_stream = None
    if rank == 0 and one_gpu and not args.no_reference_stream and (W, H) == (3840, 2160):
        ref_stream = reference_stream_on_device(engine, capi, ctx0, W, H, 17, 8)
        if ref_stream and (ref_stream["samples_differing_from_the_reference"] or ref_stream["refined_vectors_differing"]):
            raise SystemExit(f"bench: the reference's stream decodes differently on the device: {ref_stream}")

    live = None
    if rank == 0 and one_gpu and not args.no_reference_stream and not args.no_live_decoder and (W, H) == (3840, 2160):
        live = live_decoder_rates(W, H)
        if live and not live["bit_exact"]:
            raise SystemExit(f"bench: the live decode differs from the reference pass: {live}")

    if rank == 0:
        algs = [algorithmic_bytes(wl, FB) for wl in wls]
        use = np.zeros(len(wls))
        for p in lspics[1:1 + 4 * PPS]:
            use[p["content"]] += 1
        use /= use.sum()
        alg = {k: float(sum(u * a[k] for u, a in zip(use, algs))) for k in algs[0]}
        alg = {k: v for k, v in alg.items() if k in kern}
        tj = None
        try:
            tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
            if tj["workload"] != {"width": W, "height": H, "seed": args.seed}:
                tj = None
        except (OSError, KeyError, ValueError):
            tj = None

        def traffic_of(name):
            """(HBM bytes per launch group from the committed PMC passes, rocprofv3 kernel-trace average us) or (None, None)"""
            if tj is None:
                return None, None
            try:
                names = [n.strip() for n in KNAME[name].split("(")[0].split("+")]
                ks = [tj["kernels"][n] for n in names]
                per_pic = 1.0
                if name == "intra":
                    # the ordered pass: the counters are per dispatch, a picture has several (flow chunks); averaged over the stream
                    per_pic = ks[0]["dispatches"] / max(1, tj["kernels"]["k_alf"]["dispatches"])
                t = int(sum(2 * k["fetch_kib"] + k["write_kib"] for k in ks) * 1024 * per_pic)
                avg = round(sum(k["trace_avg_us"] for k in ks) * per_pic, 2) if all("trace_avg_us" in k for k in ks) else None
                return t, avg
            except (KeyError, ZeroDivisionError):
                return None, None

        def roof(name, avg_s):
            tr, ravg = traffic_of(name)
            a = alg[name] / avg_s / 1e9
            d = {"kernel": KNAME[name], "achieved": round(a, 2), "frac": round(a / HBM_PEAK_GBPS, 5), "avg_launch_us": round(avg_s * 1e6, 2),
                 "traffic": tr, "rocprof_avg_launch_us": ravg,
                 "frac_at_rocprof_duration": round(alg[name] / ravg / 1e3 / HBM_PEAK_GBPS, 5) if ravg else None,
                 "frac_isolated": round(alg[name] / isolated[name] / 1e9 / HBM_PEAK_GBPS, 5) if name in isolated and isolated[name] > 0 else None}
            return d

        rd = roof(dom, dom_avg)
        roofline = {"bound": "hbm", "kernel": rd["kernel"], "achieved": rd["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": rd["frac"],
                    "traffic": rd["traffic"], "avg_launch_us": rd["avg_launch_us"], "rocprof_avg_launch_us": rd["rocprof_avg_launch_us"],
                    "frac_at_rocprof_duration": rd["frac_at_rocprof_duration"], "frac_isolated": rd["frac_isolated"],
                    "picked_from": "survey in the timed configuration (same jobs and pictures in flight): the launch group with the largest stream time "
                                   "per picture, the ordered pass included"
                                   + ("; it is a dependency chain (levels x latency per level, DESIGN 4.1): per launch = per picture" if dom == "intra" else ""),
                    "largest_streaming_kernel": roof(stream_dom, kern[stream_dom]) if stream_dom != dom else None,
                    "survey_launch_us": {k: round(v * 1e6, 2) for k, v in survey.items()},
                    "frac_per_kernel": {k: round(alg[k] / kern[k] / 1e9 / HBM_PEAK_GBPS, 5) for k in alg if kern[k] > 0},
                    "isolated_launch_us": {k: round(v * 1e6, 2) for k, v in isolated.items()},
                    "frac_isolated_per_kernel": {k: round(alg[k] / isolated[k] / 1e9 / HBM_PEAK_GBPS, 5) for k in alg if k in isolated and isolated[k] > 0},
                    "ordered_pass": ({"levels_per_i_picture": int(wls[-1].stats["n_ilevels"]), "levels_per_b_picture": int(wls[0].stats["n_ilevels"]),
                                      "us_per_level_i_picture_isolated": round(ipic_isolated_s * 1e6 / max(1, int(wls[-1].stats["n_ilevels"])), 3) if ipic_isolated_s else None,
                                      "i_picture_pass_isolated_us": round(ipic_isolated_s * 1e6, 1) if ipic_isolated_s else None} if "intra" in kern else None),
                    "algorithmic_bytes": {k: int(v) for k, v in alg.items()},
                    "frame_frac": round(sum(alg.values()) * fps / max(world, L) / 1e9 / HBM_PEAK_GBPS, 5),
                    # SURVEY 8(d) as written: B_frame = (r_bar + 7) S + C, C = coefficients + commands consumed (the sum above also counts
                    # the read-modify-write of the transform's residual add and the inverse luma mapping's pass)
                    "frame_bytes_8d": int((float(np.dot(use, [w.stats["r_bar"] for w in wls])) + 7) * FB + float(np.dot(use, [w.stats["coef_bytes"] + 32 * (w.stats["n_tb_cmds"] + w.stats["n_mc_units"] + w.stats["n_mcx_units"] + w.stats["n_aff_units"]) for w in wls]))),
                    "frame_frac_8d": None}
        roofline["frame_frac_8d"] = round(roofline["frame_bytes_8d"] * fps / max(world, L) / 1e9 / HBM_PEAK_GBPS, 5)

        cpu = None
        if not args.no_cpu_baseline and one_gpu:
            os.sched_setaffinity(0, all_cpus)            # the CPU legs use every core of the host
            import oracle_pipeline
            wl0 = wls[0]
            t1 = time.perf_counter()
            oracle_pipeline.decode(wl0)
            t_one = time.perf_counter() - t1
            # all host cores: frame-level parallelism (one picture per thread, the reference's --framethr), bounded sample
            ncpu = os.cpu_count() or 1

            def threads_fps(n):
                t1 = time.perf_counter()
                th = [threading.Thread(target=oracle_pipeline.decode, args=(wl0,)) for _ in range(n)]
                [t.start() for t in th]
                [t.join() for t in th]
                return n / (time.perf_counter() - t1), time.perf_counter() - t1

            # every logical core of the host ("1 thread and all host cores"); the 64-thread figure of the earlier rounds beside it
            nthr = ncpu
            f_all, t_all = threads_fps(nthr)
            f_64 = threads_fps(64)[0] if ncpu > 64 else f_all
            cal = _calibration()
            cpu = {"value": round(f_all, 3), "unit": "frames/s", "cores": nthr, "kind": "port",
                   "value_1_thread": round(1.0 / t_one, 4), "value_64_threads": round(f_64, 3),
                   "sample": f"oracle/liboracle.so (scalar C restatement of the rcn path) decoding the same {W}x{H} recorded "
                             f"picture: once on 1 thread ({t_one:.2f} s), then {nthr} pictures on {nthr} threads, one picture "
                             f"per thread as the reference's frame threads do ({t_all:.2f} s); {ncpu} logical cores present",
                   "calibration": cal}
            est = _reference_estimates(cal, f_all)
            if est:
                cpu.update(est)
            if not args.no_reference_stream and (W, H) == (3840, 2160):
                ref = reference_cpu_rates(W, H, 9, ncpu)
                if ref:
                    # the reference itself on this host's cores (north_star): the headline cpu_baseline; the port's figures stay beside it
                    port = {k: cpu[k] for k in ("value", "cores", "value_1_thread", "value_64_threads", "sample") if k in cpu}
                    cpu = {"value": ref["scalar"]["fps_all_cores"], "unit": "frames/s", "cores": ncpu, "kind": "reference",
                           "measured_on": "the reference decoder's own chained stream (config.reference_stream / config.live_decoder: parse included), NOT the "
                                          "synthetic recorded stream the headline `value` is measured on: compare it with config.live_decoder (same stream, "
                                          "parse included on both sides) and config.reference_stream, not with `value`",
                           "value_1_thread": ref["scalar"]["fps_1_process"],
                           "simd": ref["simd"], "scalar": ref["scalar"], "host_side_with_the_shim": ref.get("parse_and_record"),
                           "sample": f"oracle/_ref/gen_pipe time: the reference's own slice decoder (parse + reconstruction + in-loop filters; libovvc "
                                     f"compiled from /root/reference in the build container, scalar slots) on its chained stream of 9 {W}x{H} pictures: one "
                                     f"process alone, then {ncpu} processes at once (one per logical core, each its own stream -- frame-level parallelism); "
                                     "`simd` = the same through the reference's SSE4.1 / AVX2 back-end.  The stream is the one config.reference_stream decodes "
                                     "on the GPU; the headline `value` is measured on the synthetic recorded stream of config.workload",
                           "port": port, "calibration": cal}

        st = wls[0].stats
        js = flush_stats
        out = {
            "metric": "decoded frames/sec, full rcn back-end decode step (H2D of the recorded picture + MC incl. BDOF/DMVR/"
                      "affine-PROF/GPM/CIIP + LMCS + inverse transform + ordered intra pass + deblocking + SAO + ALF/CC-ALF + D2H of refined MVs "
                      "+ wait + per-picture digest + publication to the device DPB), 4K 10-bit RA recorded stream, bit-exact vs oracle",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world if world > 1 else L, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "u16 samples / int16 coefficients / int32 accumulate", "data": "synthetic",
            "config": {"pictures_per_step": PPS * (L if world == 1 else 1), "pictures_timed": n_timed_all,
                       "step_fps": ({"min": round(min(step_fps), 1), "median": round(float(np.median(step_fps)), 1), "max": round(max(step_fps), 1),
                                     "what": "pictures/s of each of the timed steps (from the publication times of the driver's own timeline)"} if step_fps else None), "pictures_timed_this_rank": n_timed_rank,
                       "workload": f"{W}x{H} 10-bit 4:2:0 synthetic recorded random-access stream (BASELINE configs[3]): GOP {G} "
                                   f"(hierarchical B, JVET decoding order), intra period {IP}: per GOP {G - 1} B pictures with "
                                   f"{args.intra_frac:.0%} intra CUs + the key picture ({'I' if G % IP == 0 else 'I every ' + str(IP // G) + ' GOPs, else B'}); "
                                   f"reference pictures = the decoded pictures of the GOP structure (device DPB); seeds "
                                   f"{[hex(w.seed) for w in wls]}",
                       "timed_region": "one call of ovhip_stream_run (C, pthreads): frame threads take the pictures in decoding order, "
                                       "ovhip_frame_submit each (uploads -> host wait for the reference pictures -> launches -> ovhip_job_wait -> output -> publish)",
                       "timed_region_library_seconds": round(lib_seconds, 4),
                       "frame_thread_host_us_per_picture": host_us,
                       "dealing": args.dealing if (world > 1 or L > 1) else None,
                       "other_dealing": other_dealing,
                       "output": args.output + (" (per picture a PRIVATE tree fingerprint leaves the device: MD5 of MD5s over 512-byte pieces of the cropped rows, "
                                                "DESIGN 4.2 -- not the MD5 of the file dectest would write; that one is ovhip_stream's FILE_MD5 mode)" if args.output == "digest" else ""),
                       "recorder_in_timed_region": False,
                       "variants": variants,
                       "gop_size": G, "intra_period": IP,
                       "dependency_critical_path_pictures": round(gop.critical_path(gop.build_stream(4 * world, G, IP, world)), 1),
                       "intra_tasks_per_b_picture": st["n_itasks"], "intra_levels_per_b_picture": st["n_ilevels"],
                       "intra_levels_per_i_picture": wls[-1].stats["n_ilevels"],
                       "h2d_bytes_per_picture": int(mean_stat("h2d_bytes")), "d2h_bytes_per_picture": int(mean_stat("d2h_bytes")),
                       "h2d_bytes_per_step": int(mean_stat("h2d_bytes")) * PPS * L, "h2d_GBps_in_the_timed_region": round(mean_stat("h2d_bytes") * fps / 1e9, 2),
                       "ordered_pass_second_passes": second_passes[0],
                       "check": check,
                       "reference_stream": ref_stream,
                       "live_decoder": live,
                       "launches_per_b_picture": int(js.n_launches),
                       "launches_per_i_picture": int(all_stats[n_jobs].n_launches),
                       "h2d_copies_per_step": round(mean_stat("n_h2d"), 1),
                       "pre_recorded_jobs": len(jobs), "distinct_contents": len(wls),
                       "working_set_bytes": int((dpb_stats.n_live + dpb_stats.n_pool) * FB + len(jobs) * FB),
                       "dpb": {"device_pictures_allocated": int(dpb_stats.n_alloc), "begun": int(dpb_stats.n_begin), "recycled": int(dpb_stats.n_recycled),
                               "peer_copies": int(dpb_stats.n_copies), "waits_for_a_reference": int(dpb_stats.n_waits)},
                       "numa_binding": numa,
                       "pictures_in_flight_per_gpu": S, "host_threads": S + (1 if lookahead else 0), "local_devices": L,
                       "intra_lookahead_pictures": lookahead,
                       "lookahead_thread_hw_queue": {"streams_replaced": q_moved, "in_order_streams_still_sharing_it": q_sharing} if lookahead else None,
                       "picture_assignment": "decoding order; a free frame thread (pthread, own HIP stream) takes the next picture of its device",
                       "n_cu": st["n_cu"], "cu_modes": st["cu_modes"], "n_mc_units": st["n_mc_units"],
                       "n_mcx_units": st["n_mcx_units"], "n_aff_units": st["n_aff_units"], "n_tb_cmds": st["n_tb_cmds"],
                       "r_bar": round(st["r_bar"], 3), "coef_bytes": st["coef_bytes"],
                       "frame_algorithmic_bytes": int(sum(alg.values())),
                       "resident_replay_fps": round(fps_res, 2) if fps_res else None,
                       "ranks": rank_accounts,
                       "exchange_consistent": (None if rank_accounts is None else bool(
                           sum(a["pictures_sent"] for a in rank_accounts) == sum(a["pictures_received"] for a in rank_accounts)
                           and all(a["pictures_sent"] == a["pictures_sent_expected_from_the_dealing"] and a["status"] == 0 for a in rank_accounts)
                           and sum(a["pictures_decoded"] for a in rank_accounts) == n_timed_all
                           and (all(a["rccl"] is None for a in rank_accounts)
                                or sum(a["rccl"]["bytes_sent"] for a in rank_accounts) == sum(a["rccl"]["bytes_received"] for a in rank_accounts)))),
                       "transfers_this_rank": {"sent": xfers[0], "received": xfers[1], "bytes_sent": xfer_bytes_main if (world == 1 or rccl is None) else rccl.stats()["bytes_sent"],
                                               "transport": ("RCCL ncclSend / ncclRecv in the C comm thread (ovvc_rccl.hip)" if world > 1 and rccl is not None else
                                                             "torch.distributed callbacks" if world > 1 else None)},
                       "parallelism": f"{S} pictures in flight per GPU" + (f"; {args.dealing} dealing over {world} GPUs (one process per GPU), pictures other ranks "
                                      "list sent with RCCL point-to-point from the driver's comm thread (no collective)" if world > 1 else "")
                                      + (f"; ONE process, {L} logical devices ({args.dealing} dealing), reference pictures by event-ordered hipMemcpyPeerAsync" if L > 1 else "")},
            "roofline": roofline,
            "cpu_baseline": cpu,
            # how reference pictures travel between GPUs in THIS run, at the top level: a reader of the first multi-GPU
            # record must not have to infer it from the launch command
            "transport": (None if world * L == 1 else
                          f"RCCL ncclSend / ncclRecv, point-to-point, {world} ranks in one communicator (one process per GPU; ovvc_rccl.hip)" if world > 1 and rccl is not None else
                          f"torch.distributed ({'gloo, debug' if debug_gloo else 'nccl'}) send / recv callbacks, {world} ranks (RCCL transport of the C library not available)" if world > 1 else
                          f"hipMemcpyPeerAsync, event-ordered, ONE process driving {L} devices (python bench.py --gpus {L} without a launcher: no RCCL rank "
                          "exists in this run; launch with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` for the RCCL transport)"),
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()



GEN_PIPE = ROOT / "oracle" / "_ref" / "gen_pipe"


def reference_cpu_rates(W, H, n_pics, ncpu):
    """The reference's OWN slice decoder (libovvc/slicedec.c + the rcn slots, compiled from /root/reference where it lay; oracle/ref_harness/
    gen_pipe.c) decoding its chained stream of n_pics WxH pictures: one process on one core, then one process per logical core (each its own
    stream, seeds differ: frame-level parallelism as --framethr gives it), scalar slots and the x86 SSE4.1 / AVX2 back-end.
    -> dict or None when the prebuilt harness is not there."""
    import subprocess
    if not GEN_PIPE.exists():
        return None

    def batch(n, simd):
        # seeds 1000 .. 1255 were run in the build container; 1247 makes the REFERENCE fault in put_vvc_qpel_v under rcn_dmvr_mv_refine (a block on
        # the bottom picture border with far-away vectors: its window runs past the emulated border buffer) -- the parse is a random walk
        # through legal syntax, not an encoder's choice of vectors.  That seed is replaced.
        seed = lambda i: 2000 if i == 247 else 1000 + i
        cmds = [[str(GEN_PIPE), "/tmp", "time", "size", str(W), str(H), "pics", str(n_pics), "seed", str(seed(i))] + (["simd"] if simd else []) for i in range(n)]
        t0 = time.perf_counter()
        ps = [subprocess.Popen(c, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for c in cmds]
        outs = [p.communicate() for p in ps]
        wall = time.perf_counter() - t0
        ok = [o[0] for p, o in zip(ps, outs) if p.returncode == 0 and o[0].strip().startswith("{")]
        if len(ok) < n:
            bad = next((p.returncode, o[1][-300:]) for p, o in zip(ps, outs) if p.returncode != 0 or not o[0].strip().startswith("{"))
            print(f"bench: reference_cpu_rates: {n - len(ok)} of {n} gen_pipe processes failed, first: rc {bad[0]}: {bad[1]}", file=sys.stderr)
            return None
        inner = [json.loads(o.strip().splitlines()[-1])["seconds"] for o in ok]
        return {"fps": n * n_pics / wall, "wall_s": wall, "fps_inside_decoder_mean_per_process": float(np.mean([n_pics / x for x in inner]))}

    out = {}
    try:
        # a process holds its pictures (9 frames + motion planes) and the slice data: ~0.35 GB at 4K -- stay far below the host's free memory
        avail = next((int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable")), 0)
        per = int(W * H * 3 * n_pics * 1.3 + W * H * 2 * 2)
        ncpu = max(1, min(ncpu, int(0.4 * avail / per))) if avail else min(ncpu, 32)
        for simd in (0, 1):
            one, alln = batch(1, simd), batch(ncpu, simd)
            if one is None or alln is None:
                return None
            out["simd" if simd else "scalar"] = {"fps_1_process": round(one["fps_inside_decoder_mean_per_process"], 3), "fps_all_cores": round(alln["fps"], 2),
                                                 "processes": ncpu, "wall_s": round(alln["wall_s"], 2),
                                                 "fps_per_process_under_load": round(alln["fps_inside_decoder_mean_per_process"], 3)}
        # the host side of a decoder WITH the shim installed: the reference's parser + the installed slots recording (no reconstruction on
        # the host) -- what one frame thread has to do per picture before it can submit it (the caller side)
        def shim_seconds(extra):
            best = None
            for _ in range(3):
                # "null": no recorder bound; the slots must not find a device either (they would decode on it): a device index that does
                # not exist makes the shim latch at its first picture and every slot return at its top = the parse alone
                env = dict(os.environ, OVVC_HIP_DEVICE="9999") if extra else None
                p = subprocess.run([str(GEN_PIPE), "/tmp", "time", "shim"] + extra + ["size", str(W), str(H), "pics", str(n_pics)], capture_output=True, text=True, env=env)
                if p.returncode != 0 or not p.stdout.strip().startswith("{"):
                    return None
                v = json.loads(p.stdout.strip().splitlines()[-1])["seconds_shim_record_only"]
                best = v if best is None else min(best, v)
            return best

        t_rec, t_parse = shim_seconds([]), shim_seconds(["null"])
        if t_rec and t_parse:
            out["parse_and_record"] = {"fps_1_thread": round(n_pics / t_rec, 2), "parse_alone_fps_1_thread": round(n_pics / t_parse, 2),
                                       # (the slots have effects the parser reads back -- bS maps, DMVR vectors, LMCS state: without a recorder they
                                       #  return at their top and the parse itself takes other paths; where "parse alone" comes out SLOWER than
                                       #  parse + record the split is not meaningful on this host and is not reported)
                                       "recording_share_of_host_time": round(1 - t_parse / t_rec, 3) if t_parse < t_rec else None,
                                       "what": "slicedec.c's parse (CABAC, partitioning, motion-vector derivation) with the installed shim slots recording, "
                                               "one thread, pictures/s: the rate at which ONE frame thread of a real decoder can feed the device"}
    except (OSError, ValueError, KeyError, IndexError):
        return None
    return out


def live_decoder_rates(W, H, n_pics=33, threads=(1, 2, 4, 8, 16), reps=3):
    """A LIVE decode on this GPU (oracle/ref_harness/gen_pipe.c "live", tests/test_gpu_live.py): the reference's own slice decoder + parser
    (libovvc/slicedec.c, vcl_*.c) drive the INSTALLED shim -- its own device DPB, ovhip_frame_submit launching, every picture copied into
    its OVFrame -- on N frame threads over a chained stream (I + GOPs of 8, seeded slice data), the DMVR slot returning what the shim
    returns and the collocated motion planes holding what the device delivered.  Every frame and every meaningful collocated-motion
    entry is compared with the reference pass (scalar slots, one thread) inside the harness.  The stream is decoded `reps` times on the
    same frame threads (contexts, jobs, page-locked arrays warm); the last repetition is the one reported.
    -> dict or None when the prebuilt harness is not there."""
    import subprocess
    if not GEN_PIPE.exists():
        return None

    def run(tl, extra, exe=GEN_PIPE):
        cmd = [str(exe), "/tmp", "live", "threads", ",".join(str(t) for t in tl), "size", str(W), str(H), "pics", str(n_pics), "reps", str(reps), "profile"] + extra
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        except (OSError, subprocess.TimeoutExpired):
            return None
        rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
        if len(rows) != len(tl):
            print(f"bench: live_decoder_rates: gen_pipe rc {p.returncode}: {p.stderr[-400:]}", file=sys.stderr)
            return None
        # "timeline": per picture of the LAST thread count's last repetition, when its decode call returned (ms from the start)
        import re
        tline = {int(m.group(1)): float(m.group(2)) for m in re.finditer(r"timeline pic\s+(\d+) poc\s+\d+ type \d thr\s+\d+ take\s+[\d.]+ decoded\s+([\d.]+)", p.stderr)}
        if tline:
            rows[-1]["_first_picture_decoded_ms"] = tline.get(0)
        return rows, p.returncode

    def summary(d):
        n = d["pictures"]
        ms = lambda k: 1e3 * d[k] / n
        held, sync = ms("thread_seconds_with_a_picture"), ms("thread_seconds_waiting_for_collocated_rows")
        hooks, dev = ms("thread_seconds_in_shim_hooks"), ms("thread_seconds_in_shim_device_half")
        # the profile brackets every outermost hook call with two time-stamp reads: their cost is measured by the harness and taken out
        over = 1e3 * d.get("shim_profile_overhead_seconds_per_call", 0.0) * d["shim_hook_calls"] / n
        rec = max(0.0, hooks - dev - over)
        parse = max(0.0, held - sync - hooks)
        r = lambda v: round(v, 2)
        return {"pictures_per_second": round(d["pictures_per_second"], 1),
                "frame_thread_ms_per_picture": {"holding_a_picture": r(held), "parse": r(parse), "recording": r(rec), "profile_overhead": r(over),
                                                "device_half_incl_waits_for_reference_pictures": r(dev), "waiting_for_collocated_rows": r(sync)},
                "shares_of_a_frame_threads_time": {"parse": r(parse / held), "recording": r(rec / held), "device_half_and_waits": r((dev + sync) / held)},
                "shim_hook_calls_per_picture": d["shim_hook_calls"] // n}

    a = run(threads, ["timeline"])
    if a is None:
        return None
    b = run((threads[0], threads[-1]), ["noout"])
    rows, rc = a
    out = {"what": "the reference's slice decoder + parser driving the installed shim on this GPU (gen_pipe live): decoded pictures/s WITH parsing, recording, "
                   "uploads, device decode, the eager DMVR rows and the copy of every picture into its OVFrame inside; frames and collocated motion planes "
                   "compared with the reference pass in the same process",
           "stream": {"pictures": n_pics, "width": W, "height": H, "structure": "I + GOPs of 8 (I B8 B4 B2 P6 b1 b3 b5 b7, then B16 ...), seeded slice data: the parse is a "
                      "random walk through legal syntax (DESIGN 2): more and smaller coding units than an encoder would choose", "dmvr_calls": rows[0]["dmvr_calls"],
                      "repetitions_on_warm_frame_threads": reps},
           "bit_exact": rc == 0 and all(d["samples_differing"] == 0 and d["collocated_motion_entries_differing"] == 0 and d["shim_error"] == 0 and d["pictures_decoded"] == n_pics for d in rows),
           "samples_compared": int(W * H * 3 // 2 * n_pics), "collocated_motion_entries_compared": rows[0]["collocated_motion_entries_compared"] // reps,
           "host_frames_recycled": rows[0]["host_frames_recycled"],
           "by_frame_threads": {str(d["frame_threads"]): summary(d) for d in rows},
           "reference_scalar_decoder_same_stream_one_thread_fps": round(n_pics / rows[0]["reference_pass_seconds_inside_slicedec"], 2)}
    # what bounds a SHORT stream: its I picture.  The parse of the random walk's 4K I picture (the reference's own parser, one thread)
    # takes longer than everything else of the stream on 16 threads; no picture of the stream can be complete before it is
    i_ms = rows[-1].pop("_first_picture_decoded_ms", None)
    if i_ms:
        out["the_i_picture"] = {"decode_call_returned_after_ms": round(i_ms, 1), "frame_threads": rows[-1]["frame_threads"],
                                "whole_stream_ms": round(1e3 * rows[-1]["seconds"], 1),
                                "pictures_per_second_if_everything_else_were_free": round(n_pics / (1e-3 * i_ms), 1),
                                "what": "gen_pipe timeline: when the frame thread that took picture 0 (the I picture) returned from the reference's "
                                        "slicedec_decode_rect_entry -- parse (the reference's CABAC + syntax code on one thread) + the shim's hooks.  Every "
                                        "other picture of the stream depends on it: the stream's rate cannot exceed pictures / this time whatever the back-end does"}
    patched = ROOT / "oracle" / "_ref" / "patched" / "gen_pipe"
    c = run((threads[0], threads[-1]), [], patched) if patched.exists() else None
    if c is not None:
        out["patched_caller"] = {"what": "the same with shim/caller.patch applied to the reference (SURVEY 8f-2: the caller hands over whole BDOF / DMVR / affine coding units -- "
                                         "rcn_cu_inter_b, rcn_affine_cu -> ovhip_rec_cu_inter -- instead of one slot call per <= 16x16 / 4x4 block; the shim's stitching is not compiled): "
                                         "frames and collocated motion planes compared with the UNPATCHED reference pass",
                                 "by_frame_threads": {str(d["frame_threads"]): summary(d) for d in c[0]}}
        out["bit_exact"] = out["bit_exact"] and c[1] == 0 and all(d["samples_differing"] == 0 and d["collocated_motion_entries_differing"] == 0 for d in c[0])
    # band-wise submission (ovhip_frame_band; OVVC_HIP_BANDS, OFF by default): the same stream with every picture entering the device CTU row by
    # CTU row while it is parsed -- recorded beside the default so that the choice of the default is a measurement
    bw = run((threads[-1],), ["bands", "1"], patched if patched.exists() else GEN_PIPE)
    if bw is not None:
        d = bw[0][0]
        out["band_wise_submission"] = {"what": "the same stream, patched caller when built, bands of ONE CTU row (gen_pipe `bands 1` = OVVC_HIP_BANDS=1): upload + "
                                               "prediction + residuals + ordered pass of a band at the end of its CTU row, its filters with it, rows posted to the device "
                                               "DPB, dependent pictures' bands and eager DMVR rows going when the rows they read are final",
                                       "frame_threads": d["frame_threads"], "summary": summary(d), "bands_sent": d.get("bands_sent"), "bands_left_to_a_later_hook": d.get("bands_left_to_a_later_hook"),
                                       "bit_exact": bw[1] == 0 and d["samples_differing"] == 0 and d["collocated_motion_entries_differing"] == 0 and d["shim_error"] == 0,
                                       "default": "off: whole-picture submission is faster on this decoder at every band size (DESIGN 12, profiles/r06_live_*)"}
        out["bit_exact"] = out["bit_exact"] and out["band_wise_submission"]["bit_exact"]
    # the steady state: the figures above are one short stream (start-up and tail of 33 pictures on up to 16 threads); a random-access
    # sequence with an intra period of 64 decoded continuously -- I + two GOPs of 32, four such sequences back to back (gen_pipe `cont`:
    # the frame threads take the next intra period's pictures while the tail of the one before still decodes), 260 pictures
    def steady(exe):
        cmd = [str(exe), "/tmp", "live", "threads", "16,32", "size", str(W), str(H), "pics", "65", "gop", "32", "noisp", "seed", "31337", "cont", "4", "reps", "2"]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        except (OSError, subprocess.TimeoutExpired):
            return None
        rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
        if len(rows) != 2 or p.returncode:
            print(f"bench: live_decoder_rates: steady state: gen_pipe rc {p.returncode}: {p.stderr[-400:]}", file=sys.stderr)
            return None
        return {str(d["frame_threads"]): {"pictures_per_second": round(d["pictures_per_second"], 1), "pictures": d["pictures"],
                                          "bit_exact": d["samples_differing"] == 0 and d["collocated_motion_entries_differing"] == 0 and d["shim_error"] == 0 and d["pictures_decoded"] == d["pictures"],
                                          "share_of_a_frame_threads_time_waiting_for_collocated_rows": round(d["thread_seconds_waiting_for_collocated_rows"] / max(1e-9, d["thread_seconds_with_a_picture"]), 2)}
                for d in rows}
    ss = {"unpatched_caller": steady(GEN_PIPE), "patched_caller": steady(patched) if patched.exists() else None}
    if ss["unpatched_caller"] or ss["patched_caller"]:
        ss["what"] = ("a random-access sequence decoded continuously: I + two GOPs of 32 (65 pictures, seeded slice data, ISP off: nearly every such 4K "
                      "stream holds a 64x2 ISP partition, whose result the reference itself leaves undefined), four of them back to back = 260 pictures "
                      "with an I picture every 65, on 16 and 32 frame threads; every frame and collocated motion entry compared with the reference pass")
        out["steady_state"] = ss
        out["bit_exact"] = out["bit_exact"] and all(v["bit_exact"] for side in (ss["unpatched_caller"], ss["patched_caller"]) if side for v in side.values())
    # ISP on over a long stream: nearly every long 4K random-walk stream holds a 64x8 coding unit split horizontally (64x2 partitions), for
    # which the reference's own result is undefined (rcn_Xx2_tb, rcn_transform_tree.c:985-1009): the back-end reconstructs them as H.266
    # defines them (parity unpinned there: tests/spec_isp64x2.py), so the stream is DECODED (shim_error 0) but cannot be compared picture
    # by picture with the reference pass from the first such coding unit on
    try:
        p = subprocess.run([str(patched if patched.exists() else GEN_PIPE), "/tmp", "live", "threads", "16", "size", str(W), str(H), "pics", "129", "gop", "32",
                            "seed", "4242", "reps", "2", "allow64x2"], capture_output=True, text=True, timeout=600)
        rows2 = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    except (OSError, subprocess.TimeoutExpired):
        rows2 = []
    if rows2:
        d = rows2[-1]
        out["isp_on_long_stream"] = {"what": "129 pictures, GOP 32, ISP ON, 16 frame threads (gen_pipe `allow64x2`): the stream holds 64x2 ISP partitions -- no reference result "
                                             "exists for them (the reference pass reads memory nothing wrote; on some seeds it crashes), the back-end follows H.266",
                                     "pictures_per_second": round(d["pictures_per_second"], 1), "pictures_decoded": d["pictures_decoded"], "shim_error": d["shim_error"],
                                     "coding_units_split_into_64x2_partitions": d.get("coding_units_split_into_64x2_isp_partitions"),
                                     "frames_differing_from_the_reference_pass_expected": d["frames_differing"]}
    if b is not None:
        out["output_none"] = {"what": "the same with OVHIP_OUT_NONE (pictures stay on the device; an application takes them through ovhip_shim_frame_output / _digest): "
                                      "collocated motion planes compared, frames not",
                              "by_frame_threads": {str(d["frame_threads"]): summary(d) for d in b[0]}}
        out["bit_exact"] = out["bit_exact"] and b[1] == 0
    return out


def reference_stream_on_device(engine, capi, ctx, W, H, n_pics, reps, extra=()):
    """The same stream through the HIP engine: gen_pipe writes the reference's frames and what the installed shim slots recorded; every
    picture is decoded on the device from the DEVICE's earlier pictures (ovhip_job_flush / _wait: uploads included) and compared byte for
    byte with the reference's frame; then the chain is timed, one picture in flight."""
    import subprocess, tempfile, shutil
    import ctypes as C
    import pipe_cases
    if not GEN_PIPE.exists():
        return None
    d = tempfile.mkdtemp(prefix="ovvc_refstream_")
    try:
        base = [str(GEN_PIPE), d]
        tail = ["size", str(W), str(H), "pics", str(n_pics)] + [str(a) for a in extra]        # (extra: gen_pipe's seed / variant / qp / tiles arguments)
        subprocess.check_call(base + tail, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        subprocess.check_call(base + ["shim"] + tail, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        P = pipe_cases.Pipe("pipe", d)
    except (OSError, subprocess.CalledProcessError):
        shutil.rmtree(d, ignore_errors=True)
        return None
    shutil.rmtree(d, ignore_errors=True)
    jobs, wls, dst = [], [], []
    for k in range(P.n):
        wl = P.workload(k, {i: None for i in range(k)})
        j = engine.Job(ctx, P.w, P.h)
        j.load_workload(wl)
        jobs.append(j); wls.append(wl); dst.append(ctx.new_pic(P.w, P.h))

    def chain():
        for k in range(P.n):
            jobs[k].flush(dst[k], [dst[i] for i in P.ref_indices(k)], None)
            jobs[k].wait()

    chain()
    differ, mv_differ = 0, 0
    for k in range(P.n):
        got = dst[k].download()
        differ += int(sum(int((a != b).sum()) for a, b in zip(got, P.frames[k])))
        calls = P.dmvr_calls(k)
        if len(calls):
            is_dmvr = (wls[k].mcx_units["flags"] & 64) != 0
            mv_differ += int((jobs[k].refined_mvs()[is_dmvr] != calls[:, 8:12]).any(axis=1).sum())
    t0 = time.perf_counter()
    for _ in range(reps):
        chain()
    dt = time.perf_counter() - t0
    # ---- the same pictures through the C stream driver with 16 pictures in flight (next to the synthetic figure): the
    #      stream = `copies` independent repetitions of the recorded pictures, a repetition's pictures referencing that repetition's own
    #      earlier pictures; jobs are shared between repetitions (a job is in flight once at a time), digests of every repetition's
    #      pictures must equal those of the one-at-a-time chain above
    in_flight = None
    try:
        dpb = engine.Dpb((0,))
        contents = []
        for k in range(P.n):
            contents.append({"params": jobs[k].params, "calllog": None, "n_ref_slots": max(1, len(P.ref_indices(k)))})
        copies = max(2, min(8, reps))
        spics = []
        for c in range(copies):
            for k in range(P.n):
                spics.append({"content": k, "job": k, "poc": c * 1000 + int(P.info[k][0]), "device": 0, "refs": [c * P.n + r for r in P.ref_indices(k)]})
        st = engine.Stream(dpb, P.w, P.h, contents, jobs, threads_per_device=16)
        arr = engine.Stream.pics_array(spics)
        want = [dst[k].digest() for k in range(P.n)]
        res, dg = st.run(arr, len(spics), 0, len(spics), digests=True)            # warm (and checked)
        bad = sum(1 for i in range(len(spics)) if bytes(dg[i]) != want[i % P.n])
        t0 = time.perf_counter()
        res, _ = st.run(arr, len(spics), 0, len(spics))
        dt16 = time.perf_counter() - t0
        in_flight = {"pictures_in_flight": 16, "fps": round(len(spics) / dt16, 1), "pictures": len(spics), "independent_repetitions_of_the_stream": copies,
                     "digests_differing_from_the_one_at_a_time_chain": bad, "second_passes": int(res.n_second_passes)}
        st.close(); dpb.close()
    except Exception as e:                                                           # noqa: BLE001  (a secondary figure must not take the line down)
        in_flight = {"error": repr(e)[:300]}
    n_units = {k: int(sum(len(getattr(w, a)) for w in wls)) for k, a in (("mc", "mc_units"), ("refined", "mcx_units"), ("affine", "aff_units"), ("transform_blocks", "tb_cmds"))}
    n_units["ordered_tasks"] = int(sum(0 if w.itasks is None else len(w.itasks) for w in wls))
    n_units["dmvr_calls"] = int(len(P.dmvr))
    for j in jobs:
        j.close()
    return {"pictures": P.n, "width": P.w, "height": P.h, "samples_differing_from_the_reference": differ, "refined_vectors_differing": mv_differ,
            "fps_one_picture_in_flight": round(P.n * reps / dt, 1) if reps else None, "stream_driver_16_in_flight": in_flight, "units": n_units,
            "what": "oracle/_ref/gen_pipe (the reference's slicedec.c compiled where it lay, driven over seeded slice data: DESIGN 2) decoded "
                    f"{P.n} chained {P.w}x{P.h} pictures (I B B B P b b b b of a GOP of 8) with the reference's scalar slots and recorded the "
                    "same parse through the installed shim slots; here the recorded stream went through ovhip_job_flush picture by picture, each "
                    "from the device's own earlier pictures, and every frame and every DMVR vector was compared with the reference's"}

def _bind_to_gpu_numa_node(torch, idx):
    """Frame threads and their page-locked command buffers on the CPUs / memory of the GPU's own NUMA node: the recorder arrays are
    written by a frame thread and read by the GPU's DMA engines; across sockets every access is remote (the recorded_in_run variant
    lost a third of its rate from 16 to 32 threads).  -> {"node": n, "cpus": k} or None."""
    try:
        p = torch.cuda.get_device_properties(idx)
        bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if len(cpus) < 8:
            return None
        os.sched_setaffinity(0, cpus)
        return {"node": node, "cpus": len(cpus)}
    except (OSError, ValueError, AttributeError):
        return None


def _calibration():
    """Reference scalar C (and SIMD) vs the oracle port on identical slot-level cases, timed in the build container (committed; the
    reference does not travel to the GPU box)."""
    try:
        return json.loads((ROOT / "profiles" / "cpu_calibration.json").read_text())
    except (OSError, ValueError):
        return None


def _reference_estimates(cal, port_fps):
    """port frames/s -> what the reference's scalar / SIMD slots would reach on the same host: the port's time is split over the stages
    in the proportions the port spends there on a 4K B picture (profiles/cpu_calibration.json: picture_share), each stage divided by its
    measured port-over-reference ratio."""
    try:
        share = cal["picture_share"]
        st = cal["stages"]
        out = {}
        for key, name in (("port_over_reference", "reference_scalar_estimate"), ("port_over_reference_simd", "reference_simd_estimate")):
            if not all(key in st[k] for k in share):
                continue
            t = sum(share[k] / st[k][key] for k in share)
            out[name] = round(port_fps / t, 2)
        if out:
            out["estimate_method"] = ("value x (sum over stages of the port's time share / measured port-over-reference ratio)^-1, ratios from "
                                      "profiles/cpu_calibration.json (reference's own slots timed in the build container on the fixture cases)")
        return out
    except (KeyError, TypeError, ZeroDivisionError):
        return None


if __name__ == "__main__":
    main()
