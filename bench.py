#!/usr/bin/env python
"""bench.py -- decoded frames/s of the MI355X rcn back-end, one DECODE STEP per picture.

A "step" is what the reference-side shim does at the end of a parsed picture (shim/rcn_hip.c flush_picture ->
ovhip_job_flush, all in C): asynchronous H2D of the picture's recorded command buffers + coefficient arena + deblocking
edge lists + filter parameters out of page-locked memory, the launch chain of the whole rcn path (prediction incl.
BDOF / DMVR / affine-PROF / GPM / CIIP + LMCS, inverse quantisation / LFNST / transforms + residual, deblocking, SAO,
ALF / CC-ALF; intra prediction of the picture's intra CUs as a dependency-ordered pass), and the D2H of the DMVR-refined
motion vectors.  The workload is a synthetic recorded 3840x2160 10-bit 4:2:0 random-access stream (BASELINE.json
configs[3]): B pictures with `--intra-frac` of their CUs intra, and `--i-sets` of the picture sets an I picture.

Nothing is replayed out of cache: the steps rotate over `--sets` picture sets (reference pictures, intra picture,
destination, command buffers; distinct addresses, `--contents` distinct recorded pictures), sized so that the working
set is several times the 256 MiB Infinity Cache.  `--in-flight S` pictures are in flight per GPU (one HIP stream + one
host thread each: the reference's frame threads, ovdec.c:188-248).

N > 1: one process per GPU, frames sharded --framethr style; reference pictures move between ranks with RCCL
point-to-point only where the GOP's reference lists need them (openvvc_amd/gop.py); no collective on the data path.

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
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
         "sao": "k_sao", "alf": "k_alf", "intra": "k_intra_level (all levels)", "h2d": "H2D copies"}


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
    ev, eh = capi.dbf_compact(wl.dbf_planes, 0), capi.dbf_compact(wl.dbf_planes, 1)
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
    ap.add_argument("--steps", type=int, default=240)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--seed", type=int, default=0x266)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=6, help="pictures in flight per GPU (one HIP stream + one host thread each)")
    ap.add_argument("--sets", type=int, default=32, help="picture sets the steps rotate over = one intra period (distinct addresses; working set = sets x ~100 MB at 4K)")
    ap.add_argument("--contents", type=int, default=2, help="distinct recorded pictures (seeds) among the sets")
    ap.add_argument("--intra-frac", type=float, default=0.12, help="share of intra CUs in the B pictures")
    ap.add_argument("--i-sets", type=int, default=1, help="picture sets that hold an I picture (all CUs intra)")
    ap.add_argument("--intra-ctu", action="store_true", help="ordered pass as the one-launch CTU wavefront instead of one launch per level")
    ap.add_argument("--host-threads", type=int, default=-1, help="host threads issuing the flushes (-1: one per picture in flight)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from openvvc_amd import capi, engine, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    W, H = args.width, args.height
    S = max(1, args.in_flight)
    K = max(S, (args.sets + S - 1) // S * S)
    tools = synth.INTRA_TOOLS if (args.intra_frac > 0 or args.i_sets) else synth.ALL_TOOLS
    wls = [synth.make_workload(W, H, args.seed + 1000 * c + rank, tools=tools, intra_frac=args.intra_frac)
           for c in range(max(1, min(args.contents, K)))]
    n_b = len(wls)
    if args.i_sets:
        wls.append(synth.make_workload(W, H, args.seed + 7777 + rank, tools=synth.INTRA_TOOLS, intra_frac=1.0))
    content_of = [n_b if k < args.i_sets else k % n_b for k in range(K)]       # set 0.. i_sets-1: the I picture
    FB = wls[0].frame_bytes

    # ---- picture sets: every set has its own reference pictures, intra picture, destination and job (= command buffers,
    # tmp picture), at distinct addresses.  Pictures live in torch tensors so that RCCL can move them (N > 1).
    ctxs = [engine.Context(local_rank) for _ in range(S)]
    ext = [torch.cuda.ExternalStream(c.stream, device=dev) for c in ctxs]

    def torch_pic(ctx, planes=None):
        t = torch.empty(H * W * 3 // 2, dtype=torch.int16, device=dev)
        ysz, csz = H * W, (H // 2) * (W // 2)
        s = capi.Pic(t.data_ptr(), t.data_ptr() + 2 * ysz, t.data_ptr() + 2 * (ysz + csz), W, H, W, W // 2)
        p = engine.DevPic(ctx, s, owns=False)
        if planes is not None:
            p.upload(*planes)
        return t, p

    class Set:
        pass

    sets = []
    for k in range(K):
        st = Set()
        st.slot = k % S
        st.ctx = ctxs[st.slot]
        st.lock = threading.Lock()
        st.wl = wls[content_of[k]]
        st.job = engine.Job(st.ctx, W, H)
        st.job.load_workload(st.wl)
        st.ref_t, st.refs = zip(*[torch_pic(st.ctx, r) for r in st.wl.refs])
        st.ref_t, st.refs = list(st.ref_t), list(st.refs)
        st.intra = torch_pic(st.ctx, st.wl.intra)[1] if st.wl.intra is not None else None
        st.dst_t, st.dst = torch_pic(st.ctx)
        st.spare_t, st.spare = torch_pic(st.ctx, st.wl.refs[1]) if world > 1 else (None, None)
        sets.append(st)
    torch.cuda.synchronize(dev)
    working_set = K * ((len(wls[0].refs) + 2 + (wls[0].intra is not None)) * FB)

    # ---- N > 1: frames shard across ranks; a decoded picture is pushed to the next rank, where it replaces reference 1
    # of that slot's next picture -- issued on the slot's stream, never waited for on the host
    def exchange(st, slot):
        with torch.cuda.stream(ext[slot]):
            ops = [dist.P2POp(dist.isend, st.dst_t, (rank + 1) % world), dist.P2POp(dist.irecv, st.spare_t, (rank - 1) % world)]
            dist.batch_isend_irecv(ops)
        st.ref_t[1], st.spare_t = st.spare_t, st.ref_t[1]
        st.refs[1], st.spare = st.spare, st.refs[1]

    lv = capi.STAGE_INTRA_CTU if args.intra_ctu else 0
    nthreads = S if args.host_threads < 0 else max(1, min(args.host_threads, S))

    def run_steps(first, n, resident=False):
        """Steps [first, first + n): step i decodes picture set i mod K.  One host thread per picture in flight, each with its
        own HIP stream; a thread that became free takes the next picture (the reference's frame threads, ovdec.c:188-248) and,
        like the shim's flush_picture, waits for its picture before it takes another."""
        def one(i, slot, wait):
            st = sets[i % K]
            with st.lock:
                st.job.bind(ctxs[slot])
                st.job.params.stages = ((capi.STAGE_ALL | capi.STAGE_RESIDENT) if resident else capi.STAGE_ALL) | lv
                st.job.flush(st.dst, st.refs, st.intra)
                if world > 1:
                    exchange(st, slot)
                if wait:
                    st.job.wait()
        if nthreads == 1 or world > 1:
            for i in range(first, first + n):
                one(i, i % S, False)
            return
        errs, nxt, nlock = [], [first], threading.Lock()

        def worker(slot):
            try:
                while True:
                    with nlock:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= first + n:
                        return
                    one(i, slot, True)
            except Exception as e:          # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=worker, args=(s,)) for s in range(nthreads)]
        [t.start() for t in th]
        [t.join() for t in th]
        if errs:
            raise errs[0]

    def barrier():
        if world > 1:
            dist.barrier()
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize(dev)

    def timed(n, resident=False):
        barrier()
        t0 = time.perf_counter()
        run_steps(0, n, resident)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def set_timer(name):
        for st in sets:
            st.job.time_stage(name)

    def read_timer():
        tot, cnt = 0.0, 0
        for st in sets:
            s, n = st.job.stage_time()
            tot += s; cnt += n
        return tot / max(cnt, 1)

    run_steps(0, max(args.warmup, K))            # every picture set flushed at least once
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
        run_steps(0, 2 * K)
        barrier()
        survey[name] = read_timer()
    kern = {k: v for k, v in survey.items() if k != "h2d"}
    dom = max(kern, key=kern.get)
    if world > 1:
        pick = torch.tensor([present.index(dom)], dtype=torch.int64, device=dev)
        dist.broadcast(pick, 0)
        dom = present[int(pick.item())]

    # ---- timed region: EXACTLY --steps decode steps, only the dominant launch group bracketed
    set_timer(dom)
    dt = timed(args.steps)
    dom_avg = read_timer()
    set_timer(None)
    ms_per_step = dt * 1e3 / args.steps
    fps = world * args.steps / dt

    # secondary figure: the round-1 measurement (device-resident replay of the same command buffers, no H2D / D2H)
    dt_res = timed(min(args.steps, 120), resident=True)
    fps_res = world * min(args.steps, 120) / dt_res

    if rank == 0:
        algs = [algorithmic_bytes(wl, FB) for wl in wls]
        use = np.bincount(content_of, minlength=len(wls)).astype(np.float64)
        use /= use.sum()
        alg = {k: float(sum(u * a[k] for u, a in zip(use, algs))) for k in algs[0]}
        alg = {k: v for k, v in alg.items() if k in kern}
        achieved = alg[dom] / dom_avg / 1e9
        traffic = None
        try:
            tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
            if tj["workload"] == {"width": W, "height": H, "seed": args.seed}:
                names = [n.strip() for n in KNAME[dom].split("(")[0].split("+")]
                ks = [tj["kernels"][n] for n in names]
                traffic = int(sum(2 * k["fetch_kib"] + k["write_kib"] for k in ks) * 1024)
        except (OSError, KeyError, ValueError):
            traffic = None
        roofline = {"bound": "hbm", "kernel": KNAME[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "avg_launch_us": round(dom_avg * 1e6, 2),
                    "picked_from": "survey in the timed configuration (same rotation and pictures in flight)",
                    "survey_launch_us": {k: round(v * 1e6, 2) for k, v in survey.items()},
                    "frac_per_kernel": {k: round(alg[k] / kern[k] / 1e9 / HBM_PEAK_GBPS, 5) for k in alg},
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
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16 samples / int16 coefficients / int32 accumulate", "data": "synthetic",
            "config": {"workload": f"{W}x{H} 10-bit 4:2:0 synthetic recorded RA pictures (BASELINE configs[3]): {K - args.i_sets} B "
                                   f"picture sets with {args.intra_frac:.0%} intra CUs + {args.i_sets} I picture set(s), seeds "
                                   f"{[hex(w.seed) for w in wls]}, per-picture flush in C (ovhip_job_flush)",
                       "intra_tasks_per_b_picture": st["n_itasks"], "intra_levels_per_b_picture": st["n_ilevels"],
                       "intra_levels_per_i_picture": wls[-1].stats["n_ilevels"] if args.i_sets else None,
                       "h2d_bytes_per_step": int(mean_stat("h2d_bytes")), "d2h_bytes_per_step": int(mean_stat("d2h_bytes")),
                       "launches_per_step": round(mean_stat("n_launches"), 1), "h2d_copies_per_step": round(mean_stat("n_h2d"), 1),
                       "launches_per_b_picture": int(js.n_launches),
                       "launches_per_i_picture": int(all_stats[0].n_launches) if args.i_sets else None,
                       "distinct_pictures": K, "distinct_contents": len(wls), "working_set_bytes": int(working_set),
                       "pictures_in_flight_per_gpu": S,
                       "host_threads": 1 if world > 1 else nthreads,
                       "picture_assignment": "static round-robin, asynchronous" if (world > 1 or nthreads == 1)
                                             else "a free host thread takes the next picture, flushes it and waits for it (as the shim does)",
                       "recorder_in_timed_region": False,
                       "n_cu": st["n_cu"], "cu_modes": st["cu_modes"], "n_mc_units": st["n_mc_units"],
                       "n_mcx_units": st["n_mcx_units"], "n_aff_units": st["n_aff_units"], "n_tb_cmds": st["n_tb_cmds"],
                       "r_bar": round(st["r_bar"], 3), "coef_bytes": st["coef_bytes"],
                       "frame_algorithmic_bytes": int(sum(alg.values())),
                       "resident_replay_fps": round(fps_res, 2),
                       "parallelism": f"frames x{world}" + (" + RCCL p2p reference push on the picture's stream" if world > 1 else "")
                                      + f", {S} pictures in flight per GPU"},
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
