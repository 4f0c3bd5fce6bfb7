#!/usr/bin/env python
"""bench.py -- decoded frames/s of the MI355X rcn back-end on one synthetic recorded picture.

A "step" is one pass of the whole implemented hot path (prediction -> residual -> in-loop filters,
one frame-wide HIP launch per stage) over one 3840x2160 10-bit 4:2:0 recorded inter picture
(BASELINE.json configs[3]) whose reference pictures, command buffers and coefficient arena are
already resident in HBM.  `--in-flight S` (default 2) pictures are kept in flight per GPU, each on its own HIP stream with its
own buffers and no dependency between them -- the frame-level parallelism of the reference's frame threads (`--framethr`,
ovdec.c:188-248; in a random-access GOP at least every second picture is a non-reference picture): the launch tails, ramps
and latency-bound kernels of one picture are filled by the other's.  N > 1: one process per GPU, every rank decodes its own picture (frame
sharding, `--framethr` style, weak scaling); after each step the rank pushes its reconstructed
picture to the next rank over RCCL point-to-point, where it becomes a reference picture of the
next step (the reference-picture exchange of SURVEY.md 8e) -- no collective on the data path.

Prints ONE JSON line on rank 0 (contract in the task statement).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--seed", type=int, default=0x266)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=2, help="independent pictures in flight per GPU (one HIP stream each)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from openvvc_amd import capi, engine, frames, synth

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
    wl = synth.make_workload(W, H, args.seed + rank)
    S = wl.frame_bytes

    # Every picture slot runs on a torch-owned stream made current while it is driven, so that torch.cuda.Event brackets
    # its kernels and RCCL point-to-point ops order against them (a NULL stream handle would make the engine create a
    # private stream the events cannot see).
    class Slot:
        pass

    def torch_pic(ctx, planes=None):
        """A picture stored in torch int16 tensors (so RCCL can move it), viewed as ovhip_pic."""
        t = torch.empty(H * W * 3 // 2, dtype=torch.int16, device=dev)
        ysz, csz = H * W, (H // 2) * (W // 2)
        s = capi.Pic(t.data_ptr(), t.data_ptr() + 2 * ysz, t.data_ptr() + 2 * (ysz + csz), W, H, W, W // 2)
        p = engine.DevPic(ctx, s, owns=False)
        if planes is not None:
            p.upload(*planes)
        return t, p

    def make_slot():
        sl = Slot()
        sl.stream = torch.cuda.Stream(dev)
        assert sl.stream.cuda_stream, "expected a non-default HIP stream"
        torch.cuda.set_stream(sl.stream)
        sl.ctx = engine.Context(local_rank, stream=sl.stream.cuda_stream)
        sl.rp = engine.ResidentPicture(sl.ctx, wl)
        # pictures that take part in the reference exchange live in torch tensors
        sl.dst_t, sl.rp.dst = torch_pic(sl.ctx)
        sl.ref1_t, sl.rp.refs[1] = torch_pic(sl.ctx, wl.refs[1])
        sl.spare_t, sl.spare = torch_pic(sl.ctx, wl.refs[1])      # receive buffer for the exchanged reference picture
        return sl

    n_slots = max(1, args.in_flight)
    slots = [make_slot() for _ in range(n_slots)]
    rp = slots[0].rp
    torch.cuda.synchronize(dev)

    # one entry per kernel launch of the frame; launches a picture has no work for are dropped
    merged = bool(rp.mcx_units and rp.aff_units)            # k_mcxa: refined + affine units in one launch ("mcx" entry)
    present = {"mcx": rp.mcx_units, "mca": rp.aff_units and not merged, "ciip": rp.ciip_units, "lmcs_scale": rp.lmcs_regions,
               "lmcs_inv": rp.lmcs_bwd and not wl.tb_classes[3],      # else it rides in the chroma ITX launch
               "itx_c": rp.n_luma < rp.tb_cmds.count}
    stages = [k for k in rp.SUBSTAGES if present.get(k, True)]
    evs = {k: [] for k in stages}

    def step(k, timed=(), overlap=False):
        """Picture k (slot k mod S).  timed: names of the launches to bracket with HIP events (each pair costs ~7 us of
        stream time).  overlap: put the independent launches of the prediction stage on side streams (measured slower,
        see engine.py)."""
        sl = slots[k % n_slots]
        torch.cuda.set_stream(sl.stream)
        pending = {}

        def hook(name, phase):
            if name not in timed:
                return
            e = torch.cuda.Event(enable_timing=True)
            e.record(sl.stream)
            if phase == "begin":
                pending[name] = e
            else:
                evs[name].append((pending.pop(name), e))

        sl.rp.overlap = overlap
        for name in sl.rp.STAGES:
            sl.rp.run_stage(name, hook)
        if world > 1:
            # push the reconstructed picture to the rank that lists it as a reference (ring), receive
            # ours into the spare buffer, then swap it in as reference 1 of this slot's next picture
            frames.ring_exchange(dist, sl.dst_t, sl.spare_t, rank, world)
            sl.ref1_t, sl.spare_t = sl.spare_t, sl.ref1_t
            sl.rp.refs[1], sl.spare = sl.spare, sl.rp.refs[1]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(k)
    # Untimed survey pass: every launch bracketed by events, to find the dominant kernel.  Bracketing all 11
    # launches costs ~80 us of stream time per frame (measured), so the timed region below keeps the events
    # of the dominant kernel only; the survey averages are reported as `survey_launch_us`.
    barrier()
    for _ in range(min(20, max(args.steps, 1))):
        step(0, tuple(stages), overlap=False)       # one picture at a time on slot 0: isolated launch durations
    barrier()
    survey = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) * 1e-3 for k, v in evs.items()}
    dom = max(survey, key=survey.get)
    if world > 1:                                   # all ranks bracket the same launch
        pick = torch.tensor([stages.index(dom)], dtype=torch.int64, device=dev)
        dist.broadcast(pick, 0)
        dom = stages[int(pick.item())]
    evs[dom] = []
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, (dom,))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt * 1e3 / args.steps
    fps = world * args.steps / dt

    if rank == 0:
        # ---- per-kernel average launch duration (HIP events on the launch stream, timed region) ----
        kdur = dict(survey)
        kdur[dom] = float(np.mean([a.elapsed_time(b) for a, b in evs[dom]])) * 1e-3     # timed-region average
        st = wl.stats
        tb = wl.tb_cmds
        area = lambda a: int((a["w"].astype(np.int64) * a["h"]).sum())
        nref_area = lambda a: int((a["w"].astype(np.int64) * a["h"] * np.where(a["dir"] == 3, 2, 1)).sum())

        def itx_bytes(c):
            """coefficients actually stored + commands + read-modify-write of the covered samples"""
            n_samples = 1 << (c["log2_w"].astype(np.int64) + c["log2_h"])
            raster = (c["kind"] & 0x80) != 0
            sbs = np.array([bin(int(m)).count("1") for m in c["sig_sb_map"]], np.int64)
            coef = np.where(raster, 2 * n_samples, 32 * sbs).sum()
            covered = n_samples.sum() + n_samples[c["plane2"] != 0xff].sum()
            return int(coef + c.nbytes + 2 * 2 * covered)

        # algorithmic bytes per launch (DESIGN.md "Measurement"): SURVEY 8d terms split per kernel.  A luma
        # sample with its 4:2:0 chroma is 3 bytes; prediction reads nref reference blocks and writes one.
        ciip_area = int((1 << (wl.ciip_units["log2_w"].astype(np.int64) + wl.ciip_units["log2_h"])).sum()) if len(wl.ciip_units) else 0
        fused = wl.mc_units[((wl.mc_units["flags"] & 128) == 0) & (wl.mc_units["aux"] != 0)]      # CIIP blend fused into k_mc2
        alg = {
            "mcp": 3 * (nref_area(wl.mc_units) + area(wl.mc_units) + area(fused)) + wl.mc_units.nbytes,   # + planar samples read
            "mcx": 3 * 3 * area(wl.mcx_units) + wl.mcx_units.nbytes + 16 * len(wl.mcx_units),
            "mca": 3 * (nref_area(wl.aff_units) + area(wl.aff_units)) + wl.aff_units.nbytes + wl.aff_side.nbytes,
            "ciip": 3 * 3 * ciip_area + wl.ciip_units.nbytes,              # intra read + inter read-modify-write
            "itx_l": itx_bytes(tb[:rp.n_luma]),
            "lmcs_scale": st["n_lmcs_regions"] * (128 * 2 + 8 + 2),
            "itx_c": itx_bytes(tb[rp.n_luma:]),
            "lmcs_inv": 2 * 2 * W * H,                                     # luma plane read + write
            "dbf": 2 * S + rp.dbf_v.nbytes + rp.dbf_h.nbytes,                # read + write the picture, compact edge lists
            "sao": 2 * S + wl.sao_params.nbytes,
            "alf": 2 * S + rp.alf.nbytes,
        }
        if merged:
            alg["mcx"] += alg.pop("mca")
        if "lmcs_inv" not in kdur and rp.lmcs_bwd:
            alg["itx_c"] += alg.pop("lmcs_inv")
        alg = {k: v for k, v in alg.items() if k in kdur}
        achieved = alg[dom] / kdur[dom] / 1e9
        kname = {"mcp": "k_mc2", "mcx": "k_mcxa" if merged else "k_mcx", "mca": "k_mca", "ciip": "k_ciip", "itx_l": "k_itx_all (luma commands)",
                 "itx_c": "k_itx_all (chroma commands + inverse-LMCS rider)", "lmcs_scale": "k_lmcs_scale", "lmcs_inv": "k_lmcs_inverse",
                 "dbf": "k_dbf_list<0> + k_dbf_list<1>", "sao": "k_sao", "alf": "k_alf"}
        # HBM traffic of the dominant kernel: rocprofv3 PMC passes of this same command cannot run inside the timed
        # process, so the committed summary of the latest pass (profiles/traffic.json, per dispatch) is quoted when it
        # was taken on this workload; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.
        traffic = None
        try:
            tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
            if tj["workload"] == {"width": W, "height": H, "seed": args.seed}:
                names = [n.strip() for n in kname[dom].split("(")[0].split("+")]
                ks = [tj["kernels"][n] for n in names]
                traffic = int(sum(2 * k["fetch_kib"] + k["write_kib"] for k in ks) * 1024)
        except (OSError, KeyError, ValueError):
            traffic = None
        # `achieved` is what the spec asks for: algorithmic bytes / the launch duration seen in the timed region -- with
        # S > 1 pictures in flight the kernel shares the chip with the other pictures' kernels, so its own launch stretches
        # while the chip delivers more frames; `isolated_*` is the same kernel alone on the chip (survey pass).
        roofline = {"bound": "hbm", "kernel": kname[dom], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "avg_launch_us": round(kdur[dom] * 1e6, 2),
                    "isolated_launch_us": round(survey[dom] * 1e6, 2),
                    "isolated_frac": round(alg[dom] / survey[dom] / 1e9 / HBM_PEAK_GBPS, 5),
                    "survey_launch_us": {k: round(v * 1e6, 2) for k, v in survey.items()},
                    "algorithmic_bytes": {k: int(v) for k, v in alg.items()}}

        cpu = None
        if not args.no_cpu_baseline:
            import oracle_pipeline
            reps, tc = 0, 0.0
            while tc < 10.0 and reps < 64:                     # ~10 s of scalar CPU work
                t1 = time.perf_counter()
                oracle_pipeline.decode(wl)
                tc += time.perf_counter() - t1
                reps += 1
            cpu = {"value": round(reps / tc, 4), "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": f"oracle/liboracle.so (scalar C restatement, 1 thread) decoding the same {W}x{H} "
                             f"recorded picture {reps}x in {tc:.2f} s on this box's host CPU "
                             f"({os.cpu_count()} logical cores present)"}

        out = {
            "metric": "decoded frames/sec, full rcn back-end (MC incl. BDOF/DMVR/affine-PROF/GPM/CIIP + LMCS + inverse "
                      "transform + deblocking + SAO + ALF/CC-ALF), 4K 10-bit RA recorded picture, bit-exact vs oracle",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16 samples / int16 coefficients / int32 accumulate", "data": "synthetic",
            "config": {"workload": f"{W}x{H} 10-bit 4:2:0 synthetic recorded inter picture (BASELINE configs[3]), "
                                   f"seed {hex(args.seed)}, stages {'+'.join(stages)}",
                       "n_cu": st["n_cu"], "cu_modes": st["cu_modes"], "n_mc_units": st["n_mc_units"],
                       "n_mcx_units": st["n_mcx_units"], "n_aff_units": st["n_aff_units"], "n_tb_cmds": st["n_tb_cmds"],
                       "r_bar": round(st["r_bar"], 3), "coef_bytes": st["coef_bytes"],
                       "frame_algorithmic_bytes": int(sum(alg.values())),
                       "pictures_in_flight_per_gpu": n_slots,
                       "parallelism": f"frames x{world}" + (" + RCCL p2p reference exchange" if world > 1 else "")
                                      + f", {n_slots} independent pictures in flight per GPU (one HIP stream each)"},
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
