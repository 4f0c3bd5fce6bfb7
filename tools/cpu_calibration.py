#!/usr/bin/env python
"""profiles/cpu_calibration.json: the reference's scalar C slots (oracle/_ref, compiled from /root/reference) against the
oracle port (oracle/liboracle.so) on the SAME cases -- the cases of tests/golden/*.ovg -- timed in the build container, one
thread, best of five.  bench.py's cpu_baseline is the port (the reference does not travel to the GPU box); this file says how
the port's speed relates to the reference's, stage by stage.

usage (this container, after `make -C oracle`): python tools/cpu_calibration.py > profiles/cpu_calibration.json"""
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import golden_cases   # noqa: E402
import golden_io      # noqa: E402
import oracle_lib     # noqa: E402
from oracle_lib import HostPic   # noqa: E402
from openvvc_amd import capi     # noqa: E402


def best(fn, n=5):
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def main():
    with tempfile.TemporaryDirectory() as d:
        out = subprocess.run([str(ROOT / "oracle" / "_ref" / "gen_golden"), d, "time"], capture_output=True, text=True, check=True)
    ref = json.loads(out.stdout.strip().splitlines()[-1])
    with tempfile.TemporaryDirectory() as d:
        out = subprocess.run([str(ROOT / "oracle" / "_ref" / "gen_golden"), d, "simd", "time"], capture_output=True, text=True, check=True)
    simd = json.loads(out.stdout.strip().splitlines()[-1])
    port = {}
    # itx: the 425 rcn_tu_st / rcn_tu_c cases
    pic, cmds, coefs, _, _ = golden_cases.itx_cases()
    port["itx"] = best(lambda: oracle_lib.itx(pic, cmds, coefs))
    # mc: the 750 rcn_mcp_b* cases, all units in one call
    refs, descs, _, _ = golden_cases.mc_cases()
    rec = capi.Recorder(refs[0].w, refs[0].h)
    units = []
    for dsc in descs:
        rec.reset(); rec.pu(dsc); units.append(rec.mc_units().copy())
    units = np.concatenate(units)
    dst = HostPic(refs[0].w, refs[0].h)
    port["mc"] = best(lambda: oracle_lib.mc(dst, refs, units))
    # in-loop filters: the three pictures of each fixture
    dc = golden_cases.dbf_cases()
    port["dbf"] = best(lambda: [oracle_lib.dbf(p.copy(), planes) for p, planes, _ in dc]) - best(lambda: [p.copy() for p, _, _ in dc])
    sc = golden_cases.sao_cases()
    port["sao"] = best(lambda: [oracle_lib.sao(HostPic(p.w, p.h), p, prm) for p, prm, _ in sc]) - best(lambda: [HostPic(p.w, p.h) for p, _, _ in sc])
    ac = golden_cases.alf_cases()
    port["alf"] = best(lambda: [oracle_lib.alf(HostPic(p.w, p.h), p, a) for p, a, _ in ac]) - best(lambda: [HostPic(p.w, p.h) for p, _, _ in ac])
    # intra: the 6506 intra_pred / intra_pred_mrl / rcn_intra_mip / intra_pred_c cases
    g = golden_io.load("intra.ovg")
    tasks = np.frombuffer(g["task"].tobytes(), dtype=capi.ITASK_DTYPE)
    ip = HostPic(g["pic_y"].shape[1], g["pic_y"].shape[0], g["pic_y"], g["pic_cb"], g["pic_cr"])
    port["intra"] = best(lambda: oracle_lib.intra_tasks(ip, tasks))
    stages = {k: {"reference_s": round(ref[k], 6), "reference_simd_s": round(simd[k], 6), "port_s": round(port[k], 6),
                  "port_over_reference": round(port[k] / ref[k], 3), "port_over_reference_simd": round(port[k] / simd[k], 3)} for k in port if k in ref}
    # where the port spends its time on the picture bench.py's cpu_baseline decodes (a B picture with every tool, 12 % intra CUs;
    # 1920x1080 here, the shares do not depend on the size): cumulative stage masks, differences
    import oracle_pipeline
    from openvvc_amd import synth
    wl = synth.make_workload(1920, 1080, 0x266, tools=synth.INTRA_TOOLS, intra_frac=0.12)
    cum, prev, share = [], 0.0, {}
    for k, name in enumerate(oracle_pipeline.STAGES):
        t = best(lambda: oracle_pipeline.decode(wl, stages=oracle_pipeline.STAGES[:k + 1]), 3)
        share[name] = max(t - prev, 0.0)
        prev = t
    # the "itx" stage of such a picture contains the ordered intra pass: split it with the stand-alone intra timing ratio
    wl0 = synth.make_workload(1920, 1080, 0x266, tools=synth.ALL_TOOLS, intra_frac=0.0)
    t_itx0 = best(lambda: oracle_pipeline.decode(wl0, stages=("mc", "itx")), 3) - best(lambda: oracle_pipeline.decode(wl0, stages=("mc",)), 3)
    intra_part = max(share["itx"] - t_itx0, 0.0)
    share["intra"] = intra_part; share["itx"] = share["itx"] - intra_part
    tot = sum(share.values())
    share = {k: round(v / tot, 4) for k, v in share.items()}
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    json.dump({"what": "seconds inside the reference's scalar slots (oracle/_ref/gen_golden time: rcn_tu_st / rcn_tu_c, rcn_mcp_b*, "
                       "df.rcn_dbf_ctu, sao.rcn_sao_filter_line, alf.rcn_alf_filter_line, intra_pred*) vs seconds inside the oracle "
                       "port (oracle/liboracle.so) on the same cases of tests/golden/*.ovg; one thread, best of five",
               "host": cpu, "cores_present": os.cpu_count(), "stages": stages, "picture_share": share,
               "simd": "the reference's x86 back-end (libovvc/x86/*_sse.c, *_avx2.c, -msse4.1 -mavx2 -DBITDEPTH=10) installed over the scalar table in "
                       "the order of rcn.c:216-254 by `gen_golden <dir> simd time`; rcn_sao_sse.c needs the autoconf-generated ovconfig.h and is not "
                       "built (SAO runs the AVX2 override).  Its results are NOT byte-identical to the scalar slots on every harness case: "
                       "`gen_golden <dir> simd` reproduces dbf / gpm / intra / lmcs / mc / mca / mcx.ovg byte for byte and differs on alf / sao / itx / "
                       "intra_ctu / isp.ovg (the harness draws out-of-range corner inputs); the scalar slots stay the oracle",
               "note": "port_over_reference > 1: the port is slower than the reference's scalar C on that stage, i.e. bench.py's "
                       "cpu_baseline UNDERSTATES what the reference's scalar path would reach by about that factor, and its SIMD "
                       "path by port_over_reference_simd; bench.py prints both estimates (reference_scalar_estimate, "
                       "reference_simd_estimate = value / sum over stages of picture_share / ratio)"}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
