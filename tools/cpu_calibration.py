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
    stages = {k: {"reference_s": round(ref[k], 6), "port_s": round(port[k], 6), "port_over_reference": round(port[k] / ref[k], 3)} for k in port if k in ref}
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    json.dump({"what": "seconds inside the reference's scalar slots (oracle/_ref/gen_golden time: rcn_tu_st / rcn_tu_c, rcn_mcp_b*, "
                       "df.rcn_dbf_ctu, sao.rcn_sao_filter_line, alf.rcn_alf_filter_line, intra_pred*) vs seconds inside the oracle "
                       "port (oracle/liboracle.so) on the same cases of tests/golden/*.ovg; one thread, best of five",
               "host": cpu, "cores_present": os.cpu_count(), "stages": stages,
               "note": "port_over_reference > 1: the port is slower than the reference's scalar C on that stage, i.e. bench.py's "
                       "cpu_baseline UNDERSTATES what the reference's scalar path would reach by about that factor; the reference's "
                       "SIMD back-ends (SSE4 / AVX2, rcn.c:214-299) are not built here (x86 intrinsics need -msse4.1 / -mavx2 and "
                       "their own dispatch) and are faster still"}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
