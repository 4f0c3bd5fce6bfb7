"""TEST INFRASTRUCTURE: run a synthetic Workload through the CPU oracle, stage by stage, in the
same order as engine.ResidentPicture.decode()."""
import numpy as np

import oracle_lib
from oracle_lib import HostPic

STAGES = ("mc", "itx", "dbf", "sao", "alf")


def decode(wl, rows=None, stages=STAGES):
    """rows=(y0, y1): only the MC / ITX commands whose blocks start inside luma rows [y0, y1)
    (used by band checks before the in-loop filters); the picture buffers stay full size."""
    refs = [HostPic(wl.w, wl.h, *r) for r in wl.refs]
    dst = HostPic(wl.w, wl.h)
    units, cmds = wl.mc_units, wl.tb_cmds
    if rows is not None:
        y0, y1 = rows
        units = units[(units["y"] >= y0) & (units["y"] < y1)]
        ly = np.where(cmds["plane"] == 0, cmds["y"], cmds["y"].astype(np.int32) * 2)
        cmds = cmds[(ly >= y0) & (ly < y1)]
    if "mc" in stages:
        oracle_lib.mc(dst, refs, units)
    if "itx" in stages:
        oracle_lib.itx(dst, cmds, wl.coefs)
    if "dbf" in stages:
        oracle_lib.dbf(dst, wl.dbf_planes)
    if "sao" in stages:
        tmp = HostPic(wl.w, wl.h)
        oracle_lib.sao(tmp, dst, wl.sao_params)
        if "alf" in stages:
            oracle_lib.alf(dst, tmp, wl.alf)
        else:
            dst = tmp
    return dst
