"""TEST INFRASTRUCTURE: run a synthetic Workload through the CPU oracle, stage by stage, in the
same order as engine.ResidentPicture.decode()."""
import numpy as np

import oracle_lib
from oracle_lib import HostPic


def decode(wl, rows=None):
    """rows=(y0, y1): only the commands whose blocks start inside luma rows [y0, y1) (bounded CPU
    baseline sample); the picture buffers stay full size."""
    refs = [HostPic(wl.w, wl.h, *r) for r in wl.refs]
    dst = HostPic(wl.w, wl.h)
    units, cmds = wl.mc_units, wl.tb_cmds
    if rows is not None:
        y0, y1 = rows
        units = units[(units["y"] >= y0) & (units["y"] < y1)]
        ly = np.where(cmds["plane"] == 0, cmds["y"], cmds["y"].astype(np.int32) * 2)
        cmds = cmds[(ly >= y0) & (ly < y1)]
    oracle_lib.mc(dst, refs, units)
    oracle_lib.itx(dst, cmds, wl.coefs)
    return dst
