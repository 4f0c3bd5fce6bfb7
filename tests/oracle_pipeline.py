"""TEST INFRASTRUCTURE: run a synthetic Workload through the CPU oracle, stage by stage, in the
same order as engine.ResidentPicture.decode()."""
import numpy as np

import oracle_lib
from oracle_lib import HostPic

STAGES = ("mc", "itx", "dbf", "sao", "alf")


def _rows(a, y0, y1, scale_chroma=False):
    if a is None or not len(a):
        return a
    y = a["y"].astype(np.int32)
    if scale_chroma:
        y = np.where(a["plane"] == 0, y, y * 2)
    return a[(y >= y0) & (y < y1)]


def decode(wl, rows=None, stages=STAGES, want_mvs=False):
    """rows=(y0, y1): only the prediction / transform commands whose blocks start inside luma rows [y0, y1)
    (used by band checks before the in-loop filters; meaningful without LMCS chroma scaling only when the
    band starts at row 0).  The picture buffers stay full size."""
    refs = [HostPic(wl.w, wl.h, *r) for r in wl.refs]
    dst = HostPic(wl.w, wl.h)
    units, ux, ua, uc, cmds = wl.mc_units, wl.mcx_units, wl.aff_units, wl.ciip_units, wl.tb_cmds
    luma, chroma = cmds[:wl.n_luma_cmds], cmds[wl.n_luma_cmds:]
    regions = wl.lmcs_regions
    if rows is not None:
        y0, y1 = rows
        units, ux, ua, uc = (_rows(a, y0, y1) for a in (units, ux, ua, uc))
        luma, chroma = _rows(luma, y0, y1, True), _rows(chroma, y0, y1, True)
    mvs = None
    if "mc" in stages:
        oracle_lib.mc(dst, refs, units, wl.lmcs_fwd, intra=HostPic(wl.w, wl.h, *wl.intra) if wl.intra is not None else None)
        if ux is not None and len(ux):
            mvs = oracle_lib.mc_ex(dst, refs, ux, wl.lmcs_fwd)
        if ua is not None and len(ua):
            oracle_lib.mca(dst, refs, ua, wl.aff_side, wl.lmcs_fwd)
        if uc is not None and len(uc):
            oracle_lib.ciip(dst, HostPic(wl.w, wl.h, *wl.intra), uc)
    if "itx" in stages and wl.itasks is not None and len(wl.itasks):
        # picture with ordered tasks: the blocks of those tasks STORE their residual; then the ordered pass in decoding
        # order (the device: level by level); the inverse luma mapping only after it
        assert rows is None
        res = HostPic(wl.w, wl.h)                 # uint16 planes holding int16 bits
        scales = np.zeros(max(1, 0 if regions is None else len(regions)), np.int16)
        oracle_lib.itx_res(dst, luma, wl.coefs, None, res)
        if wl.lmcs is not None and regions is not None and len(regions):
            scales = oracle_lib.lmcs_scale(dst, regions, wl.lmcs)       # skips the regions marked `ordered`
        oracle_lib.itx_res(dst, chroma, wl.coefs, scales, res)
        oracle_lib.intra_tasks(dst, wl.itasks, (res.y.view(np.int16), res.cb.view(np.int16), res.cr.view(np.int16)), regions, wl.lmcs, scales)
        if wl.lmcs is not None:
            oracle_lib.lmcs_inverse(dst, wl.lmcs_bwd)
    elif "itx" in stages:
        if wl.lmcs is None:
            oracle_lib.itx(dst, cmds if rows is None else np.concatenate([luma, chroma]), wl.coefs)
        else:
            oracle_lib.itx(dst, luma, wl.coefs)
            scales = oracle_lib.lmcs_scale(dst, regions, wl.lmcs)
            oracle_lib.itx_ex(dst, chroma, wl.coefs, scales)
            oracle_lib.lmcs_inverse(dst, wl.lmcs_bwd)
    if "dbf" in stages:
        oracle_lib.dbf(dst, wl.dbf_planes)
    if "sao" in stages:
        tmp = HostPic(wl.w, wl.h)
        oracle_lib.sao(tmp, dst, wl.sao_params)
        if "alf" in stages:
            oracle_lib.alf(dst, tmp, wl.alf)
        else:
            dst = tmp
    return (dst, mvs) if want_mvs else dst
