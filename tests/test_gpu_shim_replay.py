"""GPU: the command streams the installed slots of the reference-side override block recorded (tests/golden/shim_*.ovg,
shim_cases.py) replayed through the HIP engine reproduce the bytes the reference's scalar slots produced."""
import numpy as np
import pytest

import golden_cases
import golden_io
from oracle_lib import HostPic
from openvvc_amd import capi, engine
from shim_cases import ShimStream

pytestmark = pytest.mark.gpu
BAND = 128


@pytest.fixture(scope="module")
def ctx(built_lib):
    c = engine.Context(0)
    yield c
    c.close()


def test_shim_tu_streams_gpu(ctx):
    """All 425 rcn_tu_st / rcn_tu_c cases + the 12 transform trees, each on its own 128-row band, ONE launch per stream."""
    g = golden_io.load("itx.ovg")
    for stream, key_n, tree_of in (("shim_itx.ovg", "desc", None), ("shim_itx_tree.ovg", "tt_desc", 0)):
        s = ShimStream(stream)
        n = s.n
        tall = ctx.upload_pic(np.tile(g["pred_y"], (n, 1)), np.tile(g["pred_cb"], (n, 1)), np.tile(g["pred_cr"], (n, 1)))
        cmds, coefs, rects, coef_base = [], [], [], 0
        for i in range(n):
            c = s.case(i)
            tb = c["tb"]
            tb["y"] += np.where(tb["plane"] == 0, i * BAND, i * (BAND // 2)).astype(np.uint16)
            tb["coef_off"] += coef_base
            coef_base += len(c["coef"])
            cmds.append(tb); coefs.append(c["coef"])
            if tree_of is None:
                d = capi.TuDesc.from_buffer_copy(g["desc"][i].tobytes())
                x0, y0, w, h, eo = d.x0, d.y0, 1 << d.log2_tb_w, 1 << d.log2_tb_h, g["exp_off"][i]
                if d.tree == 0:
                    rects += [(0, x0, y0 + i * BAND, w, h, int(eo[0])), (1, x0 >> 1, (y0 >> 1) + i * 64, w >> 1, h >> 1, int(eo[1])),
                              (2, x0 >> 1, (y0 >> 1) + i * 64, w >> 1, h >> 1, int(eo[2]))]
                else:
                    rects += [(1, x0, y0 + i * 64, w, h, int(eo[1])), (2, x0, y0 + i * 64, w, h, int(eo[2]))]
            else:
                d = capi.TtDesc.from_buffer_copy(g["tt_desc"][i].tobytes())
                w, h, eo = 1 << d.log2_w, 1 << d.log2_h, g["tt_exp_off"][i]
                rects += [(0, 0, i * BAND, w, h, int(eo[0])), (1, 0, i * 64, w >> 1, h >> 1, int(eo[1])), (2, 0, i * 64, w >> 1, h >> 1, int(eo[2]))]
        ctx.itx(tall, ctx.upload(np.concatenate(cmds)), ctx.upload(np.concatenate(coefs)))
        ctx.sync()
        y, cb, cr = tall.download()
        golden_cases.check_rects(HostPic(128, BAND * n, y, cb, cr), rects, g["exp"], f"{stream} HIP vs reference")


def _tall(ctx, rw, rh, n):
    fill = np.full((rh * n, rw), 0xABAB, np.uint16)
    return ctx.upload_pic(fill, fill[: rh * n // 2, : rw // 2], fill[: rh * n // 2, : rw // 2])


def _rects(d, exp_off, i, rh):
    w, h = 1 << d.log2_w, 1 << d.log2_h
    return [(0, d.x0, d.y0 + i * rh, w, h, int(exp_off[i, 0])), (1, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 1])),
            (2, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 2]))]


def test_shim_prediction_streams_gpu(ctx):
    """rcn_mcp_b*, GPM / CIIP, BDOF / DMVR (with the refined vectors) and affine streams."""
    # ---- plain + GPM / CIIP
    for stream, loader in (("shim_mc.ovg", "mc"), ("shim_gpm.ovg", "gpm")):
        if loader == "mc":
            refs, descs, exp_off, exp = golden_cases.mc_cases(); intra = None; n_gpm = len(descs)
        else:
            refs, intra_h, descs, _, n_gpm, exp_off, exp = golden_cases.gpm_cases()
            intra = ctx.upload_pic(intra_h.y, intra_h.cb, intra_h.cr)
        s = ShimStream(stream)
        rw, rh, n = refs[0].w, refs[0].h, len(descs)
        drefs = s.refs([ctx.upload_pic(r.y, r.cb, r.cr) for r in refs])
        tall = _tall(ctx, rw, rh, n)
        rects = []
        for i, d in enumerate(descs):
            c = s.case(i)
            band = tall.band(i * rh, rh)
            ctx.mc(band, drefs, ctx.upload(c["mc"]))
            if len(c["ciip"]):
                ctx.ciip(band, intra, ctx.upload(c["ciip"]))
            rects += _rects(d, exp_off, i, rh)
        ctx.sync()
        y, cb, cr = tall.download()
        golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, f"{stream} HIP vs reference")
    # ---- refined
    refs, descs, exp_off, exp, exp_mv = golden_cases.mcx_cases()
    s = ShimStream("shim_mcx.ovg")
    rw, rh, n = refs[0].w, refs[0].h, len(descs)
    drefs = s.refs([ctx.upload_pic(r.y, r.cb, r.cr) for r in refs])
    tall = _tall(ctx, rw, rh, n)
    rects, mv_checks = [], []
    for i, d in enumerate(descs):
        c = s.case(i)
        band = tall.band(i * rh, rh)
        if len(c["mc"]):
            ctx.mc(band, drefs, ctx.upload(c["mc"]))
        mv = ctx.alloc(len(c["mcx"]) * 16)
        ctx.mcx(band, drefs, ctx.upload(c["mcx"]), mv_out=mv)
        if d.refine & capi.PU_DMVR:
            mv_checks.append((i, mv, exp_mv[int(exp_off[i, 3]) // 4:int(exp_off[i, 3]) // 4 + len(c["mcx"])]))
        rects += _rects(d, exp_off, i, rh)
    ctx.sync()
    for i, buf, want in mv_checks:
        assert np.array_equal(buf.download(np.int32).reshape(-1, 4), want), f"shim mcx case {i}: refined MVs differ"
    y, cb, cr = tall.download()
    golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, "shim_mcx.ovg HIP vs reference")
    # ---- affine
    refs, cases, exp_off, exp = golden_cases.mca_cases()
    s = ShimStream("shim_mca.ovg")
    n = len(cases)
    drefs = s.refs([ctx.upload_pic(r.y, r.cb, r.cr) for r in refs])
    tall = _tall(ctx, rw, rh, n)
    rects = []
    for i, (d, _, _) in enumerate(cases):
        c = s.case(i)
        ctx.mca(tall.band(i * rh, rh), drefs, ctx.upload(c["aff"]), ctx.upload(c["side"]))
        rects += _rects(d, exp_off, i, rh)
    ctx.sync()
    y, cb, cr = tall.download()
    golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, "shim_mca.ovg HIP vs reference")


def test_shim_dbf_edge_lists_gpu(ctx):
    """df.rcn_dbf_ctu(_truncated) through the installed table -> edge lists -> k_dbf_list with the per-slice offset table."""
    import ctypes as C
    gd = golden_io.load("shim_dbf.ovg")
    for i, (pic, _, exp) in enumerate(golden_cases.dbf_cases()):
        d = ctx.upload_pic(pic.y, pic.cb, pic.cr)
        ev = np.frombuffer(gd[f"p{i}_edges_v"].tobytes(), dtype=capi.DBF_EDGE_DTYPE)
        eh = np.frombuffer(gd[f"p{i}_edges_h"].tobytes(), dtype=capi.DBF_EDGE_DTYPE)
        offs = capi.DbfOffsets.from_buffer_copy(gd[f"p{i}_offsets"].tobytes())
        dv, dh = ctx.upload(ev), ctx.upload(eh)
        ctx._chk(ctx.lib.ovhip_dbf_launch_edges_ex(ctx.h, C.byref(d.s), dv.ptr, len(ev), dh.ptr, len(eh), C.byref(offs)), "dbf_launch_edges_ex")
        ctx.sync()
        y, cb, cr = d.download()
        for name, a, b in (("Y", y, exp.y), ("Cb", cb, exp.cb), ("Cr", cr, exp.cr)):
            assert np.array_equal(a, b), f"dbf picture {i} plane {name}: {int((a != b).sum())} samples differ"
