"""Turn the golden fixtures (reference inputs/outputs) into recorded command buffers.

Every case of a fixture is laid out on its own 128-row band of one tall picture so that a whole
fixture is ONE oracle call / ONE kernel launch."""
import ctypes as C

import numpy as np

import golden_io
from openvvc_amd import capi
from oracle_lib import HostPic

BAND = 128


def itx_cases():
    g = golden_io.load("itx.ovg")
    n = g["desc"].shape[0]
    pic = HostPic(128, BAND * n,
                  np.tile(g["pred_y"], (n, 1)), np.tile(g["pred_cb"], (n, 1)), np.tile(g["pred_cr"], (n, 1)))
    rec = capi.Recorder(128, BAND * n)
    coefs = np.ascontiguousarray(g["coefs"])
    rects = []       # (plane, x, y, w, h, expected_flat_offset)
    for i in range(n):
        st = capi.TuState.from_buffer_copy(g["state"][i].tobytes())
        d = capi.TuDesc.from_buffer_copy(g["desc"][i].tobytes())
        tree = d.tree
        x0, y0, w, h = d.x0, d.y0, 1 << d.log2_tb_w, 1 << d.log2_tb_h
        d.y0 = y0 + i * (BAND // 2 if tree == 2 else BAND)
        for comp in range(3):
            ln = int(g["coef_len"][i, comp])
            d.coef[comp] = coefs[int(g["coef_off"][i, comp]):].ctypes.data if ln else None
        rec.tu(st, d)
        eo = g["exp_off"][i]
        if tree == 0:
            rects.append((0, x0, y0 + i * BAND, w, h, int(eo[0])))
            rects.append((1, x0 >> 1, (y0 >> 1) + i * (BAND // 2), w >> 1, h >> 1, int(eo[1])))
            rects.append((2, x0 >> 1, (y0 >> 1) + i * (BAND // 2), w >> 1, h >> 1, int(eo[2])))
        else:
            rects.append((1, x0, y0 + i * (BAND // 2), w, h, int(eo[1])))
            rects.append((2, x0, y0 + i * (BAND // 2), w, h, int(eo[2])))
    return pic, rec.tb_cmds(), rec.coefs(), rects, g["exp"]


def tt_cases():
    """Whole transform trees (tmp.rcn_transform_tree): (picture, commands, coefficient arena, rects, expected)."""
    g = golden_io.load("itx.ovg")
    n = g["tt_desc"].shape[0]
    pic = HostPic(128, BAND * n,
                  np.tile(g["pred_y"], (n, 1)), np.tile(g["pred_cb"], (n, 1)), np.tile(g["pred_cr"], (n, 1)))
    rec = capi.Recorder(128, BAND * n)
    rects = []
    for i in range(n):
        st = capi.TuState.from_buffer_copy(g["tt_state"][i].tobytes())
        d = capi.TtDesc.from_buffer_copy(g["tt_desc"][i].tobytes())
        off, used = (int(v) for v in g["tt_coef_off"][i])
        co = g["tt_coefs"][off:off + 3 * used]
        d.y0 = i * BAND
        rec.transform_tree(st, d, g["tt_info"][i].tobytes(), co[:used], co[used:2 * used], co[2 * used:])
        w, h = 1 << d.log2_w, 1 << d.log2_h
        eo = g["tt_exp_off"][i]
        rects.append((0, 0, i * BAND, w, h, int(eo[0])))
        rects.append((1, 0, i * (BAND // 2), w >> 1, h >> 1, int(eo[1])))
        rects.append((2, 0, i * (BAND // 2), w >> 1, h >> 1, int(eo[2])))
    return pic, rec.tb_cmds(), rec.coefs(), rects, g["exp"]


def mc_cases():
    g = golden_io.load("mc.ovg")
    n = g["desc"].shape[0]
    _, rh, rw = g["ref_y"].shape
    refs = [HostPic(rw, rh, g["ref_y"][k], g["ref_cb"][k], g["ref_cr"][k]) for k in range(3)]
    descs = [capi.PuDesc.from_buffer_copy(g["desc"][i].tobytes()) for i in range(n)]
    return refs, descs, g["exp_off"], g["exp"]


def mcx_cases():
    """BDOF / DMVR: (refs, descs, exp_off [n,4] (Y, Cb, Cr sample offsets + index of the first refined MV), exp, exp_mv [m,4])."""
    g = golden_io.load("mcx.ovg")
    n = g["desc"].shape[0]
    _, rh, rw = g["ref_y"].shape
    refs = [HostPic(rw, rh, g["ref_y"][k], g["ref_cb"][k], g["ref_cr"][k]) for k in range(3)]
    descs = [capi.PuDesc.from_buffer_copy(g["desc"][i].tobytes()) for i in range(n)]
    return refs, descs, g["exp_off"], g["exp"], g["exp_mv"].reshape(-1, 4)


def mca_cases():
    """Affine CUs: (refs, [(AffineDesc, mv0 [rows, cols, 2], mv1)], exp_off [n,4], exp)."""
    g = golden_io.load("mca.ovg")
    n = g["desc"].shape[0]
    _, rh, rw = g["ref_y"].shape
    refs = [HostPic(rw, rh, g["ref_y"][k], g["ref_cb"][k], g["ref_cr"][k]) for k in range(3)]
    cases = []
    for i in range(n):
        d = capi.AffineDesc.from_buffer_copy(g["desc"][i].tobytes())
        nsx, nsy = (1 << d.log2_w) >> 2, (1 << d.log2_h) >> 2
        o = int(g["exp_off"][i, 3])
        mv = g["mvs"][o:o + 4 * nsx * nsy].reshape(2, nsy, nsx, 2)
        cases.append((d, mv[0].copy(), mv[1].copy()))
    return refs, cases, g["exp_off"], g["exp"]


def lmcs_cases():
    """(luma picture, [(LmcsData, LmcsLuts expected)], regions int32 [n,6] = (set, x, y, abv_mask, lft_mask, scale), inverse [k,2,128,128])."""
    g = golden_io.load("lmcs.ovg")
    sets = [(capi.LmcsData.from_buffer_copy(g["data"][i].tobytes()), capi.LmcsLuts.from_buffer_copy(g["luts"][i].tobytes()))
            for i in range(g["data"].shape[0])]
    return g["pic_y"], sets, g["regions"], g["inverse"]


def ciip_planar_cases():
    """The CIIP cases of gpm.ovg that ran the reference's real planar prediction: (current picture HostPic, first case index)."""
    g = golden_io.load("gpm.ovg")
    h, w = g["cur_y"].shape
    return HostPic(w, h, g["cur_y"], g["cur_cb"], g["cur_cr"]), g["desc"].shape[0] - int(g["n_ciip_planar"][0])


def gpm_cases():
    """GPM + CIIP: (refs, intra HostPic, descs, ciip_modes [n,2], n_gpm, exp_off, exp); the first n_gpm cases are GPM, the
    last n_ciip_planar ones CIIP with the real planar prediction (ciip_planar_cases), the ones between CIIP with the
    reference's intra slots replaced by a stub that delivers the `intra` picture."""
    g = golden_io.load("gpm.ovg")
    n = g["desc"].shape[0]
    _, rh, rw = g["ref_y"].shape
    refs = [HostPic(rw, rh, g["ref_y"][k], g["ref_cb"][k], g["ref_cr"][k]) for k in range(3)]
    intra = HostPic(rw, rh, g["intra_y"], g["intra_cb"], g["intra_cr"])
    descs = [capi.PuDesc.from_buffer_copy(g["desc"][i].tobytes()) for i in range(n)]
    return refs, intra, descs, g["ciip_modes"], int(g["n_gpm"][0]), g["exp_off"], g["exp"]


def check_rects(pic: HostPic, rects, exp, what=""):
    planes = pic.planes()
    bad = []
    for k, (p, x, y, w, h, off) in enumerate(rects):
        if w == 0 or h == 0:
            continue
        got = planes[p][y:y + h, x:x + w]
        want = exp[off:off + w * h].reshape(h, w)
        if not np.array_equal(got, want):
            bad.append((k, p, x, y, w, h, int(np.abs(got.astype(int) - want.astype(int)).max())))
    assert not bad, f"{what}: {len(bad)} / {len(rects)} rectangles differ, first: {bad[:5]}"


def dbf_cases():
    """[(unfiltered HostPic, planes dict from the recorder, expected HostPic)] from dbf.ovg."""
    g = golden_io.load("dbf.ovg")
    out = []
    pi = 0
    while f"p{pi}_in_y" in g:
        y = g[f"p{pi}_in_y"]
        h, w = y.shape
        rec = capi.Recorder(w, h)
        mv = g.get(f"p{pi}_mvctx")                    # B-slice picture: motion contexts for the MV-based bS pre-pass
        for k, raw in enumerate(g[f"p{pi}_ctus"]):
            rec.dbf_ctu(raw.tobytes(), mv[k].tobytes() if mv is not None else None)
        planes = rec.dbf_planes()
        planes["edges"] = [rec.dbf_edges(0), rec.dbf_edges(1)]        # what the per-CTU recorder emitted directly
        out.append((HostPic(w, h, y, g[f"p{pi}_in_cb"], g[f"p{pi}_in_cr"]), planes,
                    HostPic(w, h, g[f"p{pi}_exp_y"], g[f"p{pi}_exp_cb"], g[f"p{pi}_exp_cr"])))
        pi += 1
    return out


def sao_cases():
    """[(deblocked HostPic, SAO params (structured array), expected HostPic)] from sao.ovg."""
    g = golden_io.load("sao.ovg")
    out, pi = [], 0
    while f"p{pi}_in_y" in g:
        y = g[f"p{pi}_in_y"]
        h, w = y.shape
        prm = np.frombuffer(g[f"p{pi}_params"].tobytes(), dtype=capi.SAO_CTU_DTYPE).copy()
        out.append((HostPic(w, h, y, g[f"p{pi}_in_cb"], g[f"p{pi}_in_cr"]), prm,
                    HostPic(w, h, g[f"p{pi}_exp_y"], g[f"p{pi}_exp_cb"], g[f"p{pi}_exp_cr"])))
        pi += 1
    return out


def alf_cases():
    """[(post-SAO HostPic, ALF tables dict, expected HostPic)] from alf.ovg."""
    g = golden_io.load("alf.ovg")
    out, pi = [], 0
    while f"p{pi}_in_y" in g:
        y = g[f"p{pi}_in_y"]
        h, w = y.shape
        alf = {"ctus": np.frombuffer(g[f"p{pi}_ctus"].tobytes(), dtype=capi.ALF_CTU_DTYPE).copy()}
        for k in ("luma_coeff", "luma_clip", "chroma_coeff", "chroma_clip", "cc_coeff"):
            alf[k] = g[f"p{pi}_{k}"]
        out.append((HostPic(w, h, y, g[f"p{pi}_in_cb"], g[f"p{pi}_in_cr"]), alf,
                    HostPic(w, h, g[f"p{pi}_exp_y"], g[f"p{pi}_exp_cb"], g[f"p{pi}_exp_cr"])))
        pi += 1
    return out
