"""ctypes binding of the C ABI declared in include/ovvc_hip.h.

Host-side mirror used by the Python harness (tests, bench, smoke).  The structures below are
field-for-field the C structs; `tests/test_capi.py` checks sizes against the library.  The
library is the product: if libovvc_hip.so is missing this module raises -- there is no
Python or CPU fallback for the engine entry points.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / os.environ.get("OVVC_HIP_LIB_NAME", "libovvc_hip.so")      # (the variable: A / B runs of kernel variants, tools/ only)

# ---- constants (include/ovvc_hip.h) ----
OVHIP_ABI_VERSION = 8
OVHIP_OK, OVHIP_ENODEV, OVHIP_ENOMEM, OVHIP_EINVAL, OVHIP_ELAUNCH, OVHIP_EUNSUP, OVHIP_EREF = 0, -1, -2, -3, -4, -5, -6
DST_VII, DCT_VIII, DCT_II = 0, 1, 2
TB_TR, TB_DC, TB_TS, TB_TS_RAW = 0, 1, 2, 3
TB_FLAG_RASTER = 0x80
RES_ADD, RES_SUB, RES_ADD_HALF, RES_SUB_HALF, RES_SCALE = 0, 1, 2, 3, 4
MC_HPEL_FILT, MC_FILT_4x4, MC_NO_LUMA, MC_NO_CHROMA, MC_LMCS, MC_BDOF, MC_DMVR = 1, 2, 4, 8, 16, 32, 64
PU_BDOF, PU_DMVR, PU_GPM = 1, 2, 4
MC_GPM = 128


class Pic(C.Structure):
    _fields_ = [("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p),
                ("w", C.c_int32), ("h", C.c_int32), ("stride_y", C.c_int32), ("stride_c", C.c_int32)]


class TbCmd(C.Structure):
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("plane", C.c_uint8), ("log2_w", C.c_uint8),
                ("log2_h", C.c_uint8), ("kind", C.c_uint8), ("tr_h", C.c_uint8), ("tr_v", C.c_uint8),
                ("lfnst", C.c_uint8), ("res_mode", C.c_uint8), ("plane2", C.c_uint8),
                ("res_mode2", C.c_uint8), ("dq_shift", C.c_uint8), ("dq_neg", C.c_uint8),
                ("dq_scale", C.c_int16), ("c_scale", C.c_int16), ("coef_off", C.c_uint32),
                ("sig_sb_map", C.c_uint64)]


class McUnit(C.Structure):
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("w", C.c_uint8), ("h", C.c_uint8),
                ("dir", C.c_uint8), ("flags", C.c_uint8), ("ref0", C.c_uint8), ("ref1", C.c_uint8),
                ("w0", C.c_int8), ("w1", C.c_int8), ("mv0x", C.c_int32), ("mv0y", C.c_int32),
                ("mv1x", C.c_int32), ("mv1y", C.c_int32), ("aux", C.c_uint32)]


class TuState(C.Structure):
    _fields_ = [("qp_y", C.c_uint8), ("qp_cb", C.c_uint8), ("qp_cr", C.c_uint8), ("qp_jcbcr", C.c_uint8),
                ("qp_y_skip", C.c_uint8), ("qp_cb_skip", C.c_uint8), ("qp_cr_skip", C.c_uint8),
                ("qp_jcbcr_skip", C.c_uint8), ("dep_quant", C.c_uint8), ("mts_implicit", C.c_uint8),
                ("sh_ts_disabled", C.c_uint8), ("ict_type", C.c_uint8), ("lmcs_scale_c", C.c_uint8),
                ("pad", C.c_uint8 * 3), ("lmcs_chroma_scale", C.c_int16), ("intra_mode", C.c_int8),
                ("lfnst_mode_c", C.c_int8)]


class TuDesc(C.Structure):
    _fields_ = [("x0", C.c_uint16), ("y0", C.c_uint16), ("log2_tb_w", C.c_uint8), ("log2_tb_h", C.c_uint8),
                ("tree", C.c_uint8), ("cbf_mask", C.c_uint8), ("cu_flags", C.c_uint16),
                ("tr_skip_mask", C.c_uint8), ("cu_mts_flag", C.c_uint8), ("cu_mts_idx", C.c_uint8),
                ("lfnst_flag", C.c_uint8), ("lfnst_idx", C.c_uint8), ("pad", C.c_uint8),
                ("last_pos", C.c_uint16 * 3), ("sig_sb_map", C.c_uint64 * 3),
                ("coef", C.c_void_p * 3)]


class TuInfo(C.Structure):
    _fields_ = [("cbf_mask", C.c_uint8), ("tr_skip_mask", C.c_uint8), ("cu_mts_flag", C.c_uint8), ("cu_mts_idx", C.c_uint8),
                ("lfnst_flag", C.c_uint8), ("lfnst_idx", C.c_uint8), ("pos_offset", C.c_uint16),
                ("last_pos", C.c_uint16 * 3), ("pad", C.c_uint16), ("sig_sb_map", C.c_uint64 * 3)]


class TtDesc(C.Structure):
    _fields_ = [("x0", C.c_uint16), ("y0", C.c_uint16), ("log2_w", C.c_uint8), ("log2_h", C.c_uint8),
                ("log2_max_tb_s", C.c_uint8), ("tree", C.c_uint8), ("cu_flags", C.c_uint16), ("pad", C.c_uint16),
                ("tu_info", C.c_void_p), ("residual", C.c_void_p * 3)]


class PuDesc(C.Structure):
    _fields_ = [("x0", C.c_uint16), ("y0", C.c_uint16), ("log2_w", C.c_uint8), ("log2_h", C.c_uint8),
                ("inter_dir", C.c_uint8), ("ref_idx0", C.c_uint8), ("ref_idx1", C.c_uint8),
                ("bcw_idx_plus1", C.c_uint8), ("prec_amvr_half", C.c_uint8), ("planes", C.c_uint8),
                ("lmcs", C.c_uint8), ("refine", C.c_uint8), ("mv0x", C.c_int32), ("mv0y", C.c_int32),
                ("mv1x", C.c_int32), ("mv1y", C.c_int32), ("poc0", C.c_int32), ("poc1", C.c_int32),
                ("ref0", C.c_uint8), ("ref1", C.c_uint8), ("gpm_split_dir", C.c_uint8), ("ciip_wt", C.c_uint8)]


class AffineDesc(C.Structure):
    _fields_ = [("x0", C.c_uint16), ("y0", C.c_uint16), ("log2_w", C.c_uint8), ("log2_h", C.c_uint8),
                ("inter_dir", C.c_uint8), ("bcw_idx_plus1", C.c_uint8), ("prof_dir", C.c_uint8), ("lmcs", C.c_uint8),
                ("ref0", C.c_uint8), ("ref1", C.c_uint8), ("poc0", C.c_int32), ("poc1", C.c_int32),
                ("mv_stride", C.c_int32), ("mv0", C.c_void_p), ("mv1", C.c_void_p), ("dmv_scale", (C.c_int16 * 16) * 4)]


class IspDesc(C.Structure):
    """ovhip_isp_desc: one intra-sub-partition CU as tmp.recon_isp_subtree_v / _h receive it"""
    _fields_ = [("x0", C.c_uint16), ("y0", C.c_uint16), ("log2_cb_w", C.c_uint8), ("log2_cb_h", C.c_uint8), ("vertical", C.c_uint8),
                ("intra_mode", C.c_uint8), ("cbf_mask", C.c_uint8), ("lfnst_flag", C.c_uint8), ("lfnst_idx", C.c_uint8), ("mts_enabled", C.c_uint8),
                ("corner", C.c_uint8 * 4), ("avl_abv", C.c_uint8 * 4), ("avl_lft", C.c_uint8 * 4), ("last_pos", C.c_uint16 * 4),
                ("sig_sb_map", C.c_uint64 * 4), ("coef", C.c_void_p)]


class LmcsData(C.Structure):
    _fields_ = [("min_bin_idx", C.c_uint8), ("delta_max_bin_idx", C.c_uint8), ("crs_offset", C.c_int16),
                ("cw_delta", C.c_int16 * 16)]


class LmcsLuts(C.Structure):
    _fields_ = [("fwd_lut", C.c_uint16 * 1024), ("bwd_lut", C.c_uint16 * 1024), ("wnd_bnd", C.c_uint16 * 17),
                ("min_idx", C.c_uint8), ("max_idx", C.c_uint8), ("crs_offset", C.c_int16), ("pad", C.c_uint16)]


DBF_EDGE_DTYPE = np.dtype([("ux", "<u2"), ("uy", "<u2"), ("word", "<u2"), ("comp", "u1"), ("pad", "u1")])
assert DBF_EDGE_DTYPE.itemsize == 8
CIIP_UNIT_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2_w", "u1"), ("log2_h", "u1"), ("wt", "u1"), ("chroma_inter", "u1")])
assert CIIP_UNIT_DTYPE.itemsize == 8
LMCS_REGION_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("n_abv", "u1"), ("n_lft", "u1"), ("ordered", "u1"), ("pad", "u1")])
assert LMCS_REGION_DTYPE.itemsize == 8


ITASK_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2_w", "u1"), ("log2_h", "u1"), ("kind", "u1"), ("mode", "u1"),
                        ("flags", "<u2"), ("avl_lft", "u1"), ("avl_abv", "u1"), ("mrl_idx", "u1"), ("ciip_wt", "u1"),
                        ("c_scale", "<i2"), ("level", "<u2"), ("ctu_deps", "<u2"),
                        ("isp_log2_cb_w", "u1"), ("isp_log2_cb_h", "u1"), ("isp_off_x", "u1"), ("isp_off_y", "u1"), ("isp_log2_pb", "u1"), ("isp_res_mask", "u1"), ("pad", "<u2", 3)])
assert ITASK_DTYPE.itemsize == 32
ICTU_DTYPE = np.dtype([("cx", "<u2"), ("cy", "<u2"), ("first", "<u4"), ("n", "<u4"), ("deps", "<u4")])
assert ICTU_DTYPE.itemsize == 16
IT_LUMA, IT_CHROMA, IT_REGION, IT_RES_C = 0, 1, 2, 3
IF_CORNER, IF_MIP, IF_MIP_TR, IF_BDPCM, IF_BDPCM_VER, IF_RES_Y, IF_RES_CB, IF_RES_CR, IF_RES_SCALE, IF_SCALE_IDX, IF_ISP, IF_CORNER_L = (1 << k for k in range(12))


class ITask(C.Structure):
    """ovhip_itask (include/ovvc_hip.h)."""
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("log2_w", C.c_uint8), ("log2_h", C.c_uint8), ("kind", C.c_uint8), ("mode", C.c_uint8),
                ("flags", C.c_uint16), ("avl_lft", C.c_uint8), ("avl_abv", C.c_uint8), ("mrl_idx", C.c_uint8), ("ciip_wt", C.c_uint8),
                ("c_scale", C.c_int16), ("level", C.c_uint16), ("ctu_deps", C.c_uint16),
                ("isp_log2_cb_w", C.c_uint8), ("isp_log2_cb_h", C.c_uint8), ("isp_off_x", C.c_uint8), ("isp_off_y", C.c_uint8), ("isp_log2_pb", C.c_uint8), ("isp_res_mask", C.c_uint8), ("pad", C.c_uint16 * 3)]


assert C.sizeof(ITask) == 32


class DbfMvCtx(C.Structure):
    _fields_ = [("cu_edge_ver", C.c_uint64 * 33), ("cu_edge_hor", C.c_uint64 * 33),
                ("map0_h", C.c_uint64 * 33), ("map0_v", C.c_uint64 * 33), ("map1_h", C.c_uint64 * 33), ("map1_v", C.c_uint64 * 33),
                ("ibc_h", C.c_uint64 * 33), ("ibc_v", C.c_uint64 * 33),
                ("dist_ref0", C.c_int16 * 16), ("dist_ref1", C.c_int16 * 16),
                ("mvs0", C.c_void_p), ("mvs1", C.c_void_p), ("mv_bytes", C.c_int32)]


class DbfPlanes(C.Structure):
    _fields_ = [("luma_v", C.c_void_p), ("luma_h", C.c_void_p), ("cb_v", C.c_void_p), ("cr_v", C.c_void_p),
                ("cb_h", C.c_void_p), ("cr_h", C.c_void_p), ("w4", C.c_int32), ("h4", C.c_int32),
                ("beta_offset", C.c_int16), ("tc_offset", C.c_int16)]


SAO_CTU_DTYPE = np.dtype([("type", "u1", 3), ("band_position", "u1", 3), ("eo_class", "u1", 3), ("border", "u1"), ("pad", "u1", 2),
                          ("offset_val", "<i2", (3, 5)), ("pad2", "u1", 2)])
assert SAO_CTU_DTYPE.itemsize == 44

ALF_CTU_DTYPE = np.dtype([("flags", "u1"), ("luma_set", "u1"), ("cb_alt", "u1"), ("cr_alt", "u1"),
                          ("cc_cb_idx", "u1"), ("cc_cr_idx", "u1"), ("border", "u1"), ("pad", "u1")])
BORDER_LEFT, BORDER_RIGHT, BORDER_UPPER, BORDER_BOTTOM, BORDER_ONE_ROW = 1, 2, 4, 8, 16   # OVHIP_BORDER_*
ALF_LUMA_SET_SIZE = 4 * 25 * 13


class AlfPic(C.Structure):
    _fields_ = [("ctus", C.c_void_p), ("luma_coeff", C.c_void_p), ("luma_clip", C.c_void_p),
                ("chroma_coeff", C.c_void_p), ("chroma_clip", C.c_void_p), ("cc_coeff", C.c_void_p),
                ("class_scratch", C.c_void_p), ("log2_ctu_s", C.c_int32)]


ALF_TABLES = (("ctus", ALF_CTU_DTYPE), ("luma_coeff", np.int16), ("luma_clip", np.int16), ("chroma_coeff", np.int16),
              ("chroma_clip", np.int16), ("cc_coeff", np.int16))

DBF_CTU_SIZE = 8 * (49 * 6 + 33 * 12) + 3 * 34 * 33 + 2 * 2 + 2 + 3 + 2 + 1 + 4 * 2
DBF_CTU_SIZE = (DBF_CTU_SIZE + 7) & ~7          # struct alignment (uint64 members)
_M49, _M33 = ("<u8", (49,)), ("<u8", (33,))
DBF_CTU_DTYPE = np.dtype(                      # ovhip_dbf_ctu, field by field
    [(n, *_M49) for n in ("ctb_bound_ver", "ctb_bound_hor", "ctb_bound_ver_c", "ctb_bound_hor_c", "aff_edg_ver", "aff_edg_hor")]
    + [(n, *_M33) for n in ("bs2_ver", "bs2_hor", "bs2c_ver", "bs2c_hor", "bs1_ver", "bs1_hor", "bs1cb_ver", "bs1cb_hor",
                            "bs1cr_ver", "bs1cr_hor", "affine_ver", "affine_hor")]
    + [(n, "u1", (34 * 33,)) for n in ("qp_y", "qp_cb", "qp_cr")]
    + [("beta_offset", "<i2"), ("tc_offset", "<i2"), ("disable_v", "u1"), ("disable_h", "u1"), ("log2_ctu_s", "u1"),
       ("last_x", "u1"), ("last_y", "u1"), ("ctu_lft", "u1"), ("ctu_abv", "u1"), ("pad", "u1"),
       ("ctu_w", "<u2"), ("ctu_h", "<u2"), ("ctb_x", "<u2"), ("ctb_y", "<u2")], align=True)
assert DBF_CTU_DTYPE.itemsize == DBF_CTU_SIZE, (DBF_CTU_DTYPE.itemsize, DBF_CTU_SIZE)


class DbfView(C.Structure):
    """ovhip_dbf_view: ovhip_dbf_ctu with every array by pointer (into the caller's struct DBFInfo)"""
    _fields_ = [(n, C.c_void_p) for n in ("ctb_bound_ver", "ctb_bound_hor", "ctb_bound_ver_c", "ctb_bound_hor_c", "aff_edg_ver", "aff_edg_hor",
                                          "bs2_ver", "bs2_hor", "bs2c_ver", "bs2c_hor", "bs1_ver", "bs1_hor", "bs1cb_ver", "bs1cb_hor",
                                          "bs1cr_ver", "bs1cr_hor", "affine_ver", "affine_hor", "qp_y", "qp_cb", "qp_cr")] + \
               [("beta_offset", C.c_int16), ("tc_offset", C.c_int16), ("disable_v", C.c_uint8), ("disable_h", C.c_uint8), ("log2_ctu_s", C.c_uint8),
                ("last_x", C.c_uint8), ("last_y", C.c_uint8), ("ctu_lft", C.c_uint8), ("ctu_abv", C.c_uint8), ("pad", C.c_uint8),
                ("ctu_w", C.c_uint16), ("ctu_h", C.c_uint16), ("ctb_x", C.c_uint16), ("ctb_y", C.c_uint16)]


FE_BEGIN, FE_REF, FE_DMVR_ROWS, FE_DMVR_BEGIN, FE_DMVR_COLLECT, FE_SUBMIT, FE_FAIL = range(1, 8)
FRAME_EVENT_DTYPE = np.dtype([("op", "<u4"), ("frame", "<i4"), ("key", "<u8"), ("tag", "<u8"), ("a", "<i8"), ("b", "<i8"), ("result", "<i8")])
assert FRAME_EVENT_DTYPE.itemsize == 48
FRAME_TRACE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)

DBF_PLANE_NAMES = ("luma_v", "luma_h", "cb_v", "cr_v", "cb_h", "cr_h")


class DbfOffsets(C.Structure):
    _fields_ = [("beta", C.c_int8 * 8), ("tc", C.c_int8 * 8)]


class BandCounts(C.Structure):
    """ovhip_band_counts: lengths of the recorder's arrays where a band of CTU rows ends (ovhip_job_band)"""
    _fields_ = [(n, C.c_uint32) for n in ("n_tb", "n_coef", "n_mc", "n_mcx", "n_aff", "n_side", "n_reg", "n_itask", "n_edge_v", "n_edge_h")]


class JobParams(C.Structure):
    """ovhip_job_params: picture-level side information (host pointers)."""
    _fields_ = [("lmcs", C.c_void_p), ("sao", C.c_void_p), ("alf_ctus", C.c_void_p),
                ("alf_luma_coeff", C.c_void_p), ("alf_luma_clip", C.c_void_p),
                ("alf_chroma_coeff", C.c_void_p), ("alf_chroma_clip", C.c_void_p), ("alf_cc_coeff", C.c_void_p),
                ("log2_ctu_s", C.c_int32), ("stages", C.c_uint32), ("wait_events", C.POINTER(C.c_void_p)), ("n_wait_events", C.c_uint32),
                ("before_launch", C.c_void_p), ("before_launch_user", C.c_void_p), ("tmvp_cells", C.c_uint32), ("wait_on_host", C.c_uint32),
                ("flow_workers", C.c_uint32)]


BEFORE_LAUNCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class JobStats(C.Structure):
    _fields_ = [("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64), ("n_launches", C.c_uint32), ("n_h2d", C.c_uint32),
                ("n_tb", C.c_uint32), ("n_mc", C.c_uint32), ("n_mcx", C.c_uint32), ("n_aff", C.c_uint32),
                ("n_edges_v", C.c_uint32), ("n_edges_h", C.c_uint32), ("n_regions", C.c_uint32),
                ("n_itasks", C.c_uint32), ("n_ilevels", C.c_uint32), ("n_ordered_retries", C.c_uint32), ("flow_shift", C.c_uint32),
                ("host_us_prepare", C.c_uint32), ("host_us_upload", C.c_uint32), ("host_us_wait", C.c_uint32), ("host_us_launch", C.c_uint32)]


TMVP_CELL_DTYPE = np.dtype([("cell", "<u4"), ("mv0x", "<i4"), ("mv0y", "<i4"), ("mv1x", "<i4"), ("mv1y", "<i4")])
TMVP_NONE = 0xffffffff
REC_TB, REC_COEF, REC_MC, REC_MCX, REC_AFF, REC_SIDE, REC_REGION, REC_CIIP, REC_EDGE_V, REC_EDGE_H, REC_ITASK = range(11)
TIME_STAGES = ("mc", "mcxa", "itx_luma", "lmcs_scale", "itx_chroma", "dbf", "sao", "alf", "intra", "h2d")
STAGE_MC, STAGE_ITX, STAGE_DBF, STAGE_SAO, STAGE_ALF, STAGE_INTRA = 1, 2, 4, 8, 16, 32
STAGE_ALL, STAGE_RESIDENT, STAGE_INTRA_CTU, STAGE_INTRA_LEVELS = 63, 0x40000000, 0x20000000, 0x10000000


def dbf_plane_shapes(w4: int, h4: int) -> dict:
    w4c, h4c = (w4 + 1) // 2, (h4 + 1) // 2
    return {"luma_v": (h4, w4), "luma_h": (h4, w4), "cb_v": (h4, w4c), "cr_v": (h4, w4c),
            "cb_h": (h4c, w4), "cr_h": (h4c, w4)}


TB_CMD_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("plane", "u1"), ("log2_w", "u1"), ("log2_h", "u1"),
                         ("kind", "u1"), ("tr_h", "u1"), ("tr_v", "u1"), ("lfnst", "u1"), ("res_mode", "u1"),
                         ("plane2", "u1"), ("res_mode2", "u1"), ("dq_shift", "u1"), ("dq_neg", "u1"),
                         ("dq_scale", "<i2"), ("c_scale", "<i2"), ("coef_off", "<u4"), ("sig_sb_map", "<u8")])
MC_UNIT_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("dir", "u1"), ("flags", "u1"),
                          ("ref0", "u1"), ("ref1", "u1"), ("w0", "i1"), ("w1", "i1"), ("mv0x", "<i4"),
                          ("mv0y", "<i4"), ("mv1x", "<i4"), ("mv1y", "<i4"), ("aux", "<u4")])
assert TB_CMD_DTYPE.itemsize == C.sizeof(TbCmd) == 32
assert MC_UNIT_DTYPE.itemsize == C.sizeof(McUnit) == 32
AFF_UNIT_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("dir", "u1"), ("flags", "u1"),
                           ("ref0", "u1"), ("ref1", "u1"), ("w0", "i1"), ("w1", "i1"), ("prof_dir", "u1"),
                           ("ident_c", "u1"), ("ident_l", "<u2"), ("side_off", "<u4"), ("prof_off", "<u4"),
                           ("pad", "<u4", 2)])
assert AFF_UNIT_DTYPE.itemsize == 32
AFF_PROF, AFF_NO_CHROMA, AFF_LMCS = 1, 8, 16

_lib = None


def dbf_compact(planes: dict, direction: int) -> np.ndarray:
    """Host: compact edge list (DBF_EDGE_DTYPE) of one direction from the dense planes dict (numpy uint16 arrays)."""
    keep = {k: np.ascontiguousarray(planes[k], dtype=np.uint16) for k in DBF_PLANE_NAMES}
    s = DbfPlanes(*[keep[k].ctypes.data for k in DBF_PLANE_NAMES], planes["w4"], planes["h4"],
                  planes["beta_offset"], planes["tc_offset"])
    n = load().ovhip_dbf_compact(C.byref(s), direction, None, 0)
    if n < 0:
        raise ValueError(f"ovhip_dbf_compact -> {n}")
    out = np.zeros(n, DBF_EDGE_DTYPE)
    if n:
        load().ovhip_dbf_compact(C.byref(s), direction, out.ctypes.data, n)
    return out


def lmcs_build(data: "LmcsData") -> "LmcsLuts":
    """Host: the LMCS tables of one APS (rcn_init_lmcs)."""
    out = LmcsLuts()
    r = load().ovhip_lmcs_build(C.byref(data), C.byref(out))
    if r < 0:
        raise ValueError(f"ovhip_lmcs_build -> {r}")
    return out


class Window(C.Structure):
    """ovhip_window = OVFrame.output_window: offsets in chroma sample units"""
    _fields_ = [("offset_lft", C.c_uint16), ("offset_rgt", C.c_uint16), ("offset_abv", C.c_uint16), ("offset_blw", C.c_uint16)]


# ---- frame threads, device DPB, stream driver (include/ovvc_hip.h) ----
DPB_PIC_ALLOC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.POINTER(Pic))
DPB_PIC_FREE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(Pic))
DPB_COPY_START_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(Pic), C.c_int, C.POINTER(Pic), C.POINTER(C.c_void_p))
DPB_COPY_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)
DPB_COPY_DONE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)
DPB_PIC_CLEAR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(Pic))
DPB_EVENT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)


class DpbOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("pic_alloc", DPB_PIC_ALLOC_FN), ("pic_free", DPB_PIC_FREE_FN),
                ("copy_start", DPB_COPY_START_FN), ("copy_wait", DPB_COPY_WAIT_FN), ("copy_done", DPB_COPY_DONE_FN),
                ("pic_clear", DPB_PIC_CLEAR_FN), ("event_query", DPB_EVENT_FN), ("event_wait", DPB_EVENT_FN)]


class DpbStats(C.Structure):
    _fields_ = [("n_live", C.c_uint32), ("n_pool", C.c_uint32), ("n_begin", C.c_uint64), ("n_alloc", C.c_uint64),
                ("n_recycled", C.c_uint64), ("n_copies", C.c_uint64), ("copy_bytes", C.c_uint64), ("n_failed", C.c_uint64),
                ("n_waits", C.c_uint64)]


OUT_NONE, OUT_DIGEST, OUT_PLANES, OUT_PACKED = 0, 1, 2, 3


class FrameOutput(C.Structure):
    _fields_ = [("mode", C.c_int32), ("window", Window), ("y", C.c_void_p), ("cb", C.c_void_p), ("cr", C.c_void_p),
                ("stride_y", C.c_int32), ("stride_c", C.c_int32), ("packed", C.c_void_p), ("digest", C.c_uint8 * 16)]


STREAM_MAX_REFS = 8
STREAM_RECORD, STREAM_DIGESTS, STREAM_RESIDENT, STREAM_KEEP, STREAM_FILE_MD5, STREAM_HOLD_ALL = 1, 2, 4, 8, 16, 32


class StreamContent(C.Structure):
    _fields_ = [("calllog", C.c_void_p), ("calllog_bytes", C.c_size_t), ("params", JobParams), ("n_ref_slots", C.c_uint32)]


class StreamPic(C.Structure):
    _fields_ = [("content", C.c_uint32), ("job", C.c_uint32), ("poc", C.c_int32), ("device", C.c_uint16), ("n_refs", C.c_uint16),
                ("refs", C.c_uint32 * STREAM_MAX_REFS), ("owner", C.c_int32), ("send_mask", C.c_uint32)]


XFER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.POINTER(Pic), C.c_int)


class StreamXfer(C.Structure):
    _fields_ = [("user", C.c_void_p), ("send", XFER_FN), ("recv", XFER_FN)]


class StreamCfg(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("flags", C.c_uint32), ("threads_per_device", C.c_int32), ("output", C.c_int32),
                ("window", Window), ("extra_stages", C.c_uint32), ("rank", C.c_int32), ("xfer", C.POINTER(StreamXfer)),
                ("intra_lookahead", C.c_int32), ("ahead_own_queue", C.c_int32)]


class StreamResult(C.Structure):
    _fields_ = [("seconds", C.c_double), ("n_decoded", C.c_uint64), ("n_second_passes", C.c_uint64), ("n_received", C.c_uint64),
                ("n_sent", C.c_uint64), ("out_frames", C.c_uint64), ("out_bytes", C.c_uint64), ("out_md5", C.c_uint8 * 16),
                ("record_seconds", C.c_double), ("host_seconds", C.c_double * 5), ("status", C.c_int32), ("error", C.c_char * 192),
                ("trace", C.c_void_p)]


class Md5State(C.Structure):
    _fields_ = [("h", C.c_uint32 * 4), ("n_bytes", C.c_uint64), ("buf", C.c_uint8 * 64)]


def md5(data: bytes) -> bytes:
    """ovhip_md5_* over a byte string (what the library uses for the digest of the row digests)"""
    lib = load()
    st = Md5State()
    lib.ovhip_md5_init(C.byref(st))
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data) if data else None
    lib.ovhip_md5_update(C.byref(st), buf, len(data))
    out = (C.c_uint8 * 16)()
    lib.ovhip_md5_final(C.byref(st), out)
    return bytes(out)


def load(path: os.PathLike | None = None) -> C.CDLL:
    """Load libovvc_hip.so (built in-tree by __graft_entry__.build()).  Fails loudly."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = Path(path) if path else LIB_PATH
    if not p.exists():
        raise RuntimeError(f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP engine has no CPU fallback)")
    try:  # share torch's HIP runtime when torch is in the process (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(str(p))
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    P = C.POINTER
    sigs = {
        "ovhip_abi_version": (C.c_int, []),
        "ovhip_rec_create": (vp, [i32, i32]),
        "ovhip_rec_destroy": (None, [vp]),
        "ovhip_rec_reset": (None, [vp]),
        "ovhip_rec_tu": (C.c_int, [vp, P(TuState), P(TuDesc)]),
        "ovhip_rec_pu": (C.c_int, [vp, P(PuDesc)]),
        "ovhip_rec_dbf_ctu": (C.c_int, [vp, vp]),
        "ovhip_rec_dbf_row": (C.c_int, [vp, vp, C.c_size_t]),
        "ovhip_rec_dbf_mv_prepass_view": (C.c_int, [vp, vp, vp, vp]),
        "ovhip_rec_cu_inter": (C.c_int, [vp, vp, vp]),
        "ovhip_rec_dbf_planes": (C.c_int, [vp, P(DbfPlanes)]),
        "ovhip_dbf_launch": (C.c_int, [vp, P(Pic), P(DbfPlanes)]),
        "ovhip_sao_launch": (C.c_int, [vp, P(Pic), P(Pic), vp, i32]),
        "ovhip_alf_launch": (C.c_int, [vp, P(Pic), P(Pic), P(AlfPic)]),
        "ovhip_rec_tb_cmds": (vp, [vp, P(C.c_size_t)]),
        "ovhip_rec_coefs": (vp, [vp, P(C.c_size_t)]),
        "ovhip_rec_mc_units": (vp, [vp, P(C.c_size_t)]),
        "ovhip_rec_mcx_units": (vp, [vp, P(C.c_size_t)]),
        "ovhip_rec_affine_cu": (C.c_int, [vp, P(AffineDesc)]),
        "ovhip_rec_transform_tree": (C.c_int, [vp, P(TuState), P(TtDesc)]),
        "ovhip_rec_lmcs_region": (C.c_int, [vp, C.c_int32, C.c_int32, u32, u32]),
        "ovhip_dbf_compact": (C.c_int64, [P(DbfPlanes), C.c_int, vp, C.c_size_t]),
        "ovhip_rec_dbf_mv_prepass": (C.c_int, [vp, P(DbfMvCtx)]),
        "ovhip_dbf_launch_edges": (C.c_int, [vp, P(Pic), vp, u32, vp, u32, C.c_int32, C.c_int32]),
        "ovhip_rec_ciip": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
        "ovhip_rec_ciip_units": (vp, [vp, P(C.c_size_t)]),
        "ovhip_ciip_launch": (C.c_int, [vp, P(Pic), P(Pic), vp, u32]),
        "ovhip_rec_lmcs_regions": (vp, [vp, P(C.c_size_t)]),
        "ovhip_rec_tb_cmds_split": (vp, [vp, C.c_size_t * 4, P(C.c_size_t)]),
        "ovhip_itx_launch_classes": (C.c_int, [vp, P(Pic), vp, u32, u32, vp, vp]),
        "ovhip_itx_launch_chroma_lmcs": (C.c_int, [vp, P(Pic), vp, u32, u32, vp, vp, vp]),
        "ovhip_lmcs_build": (C.c_int, [P(LmcsData), P(LmcsLuts)]),
        "ovhip_lmcs_scale_launch": (C.c_int, [vp, P(Pic), vp, u32, P(LmcsLuts), vp]),
        "ovhip_lmcs_inverse_launch": (C.c_int, [vp, P(Pic), vp]),
        "ovhip_rec_aff_units": (vp, [vp, P(C.c_size_t)]),
        "ovhip_rec_aff_side": (vp, [vp, P(C.c_size_t)]),
        "ovhip_ctx_create": (C.c_int, [P(vp), C.c_int, vp]),
        "ovhip_ctx_destroy": (None, [vp]),
        "ovhip_ctx_sync": (C.c_int, [vp]),
        "ovhip_ctx_shares_queue": (C.c_int, [vp, vp]),
        "ovhip_ctx_new_stream": (C.c_int, [vp]),
        "ovhip_ctx_fork": (C.c_int, [vp, C.c_int]),
        "ovhip_ctx_join": (C.c_int, [vp]),
        "ovhip_last_error": (C.c_char_p, [vp]),
        "ovhip_ctx_stream": (vp, [vp]),
        "ovhip_malloc": (C.c_int, [vp, C.c_size_t, P(vp)]),
        "ovhip_free": (C.c_int, [vp, vp]),
        "ovhip_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "ovhip_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "ovhip_pic_alloc": (C.c_int, [vp, i32, i32, P(Pic)]),
        "ovhip_pic_free": (C.c_int, [vp, P(Pic)]),
        "ovhip_pic_upload": (C.c_int, [vp, P(Pic), vp, vp, vp, i32, i32]),
        "ovhip_pic_download": (C.c_int, [vp, P(Pic), vp, vp, vp, i32, i32]),
        "ovhip_itx_launch": (C.c_int, [vp, P(Pic), vp, u32, vp, vp]),
        "ovhip_mc_launch": (C.c_int, [vp, P(Pic), P(Pic), u32, vp, u32, vp, P(Pic)]),
        "ovhip_ciip_weight": (C.c_int, [C.c_int32, C.c_int32]),
        "ovhip_mcx_launch": (C.c_int, [vp, P(Pic), P(Pic), u32, vp, u32, vp, vp]),
        "ovhip_mca_launch": (C.c_int, [vp, P(Pic), P(Pic), u32, vp, u32, vp, vp]),
        "ovhip_mcxa_launch": (C.c_int, [vp, P(Pic), P(Pic), u32, vp, u32, vp, vp, u32, vp, vp]),
        "ovhip_rec_create_ex": (vp, [i32, i32, vp]),
        "ovhip_rec_set_dense_dbf_planes": (None, [vp, C.c_int]),
        "ovhip_rec_dbf_edges": (vp, [vp, C.c_int, P(C.c_size_t), P(DbfOffsets)]),
        "ovhip_dbf_launch_edges_ex": (C.c_int, [vp, P(Pic), vp, u32, vp, u32, P(DbfOffsets)]),
        "ovhip_dmvr_search_launch": (C.c_int, [vp, P(Pic), P(Pic), u32, vp, u32, vp]),
        "ovhip_rec_append_raw": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
        "ovhip_itx_launch_classes_res": (C.c_int, [vp, P(Pic), P(Pic), vp, u32, u32, vp, vp]),
        "ovhip_intra_level_launch": (C.c_int, [vp, P(Pic), P(Pic), vp, u32, vp, vp, vp, i32, u32]),
        "ovhip_intra_level_geom": (u32, [vp, C.c_size_t]),
        "ovhip_intra_sync_words": (C.c_size_t, [i32, i32, i32]),
        "ovhip_intra_flow_words": (C.c_size_t, [i32, i32]),
        "ovhip_intra_flow_items": (C.c_size_t, [vp, C.c_size_t, vp, C.c_size_t]),
        "ovhip_intra_flow_launch": (C.c_int, [vp, P(Pic), P(Pic), vp, u32, vp, u32, vp, vp, vp, i32, vp, u32, vp, i32, i32]),
        "ovhip_intra_ctu_launch": (C.c_int, [vp, P(Pic), P(Pic), vp, vp, u32, vp, vp, vp, i32, vp, u32, vp]),
        "ovhip_rec_itask_levels": (u32, [vp]),
        "ovhip_rec_isp_cu": (C.c_int, [vp, vp, vp]),
        "ovhip_isp_geometry": (None, [i32, i32, i32, P(i32), P(i32), P(i32), P(i32)]),
        "ovhip_job_bind": (C.c_int, [vp, vp]),
        "ovhip_rec_set_ctu_size": (C.c_int, [vp, i32]),
        "ovhip_rec_itasks_by_ctu": (vp, [vp, i32, P(C.c_size_t), P(vp), P(C.c_size_t)]),
        "ovhip_rec_tu_intra": (C.c_int, [vp, P(TuState), P(TuDesc), P(ITask), P(ITask)]),
        "ovhip_rec_itasks": (vp, [vp, P(C.c_size_t)]),
        "ovhip_rec_itasks_sorted": (vp, [vp, P(C.c_size_t), P(P(C.c_uint32)), P(C.c_uint32)]),
        "ovhip_rec_set_dbf_offsets": (C.c_int, [vp, P(DbfOffsets), C.c_int]),
        "ovhip_job_create": (C.c_int, [vp, i32, i32, P(vp)]),
        "ovhip_job_destroy": (None, [vp]),
        "ovhip_job_recorder": (vp, [vp]),
        "ovhip_job_begin": (C.c_int, [vp]),
        "ovhip_job_flush": (C.c_int, [vp, P(Pic), P(Pic), u32, P(Pic), P(JobParams)]),
        "ovhip_frame_band": (C.c_int, [vp, P(JobParams), i32, i32, P(FrameOutput)]),
        "ovhip_frame_set_band_mode": (C.c_int, [vp, C.c_int]),
        "ovhip_frame_band_upto": (C.c_int, [vp, P(JobParams), i32, P(BandCounts), i32]),
        "ovhip_frame_dmvr_rows_begin_upto": (C.c_int64, [vp, i32, C.c_size_t]),
        "ovhip_job_dmvr_rows_begin_upto": (C.c_int64, [vp, P(Pic), u32, i32, C.c_size_t]),
        "ovhip_frame_band_stats": (C.c_int, [vp, P(i32), P(i32)]),
        "ovhip_dpb_post_rows": (C.c_int, [vp, vp, i32, vp, vp]),
        "ovhip_dpb_rows_tag": (C.c_int, [vp, vp, C.c_uint64, C.c_int, i32, C.c_int, C.c_int, P(Pic), P(vp)]),
        "ovhip_job_band": (C.c_int, [vp, P(Pic), P(Pic), u32, P(JobParams), P(BandCounts), i32, i32]),
        "ovhip_job_band_active": (C.c_int, [vp]),
        "ovhip_job_band_busy": (C.c_int, [vp]),
        "ovhip_job_band_progress": (C.c_int, [vp, P(i32), P(vp), P(vp)]),
        "ovhip_rec_counts": (None, [vp, P(BandCounts)]),
        "ovhip_sao_launch_rows": (C.c_int, [vp, P(Pic), P(Pic), vp, i32, i32, i32]),
        "ovhip_alf_launch_rows": (C.c_int, [vp, P(Pic), P(Pic), P(AlfPic), i32, i32]),
        "ovhip_job_wait": (C.c_int, [vp]),
        "ovhip_job_refined_mvs": (vp, [vp, P(C.c_size_t)]),
        "ovhip_job_dmvr_rows": (C.c_int64, [vp, P(Pic), u32]),
        "ovhip_job_dmvr_rows_begin": (C.c_int64, [vp, P(Pic), u32, i32]),
        "ovhip_job_dmvr_rows_collect": (C.c_int64, [vp]),
        "ovhip_job_last_stats": (C.c_int, [vp, P(JobStats)]),
        "ovhip_job_time_stage": (C.c_int, [vp, C.c_int]),
        "ovhip_job_stage_time": (C.c_int, [vp, P(C.c_double), P(C.c_uint64)]),
        "ovhip_lmcs_scale_prepare_launch": (C.c_int, [vp, P(Pic), vp, u32, vp, vp, vp, u32, vp, u32]),
        "ovhip_lmcs_inverse_untag_launch": (C.c_int, [vp, P(Pic), vp, vp, u32]),
        "ovhip_intra_flow_untag_launch": (C.c_int, [vp, P(Pic), vp, u32, i32]),
        "ovhip_tmvp_cells_launch": (C.c_int, [vp, vp, u32, vp, i32, i32, vp]),
        "ovhip_job_tmvp_cells": (vp, [vp, P(C.c_size_t)]),
        "ovhip_output_bytes": (C.c_size_t, [i32, i32, P(Window)]),
        "ovhip_output_rows": (C.c_size_t, [i32, i32, P(Window)]),
        "ovhip_output_pack_launch": (C.c_int, [vp, P(Pic), P(Window), vp]),
        "ovhip_output_row_md5_launch": (C.c_int, [vp, P(Pic), P(Window), vp]),
        "ovhip_output_bands": (C.c_size_t, [i32, i32, P(Window)]),
        "ovhip_output_tree_md5_launch": (C.c_int, [vp, P(Pic), P(Window), vp]),
        "ovhip_pic_output": (C.c_int, [vp, P(Pic), P(Window), vp]),
        "ovhip_pic_digest": (C.c_int, [vp, P(Pic), P(Window), vp]),
        "ovhip_md5_init": (None, [P(Md5State)]),
        "ovhip_md5_update": (None, [P(Md5State), vp, C.c_size_t]),
        "ovhip_md5_final": (None, [P(Md5State), vp]),
        "ovhip_d2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "ovhip_host_alloc": (vp, [C.c_size_t]),
        "ovhip_host_free": (None, [vp]),
        "ovhip_job_test_abort_next_flow": (C.c_int, [vp]),
        "ovhip_dpb_create": (C.c_int, [P(vp), P(C.c_int), C.c_int]),
        "ovhip_dpb_create_ex": (C.c_int, [P(vp), C.c_int, P(DpbOps)]),
        "ovhip_dpb_destroy": (None, [vp]),
        "ovhip_dpb_n_devices": (C.c_int, [vp]),
        "ovhip_dpb_device": (C.c_int, [vp, C.c_int]),
        "ovhip_dpb_begin": (C.c_int, [vp, vp, C.c_int, i32, i32, P(Pic)]),
        "ovhip_dpb_want": (C.c_int, [vp, vp, C.c_int]),
        "ovhip_dpb_begin_tag": (C.c_int, [vp, vp, C.c_uint64, C.c_int, i32, i32, P(Pic)]),
        "ovhip_dpb_want_tag": (C.c_int, [vp, vp, C.c_uint64, C.c_int]),
        "ovhip_dpb_acquire_tag": (C.c_int, [vp, vp, C.c_uint64, C.c_int, P(Pic), P(vp)]),
        "ovhip_dpb_publish": (C.c_int, [vp, vp, C.c_int]),
        "ovhip_dpb_acquire": (C.c_int, [vp, vp, C.c_int, P(Pic), P(vp)]),
        "ovhip_dpb_wait_copy": (C.c_int, [vp, C.c_int, vp]),
        "ovhip_dpb_poll_tag": (C.c_int, [vp, vp, C.c_uint64]),
        "ovhip_frame_refs_ready": (C.c_int, [vp]),
        "ovhip_dpb_set_unknown_key_timeout": (None, [vp, C.c_int]),
        "ovhip_dpb_unpin": (C.c_int, [vp, vp]),
        "ovhip_dpb_release": (C.c_int, [vp, vp]),
        "ovhip_dpb_lookup": (C.c_int, [vp, vp, P(C.c_int), P(Pic)]),
        "ovhip_dpb_shutdown": (None, [vp]),
        "ovhip_dpb_get_stats": (C.c_int, [vp, P(DpbStats)]),
        "ovhip_frame_create": (C.c_int, [vp, C.c_int, i32, i32, P(vp)]),
        "ovhip_frame_destroy": (None, [vp]),
        "ovhip_frame_ctx": (vp, [vp]),
        "ovhip_frame_job": (vp, [vp]),
        "ovhip_frame_recorder": (vp, [vp]),
        "ovhip_frame_begin": (C.c_int, [vp, vp]),
        "ovhip_frame_ref": (C.c_int, [vp, vp]),
        "ovhip_frame_set_trace": (None, [vp, vp]),
        "ovhip_rccl_unique_id": (C.c_int, [vp]),
        "ovhip_rccl_create": (C.c_int, [P(vp), vp, C.c_int, C.c_int, C.c_int]),
        "ovhip_rccl_destroy": (None, [vp]),
        "ovhip_rccl_xfer": (vp, [vp]),
        "ovhip_rccl_last_error": (C.c_char_p, [vp]),
        "ovhip_rccl_stats": (C.c_int, [vp, P(C.c_uint64)]),
        "ovhip_rccl_self_exchange": (C.c_int, [vp, P(Pic), P(Pic)]),
        "ovhip_frame_begin_tag": (C.c_int, [vp, vp, C.c_uint64]),
        "ovhip_frame_ref_tag": (C.c_int, [vp, vp, C.c_uint64]),
        "ovhip_frame_ref_at": (C.c_int, [vp, C.c_int, vp]),
        "ovhip_frame_dmvr_rows": (C.c_int64, [vp]),
        "ovhip_frame_dmvr_rows_begin": (C.c_int64, [vp, i32]),
        "ovhip_frame_dmvr_rows_collect": (C.c_int64, [vp]),
        "ovhip_frame_submit": (C.c_int, [vp, vp, P(Pic), P(JobParams), P(FrameOutput)]),
        "ovhip_frame_fail": (C.c_int, [vp, C.c_int]),
        "ovhip_frame_last_error": (C.c_char_p, [vp]),
        "ovhip_calllog_create": (vp, []),
        "ovhip_calllog_destroy": (None, [vp]),
        "ovhip_calllog_reset": (None, [vp]),
        "ovhip_calllog_data": (vp, [vp, P(C.c_size_t)]),
        "ovhip_rec_set_calllog": (None, [vp, vp]),
        "ovhip_calllog_replay": (C.c_int64, [vp, C.c_size_t, vp]),
        "ovhip_stream_create": (C.c_int, [P(vp), vp, P(StreamCfg), P(StreamContent), u32, P(vp), u32]),
        "ovhip_stream_destroy": (None, [vp]),
        "ovhip_stream_run": (C.c_int, [vp, P(StreamPic), u32, u32, u32, u32, vp, P(StreamResult)]),
        "ovhip_stream_frame": (vp, [vp, C.c_int, C.c_int]),
        "ovhip_stream_queue_info": (C.c_int, [vp, P(C.c_int), P(C.c_int)]),
        "ovhip_stream_key": (vp, [vp, u32]),
    }
    ab_build = "OVVC_HIP_LIB_NAME" in os.environ       # tools/ab_lib.sh: an OLDER build of the library beside the current one (its newer entry points are not called)
    for name, (res, args) in sigs.items():
        if ab_build and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError = missing export: fail loudly
        fn.restype, fn.argtypes = res, args
    if lib.ovhip_abi_version() != OVHIP_ABI_VERSION and not ab_build:
        raise RuntimeError("libovvc_hip.so ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "ovhip_abi_version", "ovhip_rec_create", "ovhip_rec_destroy", "ovhip_rec_reset", "ovhip_rec_tu",
    "ovhip_rec_pu", "ovhip_rec_cu_inter", "ovhip_rec_dbf_row", "ovhip_rec_dbf_mv_prepass_view", "ovhip_rec_dbf_ctu", "ovhip_rec_dbf_planes", "ovhip_dbf_launch", "ovhip_sao_launch", "ovhip_alf_launch", "ovhip_rec_tb_cmds", "ovhip_rec_coefs", "ovhip_rec_mc_units", "ovhip_rec_mcx_units", "ovhip_mcx_launch", "ovhip_rec_affine_cu", "ovhip_rec_transform_tree", "ovhip_rec_lmcs_region", "ovhip_dbf_compact", "ovhip_rec_dbf_mv_prepass", "ovhip_dbf_launch_edges", "ovhip_rec_ciip", "ovhip_ciip_weight", "ovhip_rec_ciip_units", "ovhip_ciip_launch", "ovhip_rec_lmcs_regions", "ovhip_rec_tb_cmds_split", "ovhip_itx_launch_classes", "ovhip_itx_launch_chroma_lmcs", "ovhip_lmcs_build", "ovhip_lmcs_scale_launch", "ovhip_lmcs_inverse_launch", "ovhip_rec_aff_units", "ovhip_rec_aff_side", "ovhip_mca_launch", "ovhip_mcxa_launch", "ovhip_ctx_create",
    "ovhip_ctx_destroy", "ovhip_ctx_sync", "ovhip_ctx_shares_queue", "ovhip_ctx_new_stream", "ovhip_ctx_fork", "ovhip_ctx_join", "ovhip_last_error", "ovhip_ctx_stream", "ovhip_malloc",
    "ovhip_free", "ovhip_h2d", "ovhip_d2h", "ovhip_pic_alloc", "ovhip_pic_free", "ovhip_pic_upload",
    "ovhip_pic_download", "ovhip_itx_launch", "ovhip_mc_launch",
    "ovhip_rec_create_ex", "ovhip_rec_set_dense_dbf_planes", "ovhip_rec_dbf_edges", "ovhip_dbf_launch_edges_ex",
    "ovhip_dmvr_search_launch", "ovhip_rec_append_raw", "ovhip_rec_set_dbf_offsets", "ovhip_rec_tu_intra", "ovhip_rec_itasks", "ovhip_rec_itasks_sorted", "ovhip_itx_launch_classes_res", "ovhip_intra_level_launch", "ovhip_intra_level_geom", "ovhip_intra_sync_words", "ovhip_intra_ctu_launch", "ovhip_intra_flow_words", "ovhip_intra_flow_items", "ovhip_intra_flow_launch",
    "ovhip_rec_itask_levels", "ovhip_rec_isp_cu", "ovhip_isp_geometry", "ovhip_rec_itasks_by_ctu", "ovhip_rec_set_ctu_size", "ovhip_job_bind", "ovhip_job_create", "ovhip_job_destroy", "ovhip_job_recorder", "ovhip_job_begin",
    "ovhip_frame_band", "ovhip_frame_band_upto", "ovhip_frame_dmvr_rows_begin_upto", "ovhip_job_dmvr_rows_begin_upto", "ovhip_frame_set_band_mode", "ovhip_frame_band_stats", "ovhip_dpb_post_rows", "ovhip_dpb_rows_tag",
    "ovhip_job_band", "ovhip_job_band_active", "ovhip_job_band_busy", "ovhip_job_band_reserve", "ovhip_job_reserve_for_picture", "ovhip_rec_reserve_for_picture", "ovhip_job_band_progress", "ovhip_rec_counts", "ovhip_sao_launch_rows", "ovhip_alf_launch_rows",
    "ovhip_job_flush", "ovhip_job_wait", "ovhip_job_refined_mvs", "ovhip_job_dmvr_rows", "ovhip_job_dmvr_rows_begin", "ovhip_job_dmvr_rows_collect", "ovhip_job_last_stats", "ovhip_job_time_stage", "ovhip_job_stage_time",
    "ovhip_output_bytes", "ovhip_output_rows", "ovhip_output_pack_launch", "ovhip_output_row_md5_launch", "ovhip_output_bands", "ovhip_output_tree_md5_launch", "ovhip_pic_output", "ovhip_pic_digest",
    "ovhip_md5_init", "ovhip_md5_update", "ovhip_md5_final", "ovhip_tmvp_cells_launch", "ovhip_job_tmvp_cells", "ovhip_intra_flow_untag_launch", "ovhip_lmcs_inverse_untag_launch", "ovhip_lmcs_scale_prepare_launch",
    "ovhip_host_alloc", "ovhip_host_free", "ovhip_d2d", "ovhip_job_test_abort_next_flow",
    "ovhip_dpb_create", "ovhip_dpb_create_ex", "ovhip_dpb_destroy", "ovhip_dpb_n_devices", "ovhip_dpb_device", "ovhip_dpb_begin", "ovhip_dpb_want",
    "ovhip_dpb_publish", "ovhip_dpb_acquire", "ovhip_dpb_wait_copy", "ovhip_dpb_poll_tag", "ovhip_frame_refs_ready", "ovhip_dpb_set_unknown_key_timeout", "ovhip_dpb_unpin", "ovhip_dpb_release", "ovhip_dpb_lookup", "ovhip_dpb_shutdown",
    "ovhip_dpb_get_stats", "ovhip_dpb_begin_tag", "ovhip_dpb_want_tag", "ovhip_dpb_acquire_tag", "ovhip_frame_begin_tag", "ovhip_frame_ref_tag", "ovhip_frame_set_trace",
    "ovhip_rccl_unique_id", "ovhip_rccl_create", "ovhip_rccl_destroy", "ovhip_rccl_xfer", "ovhip_rccl_last_error", "ovhip_rccl_stats", "ovhip_rccl_self_exchange",
    "ovhip_frame_create", "ovhip_frame_destroy", "ovhip_frame_ctx", "ovhip_frame_job", "ovhip_frame_recorder", "ovhip_frame_begin", "ovhip_frame_ref",
    "ovhip_frame_ref_at", "ovhip_frame_dmvr_rows", "ovhip_frame_dmvr_rows_begin", "ovhip_frame_dmvr_rows_collect", "ovhip_frame_submit", "ovhip_frame_fail", "ovhip_frame_last_error",
    "ovhip_calllog_create", "ovhip_calllog_destroy", "ovhip_calllog_reset", "ovhip_calllog_data", "ovhip_rec_set_calllog", "ovhip_calllog_replay",
    "ovhip_stream_create", "ovhip_stream_destroy", "ovhip_stream_run", "ovhip_stream_frame", "ovhip_stream_key", "ovhip_stream_queue_info",
]


class Recorder:
    """Host-side recorder (ovhip_rec_*): turns reference-style TU/PU calls into command buffers."""

    def __init__(self, pic_w: int, pic_h: int):
        self.lib = load()
        self.h = self.lib.ovhip_rec_create(pic_w, pic_h)
        if not self.h:
            raise MemoryError("ovhip_rec_create")
        self._keep = []

    def close(self):
        if self.h:
            self.lib.ovhip_rec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.lib.ovhip_rec_reset(self.h)

    def start_calllog(self):
        """Every recorder call from now on is also serialised (ovhip_rec_set_calllog); take_calllog() returns the bytes."""
        self._log = self.lib.ovhip_calllog_create()
        self.lib.ovhip_rec_set_calllog(self.h, self._log)

    def take_calllog(self) -> np.ndarray:
        n = C.c_size_t()
        p = self.lib.ovhip_calllog_data(self._log, C.byref(n))
        if not p and n.value:
            raise MemoryError("call log")
        out = np.frombuffer((C.c_char * n.value).from_address(p), dtype=np.uint8).copy() if n.value else np.zeros(0, np.uint8)
        self.lib.ovhip_rec_set_calllog(self.h, None)
        self.lib.ovhip_calllog_destroy(self._log)
        self._log = None
        return out

    def replay(self, log: np.ndarray) -> int:
        n = self.lib.ovhip_calllog_replay(log.ctypes.data, log.nbytes, self.h)
        if n < 0:
            raise ValueError(f"ovhip_calllog_replay -> {n}")
        return int(n)

    def tu(self, st: TuState, d: TuDesc) -> int:
        r = self.lib.ovhip_rec_tu(self.h, C.byref(st), C.byref(d))
        if r < 0:
            raise ValueError(f"ovhip_rec_tu -> {r}")
        return r

    def tu_intra(self, st: TuState, d: TuDesc, task_l: "ITask | None", task_c: "ITask | None") -> int:
        """A TU with the ordered (intra / CIIP) prediction tasks the reference runs around its residuals."""
        r = self.lib.ovhip_rec_tu_intra(self.h, C.byref(st), C.byref(d), C.byref(task_l) if task_l is not None else None,
                                        C.byref(task_c) if task_c is not None else None)
        if r < 0:
            raise ValueError(f"ovhip_rec_tu_intra -> {r}")
        return r

    def itasks(self) -> np.ndarray:
        return self._arr(self.lib.ovhip_rec_itasks, ITASK_DTYPE)

    def itasks_sorted(self):
        """(tasks sorted by level, level_start uint32 [n_levels + 1])"""
        n, nl, ls = C.c_size_t(), C.c_uint32(), C.POINTER(C.c_uint32)()
        p = self.lib.ovhip_rec_itasks_sorted(self.h, C.byref(n), C.byref(ls), C.byref(nl))
        if not n.value:
            return np.zeros(0, ITASK_DTYPE), np.zeros(1, np.uint32)
        t = np.frombuffer((C.c_char * (n.value * 32)).from_address(p), dtype=ITASK_DTYPE).copy()
        return t, np.array([ls[i] for i in range(nl.value + 1)], np.uint32)

    def itasks_by_ctu(self, log2_ctu: int = 7):
        """(tasks grouped by CTU, CTU descriptors ICTU_DTYPE)"""
        n, nc, cp = C.c_size_t(), C.c_size_t(), C.c_void_p()
        p = self.lib.ovhip_rec_itasks_by_ctu(self.h, log2_ctu, C.byref(n), C.byref(cp), C.byref(nc))
        if not n.value:
            return np.zeros(0, ITASK_DTYPE), np.zeros(0, ICTU_DTYPE)
        t = np.frombuffer((C.c_char * (n.value * 32)).from_address(p), dtype=ITASK_DTYPE).copy()
        c = np.frombuffer((C.c_char * (nc.value * 16)).from_address(cp.value), dtype=ICTU_DTYPE).copy()
        return t, c

    def transform_tree(self, st: "TuState", d: TtDesc, infos: bytes, res_cb: np.ndarray, res_cr: np.ndarray, res_y: np.ndarray) -> int:
        """tmp.rcn_transform_tree: infos = 16 ovhip_tu_info structs, res_* = the CTU's residual_cb / _cr / _y buffers."""
        ib = (C.c_char * len(infos)).from_buffer_copy(infos)
        keep = [np.ascontiguousarray(a, dtype=np.int16) for a in (res_cb, res_cr, res_y)]
        d.tu_info = C.addressof(ib)
        for k in range(3):
            d.residual[k] = keep[k].ctypes.data
        r = self.lib.ovhip_rec_transform_tree(self.h, C.byref(st), C.byref(d))
        d.tu_info = None
        if r < 0:
            raise ValueError(f"ovhip_rec_transform_tree -> {r}")
        return r

    def isp_cu(self, st: TuState, d: IspDesc) -> int:
        r = self.lib.ovhip_rec_isp_cu(self.h, C.byref(st), C.byref(d))
        if r < 0:
            raise ValueError(f"ovhip_rec_isp_cu -> {r}")
        return r

    def isp_geometry(self, log2_w: int, log2_h: int, vertical: int):
        """(log2 partition size, partitions, log2 prediction-call size, prediction calls) of an ISP CU"""
        a, b, c, d = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self.lib.ovhip_isp_geometry(log2_w, log2_h, vertical, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return a.value, b.value, c.value, d.value

    def pu(self, d: PuDesc) -> int:
        r = self.lib.ovhip_rec_pu(self.h, C.byref(d))
        if r < 0:
            raise ValueError(f"ovhip_rec_pu -> {r}")
        return r

    def affine_cu(self, d: AffineDesc, mv0: np.ndarray, mv1: np.ndarray) -> int:
        """mv0 / mv1: int32 [rows, mv_stride, 2] sub-block motion fields (host)."""
        mv0 = np.ascontiguousarray(mv0, dtype=np.int32)
        mv1 = np.ascontiguousarray(mv1, dtype=np.int32)
        d.mv0, d.mv1 = mv0.ctypes.data, mv1.ctypes.data
        r = self.lib.ovhip_rec_affine_cu(self.h, C.byref(d))
        d.mv0 = d.mv1 = None
        if r < 0:
            raise ValueError(f"ovhip_rec_affine_cu -> {r}")
        return r

    def dbf_ctu(self, raw: bytes, mvctx: bytes | None = None):
        """raw: one ovhip_dbf_ctu struct (what df.rcn_dbf_ctu receives).  mvctx (P / B slices): one ovhip_dbf_mv_ctx
        followed by the two OVMV[34 * 34] motion arrays; the MV-based bS pre-pass is applied to the CTU first."""
        assert len(raw) == DBF_CTU_SIZE, (len(raw), DBF_CTU_SIZE)
        buf = (C.c_char * len(raw)).from_buffer_copy(raw)
        if mvctx is not None:
            hs = C.sizeof(DbfMvCtx)
            mc = DbfMvCtx.from_buffer_copy(mvctx[:hs])
            n = 34 * 34 * mc.mv_bytes
            assert len(mvctx) == hs + 2 * n, (len(mvctx), hs, n)
            m0 = (C.c_char * n).from_buffer_copy(mvctx[hs:hs + n])
            m1 = (C.c_char * n).from_buffer_copy(mvctx[hs + n:])
            mc.mvs0, mc.mvs1 = C.addressof(m0), C.addressof(m1)
            r = self.lib.ovhip_rec_dbf_mv_prepass(C.addressof(buf), C.byref(mc))
            if r < 0:
                raise ValueError(f"ovhip_rec_dbf_mv_prepass -> {r}")
        r = self.lib.ovhip_rec_dbf_ctu(self.h, C.addressof(buf))
        if r < 0:
            raise ValueError(f"ovhip_rec_dbf_ctu -> {r}")

    def dbf_edges(self, direction: int):
        """The compact edge list (DBF_EDGE_DTYPE) ovhip_rec_dbf_ctu emitted for one direction + the offset table."""
        n = C.c_size_t()
        offs = DbfOffsets()
        p = self.lib.ovhip_rec_dbf_edges(self.h, direction, C.byref(n), C.byref(offs))
        if not n.value:
            return np.zeros(0, DBF_EDGE_DTYPE), offs
        buf = (C.c_char * (n.value * DBF_EDGE_DTYPE.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=DBF_EDGE_DTYPE).copy(), offs

    def append_raw(self, which: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        r = self.lib.ovhip_rec_append_raw(self.h, which, arr.ctypes.data, len(arr))
        if r < 0:
            raise ValueError(f"ovhip_rec_append_raw({which}) -> {r}")

    def dbf_planes(self) -> dict:
        """Host copies of the picture-level edge planes + offsets."""
        pl = DbfPlanes()
        r = self.lib.ovhip_rec_dbf_planes(self.h, C.byref(pl))
        if r < 0:
            raise ValueError(f"ovhip_rec_dbf_planes -> {r}")
        out = {"w4": pl.w4, "h4": pl.h4, "beta_offset": pl.beta_offset, "tc_offset": pl.tc_offset}
        for name, shape in dbf_plane_shapes(pl.w4, pl.h4).items():
            n = shape[0] * shape[1]
            buf = (C.c_uint16 * n).from_address(getattr(pl, name))
            out[name] = np.frombuffer(buf, dtype=np.uint16).reshape(shape).copy()
        return out

    def _arr(self, fn, dtype):
        n = C.c_size_t(0)
        p = fn(self.h, C.byref(n))
        if not n.value:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n.value * np.dtype(dtype).itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=dtype).copy()

    def tb_cmds(self) -> np.ndarray:
        return self._arr(self.lib.ovhip_rec_tb_cmds, TB_CMD_DTYPE)

    def coefs(self) -> np.ndarray:
        return self._arr(self.lib.ovhip_rec_coefs, np.int16)

    def mc_units(self) -> np.ndarray:
        return self._arr(self.lib.ovhip_rec_mc_units, MC_UNIT_DTYPE)

    def lmcs_region(self, x0: int, y0: int, abv_mask: int, lft_mask: int) -> int:
        r = self.lib.ovhip_rec_lmcs_region(self.h, x0, y0, abv_mask, lft_mask)
        if r < 0:
            raise ValueError(f"ovhip_rec_lmcs_region -> {r}")
        return r

    def ciip(self, x0: int, y0: int, log2_w: int, log2_h: int, mode_abv: int, mode_lft: int) -> int:
        r = self.lib.ovhip_rec_ciip(self.h, x0, y0, log2_w, log2_h, mode_abv, mode_lft)
        if r < 0:
            raise ValueError(f"ovhip_rec_ciip -> {r}")
        return r

    def ciip_units(self) -> np.ndarray:
        return self._arr(self.lib.ovhip_rec_ciip_units, CIIP_UNIT_DTYPE)

    def lmcs_regions(self) -> np.ndarray:
        return self._arr(self.lib.ovhip_rec_lmcs_regions, LMCS_REGION_DTYPE)

    def tb_cmds_split(self):
        """(commands reordered [luma > 16, luma <= 16x16, chroma > 16, chroma <= 16x16], the four counts)."""
        counts, n = (C.c_size_t * 4)(), C.c_size_t()
        p = self.lib.ovhip_rec_tb_cmds_split(self.h, counts, C.byref(n))
        if not n.value:
            return np.zeros(0, TB_CMD_DTYPE), (0, 0, 0, 0)
        buf = (C.c_char * (n.value * TB_CMD_DTYPE.itemsize)).from_address(p)
        return np.frombuffer(buf, dtype=TB_CMD_DTYPE).copy(), tuple(int(c) for c in counts)

    def aff_units(self) -> np.ndarray:
        return self._arr(self.lib.ovhip_rec_aff_units, AFF_UNIT_DTYPE)

    def aff_side(self) -> np.ndarray:
        return self._arr(self.lib.ovhip_rec_aff_side, np.dtype("<i4"))

    def mcx_units(self) -> np.ndarray:
        """The BDOF / DMVR units (ovhip_mcx_launch)."""
        return self._arr(self.lib.ovhip_rec_mcx_units, MC_UNIT_DTYPE)
