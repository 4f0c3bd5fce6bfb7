"""Synthetic "recorded picture" generator: what the host side of the decoder
(CABAC + drv) would hand to the rcn back-end for one inter picture, produced from a seed instead
of a bitstream (no .266 stream exists in this environment).

The generator speaks the reference's vocabulary -- coding units, prediction units with motion
vectors in 1/16 pel, transform units with `struct TUInfo`-style flags and sub-block-major
coefficient buffers -- and pushes every unit through the product's own host recorder
(ovhip_rec_pu / ovhip_rec_tu), so benches and tests exercise the same host logic a shim would.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

LM_FRAC = 0.35      # share of intra chroma blocks predicted by a cross-component linear model (LM / MDLM)
ISP_FRAC = 0.08     # share of the eligible intra CUs coded with intra sub-partitions (the reference decoder's random-walk streams, tests/golden/
                    # shim_pipe*.ovg: 31 of 581 and 21 of 219 intra CUs; a third of their prediction calls are 1 or 2 rows high)

from . import capi

DEFAULT_SEED = 0x266


def smooth_plane(rs: np.random.RandomState, h: int, w: int) -> np.ndarray:
    """Low-pass random 10-bit content (so filter decisions are non-degenerate) + a little noise."""
    gh, gw = h // 16 + 2, w // 16 + 2
    g = rs.randint(64, 960, size=(gh, gw)).astype(np.float32)
    yy = np.linspace(0, gh - 1.001, h, dtype=np.float32)
    xx = np.linspace(0, gw - 1.001, w, dtype=np.float32)
    y0 = yy.astype(np.int32); x0 = xx.astype(np.int32)
    fy = (yy - y0)[:, None]; fx = (xx - x0)[None, :]
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    p = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    p += rs.randint(-12, 13, size=(h, w))
    return np.clip(p, 0, 1023).astype(np.uint16)


def random_picture(rs, w, h):
    return smooth_plane(rs, h, w), smooth_plane(rs, h // 2, w // 2), smooth_plane(rs, h // 2, w // 2)


def partition(rs: np.random.RandomState, pic_w: int, pic_h: int, ctu: int = 128, min_cu: int = 8, max_cu: int = 128):
    """Random QT/BT partition of every CTU into CUs (x, y, log2w, log2h), RA-like size mix.
    Blocks crossing the picture border are split until they fit (implicit split)."""
    out = []
    # split probability by max(log2w, log2h): favours 8..32 blocks like RA streams
    p_split = {7: 0.93, 6: 0.82, 5: 0.62, 4: 0.38, 3: 0.0}

    def rec(x, y, w, h):
        if x >= pic_w or y >= pic_h:
            return
        if x + w > pic_w or y + h > pic_h:
            if w > min_cu and h > min_cu:
                hw, hh = w // 2, h // 2
                rec(x, y, hw, hh); rec(x + hw, y, hw, hh); rec(x, y + hh, hw, hh); rec(x + hw, y + hh, hw, hh)
            elif w > min_cu:
                rec(x, y, w // 2, h); rec(x + w // 2, y, w // 2, h)
            elif h > min_cu:
                rec(x, y, w, h // 2); rec(x, y + h // 2, w, h // 2)
            return
        m = max(w, h)
        l2 = m.bit_length() - 1
        if m > min_cu and (m > max_cu or rs.random_sample() < p_split.get(l2, 0.0)):
            k = rs.randint(0, 4)
            if k < 2 and w == h:
                hw, hh = w // 2, h // 2
                rec(x, y, hw, hh); rec(x + hw, y, hw, hh); rec(x, y + hh, hw, hh); rec(x + hw, y + hh, hw, hh)
            elif (k == 2 and w > min_cu) or h <= min_cu:
                rec(x, y, w // 2, h); rec(x + w // 2, y, w // 2, h)
            else:
                rec(x, y, w, h // 2); rec(x, y + h // 2, w, h // 2)
            return
        out.append((x, y, w.bit_length() - 1, h.bit_length() - 1))

    for cy in range(0, pic_h, ctu):
        for cx in range(0, pic_w, ctu):
            rec(cx, cy, ctu, ctu)
    return np.array(out, dtype=np.int32)


@dataclass
class Workload:
    """One recorded inter picture, host side."""
    w: int
    h: int
    seed: int
    refs: list                      # [(y, cb, cr)] reference pictures
    ref_pocs: list
    cus: np.ndarray                 # (n, 4) x, y, log2w, log2h
    mc_units: np.ndarray            # capi.MC_UNIT_DTYPE: plain uni / bi / BCW / GPM units
    tb_cmds: np.ndarray             # capi.TB_CMD_DTYPE, luma commands first (n_luma_cmds), then chroma
    coefs: np.ndarray               # int16 arena
    n_luma_cmds: int = 0
    tb_classes: tuple = (0, 0, 0, 0)  # counts: luma > 16, luma <= 16x16, chroma > 16, chroma <= 16x16
    mcx_units: np.ndarray = None    # BDOF / DMVR units
    aff_units: np.ndarray = None    # affine (+PROF) units and their side arena
    aff_side: np.ndarray = None
    ciip_units: np.ndarray = None   # CIIP blends; `intra` stands in for the caller's planar prediction
    intra: tuple = None
    lmcs: "capi.LmcsLuts" = None    # None: LMCS off
    lmcs_regions: np.ndarray = None
    dbf_ctus: np.ndarray = None     # capi.DBF_CTU_DTYPE: what df.rcn_dbf_ctu receives, CTU by CTU
    dbf_planes: dict = None         # picture-level deblocking edge planes ovhip_rec_dbf_ctu derived (include/ovvc_hip.h)
    dbf_edges: list = None          # [vertical, horizontal] edge lists it emitted (capi.DBF_EDGE_DTYPE)
    sao_params: np.ndarray = None   # capi.SAO_CTU_DTYPE per CTU
    alf: dict = None                # ALF tables + per-CTU parameters
    itasks: np.ndarray = None       # capi.ITASK_DTYPE, decoding order: intra / CIIP / ordered-scale tasks (None: none)
    calllog: np.ndarray = None      # uint8: the recorder calls that produced all of the above, serialised (ovhip_calllog_replay)
    stats: dict = field(default_factory=dict)

    @property
    def frame_bytes(self) -> int:
        return self.w * self.h * 3    # 4:2:0, 2 bytes / sample

    @property
    def lmcs_fwd(self):
        return None if self.lmcs is None else np.frombuffer(bytes(self.lmcs), np.uint16)[:1024].copy()

    @property
    def lmcs_bwd(self):
        return None if self.lmcs is None else np.frombuffer(bytes(self.lmcs), np.uint16)[1024:2048].copy()


def _coef_values(rs, n):
    """Laplacian-ish quantised levels: mostly small, occasionally large."""
    mag = np.minimum(rs.geometric(0.35, size=n), 600).astype(np.int32)
    big = rs.random_sample(n) < 0.01
    mag = np.where(big, rs.randint(100, 4000, size=n), mag)
    sign = np.where(rs.random_sample(n) < 0.5, -1, 1)
    return (mag * sign).astype(np.int16)


def _lmcs_tables(rs) -> "capi.LmcsLuts":
    """A mild piecewise-linear luma mapping (16 windows), like CTC HDR/SDR LMCS parameter sets."""
    d = capi.LmcsData()
    d.min_bin_idx, d.delta_max_bin_idx = 1, 1
    d.crs_offset = int(rs.randint(-3, 4))
    for i in range(16):
        d.cw_delta[i] = int(rs.randint(-14, 15))
    return capi.lmcs_build(d)


# coding tools of an inter picture; "base" = translational uni / bi / BCW prediction only
ALL_TOOLS = ("bdof", "dmvr", "affine", "gpm", "ciip", "lmcs")
# + "intra": intra CUs (intra_frac of the CUs <= 64x64; 1.0 = an I picture) predicted on the device in dependency order,
# and CIIP's planar part computed there too instead of being read from a caller-supplied picture
# + "isp": ISP_FRAC of those CUs as intra-sub-partition CUs (tmp.recon_isp_subtree_v / _h -> ovhip_rec_isp_cu): 2 or 4 partitions predicted
# and reconstructed one after the other, horizontal ones down to 1 and 2 rows
INTRA_TOOLS = ALL_TOOLS + ("intra", "isp")


def make_workload(w: int, h: int, seed: int = DEFAULT_SEED, bi_frac: float = 0.6, cbf_y: float = 0.5,
                  cbf_c: float = 0.3, mv_range_px: int = 64, tools=ALL_TOOLS, intra_frac: float = 0.12, calllog: bool = False,
                  isp_64x2: bool = False) -> Workload:
    """tools: subset of ALL_TOOLS.  Rates follow JVET CTC random-access statistics in spirit: of the
    bi-predicted CUs that satisfy check_bdof() (vcl_coding_unit.c:2019-2027) and whose references lie on
    opposite sides of the picture, ~45 % use BDOF alone and ~35 % DMVR (+BDOF); ~8 % of the CUs >= 16x16
    are affine (70 % of them with PROF), ~3 % GPM, ~3 % CIIP."""
    tools = set(tools)
    rs = np.random.RandomState(seed & 0x7FFFFFFF)
    ref0 = random_picture(rs, w, h)
    # the second reference is a displaced, slightly noisy copy of the first (so that DMVR's search and
    # BDOF's gradients see correlated content, as between two real pictures of a sequence)
    ref1 = tuple(np.clip(np.roll(p, (1, 2), axis=(0, 1)).astype(np.int32) + rs.randint(-3, 4, size=p.shape), 0, 1023).astype(np.uint16)
                 for p in ref0)
    refs = [ref0, ref1]
    pocs = [8, 24]                      # current picture would be POC 16
    cus = partition(rs, w, h, max_cu=64 if ("intra" in set(tools) and intra_frac >= 1.0) else 128)      # an I picture: intra CUs are at most 64x64
    n = len(cus)
    rec = capi.Recorder(w, h)
    if calllog:
        rec.start_calllog()
    lw, lh = cus[:, 2], cus[:, 3]

    inter_dir = np.where(rs.random_sample(n) < bi_frac, 3, rs.randint(1, 3, size=n)).astype(np.int32)
    small = (lw + lh) < 6                        # 4x8 / 8x4 / ... : uni-pred only in VVC
    inter_dir = np.where(small & (inter_dir == 3), 1, inter_dir)
    mv = rs.randint(-mv_range_px * 16, mv_range_px * 16 + 1, size=(n, 4)).astype(np.int32)
    integer = rs.random_sample((n, 4)) < 0.25     # a share of integer / half positions like real MV fields
    mv = np.where(integer, mv & ~15, mv)
    ridx = rs.randint(0, 2, size=(n, 2))
    bcw = np.where((inter_dir == 3) & (rs.random_sample(n) < 0.1), rs.randint(1, 6, size=n), 0)
    hpel = rs.random_sample(n) < 0.05
    lmcs_on = "lmcs" in tools

    # ---- coding mode per CU ----
    PLAIN, BDOF, DMVR, AFFINE, GPM, CIIP, INTRA = range(7)
    mode = np.zeros(n, np.int32)
    r = rs.random_sample((n, 3))
    dev_intra = "intra" in tools
    if dev_intra:
        # intra CUs come in clusters in real pictures (occlusions, new content): one draw per 32x32 area + per-CU noise
        gx, gy = (w + 31) // 32, (h + 31) // 32
        area_hot = rs.random_sample((gy, gx)) < intra_frac * 0.8
        hot = area_hot[cus[:, 1] // 32, cus[:, 0] // 32]
        is_intra = ((hot & (rs.random_sample(n) < 0.85)) | (rs.random_sample(n) < intra_frac * 0.3) | (intra_frac >= 1.0)) & (lw <= 6) & (lh <= 6)
        if intra_frac >= 1.0:
            assert is_intra.all()
    else:
        is_intra = np.zeros(n, bool)
    if "affine" in tools:
        mode = np.where((lw >= 4) & (lh >= 4) & (r[:, 0] < 0.08), AFFINE, mode)
    if "gpm" in tools:
        mode = np.where((mode == PLAIN) & (lw >= 3) & (lh >= 3) & (lw <= 6) & (lh <= 6) & (r[:, 0] > 0.97), GPM, mode)
    if "ciip" in tools:
        mode = np.where((mode == PLAIN) & (lw + lh >= 6) & (lw <= 6) & (lh <= 6) & (r[:, 0] > 0.94) & (r[:, 0] <= 0.97), CIIP, mode)
    elig = (mode == PLAIN) & (inter_dir == 3) & (lw >= 3) & (lh >= 3) & (lw + lh >= 7) & (bcw == 0) & (ridx[:, 0] == ridx[:, 1])
    if "dmvr" in tools:
        mode = np.where(elig & (r[:, 1] < 0.35), DMVR, mode)
    if "bdof" in tools:
        mode = np.where(elig & (mode == PLAIN) & (r[:, 1] < 0.80), BDOF, mode)
    mode = np.where(is_intra, INTRA, mode)
    bcw = np.where(np.isin(mode, (GPM, CIIP)), 0, bcw)
    hpel = np.where(np.isin(mode, (AFFINE,)) | ((lw == 2) & (lh == 2)), False, hpel)
    # merge-mode (DMVR) motion is close to mirrored between the two lists
    mirrored = -mv[:, :2] + rs.randint(-24, 25, size=(n, 2))
    mv[:, 2:] = np.where((mode == DMVR)[:, None], mirrored, mv[:, 2:])

    # rpl0 = [ref0, ref1], rpl1 = [ref1, ref0]
    slot = np.array([[0, 1], [1, 0]])
    pd = capi.PuDesc()
    ad = capi.AffineDesc()
    cu_modes = (1, 2, 4, 6)             # OV_INTER, OV_INTRA, OV_MIP, OV_INTER_SKIP (cu_utils.h:132-139)
    ciip_wt = np.zeros(n, np.int32)
    for i in range(n):
        x, y, l2w, l2h = (int(v) for v in cus[i])
        m = int(mode[i])
        if m == INTRA:
            continue
        ref0s, ref1s = int(slot[0, ridx[i, 0]]), int(slot[1, ridx[i, 1]])
        if m == AFFINE:
            nsx, nsy = (1 << l2w) >> 2, (1 << l2h) >> 2
            ad.x0, ad.y0, ad.log2_w, ad.log2_h = x, y, l2w, l2h
            ad.inter_dir = int(inter_dir[i]); ad.bcw_idx_plus1 = int(bcw[i]); ad.lmcs = int(lmcs_on)
            ad.prof_dir = 0 if rs.random_sample() < 0.3 else (int(rs.randint(1, 4)) if inter_dir[i] == 3 else int(inter_dir[i]))
            ad.ref0, ad.ref1, ad.poc0, ad.poc1 = ref0s, ref1s, pocs[ref0s], pocs[ref1s]
            ad.mv_stride = nsx
            for t in range(4):
                for k in range(16):
                    ad.dmv_scale[t][k] = int(rs.randint(-12, 13))
            gx = rs.randint(-10, 11, size=(2, 2, 2))          # [list][d/dx, d/dy][mv component], 1/16 pel per sub-block
            ii, jj = np.meshgrid(np.arange(nsx), np.arange(nsy))
            fields = []
            for l in range(2):
                fx = mv[i, 2 * l] + ((gx[l, 0, 0] * ii + gx[l, 1, 0] * jj) >> 1)
                fy = mv[i, 2 * l + 1] + ((gx[l, 0, 1] * ii + gx[l, 1, 1] * jj) >> 1)
                fields.append(np.stack([fx, fy], axis=-1).astype(np.int32))
            rec.affine_cu(ad, fields[0], fields[1])
            continue
        pd.x0, pd.y0, pd.log2_w, pd.log2_h = x, y, l2w, l2h
        pd.inter_dir = int(inter_dir[i])
        pd.ref_idx0, pd.ref_idx1 = int(ridx[i, 0]), int(ridx[i, 1])
        pd.bcw_idx_plus1 = int(bcw[i]); pd.prec_amvr_half = int(hpel[i]); pd.planes = 3; pd.lmcs = int(lmcs_on)
        pd.mv0x, pd.mv0y, pd.mv1x, pd.mv1y = (int(v) for v in mv[i])
        if hpel[i]:
            pd.mv0x = (pd.mv0x & ~15) | 8; pd.mv1y = (pd.mv1y & ~15) | 8
        pd.ref0, pd.ref1 = ref0s, ref1s
        pd.poc0 = pocs[pd.ref0]; pd.poc1 = pocs[pd.ref1]
        pd.refine = 0; pd.gpm_split_dir = 0
        if m == BDOF:
            pd.refine = capi.PU_BDOF
        elif m == DMVR:
            pd.refine = capi.PU_DMVR | (capi.PU_BDOF if rs.random_sample() < 0.8 else 0)
        elif m == GPM:
            pd.refine = capi.PU_GPM; pd.inter_dir = 3; pd.gpm_split_dir = int(rs.randint(0, 64))
        pd.ciip_wt = 0
        if m == CIIP:
            # rcn_ciip(_b): inter prediction + blend with the planar prediction; without device intra the planar prediction
            # comes from a caller-supplied picture and the blend is fused into the units, with it the blend is an ordered task
            ciip_wt[i] = capi.load().ovhip_ciip_weight(cu_modes[rs.randint(0, 4)], cu_modes[rs.randint(0, 4)])
            if not dev_intra:
                pd.ciip_wt = int(ciip_wt[i])
        rec.pu(pd)

    # ---- transform units ----
    st = capi.TuState()
    st.dep_quant = 1; st.mts_implicit = 0; st.sh_ts_disabled = 0
    st.ict_type = 3 if lmcs_on else 2
    st.lmcs_scale_c = 2 if lmcs_on else 0; st.lmcs_chroma_scale = 1 << 11
    td = capi.TuDesc()
    n_tu = 0
    coef_pool = _coef_values(rs, 1 << 20)
    pool_pos = 0
    bufs = [np.zeros(32 * 32, np.int16) for _ in range(3)]
    w4, h4 = (w + 3) // 4, (h + 3) // 4
    done = np.zeros((h4 + 1, w4 + 1), bool)          # decoding progress on the 4x4 grid (what the progress bit-fields hold)
    isp_on = dev_intra and "isp" in tools
    isp = capi.IspDesc()
    isp_buf = np.zeros(64 * 64, np.int16)
    n_isp = 0

    def avail(ux, uy, nw, nh):
        """(corner, units available above from ux, units available left from uy): neighbours decoded before this block"""
        corner = bool(ux > 0 and uy > 0 and done[uy - 1, ux - 1])
        a = 0
        if uy > 0:
            while a < 2 * nw and ux + a < w4 and done[uy - 1, ux + a]:
                a += 1
        l = 0
        if ux > 0:
            while l < 2 * nh and uy + l < h4 and done[uy + l, ux - 1]:
                l += 1
        return corner, a, l

    for i in range(n):
        x, y, l2w, l2h = (int(v) for v in cus[i])
        cu_intra, cu_ciip = mode[i] == INTRA, (mode[i] == CIIP and dev_intra)
        if lmcs_on and not (x & 63) and not (y & 63):
            # rcn_lmcs_compute_chroma_scale at every 64-aligned CU (vcl_coding_unit.c:724-730); neighbours are
            # available up to the picture edge
            # available = decoded before this CU (the progress bit-fields of the reference, rcn_lmcs.c:327-332)
            n_abv = n_lft = 0
            if y > 0:
                while n_abv < 16 and (x >> 2) + n_abv < w4 and done[(y >> 2) - 1, (x >> 2) + n_abv]:
                    n_abv += 1
            if x > 0:
                while n_lft < 16 and (y >> 2) + n_lft < h4 and done[(y >> 2) + n_lft, (x >> 2) - 1]:
                    n_lft += 1
            rec.lmcs_region(x, y, (1 << n_abv) - 1, (1 << n_lft) - 1)
        qp = int(rs.randint(22, 38)) + 12
        for ty in range(0, 1 << l2h, 64):
            for tx in range(0, 1 << l2w, 64):
                tl2w, tl2h = min(l2w, 6), min(l2h, 6)
                r = rs.random_sample(6)
                cbf = (0x10 if r[0] < cbf_y else 0)
                if tl2w + tl2h >= 6:         # chroma TBs of at least 16 samples
                    if r[1] < 0.1:
                        cbf |= 0x8 | int(rs.randint(1, 4))
                    else:
                        cbf |= (0x2 if r[2] < cbf_c else 0) | (0x1 if r[3] < cbf_c else 0)
                if not cbf and not (cu_intra or cu_ciip):
                    continue
                td.x0, td.y0, td.log2_tb_w, td.log2_tb_h = x + tx, y + ty, tl2w, tl2h
                td.tree = 0; td.cbf_mask = cbf; td.cu_flags = 0
                td.tr_skip_mask = 0; td.cu_mts_flag = 0; td.cu_mts_idx = 0; td.lfnst_flag = 0; td.lfnst_idx = 0
                if max(tl2w, tl2h) <= 5 and r[4] < 0.1:
                    td.cu_mts_flag = 1; td.cu_mts_idx = int(rs.randint(0, 4))
                elif max(tl2w, tl2h) <= 5 and r[4] < 0.13 and (cbf & 0x10):
                    td.tr_skip_mask = 0x10
                task_l = task_c = None
                cu_isp = False
                if cu_intra or cu_ciip:
                    ux, uy, nw, nh = (x + tx) >> 2, (y + ty) >> 2, (1 << tl2w) >> 2, (1 << tl2h) >> 2
                    corner, a_abv, a_lft = avail(ux, uy, nw, nh)
                    task_l = capi.ITask()
                    task_l.x, task_l.y, task_l.log2_w, task_l.log2_h, task_l.kind = x + tx, y + ty, tl2w, tl2h, capi.IT_LUMA
                    task_l.flags = capi.IF_CORNER if corner else 0
                    task_l.avl_abv, task_l.avl_lft = a_abv, a_lft
                    task_c = capi.ITask()
                    task_c.x, task_c.y, task_c.log2_w, task_c.log2_h, task_c.kind = (x + tx) >> 1, (y + ty) >> 1, tl2w - 1, tl2h - 1, capi.IT_CHROMA
                    task_c.flags = task_l.flags
                    task_c.avl_abv, task_c.avl_lft = a_abv, a_lft          # 2-chroma-sample units = the same grid
                    if cu_ciip:
                        task_l.mode = 0; task_l.ciip_wt = int(ciip_wt[i])
                        task_c.mode = 0; task_c.ciip_wt = int(ciip_wt[i])
                        if tl2w <= 2:
                            task_c = None                                   # chroma of a 4-wide CIIP CU keeps the inter prediction
                    else:
                        td.cu_flags = 1 << 1                                # pred_mode_flag
                        q = rs.random_sample(4)
                        lmode = 0 if q[0] < 0.25 else 1 if q[0] < 0.35 else int(rs.randint(2, 67))
                        st.intra_mode = lmode
                        task_l.mode = lmode
                        isp_vertical = int(rs.randint(0, 2))
                        cu_isp = (isp_on and q[3] < ISP_FRAC and l2w <= 6 and l2h <= 6 and l2w + l2h >= 5
                                  # 64x2 partitions (a 64x8 CU split horizontally): the reference's result is undefined, the back-end follows the
                                  # specification (ovvc_record.c); kept out of the default workloads so that they stay what earlier rounds measured
                                  and (isp_64x2 or not (l2w == 6 and l2h == 3 and not isp_vertical)))
                        if cu_isp:
                            # ---- intra sub-partitions (recon_isp_subtree_v / _h, rcn_transform_tree.c:1087-1205): the luma of the CU in one
                            #      recorder call; the caller has marked the CU in the progress field before (vcl_transform_unit.c:1878), so the
                            #      partitions see each other; its chroma follows as a chroma transform unit (rcn_tu_c) ----
                            done[y >> 2:(y + (1 << l2h)) >> 2, x >> 2:(x + (1 << l2w)) >> 2] = True
                            l2p, n_pb, l2pred, n_pred = rec.isp_geometry(l2w, l2h, isp_vertical)
                            isp.x0, isp.y0, isp.log2_cb_w, isp.log2_cb_h, isp.vertical, isp.intra_mode = x, y, l2w, l2h, isp_vertical, lmode
                            isp.lfnst_flag = isp.lfnst_idx = 0; isp.mts_enabled = 1
                            nb_a, nb_l = ((2 << l2w) >> 2) + 1, ((2 << l2h) >> 2) + 1
                            for k in range(min(n_pred, 4)):
                                off = k << l2pred
                                px, py, off_y = (x + off, y, 0) if isp_vertical else (x, y + off, off)
                                row, col = (py >> 2) - 1 + (1 if off_y % 4 else 0), (px >> 2) - 1
                                ma = sum(1 << j for j in range(nb_a + 1)
                                         if row >= 0 and 0 <= (x >> 2) - 1 + j < w4 and done[row, (x >> 2) - 1 + j])
                                ml = sum(1 << j for j in range(nb_l + 1)
                                         if col >= 0 and 0 <= (y >> 2) - 1 + j < h4 and done[(y >> 2) - 1 + j, col])
                                isp.corner[k] = (ma & 1) | ((ml & 1) << 1)
                                isp.avl_abv[k], isp.avl_lft[k] = (ma >> 1).bit_length(), (ml >> 1).bit_length()
                            l2tw, l2th = (l2p, l2h) if isp_vertical else (l2w, l2p)
                            tbn = 1 << (l2tw + l2th)
                            isp_buf[:n_pb * tbn] = 0
                            cbfm = int(rs.randint(1, 1 << n_pb))
                            for i in range(n_pb):
                                isp.last_pos[i] = 0x0101; isp.sig_sb_map[i] = 1
                                if not (cbfm >> (n_pb - 1 - i)) & 1:
                                    continue
                                pbuf = isp_buf[i * tbn:(i + 1) * tbn]
                                if l2tw < 2 or l2th < 2:                     # 1xN / 2xN / Nx1 / Nx2: the whole block raster
                                    kk = min(tbn, int(rs.randint(1, 5)))
                                    pbuf[rs.randint(0, min(tbn, 8), size=kk)] = coef_pool[pool_pos:pool_pos + kk]
                                    pool_pos = (pool_pos + kk) % (len(coef_pool) - 4096)
                                else:                                        # sub-block storage, row of sub-blocks = min(32, width) * 4
                                    stride = min(32, 1 << l2tw)
                                    vals = coef_pool[pool_pos:pool_pos + 16].copy(); pool_pos = (pool_pos + 16) % (len(coef_pool) - 4096)
                                    vals[rs.random_sample(16) > 0.5] = 0
                                    pbuf[0:16] = vals
                                    m = 1
                                    if (1 << l2tw) >= 8 and rs.random_sample() < 0.4:
                                        vals = coef_pool[pool_pos:pool_pos + 16].copy(); pool_pos = (pool_pos + 16) % (len(coef_pool) - 4096)
                                        vals[rs.random_sample(16) > 0.3] = 0
                                        pbuf[16:32] = vals; m |= 2
                                    if (1 << l2th) >= 8 and rs.random_sample() < 0.4:
                                        vals = coef_pool[pool_pos:pool_pos + 16].copy(); pool_pos = (pool_pos + 16) % (len(coef_pool) - 4096)
                                        vals[rs.random_sample(16) > 0.3] = 0
                                        pbuf[4 * stride:4 * stride + 16] = vals; m |= 1 << 8
                                    if pbuf[0] == 0:
                                        pbuf[0] = 1
                                    isp.sig_sb_map[i] = m
                            isp.cbf_mask = cbfm
                            isp.coef = isp_buf.ctypes.data
                            st.qp_y = qp; st.qp_cb = qp - 1; st.qp_cr = qp - 1; st.qp_jcbcr = qp - 2
                            st.qp_y_skip = max(qp, 16); st.qp_cb_skip = st.qp_cr_skip = st.qp_jcbcr_skip = max(qp - 1, 16)
                            rec.isp_cu(st, isp)
                            n_isp += 1
                            # the CU's chroma: a chroma transform unit of its own (tree 2: position and size in chroma samples)
                            task_l = None
                            cbf &= ~0x10
                            td.tree = 2; td.x0, td.y0, td.log2_tb_w, td.log2_tb_h = x >> 1, y >> 1, tl2w - 1, tl2h - 1
                            td.cbf_mask = cbf; td.tr_skip_mask = 0; td.cu_mts_flag = 0
                            q[1] = 1.0                                       # (no MIP / MRL / BDPCM with ISP)
                        if q[1] < 0.08:                                      # matrix-based intra prediction
                            n_mip = 16 if (tl2w == 2 and tl2h == 2) else 8 if (tl2h == 2 or tl2w == 2 or (tl2h <= 3 and tl2w <= 3)) else 6
                            task_l.flags |= capi.IF_MIP | (capi.IF_MIP_TR if q[2] < 0.5 else 0)
                            task_l.mode = int(rs.randint(0, n_mip)); td.cu_flags |= 1 << 2; lmode = 0
                        elif q[1] < 0.14 and ((y + ty) & 127) and corner and a_abv:
                            task_l.mrl_idx = int(rs.randint(1, 3))           # multi reference line (never on the first CTU row)
                            if lmode < 2:
                                task_l.mode = lmode = int(rs.randint(2, 67))
                                st.intra_mode = lmode
                        elif q[1] < 0.17 and max(tl2w, tl2h) <= 5 and (cbf & 0x10):
                            vert = int(rs.randint(0, 2))                     # block DPCM: transform skip implied
                            task_l.flags |= capi.IF_BDPCM | (capi.IF_BDPCM_VER if vert else 0); task_l.mode = 0
                            td.cu_flags |= (1 << 8) | (vert << 10); td.tr_skip_mask = 0x10; td.cu_mts_flag = 0
                        # chroma: derived (= luma mode), one of the fixed modes, or a cross-component linear model
                        cq = rs.random_sample()
                        if cq < LM_FRAC:
                            cm = int(rs.randint(67, 70))
                            ext = min(1 << (tl2w - 1), 1 << (tl2h - 1))
                            if cm == 67:
                                task_c.avl_abv, task_c.avl_lft = int(a_abv > 0), int(a_lft > 0)
                            elif cm == 69:
                                task_c.avl_abv = min(a_abv, ((1 << (tl2w - 1)) + ext) >> 1); task_c.avl_lft = int(a_lft > 0)
                            else:
                                task_c.avl_lft = min(a_lft, ((1 << (tl2h - 1)) + ext) >> 1); task_c.avl_abv = int(a_abv > 0)
                        else:
                            cm = lmode if cq < 0.7 else (0, 1, 18, 50)[int(rs.randint(0, 4))]
                        task_c.mode = cm
                st.qp_y = qp; st.qp_cb = qp - 1; st.qp_cr = qp - 1; st.qp_jcbcr = qp - 2
                st.qp_y_skip = max(qp, 16); st.qp_cb_skip = st.qp_cr_skip = st.qp_jcbcr_skip = max(qp - 1, 16)
                for comp in range(3):
                    is_l = comp == 2
                    used = (cbf & 0x10) if is_l else ((comp == 0) if (cbf & 0x8) else (cbf & (0x1 if comp else 0x2)))
                    if not used:
                        td.coef[comp] = None
                        continue
                    cl2w = tl2w if is_l else tl2w - 1
                    cl2h = tl2h if is_l else tl2h - 1
                    buf = bufs[comp]
                    if is_l and (td.tr_skip_mask & 0x10):
                        nn = 1 << (cl2w + cl2h)          # TS residual coding: raster, final values
                        buf[:nn] = 0
                        k = min(nn, int(rs.randint(1, 9)))
                        buf[rs.randint(0, nn, size=k)] = coef_pool[pool_pos:pool_pos + k] >> 1
                        pool_pos = (pool_pos + k) % (len(coef_pool) - 4096)
                        td.sig_sb_map[comp] = 1; td.last_pos[comp] = 0x0101
                    elif cl2w < 2 or cl2h < 2:
                        nn = 1 << (cl2w + cl2h)
                        buf[:nn] = 0
                        buf[0] = coef_pool[pool_pos]; pool_pos += 1
                        td.sig_sb_map[comp] = 1; td.last_pos[comp] = 0
                    else:
                        cw, ch = min(32, 1 << cl2w), min(32, 1 << cl2h)
                        buf[:cw * ch] = 0
                        nx, ny = cw // 4, ch // 4
                        u = rs.random_sample()
                        if u < 0.25:                       # DC only
                            buf[0] = coef_pool[pool_pos] or 3; pool_pos += 1
                            td.sig_sb_map[comp] = 0; td.last_pos[comp] = 0
                        else:
                            # low-frequency cluster: SBs inside a small top-left rectangle
                            lx = min(nx, 1 + int(rs.geometric(0.55))); ly = min(ny, 1 + int(rs.geometric(0.55)))
                            m = 0
                            for sy in range(ly):
                                for sx in range(lx):
                                    if (sx or sy) and rs.random_sample() < 0.35:
                                        continue
                                    m |= 1 << (sy * 8 + sx)
                                    o = sy * 4 * cw + sx * 16
                                    dens = 0.7 if (sx == 0 and sy == 0) else 0.3
                                    vals = coef_pool[pool_pos:pool_pos + 16].copy(); pool_pos += 16
                                    vals[rs.random_sample(16) > dens] = 0
                                    buf[o:o + 16] = vals
                            if buf[0] == 0:
                                buf[0] = 1
                            td.sig_sb_map[comp] = m; td.last_pos[comp] = 0x0101
                        pool_pos %= (len(coef_pool) - 4096)
                    td.coef[comp] = buf.ctypes.data
                if task_l is not None or task_c is not None:
                    rec.tu_intra(st, td, task_l, task_c)
                else:
                    rec.tu(st, td)
                n_tu += 1
        done[y >> 2:(y + (1 << l2h)) >> 2, x >> 2:(x + (1 << l2w)) >> 2] = True

    cmds, classes = rec.tb_cmds_split()
    n_luma = classes[0] + classes[1]
    wl = Workload(w, h, seed, refs, pocs, cus, rec.mc_units(), cmds, rec.coefs(), n_luma_cmds=n_luma, tb_classes=classes)
    wl.mcx_units, wl.aff_units, wl.aff_side, wl.ciip_units = rec.mcx_units(), rec.aff_units(), rec.aff_side(), rec.ciip_units()
    if (len(wl.ciip_units) or (mode == CIIP).any()) and not dev_intra:
        wl.intra = random_picture(rs, w, h)
    if dev_intra:
        wl.itasks = rec.itasks()
    if lmcs_on:
        wl.lmcs = _lmcs_tables(rs)
        wl.lmcs_regions = rec.lmcs_regions()
    wl.dbf_ctus = make_dbf_ctus(rs, w, h, cus)
    wl.dbf_planes, wl.dbf_edges = record_dbf(rec, wl.dbf_ctus)
    if calllog:
        wl.calllog = rec.take_calllog()
    wl.sao_params = make_sao_params(rs, w, h)
    wl.alf = make_alf(rs, w, h)
    u, ux, ua = wl.mc_units, wl.mcx_units, wl.aff_units
    area = lambda a: a["w"].astype(np.int64) * a["h"]
    nref = lambda a: np.where(a["dir"] == 3, 2, 1)
    tot = area(u).sum() + area(ux).sum() + area(ua).sum()
    wl.stats = {
        "n_cu": int(n), "n_tu": int(n_tu), "n_isp_cus": int(n_isp), "n_mc_units": int(len(u)), "n_mcx_units": int(len(ux)),
        "n_aff_units": int(len(ua)), "n_ciip_units": int(len(wl.ciip_units)), "n_tb_cmds": int(len(wl.tb_cmds)),
        "n_luma_cmds": int(n_luma), "n_lmcs_regions": 0 if wl.lmcs_regions is None else int(len(wl.lmcs_regions)),
        "cu_modes": {k: int((mode == v).sum()) for k, v in (("plain", PLAIN), ("bdof", BDOF), ("dmvr", DMVR), ("affine", AFFINE), ("gpm", GPM), ("ciip", CIIP))
                     + ((("intra", INTRA),) if dev_intra else ())},
        "n_itasks": 0 if wl.itasks is None else int(len(wl.itasks)),
        "n_ilevels": 0 if wl.itasks is None or not len(wl.itasks) else int(wl.itasks["level"].max()),
        "coef_bytes": int(wl.coefs.nbytes),
        "cmd_bytes": int(u.nbytes + ux.nbytes + ua.nbytes + wl.aff_side.nbytes + wl.tb_cmds.nbytes),
        # mean reference samples fetched per output sample (block window not counted), SURVEY 8d
        "r_bar": float(((area(u) * nref(u)).sum() + 2 * area(ux).sum() + (area(ua) * nref(ua)).sum()) / max(1, tot)),
    }
    rec.close()
    return wl


# --------------------------------------------------------------------------------------------
# in-loop filter side information (deblocking edge planes, SAO / ALF parameters)
# --------------------------------------------------------------------------------------------
def _pack_bits(win):
    """win[bit, entry] (bool) -> uint64[entry]: one mask word per entry."""
    sh = np.arange(win.shape[0], dtype=np.uint64)[:, None]
    return (win.astype(np.uint64) << sh).sum(axis=0, dtype=np.uint64)


def make_dbf_ctus(rs, w, h, cus, ctu=128):
    """What df.rcn_dbf_ctu would receive for a synthetic partition, CTU by CTU in decoding order: capi.DBF_CTU_DTYPE records
    (ovhip_dbf_ctu = the CTU's struct DBFInfo arrays after dbf_load_info()'s neighbour rotation, drv_lines.c:618-761, same
    element layout -- see ovvc_record_dbf.c for the indexing).  Rules of the synthetic picture: CU boundaries and the 64-sample
    transform grid are edges; bS 2 around a few "intra" CUs, bS 1 where either side has (random) residual / motion difference, per
    component for chroma; one QP per CU.  Columns right of / rows below the CTU are still empty when the slot runs, the left /
    above neighbours' last units are there.  The edge lists, filter lengths, average QPs and the chroma "large" decisions come
    out of ovhip_rec_dbf_ctu -- nothing of that is derived here."""
    w4, h4 = (w + 3) // 4, (h + 3) // 4
    cu_id = np.zeros((h4, w4), np.int32)
    n = len(cus)
    for i in range(n):
        x, y, l2w, l2h = (int(v) for v in cus[i])
        cu_id[y >> 2:(y + (1 << l2h)) >> 2, x >> 2:(x + (1 << l2w)) >> 2] = i
    qp_cu = rs.randint(22, 46, size=n).astype(np.int32)
    flag1 = rs.random_sample(n) < 0.55           # residual / motion difference -> bS 1
    flag1c = rs.random_sample(n) < 0.35
    intra = rs.random_sample(n) < 0.04           # bS 2
    PAD, FAR = 8, 56                             # units of margin before / after the picture in the padded arrays

    def padded(m, dtype=bool):
        out = np.zeros((h4 + PAD + FAR, w4 + PAD + FAR), dtype)
        out[PAD:PAD + h4, PAD:PAD + w4] = m
        return out

    maps = {}
    for d, name in ((0, "ver"), (1, "hor")):
        ids = cu_id if d == 0 else cu_id.T
        pos = np.arange(ids.shape[1])[None, :] + np.zeros_like(ids)
        prev = np.concatenate([ids[:, :1], ids[:, :-1]], axis=1)
        bound = (ids != prev) | ((pos % 16) == 0)
        edge = bound.copy(); edge[:, 0] = False                    # the picture border carries no strength
        either = lambda f: f[ids] | f[prev]
        planes = {"bound": bound, "bs2": edge & either(intra), "bs1": edge & either(flag1),
                  "bs1cb": edge & either(flag1c), "bs1cr": edge & either(~flag1c & flag1)}
        for k, m in planes.items():
            maps[k + "_" + name] = padded(m if d == 0 else m.T)
    maps["bound_ver"][PAD:PAD + h4, PAD + w4] = True               # the picture's right / bottom border bounds its last blocks
    maps["bound_hor"][PAD + h4, PAD:PAD + w4] = True
    qp = {"qp_y": padded(qp_cu[cu_id], np.uint8), "qp_cb": padded(np.maximum(qp_cu[cu_id] - 1, 0), np.uint8),
          "qp_cr": padded(np.maximum(qp_cu[cu_id] - 2, 0), np.uint8)}

    nb = ctu >> 2
    ncx, ncy = (w + ctu - 1) // ctu, (h + ctu - 1) // ctu
    out = np.zeros(ncx * ncy, capi.DBF_CTU_DTYPE)
    for cy in range(ncy):
        for cx in range(ncx):
            c = out[cy * ncx + cx]
            ux0, uy0 = cx * nb, cy * nb
            cw, ch = min(ctu, w - cx * ctu), min(ctu, h - cy * ctu)
            nw, nh = cw >> 2, ch >> 2
            # masks of vertical edges: entry 8 + i = unit column i, bit j = unit row j
            rows = slice(PAD + uy0, PAD + uy0 + nh)
            ver = lambda m, n_ent, first: _pack_bits(np.where(np.arange(first, first + n_ent)[None, :] <= nw,
                                                              m[rows, PAD + ux0 + first:PAD + ux0 + first + n_ent], False))
            # masks of horizontal edges: entry 8 + i = unit row i, bit k = unit column k - 2 (two units of the left neighbour)
            cols = slice(PAD + ux0 - 2, PAD + ux0 + nw)
            hor = lambda m, n_ent, first: _pack_bits(np.where(np.arange(first, first + n_ent)[:, None] <= nh,
                                                              m[PAD + uy0 + first:PAD + uy0 + first + n_ent, cols], False).T)
            for dst, src in (("ctb_bound_ver", "bound_ver"), ("ctb_bound_ver_c", "bound_ver")):
                c[dst] = ver(maps[src], 49, -8)
            for dst, src in (("ctb_bound_hor", "bound_hor"), ("ctb_bound_hor_c", "bound_hor")):
                c[dst] = hor(maps[src], 49, -8)
            for dst, src in (("bs2", "bs2"), ("bs2c", "bs2"), ("bs1", "bs1"), ("bs1cb", "bs1cb"), ("bs1cr", "bs1cr")):
                c[dst + "_ver"] = ver(maps[src + "_ver"], 33, 0)
                c[dst + "_hor"] = hor(maps[src + "_hor"], 33, 0)
            for k, m in qp.items():                                  # 33 rows (the one above first) x 34 columns (two to the left)
                c[k] = m[PAD + uy0 - 1:PAD + uy0 + 32, PAD + ux0 - 2:PAD + ux0 + 32].reshape(-1)
            c["log2_ctu_s"] = ctu.bit_length() - 1
            c["last_x"], c["last_y"] = cx == ncx - 1, cy == ncy - 1
            c["ctu_lft"], c["ctu_abv"] = cx > 0, cy > 0
            c["ctu_w"], c["ctu_h"], c["ctb_x"], c["ctb_y"] = cw, ch, cx, cy
    return out


def record_dbf(rec, ctus):
    """The CTUs through ovhip_rec_dbf_ctu: (dense edge planes for the oracle, [vertical, horizontal] edge lists as emitted)."""
    for c in ctus:
        rec.dbf_ctu(c.tobytes())
    return rec.dbf_planes(), [rec.dbf_edges(0)[0], rec.dbf_edges(1)[0]]


def make_sao_params(rs, w, h, ctu=128):
    n = ((w + ctu - 1) // ctu) * ((h + ctu - 1) // ctu)
    p = np.zeros(n, capi.SAO_CTU_DTYPE)
    t = rs.randint(0, 4, size=(n, 3))
    p["type"] = np.where(t >= 2, t - 1, 0)                # 50 % off, 25 % band, 25 % edge
    p["band_position"] = rs.randint(0, 32, size=(n, 3))
    p["eo_class"] = rs.randint(0, 4, size=(n, 3))
    off = rs.randint(0, 8, size=(n, 3, 5)).astype(np.int16)
    band = (p["type"] == 1)[:, :, None]
    sign = np.where(rs.random_sample((n, 3, 5)) < 0.5, -1, 1)
    edge_sign = np.array([1, 1, 0, -1, -1], np.int16)[None, None, :]
    p["offset_val"] = np.where(band, off * sign, off * edge_sign)
    return p


def make_alf(rs, w, h, ctu=128):
    n = ((w + ctu - 1) // ctu) * ((h + ctu - 1) // ctu)
    ctus = np.zeros(n, capi.ALF_CTU_DTYPE)
    on = rs.random_sample((n, 3)) < 0.75
    ctus["flags"] = (on[:, 0] * 4 + on[:, 1] * 2 + on[:, 2]).astype(np.uint8)
    ctus["luma_set"] = rs.randint(0, 24, size=n)
    ctus["cb_alt"] = rs.randint(0, 8, size=n)
    ctus["cr_alt"] = rs.randint(0, 8, size=n)
    ctus["cc_cb_idx"] = np.where(rs.random_sample(n) < 0.25, rs.randint(1, 5, size=n), 0)
    ctus["cc_cr_idx"] = np.where(rs.random_sample(n) < 0.25, rs.randint(1, 5, size=n), 0)
    clip_lut = np.array([1024, 128, 32, 8], np.int16)
    luma_coeff = rs.randint(-12, 13, size=(24, 4 * 25, 13)).astype(np.int16)
    luma_coeff[:, :, 12] = 128
    # Non-linear clipping only exists in APS filter sets whose alf_luma_clip_flag is set (rcn_alf.c:196-240); the
    # 16 fixed sets (luma_set < 16) and the other APS sets carry clip = 1 << bitdepth, i.e. no clipping.
    luma_clip = clip_lut[rs.randint(0, 4, size=(24, 4 * 25, 13))]
    aps_clip_flag = rs.random_sample(8) < 0.5
    luma_clip[:16] = 1024
    luma_clip[16:][~aps_clip_flag] = 1024
    luma_clip[:, :, 12] = 1024
    chroma_coeff = rs.randint(-12, 13, size=(8, 7)).astype(np.int16)
    chroma_coeff[:, 6] = 128
    chroma_clip = clip_lut[rs.randint(0, 4, size=(8, 7))]
    mag = rs.randint(0, 6, size=(2, 4, 8))
    cc = np.where(mag == 0, 0, 1 << np.maximum(mag - 1, 0)) * np.where(rs.random_sample((2, 4, 8)) < 0.5, -1, 1)
    cc[:, :, 7] = 0
    return {"ctus": ctus, "luma_coeff": luma_coeff.reshape(24, -1), "luma_clip": luma_clip.reshape(24, -1).astype(np.int16),
            "chroma_coeff": chroma_coeff, "chroma_clip": chroma_clip.astype(np.int16), "cc_coeff": cc.astype(np.int16)}
