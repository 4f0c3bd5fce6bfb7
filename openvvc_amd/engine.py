"""Python host harness over the device engine (ovhip_ctx_* / ovhip_*_launch).

Plumbing only: device buffers, picture upload/download and stage launches all go through the
C ABI in libovvc_hip.so.  There is no Python or CPU implementation of any stage here."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class EngineError(RuntimeError):
    pass


class Context:
    def __init__(self, device: int = 0, stream: int | None = None):
        self.lib = capi.load()
        h = C.c_void_p()
        r = self.lib.ovhip_ctx_create(C.byref(h), device, C.c_void_p(stream) if stream else None)
        if r != 0:
            raise EngineError(f"ovhip_ctx_create(device={device}) failed: {r} (no HIP device? the engine has no CPU fallback)")
        self.h = h
        self._bufs = []

    def _chk(self, r, what):
        if r != 0:
            raise EngineError(f"{what}: {r}: {self.lib.ovhip_last_error(self.h).decode()}")

    def sync(self):
        self._chk(self.lib.ovhip_ctx_sync(self.h), "sync")

    @property
    def stream(self) -> int:
        return self.lib.ovhip_ctx_stream(self.h) or 0

    def close(self):
        if self.h:
            self.lib.ovhip_ctx_destroy(self.h)
            self.h = None

    # ---- raw buffers ----
    def upload(self, arr: np.ndarray) -> "DevBuf":
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._chk(self.lib.ovhip_malloc(self.h, arr.nbytes, C.byref(p)), "malloc")
        if arr.nbytes:
            self._chk(self.lib.ovhip_h2d(self.h, p, arr.ctypes.data, arr.nbytes), "h2d")
            self.sync()
        return DevBuf(self, p, arr.nbytes, len(arr))

    # ---- pictures ----
    def new_pic(self, w: int, h: int) -> "DevPic":
        pic = capi.Pic()
        self._chk(self.lib.ovhip_pic_alloc(self.h, w, h, C.byref(pic)), "pic_alloc")
        return DevPic(self, pic, owns=True)

    def alloc(self, nbytes: int) -> "DevBuf":
        p = C.c_void_p()
        self._chk(self.lib.ovhip_malloc(self.h, max(int(nbytes), 16), C.byref(p)), "malloc")
        return DevBuf(self, p, int(nbytes), int(nbytes))

    def upload_pic(self, y, cb, cr) -> "DevPic":
        h, w = y.shape
        p = self.new_pic(w, h)
        p.upload(y, cb, cr)
        return p

    def fork(self, k: int):
        """Route the following launches to side stream k (1..3); 0 = back to the main stream."""
        self._chk(self.lib.ovhip_ctx_fork(self.h, k), "ctx_fork")

    def join(self):
        self._chk(self.lib.ovhip_ctx_join(self.h), "ctx_join")

    # ---- stages ----
    def itx(self, dst: "DevPic", cmds: "DevBuf", coefs: "DevBuf", n: int | None = None, first: int = 0,
            lmcs_scales: "DevBuf | None" = None):
        """n commands starting at command `first`; lmcs_scales: device-derived chroma scales (lmcs_scale)."""
        n = cmds.count - first if n is None else n
        ptr = C.c_void_p(cmds.ptr.value + first * capi.TB_CMD_DTYPE.itemsize) if first else cmds.ptr
        self._chk(self.lib.ovhip_itx_launch(self.h, C.byref(dst.s), ptr, n, coefs.ptr,
                                            lmcs_scales.ptr if lmcs_scales else None), "itx_launch")

    def itx_classes(self, dst: "DevPic", cmds: "DevBuf", coefs: "DevBuf", first: int, n_large: int, n_small: int,
                    lmcs_scales: "DevBuf | None" = None):
        """Commands [first, first + n_large) of any size, then n_small commands all <= 16x16 (recorder's split order)."""
        ptr = C.c_void_p(cmds.ptr.value + first * capi.TB_CMD_DTYPE.itemsize) if first else cmds.ptr
        self._chk(self.lib.ovhip_itx_launch_classes(self.h, C.byref(dst.s), ptr, n_large, n_small, coefs.ptr,
                                                    lmcs_scales.ptr if lmcs_scales else None), "itx_launch_classes")

    def itx_chroma_lmcs(self, dst: "DevPic", cmds: "DevBuf", coefs: "DevBuf", first: int, n_large: int, n_small: int,
                        lmcs_scales: "DevBuf | None", bwd_lut: "DevBuf"):
        """The chroma commands plus the inverse LMCS mapping of luma in the same launch (ovhip_itx_launch_chroma_lmcs)."""
        ptr = C.c_void_p(cmds.ptr.value + first * capi.TB_CMD_DTYPE.itemsize) if first else cmds.ptr
        self._chk(self.lib.ovhip_itx_launch_chroma_lmcs(self.h, C.byref(dst.s), ptr, n_large, n_small, coefs.ptr,
                                                        lmcs_scales.ptr if lmcs_scales else None, bwd_lut.ptr),
                  "itx_launch_chroma_lmcs")

    def lmcs_scale(self, pic: "DevPic", regions: "DevBuf", luts: "capi.LmcsLuts", scales: "DevBuf", n: int | None = None):
        n = regions.count if n is None else n
        self._chk(self.lib.ovhip_lmcs_scale_launch(self.h, C.byref(pic.s), regions.ptr, n, C.byref(luts), scales.ptr),
                  "lmcs_scale_launch")

    def lmcs_inverse(self, pic: "DevPic", bwd_lut: "DevBuf"):
        self._chk(self.lib.ovhip_lmcs_inverse_launch(self.h, C.byref(pic.s), bwd_lut.ptr), "lmcs_inverse_launch")

    def intra_level(self, pic: "DevPic", res: "DevPic", tasks: "DevBuf", first: int, n: int, regions: "DevBuf | None" = None,
                    luts=None, scales: "DevBuf | None" = None, log2_ctu: int = 7, geom: int = 0x12):
        """One level of the ordered pass: tasks [first, first + n) of the device task list."""
        ptr = C.c_void_p(tasks.ptr.value + first * 32)
        self._chk(self.lib.ovhip_intra_level_launch(self.h, C.byref(pic.s), C.byref(res.s), ptr, n, regions.ptr if regions else None,
                                                    C.byref(luts) if luts is not None else None, scales.ptr if scales else None, log2_ctu, geom),
                  "intra_level_launch")

    def intra_ctu(self, pic: "DevPic", res: "DevPic", tasks: "DevBuf", ctus: "DevBuf", n_ctus: int, sync: "DevBuf", epoch: int,
                  regions: "DevBuf | None" = None, luts=None, scales: "DevBuf | None" = None, log2_ctu: int = 7):
        """The whole ordered pass in one launch (tasks / ctus: device copies of Recorder.itasks_by_ctu()).  sync: a zeroed
        device buffer of ovhip_intra_sync_words() uint32; epoch: fresh and != 0 for every launch on the same sync."""
        self._chk(self.lib.ovhip_intra_ctu_launch(self.h, C.byref(pic.s), C.byref(res.s), tasks.ptr, ctus.ptr, n_ctus,
                                                  regions.ptr if regions else None, C.byref(luts) if luts is not None else None,
                                                  scales.ptr if scales else None, log2_ctu, sync.ptr, epoch, None), "intra_ctu_launch")

    def tmvp_cells(self, units: "DevBuf", n: int, refined: "DevBuf", log2_ctu: int, nb_ctb_w: int) -> np.ndarray:
        out = self.alloc(4 * n * capi.TMVP_CELL_DTYPE.itemsize)
        self._chk(self.lib.ovhip_tmvp_cells_launch(self.h, units.ptr, n, refined.ptr, log2_ctu, nb_ctb_w, out.ptr), "tmvp_cells")
        self.sync()
        r = out.download(np.uint8).view(capi.TMVP_CELL_DTYPE).copy()
        out.free()
        return r

    def dbf(self, pic: "DevPic", planes: "DevDbfPlanes"):
        self._chk(self.lib.ovhip_dbf_launch(self.h, C.byref(pic.s), C.byref(planes.s)), "dbf_launch")

    def dbf_edges(self, pic: "DevPic", edges_v: "DevBuf", edges_h: "DevBuf", beta_offset: int = 0, tc_offset: int = 0):
        """Deblocking driven by the compact edge lists (capi.dbf_compact)."""
        self._chk(self.lib.ovhip_dbf_launch_edges(self.h, C.byref(pic.s), edges_v.ptr, edges_v.count, edges_h.ptr,
                                                  edges_h.count, beta_offset, tc_offset), "dbf_launch_edges")

    def sao(self, dst: "DevPic", src: "DevPic", params: "DevBuf", log2_ctu: int = 7):
        self._chk(self.lib.ovhip_sao_launch(self.h, C.byref(dst.s), C.byref(src.s), params.ptr, log2_ctu), "sao_launch")

    def alf(self, dst: "DevPic", src: "DevPic", alf: "DevAlf"):
        self._chk(self.lib.ovhip_alf_launch(self.h, C.byref(dst.s), C.byref(src.s), C.byref(alf.s)), "alf_launch")

    def mc(self, dst: "DevPic", refs: list, units: "DevBuf", lmcs_fwd: "DevBuf | None" = None, n: int | None = None,
           intra: "DevPic | None" = None):
        """intra: picture with the planar prediction of the CIIP CUs whose blend is fused into the units (aux != 0)."""
        n = units.count if n is None else n
        arr = (capi.Pic * len(refs))(*[r.s for r in refs])
        self._chk(self.lib.ovhip_mc_launch(self.h, C.byref(dst.s), arr, len(refs), units.ptr, n,
                                           lmcs_fwd.ptr if lmcs_fwd else None, C.byref(intra.s) if intra else None), "mc_launch")

    def ciip(self, dst: "DevPic", intra: "DevPic", units: "DevBuf", n: int | None = None):
        n = units.count if n is None else n
        self._chk(self.lib.ovhip_ciip_launch(self.h, C.byref(dst.s), C.byref(intra.s), units.ptr, n), "ciip_launch")

    def mca(self, dst: "DevPic", refs: list, units: "DevBuf", side: "DevBuf", lmcs_fwd: "DevBuf | None" = None,
            n: int | None = None):
        """Affine (+PROF) units; side: device copy of the recorder's affine side arena."""
        n = units.count if n is None else n
        arr = (capi.Pic * len(refs))(*[r.s for r in refs])
        self._chk(self.lib.ovhip_mca_launch(self.h, C.byref(dst.s), arr, len(refs), units.ptr, n, side.ptr,
                                            lmcs_fwd.ptr if lmcs_fwd else None), "mca_launch")

    def mcxa(self, dst: "DevPic", refs: list, xunits: "DevBuf", aunits: "DevBuf", side: "DevBuf",
             lmcs_fwd: "DevBuf | None" = None, mv_out: "DevBuf | None" = None):
        """BDOF / DMVR units and affine units in one launch (ovhip_mcxa_launch)."""
        arr = (capi.Pic * len(refs))(*[r.s for r in refs])
        self._chk(self.lib.ovhip_mcxa_launch(self.h, C.byref(dst.s), arr, len(refs), xunits.ptr, xunits.count,
                                             mv_out.ptr if mv_out else None, aunits.ptr, aunits.count, side.ptr,
                                             lmcs_fwd.ptr if lmcs_fwd else None), "mcxa_launch")

    def mcx(self, dst: "DevPic", refs: list, units: "DevBuf", lmcs_fwd: "DevBuf | None" = None,
            mv_out: "DevBuf | None" = None, n: int | None = None):
        """BDOF / DMVR units; mv_out: device int32[n][4] receiving the refined motion vectors."""
        n = units.count if n is None else n
        arr = (capi.Pic * len(refs))(*[r.s for r in refs])
        self._chk(self.lib.ovhip_mcx_launch(self.h, C.byref(dst.s), arr, len(refs), units.ptr, n,
                                            lmcs_fwd.ptr if lmcs_fwd else None,
                                            mv_out.ptr if mv_out else None), "mcx_launch")


class DevDbfPlanes:
    """Deblocking edge planes resident on the device (ovhip_dbf_planes with device pointers)."""

    def __init__(self, ctx: "Context", planes: dict):
        self.bufs = {k: ctx.upload(np.ascontiguousarray(planes[k], dtype=np.uint16).ravel()) for k in capi.DBF_PLANE_NAMES}
        self.s = capi.DbfPlanes(*[self.bufs[k].ptr for k in capi.DBF_PLANE_NAMES], planes["w4"], planes["h4"],
                                planes["beta_offset"], planes["tc_offset"])
        self.nbytes = sum(b.nbytes for b in self.bufs.values())

    def free(self):
        for b in self.bufs.values():
            b.free()


class DevAlf:
    """ALF parameter tables resident on the device (ovhip_alf_pic with device pointers).
    `alf`: dict with ctus (ALF_CTU_DTYPE), luma_coeff/luma_clip [24,1300], chroma_coeff/chroma_clip [8,7], cc_coeff [2,4,8]."""

    def __init__(self, ctx: "Context", alf: dict, w: int, h: int, log2_ctu: int = 7):
        self.bufs = {k: ctx.upload(np.ascontiguousarray(alf[k], dtype=dt).ravel()) for k, dt in capi.ALF_TABLES}
        self.scratch = ctx.upload(np.zeros(((w + 3) // 4) * ((h + 3) // 4), np.uint8))
        self.s = capi.AlfPic(*[self.bufs[k].ptr for k, _ in capi.ALF_TABLES], self.scratch.ptr, log2_ctu)
        self.nbytes = sum(b.nbytes for b in self.bufs.values())

    def free(self):
        for b in list(self.bufs.values()) + [self.scratch]:
            b.free()


class DevBuf:
    def __init__(self, ctx: Context, ptr, nbytes: int, count: int):
        self.ctx, self.ptr, self.nbytes, self.count = ctx, ptr, nbytes, count

    def free(self):
        if self.ptr:
            self.ctx.lib.ovhip_free(self.ctx.h, self.ptr)
            self.ptr = None

    def download(self, dtype=np.uint8) -> np.ndarray:
        """Synchronous device -> host copy of the whole buffer."""
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype)
        if self.nbytes:
            self.ctx._chk(self.ctx.lib.ovhip_d2h(self.ctx.h, out.ctypes.data, self.ptr, self.nbytes), "d2h")
        return out


class DevPic:
    def __init__(self, ctx: Context, s: capi.Pic, owns: bool):
        self.ctx, self.s, self.owns = ctx, s, owns

    @property
    def w(self):
        return self.s.w

    @property
    def h(self):
        return self.s.h

    def upload(self, y, cb, cr):
        y, cb, cr = (np.ascontiguousarray(a, dtype=np.uint16) for a in (y, cb, cr))
        self.ctx._chk(self.ctx.lib.ovhip_pic_upload(self.ctx.h, C.byref(self.s), y.ctypes.data, cb.ctypes.data,
                                                    cr.ctypes.data, y.shape[1], cb.shape[1]), "pic_upload")

    def download(self):
        y = np.empty((self.h, self.w), np.uint16)
        cb = np.empty((self.h // 2, self.w // 2), np.uint16)
        cr = np.empty_like(cb)
        self.ctx._chk(self.ctx.lib.ovhip_pic_download(self.ctx.h, C.byref(self.s), y.ctypes.data, cb.ctypes.data,
                                                      cr.ctypes.data, self.w, self.w // 2), "pic_download")
        return y, cb, cr

    def output(self, window=(0, 0, 0, 0)) -> np.ndarray:
        """The cropped frame in the byte layout dectest.c:372-409 writes (uint16 LE: Y rows, Cb rows, Cr rows), packed on the
        device and fetched with one D2H.  window = (lft, rgt, abv, blw) in chroma sample units."""
        win = capi.Window(*window)
        n = self.ctx.lib.ovhip_output_bytes(self.w, self.h, C.byref(win))
        if not n:
            raise EngineError("output window leaves nothing")
        out = np.empty(n // 2, np.uint16)
        self.ctx._chk(self.ctx.lib.ovhip_pic_output(self.ctx.h, C.byref(self.s), C.byref(win), out.ctypes.data), "pic_output")
        return out

    def digest(self, window=(0, 0, 0, 0)) -> bytes:
        """the picture's fingerprint: an MD5 tree over the cropped frame computed on the device (include/ovvc_hip.h, "Digest")"""
        win = capi.Window(*window)
        out = (C.c_uint8 * 16)()
        self.ctx._chk(self.ctx.lib.ovhip_pic_digest(self.ctx.h, C.byref(self.s), C.byref(win), out), "pic_digest")
        return bytes(out)

    def row_digests(self, window=(0, 0, 0, 0)) -> np.ndarray:
        win = capi.Window(*window)
        rows = self.ctx.lib.ovhip_output_rows(self.w, self.h, C.byref(win))
        buf = self.ctx.alloc(rows * 16)
        self.ctx._chk(self.ctx.lib.ovhip_output_row_md5_launch(self.ctx.h, C.byref(self.s), C.byref(win), buf.ptr), "row_md5")
        self.ctx.sync()
        d = buf.download(np.uint8).reshape(rows, 16).copy()
        buf.free()
        return d

    def band(self, y0: int, h: int) -> "DevPic":
        """A view of rows [y0, y0+h) (luma) as its own picture (no copy)."""
        s = capi.Pic(self.s.y + y0 * self.s.stride_y * 2, self.s.cb + (y0 // 2) * self.s.stride_c * 2,
                     self.s.cr + (y0 // 2) * self.s.stride_c * 2, self.s.w, h, self.s.stride_y, self.s.stride_c)
        return DevPic(self.ctx, s, owns=False)

    def free(self):
        if self.owns and self.s.y:
            self.ctx.lib.ovhip_pic_free(self.ctx.h, C.byref(self.s))


class ResidentPicture:
    """A recorded picture resident in HBM: reference pictures, command buffers, coefficient arena,
    in-loop filter side information and two picture buffers.  `decode()` enqueues every stage of
    the rcn path in the order the reference executes them (prediction -> residual -> inverse luma
    mapping -> deblocking -> SAO -> ALF/CC-ALF), each as frame-wide launches on the context stream.
    The reconstructed / deblocked picture lives in `dst`, SAO writes `tmp`, ALF writes the final
    samples back to `dst`.

    Launch order inside the two composite stages:
      "mc"  = plain / GPM units, BDOF / DMVR units (+ refined-MV write-back), affine units, CIIP blend
      "itx" = luma TBs, LMCS chroma-scale derivation (reads the luma just reconstructed), chroma TBs,
              inverse luma mapping"""

    STAGES = ("mc", "itx", "dbf", "sao", "alf")
    SUBSTAGES = ("mcp", "mcx", "mca", "ciip", "itx_l", "lmcs_scale", "itx_c", "lmcs_inv", "dbf", "sao", "alf")
    _GROUPS = {"mc": ("mcp", "mcx", "mca", "ciip"), "itx": ("itx_l", "lmcs_scale", "itx_c", "lmcs_inv")}

    def __init__(self, ctx: Context, wl, log2_ctu: int = 7, overlap: bool = False):
        self.ctx, self.wl, self.log2_ctu, self.overlap = ctx, wl, log2_ctu, overlap
        self.refs = [ctx.upload_pic(*r) for r in wl.refs]
        self.dst = ctx.new_pic(wl.w, wl.h)
        self.tmp = ctx.new_pic(wl.w, wl.h)
        self.mc_units = ctx.upload(wl.mc_units)
        self.tb_cmds = ctx.upload(wl.tb_cmds)
        self.coefs = ctx.upload(wl.coefs)
        self.dbf_planes = DevDbfPlanes(ctx, wl.dbf_planes)
        self.dbf_v = ctx.upload(wl.dbf_edges[0])              # the lists ovhip_rec_dbf_ctu emitted CTU by CTU
        self.dbf_h = ctx.upload(wl.dbf_edges[1])
        self.sao_params = ctx.upload(wl.sao_params)
        self.alf = DevAlf(ctx, wl.alf, wl.w, wl.h, log2_ctu)
        self.bufs = [self.mc_units, self.tb_cmds, self.coefs, self.sao_params, self.dbf_v, self.dbf_h]
        up = lambda a: self._keep(ctx.upload(a)) if a is not None and len(a) else None
        self.mcx_units = up(wl.mcx_units)
        self.mv_out = self._keep(ctx.alloc(16 * len(wl.mcx_units))) if self.mcx_units else None
        self.aff_units, self.aff_side = up(wl.aff_units), up(wl.aff_side)
        self.ciip_units = up(wl.ciip_units)
        self.intra = ctx.upload_pic(*wl.intra) if wl.intra is not None else None
        self.lmcs = wl.lmcs
        self.lmcs_fwd = up(wl.lmcs_fwd)
        self.lmcs_bwd = up(wl.lmcs_bwd)
        self.lmcs_regions = up(wl.lmcs_regions)
        self.lmcs_scales = self._keep(ctx.alloc(2 * len(wl.lmcs_regions))) if self.lmcs_regions else None
        self.n_luma = wl.n_luma_cmds

    def _keep(self, b):
        self.bufs.append(b)
        return b

    def run_stage(self, name: str, on_sub=None):
        """on_sub(sub_stage_name, "begin" | "end"): optional hook around every launch that goes to the main stream."""
        c = self.ctx
        if name == "mc" and self.overlap:
            # plain, refined and affine units write disjoint samples: the three kernels may share the GPU, the CIIP
            # blend follows once all of them are done.  Measured on MI355X: the two cross-stream event waits cost more
            # (frame 0.353 ms) than the overlap of ramp-up / tail gains (serial 0.324 ms), hence off by default.
            self._sub("mcp", on_sub)
            for k, sub in enumerate(("mcx", "mca"), start=1):
                c.fork(k)
                self._launch(sub)
            c.join()
            self._sub("ciip", on_sub)
            return
        if name in self._GROUPS:
            for sub in self._GROUPS[name]:
                self._sub(sub, on_sub)
            return
        self._sub(name, on_sub)

    def _sub(self, name, on_sub):
        if on_sub:
            on_sub(name, "begin")
        self._launch(name)
        if on_sub:
            on_sub(name, "end")

    def _launch(self, name: str):
        c = self.ctx
        if name == "mcp":
            c.mc(self.dst, self.refs, self.mc_units, self.lmcs_fwd, intra=self.intra)
        elif name == "mcx":
            if self.mcx_units and self.aff_units:       # both kinds: one launch (the "mca" sub-stage is then empty)
                c.mcxa(self.dst, self.refs, self.mcx_units, self.aff_units, self.aff_side, self.lmcs_fwd, self.mv_out)
            elif self.mcx_units:
                c.mcx(self.dst, self.refs, self.mcx_units, self.lmcs_fwd, self.mv_out)
        elif name == "mca":
            if self.aff_units and not self.mcx_units:
                c.mca(self.dst, self.refs, self.aff_units, self.aff_side, self.lmcs_fwd)
        elif name == "ciip":
            if self.ciip_units:
                c.ciip(self.dst, self.intra, self.ciip_units)
        elif name == "itx_l":
            k = self.wl.tb_classes
            c.itx_classes(self.dst, self.tb_cmds, self.coefs, 0, k[0], k[1])
        elif name == "lmcs_scale":
            if self.lmcs_regions:
                c.lmcs_scale(self.dst, self.lmcs_regions, self.lmcs, self.lmcs_scales)
        elif name == "itx_c":
            k = self.wl.tb_classes
            if self.lmcs is not None and k[3]:          # the inverse mapping rides in the chroma launch ("lmcs_inv" is then empty)
                c.itx_chroma_lmcs(self.dst, self.tb_cmds, self.coefs, k[0] + k[1], k[2], k[3], self.lmcs_scales, self.lmcs_bwd)
            elif k[2] + k[3]:
                c.itx_classes(self.dst, self.tb_cmds, self.coefs, k[0] + k[1], k[2], k[3], self.lmcs_scales)
        elif name == "lmcs_inv":
            if self.lmcs is not None and not self.wl.tb_classes[3]:
                c.lmcs_inverse(self.dst, self.lmcs_bwd)
        elif name == "dbf":
            c.dbf_edges(self.dst, self.dbf_v, self.dbf_h, self.wl.dbf_planes["beta_offset"], self.wl.dbf_planes["tc_offset"])
        elif name == "sao":
            c.sao(self.tmp, self.dst, self.sao_params, self.log2_ctu)
        elif name == "alf":
            c.alf(self.dst, self.tmp, self.alf)
        else:
            raise ValueError(name)

    def decode(self, stages=STAGES):
        for s in stages:
            self.run_stage(s)

    def result(self):
        self.ctx.sync()
        return self.dst.download()

    def refined_mvs(self) -> np.ndarray:
        """int32 [n_mcx_units, 4]: the motion vectors the BDOF / DMVR units finally used."""
        self.ctx.sync()
        return self.mv_out.download(np.int32).reshape(-1, 4) if self.mv_out else np.zeros((0, 4), np.int32)

    def free(self):
        for b in self.bufs + [self.dbf_planes, self.alf]:
            b.free()
        for p in self.refs + [self.dst, self.tmp] + ([self.intra] if self.intra else []):
            p.free()


class Job:
    """One picture in flight through the C-side flush (ovhip_job_*): the recorder's arrays are page-locked, `flush`
    enqueues H2D + every stage launch + the D2H of the refined motion vectors and returns without waiting."""

    def __init__(self, ctx: Context, w: int, h: int):
        self.ctx, self.lib, self.w, self.h = ctx, ctx.lib, w, h
        j = C.c_void_p()
        ctx._chk(self.lib.ovhip_job_create(ctx.h, w, h, C.byref(j)), "job_create")
        self.j = j
        self.rec = capi.Recorder.__new__(capi.Recorder)          # a view of the job's recorder (not owned)
        self.rec.lib, self.rec.h, self.rec._keep = self.lib, self.lib.ovhip_job_recorder(j), []
        self._keep = {}

    def close(self):
        if self.j:
            self.rec.h = None
            self.lib.ovhip_job_destroy(self.j)
            self.j = None

    def begin(self):
        self.ctx._chk(self.lib.ovhip_job_begin(self.j), "job_begin")

    def load_workload(self, wl):
        """Replay a recorded picture (synth.Workload) into the job's recorder: commands, arena, edge lists."""
        self.begin()
        r = self.rec
        for which, arr in ((capi.REC_COEF, wl.coefs), (capi.REC_TB, wl.tb_cmds), (capi.REC_MC, wl.mc_units),
                           (capi.REC_MCX, wl.mcx_units), (capi.REC_AFF, wl.aff_units), (capi.REC_SIDE, wl.aff_side),
                           (capi.REC_REGION, wl.lmcs_regions), (capi.REC_CIIP, wl.ciip_units), (capi.REC_ITASK, wl.itasks),
                           (capi.REC_EDGE_V, wl.dbf_edges[0]), (capi.REC_EDGE_H, wl.dbf_edges[1])):
            if arr is not None and len(arr):
                r.append_raw(which, arr)
        offs = capi.DbfOffsets()
        for i in range(8):
            offs.beta[i], offs.tc[i] = wl.dbf_planes["beta_offset"], wl.dbf_planes["tc_offset"]
        self.ctx._chk(self.lib.ovhip_rec_set_dbf_offsets(r.h, C.byref(offs), 1), "set_dbf_offsets")
        self.params = self.make_params(wl)

    def make_params(self, wl, log2_ctu: int = 7, stages: int = 0) -> "capi.JobParams":
        keep = self._keep
        keep["sao"] = np.ascontiguousarray(wl.sao_params)
        keep["alf"] = {k: np.ascontiguousarray(wl.alf[k], dtype=dt) for k, dt in capi.ALF_TABLES}
        keep["lmcs"] = wl.lmcs
        a = keep["alf"]
        p = capi.JobParams()
        p.lmcs = C.addressof(wl.lmcs) if wl.lmcs is not None else None
        p.sao = keep["sao"].ctypes.data
        p.alf_ctus = a["ctus"].ctypes.data
        p.alf_luma_coeff, p.alf_luma_clip = a["luma_coeff"].ctypes.data, a["luma_clip"].ctypes.data
        p.alf_chroma_coeff, p.alf_chroma_clip = a["chroma_coeff"].ctypes.data, a["chroma_clip"].ctypes.data
        p.alf_cc_coeff = a["cc_coeff"].ctypes.data
        p.log2_ctu_s, p.stages = log2_ctu, stages
        return p

    def flush(self, dst: "DevPic", refs: list, intra: "DevPic | None" = None, params=None):
        arr = (capi.Pic * max(len(refs), 1))(*[r.s for r in refs])
        self.ctx._chk(self.lib.ovhip_job_flush(self.j, C.byref(dst.s), arr, len(refs), C.byref(intra.s) if intra else None,
                                               C.byref(params if params is not None else self.params)), "job_flush")

    def wait(self):
        self.ctx._chk(self.lib.ovhip_job_wait(self.j), "job_wait")

    def band(self, dst: "DevPic", refs: list, row_end: int, last: bool = False, upto: "capi.BandCounts | None" = None, params=None):
        """ovhip_job_band: what was recorded since the previous call (or up to `upto`) as one band of CTU rows ending at row_end"""
        arr = (capi.Pic * max(len(refs), 1))(*[r.s for r in refs])
        self.ctx._chk(self.lib.ovhip_job_band(self.j, C.byref(dst.s), arr, len(refs), C.byref(params if params is not None else self.params),
                                              C.byref(upto) if upto is not None else None, row_end, int(last)), "job_band")

    def band_progress(self) -> int:
        rows = C.c_int32()
        self.ctx._chk(self.lib.ovhip_job_band_progress(self.j, C.byref(rows), None, None), "job_band_progress")
        return rows.value

    def flush_in_bands(self, wl, dst: "DevPic", refs: list, ctu_rows_per_band: int = 1, log2_ctu: int = 7, params=None):
        """A recorded picture (load_workload(wl) before) submitted band by band, as the live decoder's row hooks do while the picture is
        parsed: the counts of every array at each band's last CTU row come from the commands' positions (the arrays are in decoding order)."""
        rows = list(range(ctu_rows_per_band << log2_ctu, self.h, ctu_rows_per_band << log2_ctu))
        cuts = band_counts(wl, rows, log2_ctu)
        for r, c in zip(rows, cuts):
            self.band(dst, refs, r, False, c, params)
        self.band(dst, refs, self.h, True, None, params)

    def bind(self, ctx: "Context"):
        """The next flushes go to ctx's stream (a free frame thread takes the picture over)."""
        self.ctx._chk(self.lib.ovhip_job_bind(self.j, ctx.h), "job_bind")
        self.ctx = ctx

    def dmvr_rows(self, refs: list) -> int:
        arr = (capi.Pic * max(len(refs), 1))(*[r.s for r in refs])
        n = self.lib.ovhip_job_dmvr_rows(self.j, arr, len(refs))
        if n < 0:
            self.ctx._chk(int(n), "job_dmvr_rows")
        return int(n)

    def dmvr_rows_begin(self, refs: list, log2_ctu: int = 0) -> int:
        """enqueue the eager pass over the refined units recorded since the last one (log2_ctu != 0: + their TMVP plane entries)"""
        arr = (capi.Pic * max(len(refs), 1))(*[r.s for r in refs])
        n = self.lib.ovhip_job_dmvr_rows_begin(self.j, arr, len(refs), log2_ctu)
        if n < 0:
            self.ctx._chk(int(n), "job_dmvr_rows_begin")
        return int(n)

    def dmvr_rows_collect(self) -> int:
        n = self.lib.ovhip_job_dmvr_rows_collect(self.j)
        if n < 0:
            self.ctx._chk(int(n), "job_dmvr_rows_collect")
        return int(n)

    def refined_mvs(self) -> np.ndarray:
        n = C.c_size_t()
        p = self.lib.ovhip_job_refined_mvs(self.j, C.byref(n))
        if not n.value:
            return np.zeros((0, 4), np.int32)
        return np.frombuffer((C.c_int32 * (4 * n.value)).from_address(p), dtype=np.int32).reshape(-1, 4).copy()

    def tmvp_cells(self) -> np.ndarray:
        """ovhip_job_tmvp_cells after wait(): capi.TMVP_CELL_DTYPE, 4 entries per refined unit (cell == capi.TMVP_NONE: unused)"""
        n = C.c_size_t()
        p = self.lib.ovhip_job_tmvp_cells(self.j, C.byref(n))
        if not n.value:
            return np.zeros(0, capi.TMVP_CELL_DTYPE)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n.value * capi.TMVP_CELL_DTYPE.itemsize,)).view(capi.TMVP_CELL_DTYPE).copy()

    def test_abort_next_flow(self):
        """test hook: the next flush's flow launch is abandoned for real (ovhip_job_test_abort_next_flow)"""
        self.ctx._chk(self.lib.ovhip_job_test_abort_next_flow(self.j), "job_test_abort_next_flow")

    def stats(self) -> "capi.JobStats":
        s = capi.JobStats()
        self.lib.ovhip_job_last_stats(self.j, C.byref(s))
        return s

    def time_stage(self, name: "str | None"):
        self.ctx._chk(self.lib.ovhip_job_time_stage(self.j, capi.TIME_STAGES.index(name) if name else -1), "job_time_stage")

    def stage_time(self):
        """(sum of the bracketed launch group's durations in seconds, number of flushes measured)"""
        s, n = C.c_double(), C.c_uint64()
        self.ctx._chk(self.lib.ovhip_job_stage_time(self.j, C.byref(s), C.byref(n)), "job_stage_time")
        return s.value * 1e-3, n.value


def band_counts(wl, rows, log2_ctu: int = 7) -> list:
    """capi.BandCounts of a recorded picture at each luma row of `rows` (CTU-row boundaries): how many entries of every array belong
    to the CTU rows above it.  The arrays must be in decoding order (CTU rows ascending) -- checked."""
    def cut(ctu_row, what):
        ctu_row = np.asarray(ctu_row, np.int64)
        if len(ctu_row) > 1 and (np.diff(ctu_row) < 0).any():
            raise EngineError(f"band_counts: {what} are not in CTU-row order")
        return [int(np.searchsorted(ctu_row, r >> log2_ctu, side="left")) for r in rows]

    def arr(a, dt):
        return np.zeros(0, dt) if a is None or not len(a) else np.asarray(a).view(dt).reshape(-1)

    tb, mc, mcx = arr(wl.tb_cmds, capi.TB_CMD_DTYPE), arr(wl.mc_units, capi.MC_UNIT_DTYPE), arr(wl.mcx_units, capi.MC_UNIT_DTYPE)
    aff, reg, it = arr(wl.aff_units, capi.AFF_UNIT_DTYPE), arr(wl.lmcs_regions, capi.LMCS_REGION_DTYPE), arr(wl.itasks, capi.ITASK_DTYPE)
    ev, eh = arr(wl.dbf_edges[0], capi.DBF_EDGE_DTYPE), arr(wl.dbf_edges[1], capi.DBF_EDGE_DTYPE)
    n_coef = 0 if wl.coefs is None else len(wl.coefs)
    n_side = 0 if wl.aff_side is None else len(wl.aff_side)
    tb_row = (tb["y"].astype(np.int64) << (tb["plane"] != 0)) >> log2_ctu
    it_chroma = (it["kind"] == capi.IT_CHROMA) | (it["kind"] == capi.IT_RES_C)
    it_row = (it["y"].astype(np.int64) << it_chroma) >> log2_ctu
    k_tb, k_mc, k_mcx = cut(tb_row, "transform blocks"), cut(mc["y"] >> log2_ctu, "MC units"), cut(mcx["y"] >> log2_ctu, "refined units")
    k_aff, k_reg, k_it = cut(aff["y"] >> log2_ctu, "affine units"), cut(reg["y"] >> log2_ctu, "regions"), cut(it_row, "ordered tasks")
    # an edge's uy counts units of 4 LUMA samples in every plane (the 4x4-luma grid of the deblocking maps)
    k_ev = cut((ev["uy"].astype(np.int64) << 2) >> log2_ctu, "vertical edges")
    k_eh = cut((eh["uy"].astype(np.int64) << 2) >> log2_ctu, "horizontal edges")
    out = []
    for i in range(len(rows)):
        c = capi.BandCounts()
        c.n_tb, c.n_mc, c.n_mcx, c.n_aff, c.n_reg, c.n_itask, c.n_edge_v, c.n_edge_h = k_tb[i], k_mc[i], k_mcx[i], k_aff[i], k_reg[i], k_it[i], k_ev[i], k_eh[i]
        # the arenas are appended with the commands that own their entries: a band's share ends where the next band's first owner starts
        later = tb["coef_off"][k_tb[i]:]
        c.n_coef = int(later.min()) if len(later) else n_coef
        la = aff[k_aff[i]:]
        c.n_side = int(min(la["side_off"].min(), la["prof_off"][(la["flags"] & 1) != 0].min(initial=0xffffffff))) if len(la) else n_side
        out.append(c)
    return out


class Dpb:
    """Device mirror of the decoded picture buffer (ovhip_dpb_*): `devices` = HIP ordinals of the logical devices."""

    def __init__(self, devices=(0,), ops: "capi.DpbOps | None" = None, n_devices: int | None = None):
        self.lib = capi.load()
        h = C.c_void_p()
        if ops is not None:
            self._ops = ops
            r = self.lib.ovhip_dpb_create_ex(C.byref(h), n_devices or 1, C.byref(ops))
        else:
            arr = (C.c_int * len(devices))(*devices)
            r = self.lib.ovhip_dpb_create(C.byref(h), arr, len(devices))
        if r != 0:
            raise EngineError(f"ovhip_dpb_create: {r} (no HIP device? the DPB has no CPU back-end outside the tests' own)")
        self.h = h

    def close(self):
        if self.h:
            self.lib.ovhip_dpb_destroy(self.h)
            self.h = None

    def stats(self) -> "capi.DpbStats":
        st = capi.DpbStats()
        self.lib.ovhip_dpb_get_stats(self.h, C.byref(st))
        return st

    def begin(self, key: int, dev: int, w: int, h: int, tag: int = 0) -> "capi.Pic":
        pic = capi.Pic()
        r = self.lib.ovhip_dpb_begin_tag(self.h, C.c_void_p(key), tag, dev, w, h, C.byref(pic))
        if r != 0:
            raise EngineError(f"ovhip_dpb_begin: {r}")
        return pic

    def lookup(self, key: int) -> "tuple[int, capi.Pic]":
        pic, dev = capi.Pic(), C.c_int()
        r = self.lib.ovhip_dpb_lookup(self.h, C.c_void_p(key), C.byref(dev), C.byref(pic))
        if r != 0:
            raise EngineError(f"ovhip_dpb_lookup: {r}")
        return dev.value, pic


class Frame:
    """One frame thread (ovhip_frame_*): context + job on a logical device of a DPB."""

    def __init__(self, dpb: Dpb, dev: int, w: int, h: int):
        self.lib, self.dpb, self.w, self.h = dpb.lib, dpb, w, h
        f = C.c_void_p()
        r = self.lib.ovhip_frame_create(dpb.h, dev, w, h, C.byref(f))
        if r != 0:
            raise EngineError(f"ovhip_frame_create: {r}")
        self.f = f

    def _chk(self, r, what):
        if r < 0:
            raise EngineError(f"{what}: {r}: {self.lib.ovhip_frame_last_error(self.f).decode()}")
        return r

    def close(self):
        if self.f:
            self.lib.ovhip_frame_destroy(self.f)
            self.f = None

    def begin(self, key: int, tag: int = 0):
        self._chk(self.lib.ovhip_frame_begin_tag(self.f, C.c_void_p(key), tag), "frame_begin")

    def ref(self, key: int, tag: int = 0) -> int:
        return self._chk(self.lib.ovhip_frame_ref_tag(self.f, C.c_void_p(key), tag), "frame_ref")

    def ref_at(self, slot: int, key: int) -> int:
        return self._chk(self.lib.ovhip_frame_ref_at(self.f, slot, C.c_void_p(key)), "frame_ref_at")

    def recorder(self) -> "capi.Recorder":
        r = capi.Recorder.__new__(capi.Recorder)
        r.lib, r.h, r._keep = self.lib, self.lib.ovhip_frame_recorder(self.f), []
        if not r.h:
            raise EngineError("ovhip_frame_recorder")
        r.close = lambda: None                                     # owned by the frame's job
        return r

    def dmvr_rows(self) -> int:
        return int(self._chk(self.lib.ovhip_frame_dmvr_rows(self.f), "frame_dmvr_rows"))

    def dmvr_rows_begin(self, log2_ctu: int = 7) -> int:
        return int(self._chk(self.lib.ovhip_frame_dmvr_rows_begin(self.f, log2_ctu), "frame_dmvr_rows_begin"))

    def dmvr_rows_collect(self) -> int:
        return int(self._chk(self.lib.ovhip_frame_dmvr_rows_collect(self.f), "frame_dmvr_rows_collect"))

    def submit(self, params: "capi.JobParams", job: "Job | None" = None, out: "capi.FrameOutput | None" = None, check: bool = True) -> int:
        r = self.lib.ovhip_frame_submit(self.f, job.j if job is not None else None, None, C.byref(params), C.byref(out) if out is not None else None)
        return self._chk(r, "frame_submit") if check else r

    def fail(self, status: int = -3):
        self.lib.ovhip_frame_fail(self.f, status)

    def job(self) -> "Job":
        """the frame's own job as an engine.Job view (not owned)"""
        j = Job.__new__(Job)
        ctx = Context.__new__(Context)
        ctx.lib, ctx.h, ctx._bufs = self.lib, C.c_void_p(self.lib.ovhip_frame_ctx(self.f)), []
        j.ctx, j.lib, j.w, j.h, j.j, j._keep = ctx, self.lib, self.w, self.h, C.c_void_p(self.lib.ovhip_frame_job(self.f)), {}
        j.rec = self.recorder()
        return j


class Stream:
    """The C stream driver (ovhip_stream_*): frame threads per device decoding a list of pictures in decoding order.

    contents: list of dicts {"params": capi.JobParams, "calllog": np.ndarray | None, "n_ref_slots": int}; jobs: engine.Job list
    (pre-recorded pictures).  Everything referenced stays alive as long as this object."""

    def __init__(self, dpb: Dpb, w: int, h: int, contents: list, jobs: list = (), threads_per_device: int = 1, flags: int = 0,
                 output: int = 0, window=(0, 0, 0, 0), extra_stages: int = 0, rank: int = 0, xfer: "capi.StreamXfer | None" = None,
                 intra_lookahead: int = 0, ahead_own_queue: int = 0):
        self.lib, self.dpb = dpb.lib, dpb
        self._contents = (capi.StreamContent * len(contents))()
        self._keep = [contents, jobs, xfer]
        for i, c in enumerate(contents):
            sc = self._contents[i]
            log = c.get("calllog")
            sc.calllog = log.ctypes.data if log is not None else None
            sc.calllog_bytes = log.nbytes if log is not None else 0
            sc.params = c["params"]
            sc.n_ref_slots = c.get("n_ref_slots", 2)
        self._jobs = (C.c_void_p * max(1, len(jobs)))(*[j.j for j in jobs])
        cfg = capi.StreamCfg()
        cfg.w, cfg.h, cfg.flags, cfg.threads_per_device, cfg.output = w, h, flags, threads_per_device, output
        cfg.window = capi.Window(*window)
        cfg.extra_stages, cfg.rank = extra_stages, rank
        # a capi.StreamXfer of Python callbacks, or the address of a C one (RcclTransport.xfer: ovhip_rccl_xfer)
        cfg.xfer = (C.cast(C.c_void_p(xfer), C.POINTER(capi.StreamXfer)) if isinstance(xfer, int) else C.pointer(xfer)) if xfer is not None else None
        cfg.intra_lookahead, cfg.ahead_own_queue = intra_lookahead, ahead_own_queue
        self.cfg = cfg
        s = C.c_void_p()
        r = self.lib.ovhip_stream_create(C.byref(s), dpb.h, C.byref(cfg), self._contents, len(contents), self._jobs, len(jobs))
        if r != 0:
            raise EngineError(f"ovhip_stream_create: {r}")
        self.s = s

    def close(self):
        if self.s:
            self.lib.ovhip_stream_destroy(self.s)
            self.s = None

    def queue_info(self):
        """(streams replaced to clear the look-ahead thread's hardware queue, in-order streams still sharing it)"""
        a, b = C.c_int(), C.c_int()
        self.lib.ovhip_stream_queue_info(self.s, C.byref(a), C.byref(b))
        return a.value, b.value

    def picture(self, idx: int, ctx: "Context") -> "DevPic":
        """the device picture of stream picture idx (kept by OVHIP_STREAM_KEEP), as a DevPic on ctx for download()"""
        key = self.lib.ovhip_stream_key(self.s, idx)
        _dev, pic = self.dpb.lookup(key)
        return DevPic(ctx, pic, owns=False)

    @staticmethod
    def pics_array(pics: list):
        """pics: list of dicts (content, job, poc, device, refs, owner, send_mask) -> (capi.StreamPic * n)"""
        arr = (capi.StreamPic * max(1, len(pics)))()
        for i, p in enumerate(pics):
            a = arr[i]
            a.content, a.job, a.poc, a.device = p.get("content", 0), p.get("job", 0), p.get("poc", i), p.get("device", 0)
            refs = list(p.get("refs", ()))
            assert len(refs) <= capi.STREAM_MAX_REFS
            a.n_refs = len(refs)
            for k, r in enumerate(refs):
                a.refs[k] = r
            a.owner, a.send_mask = p.get("owner", 0), p.get("send_mask", 0)
        return arr

    def run(self, arr, n_total: int, first: int, n: int, flags: int = 0, digests: bool = False, check: bool = True, trace=None):
        """-> (capi.StreamResult, digests uint8 [n, 16] | None); trace: float64 [n, 8] filled with (taken, submit, published, thread, references in hand, launches enqueued, submit returned, complete on the device)"""
        res = capi.StreamResult()
        if trace is not None:
            res.trace = trace.ctypes.data
        dg = np.zeros((n, 16), np.uint8) if digests else None
        r = self.lib.ovhip_stream_run(self.s, arr, n_total, first, n, flags | (capi.STREAM_DIGESTS if digests else 0),
                                      dg.ctypes.data if dg is not None else None, C.byref(res))
        if r != 0 and check:
            raise EngineError(f"ovhip_stream_run: {r}: {res.error.decode(errors='replace')}")
        return res, dg


class RcclTransport:
    """ovhip_rccl_*: the stream driver's multi-process exchange over RCCL point-to-point (one process per GPU)."""

    def __init__(self, unique_id: bytes, rank: int, world: int, hip_device: int):
        self.lib = capi.load()
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        r = self.lib.ovhip_rccl_create(C.byref(h), buf, rank, world, hip_device)
        if r != 0:
            raise EngineError(f"ovhip_rccl_create: {r}")
        self.h = h

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        r = capi.load().ovhip_rccl_unique_id(buf)
        if r != 0:
            raise EngineError(f"ovhip_rccl_unique_id: {r}")
        return bytes(buf)

    @property
    def xfer(self) -> int:
        return int(self.lib.ovhip_rccl_xfer(self.h))

    def stats(self) -> dict:
        out = (C.c_uint64 * 4)()
        self.lib.ovhip_rccl_stats(self.h, out)
        return {"pictures_sent": int(out[0]), "bytes_sent": int(out[1]), "pictures_received": int(out[2]), "bytes_received": int(out[3])}

    def self_exchange(self, src: "DevPic", dst: "DevPic"):
        r = self.lib.ovhip_rccl_self_exchange(self.h, C.byref(src.s), C.byref(dst.s))
        if r != 0:
            raise EngineError(f"ovhip_rccl_self_exchange: {r}: {self.lib.ovhip_rccl_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.ovhip_rccl_destroy(self.h)
            self.h = None
