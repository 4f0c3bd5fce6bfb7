"""The reference-driven chained mini-stream: tests/golden/pipe.ovg + shim_pipe.ovg (oracle/ref_harness/gen_pipe.c).

pipe.ovg      : 5 pictures of 416x240 (I, B, B, B, P in decoding order) as the REFERENCE's own slice decoder left them -- parse
                (coding_quadtree / dual_tree -> coding_unit -> prediction_unit / transform_unit), the rcn slots in decode_ctu /
                decode_ctu_line order with the reference's CTU scratch and line buffers, lmcs_reshape_backward, df.rcn_dbf_ctu,
                sao / alf line filters -- each picture predicted from the pictures decoded before it, + the vectors every
                rcn_dmvr_mv_refine call returned.
shim_pipe.ovg : what the INSTALLED slots of shim/rcn_hip.c recorded for the same pictures: one command stream per picture and the
                picture-level parameters its flush would pass (SAO / ALF parameters and tables, LMCS tables, deblocking offsets,
                the reference-picture table in order of first use).

`workload(k, decoded)` turns picture k's stream into the synth.Workload the oracle pipeline (tests/oracle_pipeline.py) and the
C-side flush (engine.Job.load_workload) take; its reference pictures are the pictures the CALLER decoded before."""
import numpy as np

import golden_io
from openvvc_amd import capi
from openvvc_amd.synth import Workload
from shim_cases import ShimStream


class Pipe:
    def __init__(self, name="pipe", directory=None):
        """directory: where <name>.ovg / shim_<name>.ovg lie (default tests/golden; bench.py: a stream gen_pipe just made)"""
        g = golden_io.load(f"{name}.ovg", directory)
        self.w, self.h, self.n, self.log2_ctu = (int(v) for v in g["geometry"])
        self.info = g["info"]                      # poc, slice type, qp, n0, l0[2], n1, l1[2], tmvp, first / end DMVR call, lmcs
        self.dmvr = g["dmvr"]                      # x, y, log2 w, log2 h, mv0 in, mv1 in, mv0 out, mv1 out
        fr, s = g["frames"], self.w * self.h
        self.frames = []
        for k in range(self.n):
            f = fr[k * s * 3 // 2:(k + 1) * s * 3 // 2]
            self.frames.append((f[:s].reshape(self.h, self.w), f[s:s * 5 // 4].reshape(self.h // 2, self.w // 2),
                                f[s * 5 // 4:].reshape(self.h // 2, self.w // 2)))
        self.s = ShimStream(f"shim_{name}.ovg", directory)
        assert self.s.n == self.n
        self.g = self.s.g

    def ref_indices(self, k):
        """picture indices of picture k's reference-picture table, in the order the slots first used them"""
        np_ = int(self.g["pic_flags"][k][1])
        m = [int(v) for v in self.g["ref_map"][k][:np_]]
        assert all(0 <= v < k for v in m)
        return m

    def dmvr_calls(self, k):
        a, b = int(self.info[k][10]), int(self.info[k][11])
        return self.dmvr[a:b]

    def workload(self, k, decoded):
        """decoded: {picture index: (y, cb, cr)} -- what the caller's decoder made of the earlier pictures"""
        c, g = self.s.case(k), self.g
        lmcs_on = bool(g["pic_flags"][k][0])
        tb = c["tb"]
        luma = tb["plane"] == 0
        tb = np.concatenate([tb[luma], tb[~luma]])
        offs = g["dbf_offsets"][k]
        w4, h4 = self.w // 4, self.h // 4
        planes = edges_to_planes(c["edge_v"], c["edge_h"], w4, h4)
        planes["beta_offset"], planes["tc_offset"] = int(offs[0]), int(offs[8])
        assert (c["edge_v"]["pad"] == 0).all() and (c["edge_h"]["pad"] == 0).all()     # one slice: one offset pair
        alf = {"ctus": np.frombuffer(g["alf_ctus"][k].tobytes(), capi.ALF_CTU_DTYPE)}
        for name in ("luma_coeff", "luma_clip", "chroma_coeff", "chroma_clip", "cc_coeff"):
            alf[name] = g[name][k]
        return Workload(
            w=self.w, h=self.h, seed=0, refs=[decoded[i] for i in self.ref_indices(k)], ref_pocs=[], cus=np.zeros((0, 4), np.int32),
            mc_units=c["mc"], tb_cmds=tb, coefs=c["coef"], n_luma_cmds=int(luma.sum()),
            mcx_units=c["mcx"], aff_units=c["aff"], aff_side=c["side"], ciip_units=c["ciip"], intra=None,
            lmcs=capi.LmcsLuts.from_buffer_copy(g["luts"][k].tobytes()) if lmcs_on else None,
            lmcs_regions=c["region"], dbf_planes=planes, dbf_edges=[c["edge_v"], c["edge_h"]],
            sao_params=np.frombuffer(g["sao"][k].tobytes(), capi.SAO_CTU_DTYPE), alf=alf,
            itasks=c["itask"] if len(c["itask"]) else None)


def edges_to_planes(ev, eh, w4, h4):
    """the dense deblocking planes (include/ovvc_hip.h, ovhip_dbf_planes) the compact edge lists stand for"""
    w4c, h4c = (w4 + 1) // 2, (h4 + 1) // 2
    p = {"luma_v": np.zeros((h4, w4), np.uint16), "luma_h": np.zeros((h4, w4), np.uint16),
         "cb_v": np.zeros((h4, w4c), np.uint16), "cr_v": np.zeros((h4, w4c), np.uint16),
         "cb_h": np.zeros((h4c, w4), np.uint16), "cr_h": np.zeros((h4c, w4), np.uint16), "w4": w4, "h4": h4}
    for e, d in ((ev, "v"), (eh, "h")):
        for comp, name in ((0, "luma"), (1, "cb"), (2, "cr")):
            s = e[e["comp"] == comp]
            ux, uy = s["ux"].astype(int), s["uy"].astype(int)
            if comp and d == "v":
                ux = ux // 2
            if comp and d == "h":
                uy = uy // 2
            p[f"{name}_{d}"][uy, ux] = s["word"]
    return p
