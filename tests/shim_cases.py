"""Command streams recorded by the INSTALLED slots of the MI355X override block (shim/rcn_hip.c), tests/golden/shim_*.ovg.

oracle/ref_harness/gen_golden.c writes them in its "shim" mode: it installs rcn_init_functions_hip() on the table the
reference's own initialisers filled and drives the installed slots (tmp.rcn_tu_st, rcn_mcp_b, rcn_dmvr_mv_refine,
df.rcn_dbf_ctu, ...) with exactly the seeded OVCTUDec states of the reference run that produced tests/golden/*.ovg.
Case i of a stream therefore corresponds to case i of the reference fixture: executing the stream must reproduce the
bytes the reference wrote."""
import numpy as np

import golden_io
from openvvc_amd import capi

ARRAYS = ("tb", "coef", "mc", "mcx", "aff", "side", "region", "ciip", "edge_v", "edge_h", "itask")
DTYPES = {"tb": capi.TB_CMD_DTYPE, "coef": np.dtype("<i2"), "mc": capi.MC_UNIT_DTYPE, "mcx": capi.MC_UNIT_DTYPE,
          "aff": capi.AFF_UNIT_DTYPE, "side": np.dtype("<i4"), "region": capi.LMCS_REGION_DTYPE, "ciip": capi.CIIP_UNIT_DTYPE,
          "edge_v": capi.DBF_EDGE_DTYPE, "edge_h": capi.DBF_EDGE_DTYPE, "itask": capi.ITASK_DTYPE}


class ShimStream:
    def __init__(self, name: str, directory=None):
        g = golden_io.load(name, directory)
        self.g = g
        self.arr = {k: np.frombuffer(np.ascontiguousarray(g[k]).tobytes(), dtype=DTYPES[k]) for k in ARRAYS}
        self.off = g["case_off"]
        self.n = self.off.shape[0] - 1
        self.ref_map = [int(v) for v in g["ref_map"]] if "ref_map" in g and g["ref_map"].ndim == 1 else []

    def case(self, i: int) -> dict:
        """The arrays the slots recorded for case i (offsets inside them are relative to the case)."""
        return {k: self.arr[k][int(self.off[i, j]):int(self.off[i + 1, j])].copy() for j, k in enumerate(ARRAYS)}

    def refs(self, fixture_refs: list) -> list:
        """The reference-picture table the slots built (order of first use) out of the fixture's pictures."""
        assert all(m >= 0 for m in self.ref_map)
        return [fixture_refs[m] for m in self.ref_map]
