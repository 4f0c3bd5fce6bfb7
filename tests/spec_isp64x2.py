"""TEST INFRASTRUCTURE -- PARITY UNPINNED.  A restatement of H.266 (ITU-T H.266 8.7.3 scaling, 8.7.4 transformation) for ONE block
shape: the 64x2 transform block of a 64x8 coding unit split horizontally into intra sub-partitions.  The reference's own result for
this block is undefined (rcn_Xx2_tb de-quantises with a row stride of 32 and transforms with 64, rcn_transform_tree.c:985-1009: it reads
stack memory nothing wrote), so there is nothing of the reference's to be equal to; the back-end and the oracle reconstruct it as the
specification defines it and this file is the checker.  What is NOT independent of the rest of the repo: the 64-point DCT-II matrix
(openvvc_amd/csrc/vvc_tables.h, probed from the compiled reference; every other 64-wide block is pinned against the reference with it)."""
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LEVEL_SCALE = np.array([[40, 45, 51, 57, 64, 72], [57, 64, 72, 80, 90, 102]], np.int64)      # H.266 (1155)


def dct2_64() -> np.ndarray:
    """[k][j]: weight of input coefficient k (k < 32: the coded ones; the other 32 are zero, 8.7.4.2) for output sample j"""
    txt = (ROOT / "openvvc_amd" / "csrc" / "vvc_tables.h").read_text()
    m = re.search(r"ovt_dct2_64\[2048\]\s*=\s*\{([^}]*)\}", txt)
    return np.array([int(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()], np.int64).reshape(32, 64)


def residual_64x2(levels: np.ndarray, qp: int, dep_quant: int, bit_depth: int = 10) -> np.ndarray:
    """levels: int [2][32] (the coded area of the 64x2 block) -> residual int [2][64]"""
    log2_w, log2_h = 6, 1
    rect = (log2_w + log2_h) & 1
    q = qp + 1 if dep_quant else qp                                                  # 8.7.3: qP + 1 with dependent quantisation
    bd_shift = bit_depth + rect + ((log2_w + log2_h) // 2) + 10 - 15 + dep_quant     # (1152); log2TransformRange = 15
    ls = (16 * LEVEL_SCALE[rect][q % 6]) << (q // 6)                                 # flat scaling list m = 16
    d = np.clip((levels.astype(np.int64) * ls + (1 << (bd_shift - 1))) >> bd_shift, -32768, 32767)
    # 8.7.4.1: the vertical transform first (nTbH = 2: DCT-II [[64, 64], [64, -64]]), intermediate clipped after (e + 64) >> 7
    e0, e1 = 64 * d[0] + 64 * d[1], 64 * d[0] - 64 * d[1]
    g = np.clip((np.stack([e0, e1]) + 64) >> 7, -32768, 32767)
    # then the horizontal one: 64 outputs from the 32 coded inputs; residual = (r + 2^(bdShift - 1)) >> bdShift, bdShift = 20 - BitDepth
    r = g @ dct2_64()
    sh = 20 - bit_depth
    return np.clip((r + (1 << (sh - 1))) >> sh, -32768, 32767)
