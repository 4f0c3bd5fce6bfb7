"""TEST INFRASTRUCTURE -- CPU restatement of the reference's output path, for checking the device output path only (tests/,
smoke).  Never imported by the product.

PARITY UNPINNED for this leg: examples/dectest.c includes ovversion.h, which only the reference's build system generates (version.sh);
the file cannot be compiled here without writing a stand-in for it, so write_decoded_frame_to_file was never run -- this restatement
is pinned by nothing but its reading of the source.  It is ten lines of slicing:

write_decoded_frame_to_file (examples/dectest.c:372-409): the window offsets of OVFrame.output_window are in chroma sample
units (luma: twice, :383-388); every component writes frame_h rows of frame_w 16-bit little-endian samples starting at
(win_left, win_top) (:389-397); without a window the three planes are written whole (:399-405) -- the same bytes.
"""
import hashlib

import numpy as np


def cropped_planes(y, cb, cr, window=(0, 0, 0, 0)):
    l, r, a, b = window
    out = []
    for c, p in enumerate((y, cb, cr)):
        sh = 0 if c else 1
        h, w = p.shape
        out.append(p[a << sh:h - (b << sh), l << sh:w - (r << sh)])
    return out


def packed_frame(y, cb, cr, window=(0, 0, 0, 0)) -> bytes:
    """the bytes dectest writes for one frame"""
    return b"".join(np.ascontiguousarray(p, dtype="<u2").tobytes() for p in cropped_planes(y, cb, cr, window))


def row_digests(y, cb, cr, window=(0, 0, 0, 0)) -> np.ndarray:
    """MD5 of every cropped row's bytes: Y rows, Cb rows, Cr rows"""
    d = [hashlib.md5(np.ascontiguousarray(row, dtype="<u2").tobytes()).digest() for p in cropped_planes(y, cb, cr, window) for row in p]
    return np.frombuffer(b"".join(d), np.uint8).reshape(-1, 16)


def picture_digest(y, cb, cr, window=(0, 0, 0, 0)) -> bytes:
    """the library's per-picture fingerprint (include/ovvc_hip.h, "Digest"): leaf = MD5 of each 512-byte piece of a cropped row,
    row = MD5 of its leaf digests, band = MD5 of the digests of 8 consecutive rows of a plane, picture = MD5 of the band digests"""
    bands = []
    for p in cropped_planes(y, cb, cr, window):
        rows = []
        for row in p:
            b = np.ascontiguousarray(row, dtype="<u2").tobytes()
            rows.append(hashlib.md5(b"".join(hashlib.md5(b[o:o + 512]).digest() for o in range(0, len(b), 512))).digest())
        bands += [hashlib.md5(b"".join(rows[o:o + 8])).digest() for o in range(0, len(rows), 8)]
    return hashlib.md5(b"".join(bands)).digest()
