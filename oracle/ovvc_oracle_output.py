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


_P, _SEED = np.uint32(0x01000193), np.array([0x67452301, 0xefcdab89, 0x98badcfe, 0x10325476], np.uint32)


def _mix128(words: np.ndarray, tag: int) -> np.ndarray:
    """the fingerprint's own hash (include/ovvc_hip.h, "Digest"), vectorised over the first axis: words uint32 [k, m] -> digests
    uint32 [k, 4].  Four FNV-1a lanes (word i goes to lane i & 3), murmur3's finaliser per lane, murmur3_x86_128's last mixing."""
    with np.errstate(over="ignore"):
        k, m = words.shape
        h = np.tile(_SEED, (k, 1))
        h[:, 0] ^= np.uint32(tag)
        for i in range(m):
            h[:, i & 3] = (h[:, i & 3] ^ words[:, i]) * _P
        for j in range(4):
            v = h[:, j]
            v ^= v >> np.uint32(16); v *= np.uint32(0x85ebca6b); v ^= v >> np.uint32(13); v *= np.uint32(0xc2b2ae35); v ^= v >> np.uint32(16)
            h[:, j] = v
        h[:, 0] += h[:, 1]; h[:, 2] += h[:, 3]; h[:, 0] += h[:, 2]; h[:, 1] += h[:, 0]; h[:, 2] += h[:, 0]; h[:, 3] += h[:, 0]
    return h


def _sample_words(a: np.ndarray) -> np.ndarray:
    """uint16 [k, n] -> uint32 [k, ceil(n / 2)]: two samples per word, little endian; an odd last sample alone in its word"""
    k, n = a.shape
    if n & 1:
        a = np.concatenate([a, np.zeros((k, 1), np.uint16)], axis=1)
    a = a.astype(np.uint32)
    return a[:, 0::2] | (a[:, 1::2] << np.uint32(16))


def picture_digest(y, cb, cr, window=(0, 0, 0, 0)) -> bytes:
    """the library's per-picture fingerprint (include/ovvc_hip.h, "Digest"): leaf = mix128 of each 512-byte piece of a cropped row
    (tag: its samples), row = mix128 of its leaf digests (tag: their number), band = mix128 of the digests of 8 consecutive rows of a
    plane (tag: the rows), picture = MD5 of the band digests (Y, Cb, Cr)"""
    bands = []
    for p in cropped_planes(y, cb, cr, window):
        p = np.ascontiguousarray(p, dtype=np.uint16)
        h, w = p.shape
        nfull, tail = w // 256, w % 256
        leaves = []
        if nfull:
            lw = _sample_words(p[:, :nfull * 256].reshape(h * nfull, 256))
            leaves.append(_mix128(lw, 256).reshape(h, nfull, 4))
        if tail:
            leaves.append(_mix128(_sample_words(p[:, nfull * 256:]), tail).reshape(h, 1, 4))
        lv = np.concatenate(leaves, axis=1)                                  # [h, nseg, 4]
        rows = _mix128(lv.reshape(h, -1), lv.shape[1])                        # [h, 4]
        for o in range(0, h, 8):
            r = rows[o:o + 8]
            bands.append(_mix128(r.reshape(1, -1), len(r))[0].astype("<u4").tobytes())
    return hashlib.md5(b"".join(bands)).digest()
