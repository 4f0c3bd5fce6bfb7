"""Output path: crop + pack + digests.  CPU: the library's host MD5 against hashlib, the size helpers; GPU: the
device pack and the per-row MD5 kernel against the numpy / hashlib restatement of dectest.c:372-409 (oracle/ovvc_oracle_output.py)."""
import ctypes as C
import hashlib
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import ovvc_oracle_output as oo                                                        # noqa: E402


def test_host_md5_matches_hashlib(built_lib):
    from openvvc_amd import capi
    rs = np.random.RandomState(7)
    assert capi.md5(b"") == hashlib.md5(b"").digest()                                    # RFC 1321 test suite, first entry
    assert capi.md5(b"abc").hex() == "900150983cd24fb0d6963f7d28e17f72"
    for n in (1, 55, 56, 57, 63, 64, 65, 119, 120, 128, 1000, 7680, 100003):
        data = rs.randint(0, 256, n).astype(np.uint8).tobytes()
        assert capi.md5(data) == hashlib.md5(data).digest(), n
    # incremental updates across block boundaries
    lib = capi.load()
    data = rs.randint(0, 256, 5000).astype(np.uint8).tobytes()
    st = capi.Md5State()
    lib.ovhip_md5_init(C.byref(st))
    pos = 0
    for step in (1, 62, 1, 64, 3, 700, 129, 4040):
        chunk = data[pos:pos + step]
        lib.ovhip_md5_update(C.byref(st), (C.c_uint8 * len(chunk)).from_buffer_copy(chunk), len(chunk))
        pos += step
    assert pos == len(data)
    out = (C.c_uint8 * 16)()
    lib.ovhip_md5_final(C.byref(st), out)
    assert bytes(out) == hashlib.md5(data).digest()


def test_output_geometry(built_lib):
    from openvvc_amd import capi
    lib = capi.load()
    w = capi.Window(0, 0, 0, 0)
    assert lib.ovhip_output_bytes(3840, 2160, C.byref(w)) == 3840 * 2160 * 3                 # 1.5 samples x 2 bytes
    assert lib.ovhip_output_rows(3840, 2160, C.byref(w)) == 2160 * 2
    assert lib.ovhip_output_bytes(3840, 2160, None) == 3840 * 2160 * 3
    w = capi.Window(1, 2, 3, 4)                                                               # chroma units
    assert lib.ovhip_output_bytes(416, 240, C.byref(w)) == ((416 - 6) * (240 - 14) + 2 * (208 - 3) * (120 - 7)) * 2
    assert lib.ovhip_output_rows(416, 240, C.byref(w)) == (240 - 14) + 2 * (120 - 7)
    w = capi.Window(104, 104, 0, 0)                                                           # nothing left
    assert lib.ovhip_output_bytes(416, 240, C.byref(w)) == 0


WINDOWS = [(0, 0, 0, 0), (1, 0, 0, 3), (3, 5, 2, 1), (0, 4, 0, 4)]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(416, 240), (1920, 1080), (3840, 2160), (72, 34)])
def test_device_output_matches_dectest_layout(w, h):
    from openvvc_amd import engine
    ctx = engine.Context(0)
    rs = np.random.RandomState(w + h)
    y = rs.randint(0, 1024, (h, w)).astype(np.uint16)
    cb = rs.randint(0, 1024, (h // 2, w // 2)).astype(np.uint16)
    cr = rs.randint(0, 1024, (h // 2, w // 2)).astype(np.uint16)
    pic = ctx.upload_pic(y, cb, cr)
    for win in WINDOWS:
        got = pic.output(win)
        assert got.tobytes() == oo.packed_frame(y, cb, cr, win), (w, h, win)
        want = oo.row_digests(y, cb, cr, win)
        assert np.array_equal(pic.row_digests(win), want), (w, h, win)
        assert pic.digest(win) == oo.picture_digest(y, cb, cr, win)
        # the file's md5sum from the packed frame = what CI/checkMD5.sh compares
        from openvvc_amd import capi
        assert capi.md5(got.tobytes()) == hashlib.md5(oo.packed_frame(y, cb, cr, win)).digest()
    pic.free()
