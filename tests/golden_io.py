"""Reader of the "OVG1" fixture container written by oracle/ref_harness/ref_common.h."""
from pathlib import Path
import struct
import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"
_DT = {0: "u1", 1: "<i2", 2: "<u2", 3: "<i4", 4: "<u4", 5: "<u8", 6: "i1"}


def load(name: str, directory=None) -> dict:
    raw = (Path(directory) / name if directory is not None else GOLDEN / name).read_bytes()
    magic, n = struct.unpack_from("<II", raw, 0)
    assert magic == 0x3147564F, "not an OVG1 file"
    off, out = 8, {}
    for _ in range(n):
        nm = raw[off:off + 32].split(b"\0")[0].decode(); off += 32
        t, nd, d0, d1, d2, d3 = struct.unpack_from("<6I", raw, off); off += 24
        dims = (d0, d1, d2, d3)[:nd]
        dt = np.dtype(_DT[t])
        cnt = int(np.prod(dims))
        out[nm] = np.frombuffer(raw, dtype=dt, count=cnt, offset=off).reshape(dims).copy()
        off += cnt * dt.itemsize
    return out
