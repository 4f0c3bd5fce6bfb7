"""GPU: full recorded pictures (synthetic, seeded) through the HIP engine vs the CPU oracle,
bit-exact, plus size-independent properties at BASELINE.json's full size."""
import hashlib

import numpy as np
import pytest

import oracle_pipeline
from openvvc_amd import engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built_lib):
    c = engine.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("w,h,seed", [(416, 240, 0x266), (416, 240, 7), (1920, 1080, 0x266)])
def test_picture_matches_oracle(ctx, w, h, seed):
    wl = synth.make_workload(w, h, seed)
    rp = engine.ResidentPicture(ctx, wl)
    rp.decode()
    y, cb, cr = rp.result()
    ref = oracle_pipeline.decode(wl)
    for name, a, b in (("Y", y, ref.y), ("Cb", cb, ref.cb), ("Cr", cr, ref.cr)):
        assert np.array_equal(a, b), f"{w}x{h} seed {seed}: plane {name}: {int((a != b).sum())} samples differ"
    rp.free()


def test_4k_idempotent_and_band_exact(ctx):
    """3840x2160 (BASELINE configs[3]): decoding twice gives the same MD5 (every stage rewrites its
    whole output), and the top two CTU rows equal the oracle run on that band."""
    wl = synth.make_workload(3840, 2160, 0x266)
    rp = engine.ResidentPicture(ctx, wl)
    rp.decode()
    a = rp.result()
    rp.decode()
    b = rp.result()
    md5 = lambda planes: hashlib.md5(b"".join(p.tobytes() for p in planes)).hexdigest()
    assert md5(a) == md5(b)
    ref = oracle_pipeline.decode(wl, rows=(0, 256))
    assert np.array_equal(a[0][:256], ref.y[:256])
    assert np.array_equal(a[1][:128], ref.cb[:128]) and np.array_equal(a[2][:128], ref.cr[:128])
    rp.free()
