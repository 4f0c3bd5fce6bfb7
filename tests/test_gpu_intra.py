"""GPU: the ordered pass (k_intra_level) against the reference's intra slots (tests/golden/intra.ovg) and, as part of the
per-picture flush, against the oracle on recorded pictures with intra CUs (B-like pictures and an I picture)."""
import numpy as np
import pytest

import golden_io
import oracle_pipeline
from openvvc_amd import capi, engine, synth

pytestmark = pytest.mark.gpu
BAND = 384            # a multiple of the CTU size, so that CTU-relative tests (CCLM's first line) see the fixture's alignment


@pytest.fixture(scope="module")
def ctx(built_lib):
    c = engine.Context(0)
    yield c
    c.close()


def test_intra_tasks_gpu_match_reference(ctx):
    """6506 cases of intra_pred / intra_pred_mrl / mip.rcn_intra_mip / intra_pred_c (+ cclm.*): every case predicts on its own
    copy of the picture (one band of a tall picture), 160 cases per launch (ovhip_itask.y is 16 bits)."""
    g = golden_io.load("intra.ovg")
    tasks = np.frombuffer(g["task"].tobytes(), dtype=capi.ITASK_DTYPE)
    H, W = g["pic_y"].shape
    base = [np.zeros((BAND, W), np.uint16), np.zeros((BAND // 2, W // 2), np.uint16), np.zeros((BAND // 2, W // 2), np.uint16)]
    base[0][:H] = g["pic_y"]; base[1][:H // 2] = g["pic_cb"]; base[2][:H // 2] = g["pic_cr"]
    NB = 160
    tall_planes = [np.tile(p, (NB, 1)) for p in base]
    res = ctx.new_pic(W, BAND * NB)
    bad = []
    for b0 in range(0, len(tasks), NB):
        t = tasks[b0:b0 + NB].copy()
        k = np.arange(len(t))
        t["y"] += np.where(t["kind"] == capi.IT_LUMA, k * BAND, k * (BAND // 2)).astype(np.uint16)
        pic = ctx.upload_pic(*tall_planes)
        ctx.intra_level(pic, res, ctx.upload(t), 0, len(t))
        ctx.sync()
        y, cb, cr = pic.download()
        pic.free()
        for i in range(len(t)):
            tt = t[i]
            w, h, x, yy = 1 << int(tt["log2_w"]), 1 << int(tt["log2_h"]), int(tt["x"]), int(tt["y"])
            eo = g["exp_off"][b0 + i]
            if tt["kind"] == capi.IT_LUMA:
                ok = np.array_equal(y[yy:yy + h, x:x + w], g["exp"][eo[0]:eo[0] + w * h].reshape(h, w))
            else:
                ok = (np.array_equal(cb[yy:yy + h, x:x + w], g["exp"][eo[0]:eo[0] + w * h].reshape(h, w))
                      and np.array_equal(cr[yy:yy + h, x:x + w], g["exp"][eo[1]:eo[1] + w * h].reshape(h, w)))
            if not ok:
                bad.append((b0 + i, int(tt["kind"]), int(tt["mode"]), w, h, int(tt["flags"]), int(tt["avl_lft"]), int(tt["avl_abv"]), int(tt["mrl_idx"])))
    assert not bad, f"{len(bad)} / {len(tasks)} intra cases differ from the reference on the GPU, first: {bad[:8]}"


@pytest.mark.parametrize("w,h,seed,frac", [(416, 240, 5, 0.12), (416, 240, 6, 1.0), (832, 480, 7, 0.3), (1920, 1080, 0x266, 0.12)])
def test_picture_with_intra_matches_oracle(ctx, w, h, seed, frac):
    """Recorded pictures with intra CUs (MIP, MRL, BDPCM, CCLM / MDLM, CIIP blended on the device, ordered chroma-scale
    regions) through the C flush: level-ordered launches == the oracle's decoding-order execution, all stages."""
    wl = synth.make_workload(w, h, seed, tools=synth.INTRA_TOOLS, intra_frac=frac)
    assert wl.stats["n_itasks"] > 50 and wl.stats["n_ilevels"] > 5
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    job.flush(dst, refs, None)
    job.wait()
    got = dst.download()
    ref, mvs = oracle_pipeline.decode(wl, want_mvs=True)
    for name, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)):
        assert np.array_equal(a, b), f"{w}x{h} intra_frac {frac}: plane {name}: {int((a != b).sum())} samples differ"
    st = job.stats()
    assert st.n_itasks == wl.stats["n_itasks"] and st.n_ilevels == wl.stats["n_ilevels"]
    if mvs is not None:
        assert np.array_equal(job.refined_mvs(), mvs)
    job.close()
