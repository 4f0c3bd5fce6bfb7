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


def _inside_ctu_geometry(t):
    """The fixture draws availability at random; a decoder never marks samples available that lie right of the CTU below its
    first row, or below the CTU (not decoded yet) -- the only samples the CTU tile does not hold."""
    chroma = t["kind"] != capi.IT_LUMA
    S, unit = (64, 2) if chroma else (128, 4)
    x0, y0 = int(t["x"]), int(t["y"])
    X0, Y0 = x0 & ~(S - 1), y0 & ~(S - 1)
    mrl = 0 if (chroma or t["flags"] & capi.IF_MIP) else int(t["mrl_idx"])
    if y0 == Y0 and mrl:
        return False
    if y0 > Y0 and x0 + unit * int(t["avl_abv"]) > X0 + S:
        return False
    return y0 + unit * int(t["avl_lft"]) <= Y0 + S


def test_intra_tasks_ctu_kernel_match_reference(ctx):
    """The same cases through the one-launch pass (k_intra_ctu): every case is the only task of its CTU (no waits), the
    picture comes from / goes back through the CTU tile in LDS."""
    g = golden_io.load("intra.ovg")
    tasks = np.frombuffer(g["task"].tobytes(), dtype=capi.ITASK_DTYPE)
    H, W = g["pic_y"].shape
    base = [np.zeros((BAND, W), np.uint16), np.zeros((BAND // 2, W // 2), np.uint16), np.zeros((BAND // 2, W // 2), np.uint16)]
    base[0][:H] = g["pic_y"]; base[1][:H // 2] = g["pic_cb"]; base[2][:H // 2] = g["pic_cr"]
    NB = 160
    tall_planes = [np.tile(p, (NB, 1)) for p in base]
    res = ctx.new_pic(W, BAND * NB)
    sync = ctx.upload(np.zeros(int(ctx.lib.ovhip_intra_sync_words(W, BAND * NB, 7)), np.uint32))
    bad, n_checked = [], 0
    for epoch, b0 in enumerate(range(0, len(tasks), NB), 1):
        t = tasks[b0:b0 + NB].copy()
        k = np.arange(len(t))
        t["y"] += np.where(t["kind"] == capi.IT_LUMA, k * BAND, k * (BAND // 2)).astype(np.uint16)
        rec = capi.Recorder(W, BAND * NB)
        rec.append_raw(capi.REC_ITASK, t)
        ts, cs = rec.itasks_by_ctu(7)
        rec.close()
        assert len(cs) == len(t) and not cs["deps"].any()
        pic = ctx.upload_pic(*tall_planes)
        ctx.intra_ctu(pic, res, ctx.upload(ts), ctx.upload(cs), len(cs), sync, epoch)
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
            if not ok and _inside_ctu_geometry(tasks[b0 + i]):
                bad.append((b0 + i, int(tt["kind"]), int(tt["mode"]), w, h, int(tt["flags"]), int(tt["avl_lft"]), int(tt["avl_abv"]), int(tt["mrl_idx"])))
            n_checked += int(_inside_ctu_geometry(tasks[b0 + i]))
        # nothing but the task's block may change
        if not bad:
            for a, b, name in ((y, tall_planes[0], "Y"), (cb, tall_planes[1], "Cb"), (cr, tall_planes[2], "Cr")):
                d = a != b
                for i in range(len(t)):
                    tt = t[i]
                    if (tt["kind"] == capi.IT_LUMA) == (name == "Y"):
                        d[int(tt["y"]):int(tt["y"]) + (1 << int(tt["log2_h"])), int(tt["x"]):int(tt["x"]) + (1 << int(tt["log2_w"]))] = False
                assert not d.any(), f"plane {name}: samples outside the tasks' blocks changed"
    assert n_checked > 5500
    assert not bad, f"{len(bad)} / {n_checked} intra cases differ from the reference through k_intra_ctu, first: {bad[:8]}"


@pytest.mark.parametrize("one_launch", [0, capi.STAGE_INTRA_CTU, capi.STAGE_INTRA_LEVELS])
@pytest.mark.parametrize("w,h,seed,frac", [(416, 240, 5, 0.12), (416, 240, 6, 1.0), (832, 480, 7, 0.3), (1920, 1080, 0x266, 0.12), (1920, 1080, 0x267, 1.0)])
def test_picture_with_intra_matches_oracle(ctx, w, h, seed, frac, one_launch):
    """Recorded pictures with intra CUs (MIP, MRL, BDPCM, CCLM / MDLM, CIIP blended on the device, ordered chroma-scale
    regions) through the C flush, ordered pass as one launch per level (default) and as the one-launch CTU wavefront: both == the
    oracle's decoding-order execution, all stages."""
    wl = synth.make_workload(w, h, seed, tools=synth.INTRA_TOOLS, intra_frac=frac)
    assert wl.stats["n_itasks"] > 50 and wl.stats["n_ilevels"] > 5
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    if one_launch:
        job.params.stages = capi.STAGE_ALL | one_launch
    job.flush(dst, refs, None)
    job.wait()
    got = dst.download()
    ref, mvs = oracle_pipeline.decode(wl, want_mvs=True)
    for name, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)):
        assert np.array_equal(a, b), f"{w}x{h} intra_frac {frac} one_launch={one_launch}: plane {name}: {int((a != b).sum())} samples differ"
    st = job.stats()
    assert st.n_itasks == wl.stats["n_itasks"] and st.n_ilevels == wl.stats["n_ilevels"]
    if mvs is not None:
        assert np.array_equal(job.refined_mvs(), mvs)
    job.close()


@pytest.mark.gpu
def test_second_pass_after_an_abandoned_flow_launch(ctx, monkeypatch):
    """ovhip_job_wait decodes the picture again with one launch per level when the flow launch reports an expired wait (forced
    here through the library's test hook): same samples and vectors as the oracle, and the statistics say so."""
    w, h = 832, 480
    wl = synth.make_workload(w, h, 0x31, tools=synth.INTRA_TOOLS, intra_frac=0.3)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    job.test_abort_next_flow()            # the first pass is abandoned for real: the device's abort word is set before the launch
    job.flush(dst, refs, None)
    job.wait()
    assert job.stats().n_ordered_retries == 1
    got = dst.download()
    ref, mvs = oracle_pipeline.decode(wl, want_mvs=True)
    for name, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)):
        assert np.array_equal(a, b), f"second pass: plane {name}: {int((a != b).sum())} samples differ"
    assert np.array_equal(job.refined_mvs(), mvs)
    # and the next picture of the job takes the flow launch again
    job.begin(); job.load_workload(wl)
    job.flush(dst, refs, None); job.wait()
    assert job.stats().n_ordered_retries == 0
    assert np.array_equal(dst.download()[0], ref.y)


@pytest.mark.gpu
def test_resident_replay_after_a_second_pass(ctx):
    """A resident replay (OVHIP_STAGE_RESIDENT: no uploads, the device copies of the last full flush) of a job whose last full flush was
    a SECOND pass: that flush uploaded no item list for the flow launch (per-level launches), so the replay must not take the flow
    launch either.  (Round 5: bench.py's resident variant failed with "ovhip_intra_flow_launch: bad arguments" in one run of ten --
    whenever a picture of the main run had fallen into a second pass.)"""
    w, h = 832, 480
    wl = synth.make_workload(w, h, 0x31, tools=synth.INTRA_TOOLS, intra_frac=0.3)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    job.test_abort_next_flow()
    job.flush(dst, refs, None); job.wait()
    assert job.stats().n_ordered_retries == 1
    ref = oracle_pipeline.decode(wl)
    res = job.make_params(wl, stages=capi.STAGE_ALL | capi.STAGE_RESIDENT)
    for _ in range(2):
        dst2 = ctx.new_pic(w, h)
        job.flush(dst2, refs, None, params=res); job.wait()
        got = dst2.download()
        for name, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)):
            assert np.array_equal(a, b), f"resident replay after a second pass: plane {name}: {int((a != b).sum())} samples differ"
    # a full flush brings the flow launch back, and a replay of THAT takes it too
    job.begin(); job.load_workload(wl)
    job.flush(dst, refs, None); job.wait()
    assert job.stats().n_ordered_retries == 0
    job.flush(dst, refs, None, params=res); job.wait()
    assert job.stats().n_ordered_retries == 0 and np.array_equal(dst.download()[0], ref.y)
    job.close()


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [1, 3, 64, 100000])
def test_flow_launch_with_few_workers(ctx, workers):
    """The ordered pass as W persistent workers (ovhip_job_params.flow_workers): worker b takes the items b, b + W, ... in level
    order.  ONE worker walks the whole chain by itself, three wait for each other across hundreds of items each, 100000 is one
    workgroup per item: an I picture and a B picture with intra CUs == the oracle every time, no second pass."""
    for w, h, seed, frac in ((416, 240, 21, 1.0), (832, 480, 22, 0.25)):
        wl = synth.make_workload(w, h, seed, tools=synth.INTRA_TOOLS, intra_frac=frac)
        job = engine.Job(ctx, w, h)
        refs = [ctx.upload_pic(*r) for r in wl.refs]
        dst = ctx.new_pic(w, h)
        job.load_workload(wl)
        job.params.flow_workers = workers
        job.flush(dst, refs, None)
        job.wait()
        assert job.stats().n_ordered_retries == 0
        got = dst.download()
        ref = oracle_pipeline.decode(wl)
        for name, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)):
            assert np.array_equal(a, b), f"{w}x{h}, {workers} workers: plane {name}: {int((a != b).sum())} samples differ"
        job.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed,frac", [(416, 240, 11, 1.0), (832, 480, 12, 0.3)])
def test_picture_with_intra_without_lmcs(ctx, w, h, seed, frac):
    """The flow launch hands luma over with a tag bit that the inverse luma mapping drops; a picture without LMCS has no such pass
    and gets the bit cleared by k_flow_untag: I and B pictures without LMCS == the oracle."""
    tools = tuple(t for t in synth.INTRA_TOOLS if t != "lmcs")
    wl = synth.make_workload(w, h, seed, tools=tools, intra_frac=frac)
    assert wl.lmcs is None and wl.stats["n_itasks"] > 50
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    job.flush(dst, refs, None)
    job.wait()
    got = dst.download()
    ref = oracle_pipeline.decode(wl)
    for name, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)):
        assert np.array_equal(a, b), f"{w}x{h} without LMCS: plane {name}: {int((a != b).sum())} samples differ (max {int(a.max())})"


@pytest.mark.gpu
def test_4k_intra_picture_matches_oracle(ctx):
    """The configuration the stream's time hangs on: a 3840x2160 picture with every CU intra (69k ordered tasks, ~2100 levels)
    through the flow launches (tagged hand-over, four launches of 32768 items, riders) == the oracle, decoded three times into the
    same buffers (a new epoch each time, the previous picture's samples underneath)."""
    w, h = 3840, 2160
    wl = synth.make_workload(w, h, 0x268, tools=synth.INTRA_TOOLS, intra_frac=1.0)
    assert wl.stats["n_ilevels"] > 1500
    ref = oracle_pipeline.decode(wl)
    job = engine.Job(ctx, w, h)
    dst = ctx.new_pic(w, h)
    for rep in range(3):
        job.load_workload(wl)
        job.flush(dst, [], None)
        job.wait()
        assert job.stats().n_ordered_retries == 0
        got = dst.download()
        for name, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)):
            assert np.array_equal(a, b), f"4K I picture, decode {rep}: plane {name}: {int((a != b).sum())} samples differ"
        job.begin()


@pytest.mark.gpu
def test_4k_b_picture_with_intra_matches_oracle(ctx):
    """the picture type bench.py times -- 3840x2160, every coding tool, 12 % of the CUs intra in clusters (the
    bench's own seed) -- at ITS size against the oracle: all stages, the refined vectors, and a second decode into the same
    buffers through a real abandoned first pass."""
    w, h = 3840, 2160
    wl = synth.make_workload(w, h, 0x266, tools=synth.INTRA_TOOLS, intra_frac=0.12)
    assert wl.stats["n_itasks"] > 5000 and wl.stats["cu_modes"]["intra"] > 500 and wl.stats["n_mcx_units"] > 1000
    ref, mvs = oracle_pipeline.decode(wl, want_mvs=True)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    for rep in range(2):
        job.load_workload(wl)
        if rep:
            job.test_abort_next_flow()
        job.flush(dst, refs, None)
        job.wait()
        assert job.stats().n_ordered_retries == rep
        got = dst.download()
        for name, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)):
            assert np.array_equal(a, b), f"4K B picture with intra, decode {rep}: plane {name}: {int((a != b).sum())} samples differ"
        assert np.array_equal(job.refined_mvs(), mvs)
        job.begin()
    job.close()
