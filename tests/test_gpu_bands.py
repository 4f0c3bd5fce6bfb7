"""GPU: band-wise submission (ovhip_job_band, include/ovvc_hip.h) -- a picture handed to the device band of CTU rows by band of CTU
rows, as the live decoder's row hooks do while the picture is still being parsed (slicedec.c:815-975: the reference reconstructs a CTU
row right after parsing it).  A band's prediction / residual / ordered pass and its own filters run at once (inverse luma mapping,
deblocking, SAO and ALF over the rows the deblocking made final: all but the last 24 of the band); the band's unfiltered bottom row is
set aside for the intra prediction of the band below (the reference's saved lines, rcn_ctu.c:246-510).

Bar: bit-exact.  Bands of one CTU row, of two, of three and ONE band = the whole picture give the picture ovhip_job_flush gives, which is
the oracle's (and, through the fixtures, the reference's)."""
import copy

import numpy as np
import pytest

import oracle_pipeline
from openvvc_amd import capi, engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built_lib):
    c = engine.Context(0)
    yield c
    c.close()


def _same(got, ref, what):
    for name, a, b in (("Y", got[0], ref[0]), ("Cb", got[1], ref[1]), ("Cr", got[2], ref[2])):
        if not np.array_equal(a, b):
            ys = np.nonzero((a != b).any(axis=1))[0]
            raise AssertionError(f"{what}: plane {name}: {int((a != b).sum())} samples differ, rows {ys.min()}..{ys.max()}")


def _workload(w, h, seed, **kw):
    """(the workload as synth made it -- what the oracle decodes --, a copy whose transform blocks are in DECODING order -- what the job
    is loaded with).  synth.Workload keeps its transform blocks as ovhip_rec_tb_cmds_split left them (four classes); a band is a slice of
    the list in decoding order, which the coefficient arena's offsets give back (the arena is appended block by block)."""
    wl = synth.make_workload(w, h, seed, **kw)
    assert wl.ciip_units is None or not len(wl.ciip_units)
    dec = copy.copy(wl)
    tb = np.asarray(wl.tb_cmds).view(capi.TB_CMD_DTYPE).reshape(-1)
    dec.tb_cmds = np.ascontiguousarray(tb[np.argsort(tb["coef_off"], kind="stable")])
    return wl, dec


CASES = [
    # (w, h, seed, kwargs): B pictures with every inter tool + LMCS, with intra CUs / ISP (ordered tasks in most bands), an I picture
    (832, 480, 0x266, dict(tools=synth.INTRA_TOOLS, intra_frac=0.2)),
    (1920, 1080, 7, dict(tools=synth.INTRA_TOOLS, intra_frac=0.12)),
    (832, 480, 11, dict(tools=tuple(t for t in synth.INTRA_TOOLS if t != "lmcs"), intra_frac=0.3)),
    (416, 240, 5, dict(tools=synth.INTRA_TOOLS, intra_frac=1.0)),
    (1920, 1080, 9, dict(tools=synth.INTRA_TOOLS, intra_frac=1.0)),
]


@pytest.mark.parametrize("w,h,seed,kw", CASES)
def test_bands_give_the_picture_of_the_whole_flush(ctx, w, h, seed, kw):
    orig, wl = _workload(w, h, seed, **kw)
    ref = oracle_pipeline.decode(orig)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    job.flush(dst, refs, None)
    job.wait()
    whole = dst.download()
    _same(whole, (ref.y, ref.cb, ref.cr), f"{w}x{h} whole-picture flush vs oracle")
    n_rows = (h + 127) // 128
    for per_band in (1, 2, 3, n_rows):
        dst.upload(*[np.full_like(p, 0x155) for p in whole])              # nothing of the previous decode may survive unnoticed
        job.load_workload(wl)
        job.flush_in_bands(wl, dst, refs, per_band)
        job.wait()
        assert job.band_progress() == h
        _same(dst.download(), whole, f"{w}x{h}, bands of {per_band} CTU row(s)")
    st = job.stats()
    assert st.n_h2d <= n_rows + 1 and st.n_tb == len(wl.tb_cmds)
    job.close()


@pytest.mark.parametrize("stages", [capi.STAGE_MC | capi.STAGE_ITX | capi.STAGE_INTRA,
                                    capi.STAGE_MC | capi.STAGE_ITX | capi.STAGE_INTRA | capi.STAGE_DBF,
                                    capi.STAGE_MC | capi.STAGE_ITX | capi.STAGE_INTRA | capi.STAGE_DBF | capi.STAGE_SAO,
                                    capi.STAGE_MC | capi.STAGE_ITX | capi.STAGE_INTRA | capi.STAGE_DBF | capi.STAGE_ALF,
                                    capi.STAGE_MC | capi.STAGE_ITX | capi.STAGE_INTRA | capi.STAGE_DBF | capi.STAGE_SAO | capi.STAGE_ALF | capi.STAGE_INTRA_LEVELS])
def test_bands_with_stages_off(ctx, stages):
    """filters switched off one by one (the rows a band makes final then come from another stage), and the ordered pass as one launch
    per level (what a band falls back to when the device's flow budget is spent)"""
    w, h = 832, 480
    _, wl = _workload(w, h, 21, tools=synth.INTRA_TOOLS, intra_frac=0.2)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    job.params.stages = stages
    job.flush(dst, refs, None)
    job.wait()
    whole = dst.download()
    for per_band in (1, 2):
        dst.upload(*[np.full_like(p, 0x2aa) for p in whole])
        job.load_workload(wl)
        job.params.stages = stages
        job.flush_in_bands(wl, dst, refs, per_band)
        job.wait()
        _same(dst.download(), whole, f"stages {stages:#x}, bands of {per_band}")
    job.close()


def test_band_rows_become_final_in_order(ctx):
    """ovhip_job_band_progress: the rows a reader may use grow band by band, lag the parse by the filters' reach, and the rows reported
    final ARE final -- compared with the finished picture after every band (the stream is drained before each look)"""
    w, h = 1920, 1080
    _, wl = _workload(w, h, 3, tools=synth.INTRA_TOOLS, intra_frac=0.12)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    job.flush(dst, refs, None)
    job.wait()
    whole = dst.download()
    dst.upload(*[np.zeros_like(p) for p in whole])
    job.load_workload(wl)
    rows = list(range(128, h, 128))
    cuts = engine.band_counts(wl, rows)
    seen = []
    for r, c in zip(rows, cuts):
        job.band(dst, refs, r, False, c)
        ctx.sync()
        final = job.band_progress()
        seen.append(final)
        assert final == max(0, r - 24) and final % 8 == 0          # deblocking leaves the 8 rows above a band's end, SAO / ALF follow in steps of 8
        got = dst.download()
        _same([got[0][:final], got[1][:final // 2], got[2][:final // 2]], [whole[0][:final], whole[1][:final // 2], whole[2][:final // 2]],
              f"rows reported final after the band ending at {r}")
    job.band(dst, refs, h, True, None)
    job.wait()
    assert job.band_progress() == h and seen == sorted(seen) and seen[-1] > 0
    _same(dst.download(), whole, "finished picture")
    job.close()
