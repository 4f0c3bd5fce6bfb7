"""GPU: the C-side per-picture flush (ovhip_job_*: page-locked recorder arrays -> async H2D -> launch chain -> D2H of the
refined motion vectors) reproduces the oracle bit for bit, from any thread, and the eager per-row DMVR search returns the
vectors the full pass finally uses."""
import threading

import numpy as np
import pytest

import oracle_lib
import oracle_pipeline
from openvvc_amd import capi, engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built_lib):
    c = engine.Context(0)
    yield c
    c.close()


def _check(wl, planes, what):
    ref, mvs = oracle_pipeline.decode(wl, want_mvs=True)
    for name, a, b in (("Y", planes[0], ref.y), ("Cb", planes[1], ref.cb), ("Cr", planes[2], ref.cr)):
        assert np.array_equal(a, b), f"{what}: plane {name}: {int((a != b).sum())} samples differ"
    return mvs


@pytest.mark.parametrize("w,h,seed", [(416, 240, 0x266), (1920, 1080, 0x266)])
def test_job_flush_matches_oracle(ctx, w, h, seed):
    wl = synth.make_workload(w, h, seed)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    intra = ctx.upload_pic(*wl.intra) if wl.intra is not None else None
    dst = ctx.new_pic(w, h)
    for rep in range(2):                       # the second flush reuses every device buffer
        job.load_workload(wl)
        job.flush(dst, refs, intra)
        job.wait()
        mvs = _check(wl, dst.download(), f"{w}x{h} flush {rep}")
        assert np.array_equal(job.refined_mvs(), mvs)
    st = job.stats()
    assert st.n_launches >= 8 and st.h2d_bytes > wl.coefs.nbytes and st.d2h_bytes == 16 * len(wl.mcx_units)
    assert st.n_edges_v > 0 and st.n_edges_h > 0
    job.close()


def test_job_eager_dmvr_rows(ctx):
    """ovhip_job_dmvr_rows after a part of the refined units has been recorded: the search-only kernel returns for the
    DMVR units exactly the vectors the full flush writes later (what the shim patches into the TMVP planes before a CTU
    row is published)."""
    w, h = 832, 480
    wl = synth.make_workload(w, h, 3)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    intra = ctx.upload_pic(*wl.intra) if wl.intra is not None else None
    dst = ctx.new_pic(w, h)
    job.begin()
    ux = wl.mcx_units
    half = len(ux) // 2
    assert half > 20
    job.rec.append_raw(capi.REC_MCX, ux[:half])
    assert job.dmvr_rows(refs) == half
    early = job.refined_mvs()[:half].copy()
    job.rec.append_raw(capi.REC_MCX, ux[half:])
    assert job.dmvr_rows(refs) == len(ux)
    early2 = job.refined_mvs().copy()
    # now the whole picture
    job.load_workload(wl)
    job.flush(dst, refs, intra)
    job.wait()
    final = job.refined_mvs()
    is_dmvr = (ux["flags"] & 64) != 0
    assert is_dmvr.sum() > 10
    assert np.array_equal(early[is_dmvr[:half]], final[:half][is_dmvr[:half]])
    assert np.array_equal(early2[is_dmvr], final[is_dmvr])
    assert (final[is_dmvr] != np.stack([ux["mv0x"], ux["mv0y"], ux["mv1x"], ux["mv1y"]], axis=1)[is_dmvr]).any()
    _check(wl, dst.download(), "after eager rows")
    job.close()


def test_job_eager_dmvr_rows_in_two_halves(ctx):
    """ovhip_job_dmvr_rows_begin / _collect, the way the shim's row-end hooks use them: a pass is enqueued at the end of a CTU row
    and collected at the end of the next, more units having been recorded in between; vectors AND the TMVP plane entries of every
    pass equal the oracle's for the final vectors of the full flush (entries of DMVR units; the others are OVHIP_TMVP_NONE)."""
    w, h = 832, 480
    wl = synth.make_workload(w, h, 5)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    intra = ctx.upload_pic(*wl.intra) if wl.intra is not None else None
    dst = ctx.new_pic(w, h)
    job.begin()
    ux = wl.mcx_units
    cuts = [0, len(ux) // 5, len(ux) // 2, len(ux) - 7, len(ux)]
    assert cuts[1] > 10
    assert job.dmvr_rows_collect() == 0                               # nothing pending: returns at once
    seen = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        job.rec.append_raw(capi.REC_MCX, ux[a:b])                     # "the next row is parsed" ...
        done = job.dmvr_rows_collect()                                # ... then the pass of the row before is collected
        assert done == a
        seen.append((done, job.refined_mvs()[:done].copy(), job.tmvp_cells()[:4 * done].copy()))
        assert job.dmvr_rows_begin(refs, 7) == b                      # and the pass over the new units enqueued
    assert job.dmvr_rows_collect() == len(ux)
    mv_rows, cells_rows = job.refined_mvs().copy(), job.tmvp_cells().copy()
    job.load_workload(wl)
    job.params.tmvp_cells = 1
    job.flush(dst, refs, intra)
    job.wait()
    final, cells = job.refined_mvs(), job.tmvp_cells()
    is_dmvr = (ux["flags"] & 64) != 0
    assert is_dmvr.sum() > 10
    assert np.array_equal(mv_rows[is_dmvr], final[is_dmvr])
    want = oracle_lib.tmvp_cells(ux, final, 7, (w + 127) // 128)
    assert np.array_equal(cells, want) and np.array_equal(cells_rows, want)
    for done, mv, ce in seen:
        assert np.array_equal(mv[is_dmvr[:done]], final[:done][is_dmvr[:done]]) and np.array_equal(ce, want[:4 * done])
    _check(wl, dst.download(), "after eager rows in two halves")
    job.close()


def test_context_used_from_another_thread(ctx):
    """The HIP current device is per thread: every entry point re-selects the context's device (one context per decoder
    frame thread, possibly created elsewhere)."""
    w, h = 416, 240
    wl = synth.make_workload(w, h, 9)
    out = {}

    def worker():
        try:
            job = engine.Job(ctx, w, h)
            refs = [ctx.upload_pic(*r) for r in wl.refs]
            intra = ctx.upload_pic(*wl.intra) if wl.intra is not None else None
            dst = ctx.new_pic(w, h)
            job.load_workload(wl)
            job.flush(dst, refs, intra)
            job.wait()
            out["planes"] = dst.download()
            job.close()
        except Exception as e:          # noqa: BLE001
            out["err"] = e

    t = threading.Thread(target=worker)
    t.start(); t.join()
    assert "err" not in out, out.get("err")
    _check(wl, out["planes"], "flush from a second thread")


@pytest.mark.parametrize("w,h", [(416, 240), (1920, 1080)])
def test_stage_mask_and_filters_off(ctx, w, h):
    """stages mask / SAO and ALF switched off: dst always holds the result of the last stage that ran.  1920x1080 with only the prediction
    and transform stages on the device is BASELINE configs[1] at its stated size (inverse transform + MC on the GPU, the in-loop filters
    left to the host), compared with the oracle's picture after the same stages."""
    wl = synth.make_workload(w, h, 21)
    job = engine.Job(ctx, w, h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    intra = ctx.upload_pic(*wl.intra) if wl.intra is not None else None
    dst = ctx.new_pic(w, h)
    for stages, names in ((capi.STAGE_MC | capi.STAGE_ITX, ("mc", "itx")),
                          (capi.STAGE_MC | capi.STAGE_ITX | capi.STAGE_DBF, ("mc", "itx", "dbf")),
                          (capi.STAGE_MC | capi.STAGE_ITX | capi.STAGE_DBF | capi.STAGE_SAO, ("mc", "itx", "dbf", "sao"))):
        job.load_workload(wl)
        job.params.stages = stages
        job.flush(dst, refs, intra)
        job.wait()
        ref = oracle_pipeline.decode(wl, stages=names)
        got = dst.download()
        for name, a, b in (("Y", got[0], ref.y), ("Cb", got[1], ref.cb), ("Cr", got[2], ref.cr)):
            assert np.array_equal(a, b), f"stages {names}: plane {name} differs"
    job.close()


@pytest.mark.parametrize("on_host", [1, 0])
def test_picture_waits_for_its_reference_picture(built_lib, on_host):
    """Two pictures on two streams, the second one predicted from the first: ovhip_job_params.wait_events orders its launches
    behind the first picture's completion event -- on the host inside the flush (wait_on_host, what the stream bench does) or as a
    stream wait -- while its uploads are already under way.  Picture 2 == the oracle decoding it from the oracle's picture 1."""
    import ctypes as C
    import torch
    w, h = 832, 480
    dev = torch.device("cuda", 0)
    c1, c2 = engine.Context(0), engine.Context(0)
    s1 = torch.cuda.ExternalStream(c1.stream, device=dev)
    wl1 = synth.make_workload(w, h, 0x51, tools=synth.INTRA_TOOLS, intra_frac=0.2)
    wl2 = synth.make_workload(w, h, 0x52, tools=synth.INTRA_TOOLS, intra_frac=0.2)
    ref1 = oracle_pipeline.decode(wl1)
    wl2.refs[0] = (ref1.y.copy(), ref1.cb.copy(), ref1.cr.copy())          # picture 2's first reference = decoded picture 1
    ref2 = oracle_pipeline.decode(wl2)
    refs1 = [c1.upload_pic(*r) for r in wl1.refs]
    dst1, dst2 = c1.new_pic(w, h), c2.new_pic(w, h)
    refs2 = [dst1] + [c2.upload_pic(*r) for r in wl2.refs[1:]]
    j1, j2 = engine.Job(c1, w, h), engine.Job(c2, w, h)
    for rep in range(3):
        j1.load_workload(wl1)
        j2.load_workload(wl2)
        j1.flush(dst1, refs1, None)
        ev = torch.cuda.Event()
        s1.record_event(ev)
        handles = (C.c_void_p * 1)(ev.cuda_event)
        j2.params.wait_events = C.cast(handles, C.POINTER(C.c_void_p))
        j2.params.n_wait_events = 1
        j2.params.wait_on_host = on_host
        j2.flush(dst2, refs2, None)
        j2.wait()
        j1.wait()
        got = dst2.download()
        for name, a, b in (("Y", got[0], ref2.y), ("Cb", got[1], ref2.cb), ("Cr", got[2], ref2.cr)):
            assert np.array_equal(a, b), f"picture 2 (on_host={on_host}, decode {rep}): plane {name}: {int((a != b).sum())} samples differ"
        j1.begin(); j2.begin()
    j1.close(); j2.close(); c1.close(); c2.close()
