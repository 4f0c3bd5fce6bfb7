"""GPU: the frame-thread layer of the library -- device DPB + ovhip_frame_* + the C stream driver (ovvc_dpb.c, ovvc_frame.c,
ovvc_stream.c) -- decoding streams of dependent pictures with several pictures in flight, checked picture by picture against
the ORACLE decoding the same stream one picture at a time (its own outputs as reference pictures), not against itself."""
import ctypes as C
import hashlib
import sys
from pathlib import Path

import numpy as np
import pytest

import oracle_lib
import oracle_pipeline
from openvvc_amd import capi, engine, gop, synth

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import ovvc_oracle_output as oo                                                        # noqa: E402

pytestmark = pytest.mark.gpu


def _contents(w, h, seeds_b, seed_i, intra_frac=0.15, calllog=False):
    """B-picture contents + one I-picture content (last)"""
    wls = [synth.make_workload(w, h, s, tools=synth.INTRA_TOOLS, intra_frac=intra_frac, calllog=calllog) for s in seeds_b]
    wls.append(synth.make_workload(w, h, seed_i, tools=synth.INTRA_TOOLS, intra_frac=1.0, calllog=calllog))
    return wls


def _stream_pics(pics, n_b):
    """gop.Picture list -> stream picture dicts: I pictures show the last content, B pictures rotate over the others"""
    out = []
    for p in pics:
        out.append({"content": n_b if p.intra else p.idx % n_b, "job": p.idx, "poc": p.poc, "refs": p.refs, "device": 0})
    return out


def _oracle_stream(wls, spics):
    """the stream decoded by the oracle, one picture at a time in decoding order: planes per picture"""
    planes = []
    for p in spics:
        wl = wls[p["content"]]
        saved = list(wl.refs)
        if p["refs"]:
            for k in range(len(wl.refs)):
                wl.refs[k] = planes[p["refs"][k % len(p["refs"])]]
        o = oracle_pipeline.decode(wl)
        wl.refs[:] = saved
        planes.append((o.y.copy(), o.cb.copy(), o.cr.copy()))
    return planes


def _jobs_for(ctx, wls, spics, w, h):
    jobs = []
    for p in spics:
        j = engine.Job(ctx, w, h)
        j.load_workload(wls[p["content"]])
        jobs.append(j)
    return jobs


class _Keep:
    """make_params wants an object with a _keep dict to park the numpy arrays in"""
    def __init__(self):
        self._keep = {}


_KEEPERS = []


def _make_contents(wls):
    out = []
    for wl in wls:
        k = _Keep()
        _KEEPERS.append(k)
        out.append({"params": engine.Job.make_params(k, wl), "calllog": wl.calllog, "n_ref_slots": len(wl.refs)})
    return out


def _compare(stream, ctx, planes, idxs, what):
    for i in idxs:
        got = stream.picture(i, ctx).download()
        for name, a, b in zip(("Y", "Cb", "Cr"), got, planes[i]):
            assert np.array_equal(a, b), f"{what}: picture {i} plane {name}: {int((a != b).sum())} samples differ from the oracle"


def test_gop_of_16_in_flight_matches_oracle_picture_by_picture(built_lib):
    """one GOP of 16 pictures in flight at 832x480 checked picture by picture against the oracle."""
    w, h = 832, 480
    wls = _contents(w, h, (0x266, 0x1266, 0x2266), 0x9266)
    pics = gop.build_stream(2, 16, 32, 1)                           # I, then two GOPs of 16 (33 pictures)
    spics = _stream_pics(pics, 3)
    planes = _oracle_stream(wls, spics)
    ctx = engine.Context(0)
    dpb = engine.Dpb((0,))
    jobs = _jobs_for(ctx, wls, spics, w, h)
    st = engine.Stream(dpb, w, h, _make_contents(wls), jobs, threads_per_device=16)
    arr = st.pics_array(spics)
    for rep in range(2):                                            # the second decode re-uses every buffer of the first
        res, dg = st.run(arr, len(spics), 0, len(spics), flags=capi.STREAM_KEEP | capi.STREAM_HOLD_ALL, digests=True)
        assert res.status == 0 and res.n_decoded == len(spics)
        _compare(st, ctx, planes, range(len(spics)), f"16 in flight, decode {rep}")
        # the per-picture digest of the run = the digest of the oracle's picture
        for i in range(len(spics)):
            assert bytes(dg[i]) == oo.picture_digest(*planes[i]), i
    stats = dpb.stats()
    assert stats.n_alloc <= 40 and stats.n_recycled >= 20, (stats.n_alloc, stats.n_recycled)
    st.close(); [j.close() for j in jobs]; dpb.close(); ctx.close()


def test_second_pass_on_a_reference_picture_while_its_readers_are_in_flight(built_lib):
    """VERDICT r2 weak #2 / ADVICE (high, medium): picture A's flow launch is abandoned FOR REAL (abort word set before the launch:
    the first pass leaves A incomplete and partly tagged); B and C, which reference A, are already taken by other frame threads.
    They must be decoded from the picture ovhip_job_wait's second pass produced -- i.e. start only after A was published --
    and A itself must come out right."""
    w, h = 832, 480
    wls = _contents(w, h, (0x31, 0x32), 0x33, intra_frac=0.3)
    spics = [{"content": 2, "job": 0, "poc": 0, "refs": []},            # A: I picture
             {"content": 0, "job": 1, "poc": 8, "refs": [0, 0]},        # B <- A
             {"content": 1, "job": 2, "poc": 4, "refs": [0, 1]},        # C <- A, B
             {"content": 0, "job": 3, "poc": 2, "refs": [0, 2]}]        # D <- A, C
    planes = _oracle_stream(wls, spics)
    ctx = engine.Context(0)
    dpb = engine.Dpb((0,))
    jobs = _jobs_for(ctx, wls, spics, w, h)
    st = engine.Stream(dpb, w, h, _make_contents(wls), jobs, threads_per_device=4)
    arr = st.pics_array(spics)
    for rep in range(3):
        jobs[0].test_abort_next_flow()
        if rep == 2:
            jobs[2].test_abort_next_flow()                          # ... and a B picture with intra CUs in the chain as well
        res, _ = st.run(arr, len(spics), 0, len(spics), flags=capi.STREAM_KEEP | capi.STREAM_HOLD_ALL)
        assert res.n_second_passes == (2 if rep == 2 else 1), res.n_second_passes
        _compare(st, ctx, planes, range(len(spics)), f"abandoned first pass, decode {rep}")
    st.close(); [j.close() for j in jobs]; dpb.close(); ctx.close()


def test_recording_inside_the_run_equals_prerecorded_jobs(built_lib):
    """OVHIP_STREAM_RECORD: every frame thread replays the picture's call log into its own job (what a parse thread does), no
    pre-recorded job involved: same pictures."""
    w, h = 416, 240
    wls = _contents(w, h, (5, 6), 7, calllog=True)
    pics = gop.build_stream(3, 8, 16, 1)
    spics = _stream_pics(pics, 2)
    planes = _oracle_stream(wls, spics)
    ctx = engine.Context(0)
    dpb = engine.Dpb((0,))
    st = engine.Stream(dpb, w, h, _make_contents(wls), [], threads_per_device=6, flags=capi.STREAM_RECORD)
    arr = st.pics_array(spics)
    res, _ = st.run(arr, len(spics), 0, len(spics), flags=capi.STREAM_KEEP | capi.STREAM_HOLD_ALL)
    assert res.n_decoded == len(spics) and res.record_seconds > 0
    _compare(st, ctx, planes, range(len(spics)), "recorded in the run")
    st.close(); dpb.close(); ctx.close()


def test_continued_stream_and_dpb_churn(built_lib):
    """A 97-picture stream decoded in three runs that continue each other (warm-up / timed split of bench.py), pictures released
    as soon as their last reader is done: the DPB holds a GOP's worth of buffers, not the stream's."""
    w, h = 416, 240
    wls = _contents(w, h, (11, 12, 13), 14)
    pics = gop.build_stream(6, 16, 32, 1)
    spics = _stream_pics(pics, 3)
    # 40 jobs in rotation for the B pictures (a job holds one content for ever: job k <-> content k % 3), two for the I pictures
    n_i = 0
    for p, q in zip(spics, pics):
        if q.intra:
            p["job"], p["content"] = 40 + n_i % 2, 3
            n_i += 1
        else:
            p["job"] = q.idx % 40
            p["content"] = p["job"] % 3
    planes = _oracle_stream(wls, spics)
    ctx = engine.Context(0)
    dpb = engine.Dpb((0,))
    jobs = []
    for k in range(42):
        j = engine.Job(ctx, w, h)
        j.load_workload(wls[k % 3 if k < 40 else 3])
        jobs.append(j)
    st = engine.Stream(dpb, w, h, _make_contents(wls), jobs, threads_per_device=8)
    arr = st.pics_array(spics)
    n = len(spics)
    dgs = []
    for first, cnt in ((0, 17), (17, 48), (65, n - 65)):
        res, dg = st.run(arr, n, first, cnt, flags=capi.STREAM_KEEP, digests=True)
        assert res.n_decoded == cnt
        dgs.append(dg)
    dg = np.concatenate(dgs)
    for i in range(n):
        assert bytes(dg[i]) == oo.picture_digest(*planes[i]), f"picture {i}"
    stats = dpb.stats()
    assert stats.n_begin == n and stats.n_alloc <= 30, f"{stats.n_alloc} device pictures allocated for a {n}-picture stream"
    assert stats.n_live == 0, stats.n_live           # the stream ended: nothing could still be referenced
    st.close()
    assert dpb.stats().n_live == 0
    [j.close() for j in jobs]; dpb.close(); ctx.close()


def test_two_logical_devices_in_one_process(built_lib):
    """SURVEY 8e in the product: pictures dealt to two logical devices (both on GPU 0: the peer copy degenerates to a
    device-to-device copy), a reference picture decoded on the other device arrives by an event-ordered copy pushed to exactly
    the devices that list it."""
    w, h = 416, 240
    wls = _contents(w, h, (21, 22), 23)
    pics = gop.build_stream(3, 8, 16, 1)
    spics = _stream_pics(pics, 2)
    for p, q in zip(spics, pics):
        p["device"] = q.idx % 2                                     # picture-interleaved dealing (picture k -> device k mod G)
    planes = _oracle_stream(wls, spics)
    ctx = engine.Context(0)
    dpb = engine.Dpb((0, 0))
    jobs = _jobs_for(ctx, wls, spics, w, h)
    st = engine.Stream(dpb, w, h, _make_contents(wls), jobs, threads_per_device=3)
    arr = st.pics_array(spics)
    res, dg = st.run(arr, len(spics), 0, len(spics), flags=capi.STREAM_KEEP, digests=True)
    for i in range(len(spics)):
        assert bytes(dg[i]) == oo.picture_digest(*planes[i]), f"picture {i} (device {spics[i]['device']})"
    stats = dpb.stats()
    used = lambda p: {p["refs"][k % len(p["refs"])] for k in range(2)} if p["refs"] else set()        # the table has two entries
    cross = len({(r, p["device"]) for p in spics for r in used(p) if spics[r]["device"] != p["device"]})
    assert stats.n_copies == cross, (stats.n_copies, cross)        # one copy per (picture, device that lists it), none else
    assert stats.copy_bytes == cross * w * h * 3
    st.close(); [j.close() for j in jobs]; dpb.close(); ctx.close()


def test_output_thread_writes_the_file_dectest_would(built_lib):
    """OVHIP_OUT_PACKED + OVHIP_STREAM_FILE_MD5: the frames leave in output (POC) order, cropped and packed; the MD5 over them is
    the md5sum of the file examples/dectest.c writes (what CI/checkMD5.sh compares)."""
    w, h = 416, 240
    win = (1, 2, 0, 3)
    wls = _contents(w, h, (41, 42), 43)
    pics = gop.build_stream(2, 8, 16, 1)
    spics = _stream_pics(pics, 2)
    planes = _oracle_stream(wls, spics)
    order = sorted(range(len(spics)), key=lambda i: spics[i]["poc"])
    want = hashlib.md5(b"".join(oo.packed_frame(*planes[i], win) for i in order)).digest()
    ctx = engine.Context(0)
    dpb = engine.Dpb((0,))
    jobs = _jobs_for(ctx, wls, spics, w, h)
    st = engine.Stream(dpb, w, h, _make_contents(wls), jobs, threads_per_device=4, output=capi.OUT_PACKED, window=win)
    arr = st.pics_array(spics)
    res, _ = st.run(arr, len(spics), 0, len(spics), flags=capi.STREAM_FILE_MD5)
    assert res.out_frames == len(spics) and res.out_bytes == len(spics) * len(oo.packed_frame(*planes[0], win))
    assert bytes(res.out_md5) == want
    assert dpb.stats().n_live == 0                                  # decode, readers and output all dropped their holds
    st.close()
    st = engine.Stream(dpb, w, h, _make_contents(wls), jobs, threads_per_device=4, output=capi.OUT_DIGEST, window=win)
    res, _ = st.run(arr, len(spics), 0, len(spics))
    assert bytes(res.out_md5) == hashlib.md5(b"".join(oo.picture_digest(*planes[i], win) for i in order)).digest()
    st.close(); [j.close() for j in jobs]; dpb.close(); ctx.close()


def test_failed_picture_fails_its_readers_instead_of_hanging_them(built_lib):
    """a picture that cannot be decoded is published as failed; the pictures that reference it end with
    OVHIP_EREF and the run returns the first error."""
    w, h = 416, 240
    wls = _contents(w, h, (51,), 52)
    spics = [{"content": 1, "job": 0, "poc": 0, "refs": []},
             {"content": 0, "job": 1, "poc": 4, "refs": [0, 0]},
             {"content": 0, "job": 2, "poc": 2, "refs": [0, 1]}]
    ctx = engine.Context(0)
    dpb = engine.Dpb((0,))
    jobs = _jobs_for(ctx, wls, spics, w, h)
    conts = _make_contents(wls)
    conts[1]["params"].alf_luma_coeff = None                        # the I picture's flush is refused: ALF tables missing
    st = engine.Stream(dpb, w, h, conts, jobs, threads_per_device=3)
    arr = st.pics_array(spics)
    res, _ = st.run(arr, len(spics), 0, len(spics), check=False)
    assert res.status in (capi.OVHIP_EINVAL, capi.OVHIP_EREF) and res.error, (res.status, res.error)
    assert dpb.stats().n_failed >= 1
    # and the stream object is usable again afterwards
    conts2 = _make_contents(wls)
    st2 = engine.Stream(dpb, w, h, conts2, jobs, threads_per_device=3)
    res, _ = st2.run(arr, len(spics), 0, len(spics))
    assert res.status == 0 and res.n_decoded == 3
    st.close(); st2.close(); [j.close() for j in jobs]; dpb.close(); ctx.close()


def test_frame_api_as_the_shim_uses_it(built_lib):
    """ovhip_frame_begin / _ref / recorder / _dmvr_rows / _submit with OVHIP_OUT_PLANES: the calls shim/rcn_hip.c makes per picture,
    two frame objects (two OVCTUDecs), the second picture referencing the first; output lands in host planes only after the wait."""
    w, h = 416, 240
    wl1 = synth.make_workload(w, h, 61, tools=synth.INTRA_TOOLS, intra_frac=1.0, calllog=True)
    wl2 = synth.make_workload(w, h, 62, tools=synth.INTRA_TOOLS, intra_frac=0.2, calllog=True)
    p1 = oracle_pipeline.decode(wl1)
    wl2.refs[0] = (p1.y.copy(), p1.cb.copy(), p1.cr.copy())
    wl2.refs[1] = wl2.refs[0]
    p2, mvs = oracle_pipeline.decode(wl2, want_mvs=True)
    dpb = engine.Dpb((0,))
    f1, f2 = engine.Frame(dpb, 0, w, h), engine.Frame(dpb, 0, w, h)
    KEY1, KEY2 = 0xA000, 0xB000
    for rep in range(2):
        f1.begin(KEY1)
        f1.recorder().replay(wl1.calllog)
        f2.begin(KEY2)                                              # the second "frame thread" starts parsing before picture 1 is done
        assert f2.ref(KEY1) == 0 and f2.ref(KEY1) == 0
        assert f2.ref_at(1, KEY1) == 1
        f2.recorder().replay(wl2.calllog)
        k1, k2 = _Keep(), _Keep()
        out1 = capi.FrameOutput()
        y, cb, cr = np.zeros((h, w), np.uint16), np.zeros((h // 2, w // 2), np.uint16), np.zeros((h // 2, w // 2), np.uint16)
        out1.mode, out1.y, out1.cb, out1.cr, out1.stride_y, out1.stride_c = capi.OUT_PLANES, y.ctypes.data, cb.ctypes.data, cr.ctypes.data, w, w // 2
        f1.job().test_abort_next_flow()                             # the download must see the SECOND pass (r2 downloaded before the wait)
        f1.submit(engine.Job.make_params(k1, wl1), out=out1)
        assert np.array_equal(y, p1.y) and np.array_equal(cb, p1.cb) and np.array_equal(cr, p1.cr)
        # eager refinement in two halves, as the shim's row-end hooks run it: needs picture 1 complete
        assert f2.dmvr_rows_collect() == 0
        assert f2.dmvr_rows_begin(7) == len(wl2.mcx_units)
        assert f2.dmvr_rows_collect() == len(wl2.mcx_units)
        is_dmvr = (wl2.mcx_units["flags"] & 64) != 0
        assert is_dmvr.sum() > 5 and np.array_equal(f2.job().refined_mvs()[is_dmvr], mvs[is_dmvr])
        assert np.array_equal(f2.job().tmvp_cells(), oracle_lib.tmvp_cells(wl2.mcx_units, mvs, 7, (w + 127) // 128))
        assert f2.dmvr_rows() == len(wl2.mcx_units)                 # the synchronous form: nothing new
        out2 = capi.FrameOutput()
        out2.mode = capi.OUT_DIGEST
        f2.submit(engine.Job.make_params(k2, wl2), out=out2)
        assert bytes(out2.digest) == oo.picture_digest(p2.y, p2.cb, p2.cr)
        assert np.array_equal(f2.job().refined_mvs(), mvs)
    # a picture that was begun and dropped: its readers are told
    f1.begin(0xC000)
    f1.fail(capi.OVHIP_EUNSUP)
    f2.begin(0xD000)
    f2.ref(0xC000)
    f2.recorder().replay(wl2.calllog)
    k3 = _Keep()
    r = f2.submit(engine.Job.make_params(k3, wl2), check=False)
    assert r == capi.OVHIP_EREF
    f1.close(); f2.close(); dpb.close()
