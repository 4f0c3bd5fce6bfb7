"""GPU: the chained streams the reference's slice decoder produced (tests/golden/pipe*.ovg, test_pipe_cpu.py) through the HIP
engine and the C ABI: every picture decoded on the device FROM THE DEVICE'S OWN earlier pictures must equal the reference's frame
byte for byte; DMVR's refined vectors must equal what rcn_dmvr_mv_refine returned."""
import ctypes as C

import numpy as np
import pytest

import pipe_cases
from openvvc_amd import capi, engine

pytestmark = pytest.mark.gpu
STREAMS = ("pipe", "pipe_b", "tiles", "tiles_b")       # tiles*: every picture cut into 2 x 2 rect entries (test_pipe_cpu.py)


def _load(rec, lib, wl):
    """a recorded picture into a (job's / frame's) recorder: what Job.load_workload does after its begin"""
    for which, arr in ((capi.REC_COEF, wl.coefs), (capi.REC_TB, wl.tb_cmds), (capi.REC_MC, wl.mc_units), (capi.REC_MCX, wl.mcx_units),
                       (capi.REC_AFF, wl.aff_units), (capi.REC_SIDE, wl.aff_side), (capi.REC_REGION, wl.lmcs_regions),
                       (capi.REC_CIIP, wl.ciip_units), (capi.REC_ITASK, wl.itasks), (capi.REC_EDGE_V, wl.dbf_edges[0]), (capi.REC_EDGE_H, wl.dbf_edges[1])):
        if arr is not None and len(arr):
            rec.append_raw(which, arr)
    offs = capi.DbfOffsets()
    for i in range(8):
        offs.beta[i], offs.tc[i] = wl.dbf_planes["beta_offset"], wl.dbf_planes["tc_offset"]
    assert lib.ovhip_rec_set_dbf_offsets(rec.h, C.byref(offs), 1) == 0


class _Keep:
    def __init__(self):
        self._keep = {}


def _check(name, k, got, P, mvs, wl):
    for plane, a, b in zip("Y Cb Cr".split(), got, P.frames[k]):
        assert np.array_equal(a, b), f"{name} picture {k} plane {plane}: {int((a != b).sum())} samples differ from the reference"
    calls = P.dmvr_calls(k)
    is_dmvr = (wl.mcx_units["flags"] & 64) != 0
    assert is_dmvr.sum() == len(calls)
    if len(calls):
        assert np.array_equal(mvs[is_dmvr], calls[:, 8:12]), f"{name} picture {k}: refined vectors differ from rcn_dmvr_mv_refine's"


@pytest.mark.parametrize("name", STREAMS)
def test_job_chain_on_the_device_equals_the_reference_slice_decoder(built_lib, name):
    """ovhip_job_flush per picture; the reference pictures are the device pictures of the jobs before"""
    P = pipe_cases.Pipe(name)
    ctx = engine.Context(0)
    dev, host = {}, {}
    for k in range(P.n):
        wl = P.workload(k, {i: None for i in range(k)})          # (the references are bound on the device below)
        job = engine.Job(ctx, P.w, P.h)
        job.begin()
        _load(job.rec, job.lib, wl)
        dst = ctx.new_pic(P.w, P.h)
        job.flush(dst, [dev[i] for i in P.ref_indices(k)], None, params=job.make_params(wl))
        job.wait()
        host[k] = dst.download()
        _check(name, k, host[k], P, job.refined_mvs(), wl)
        assert job.stats().n_ordered_retries == 0
        dev[k] = dst
        job.close()
    ctx.close()


@pytest.mark.parametrize("name", STREAMS)
def test_frame_api_chain_equals_the_reference_slice_decoder(built_lib, name):
    """the calls shim/rcn_hip.c makes per picture (ovhip_frame_begin / _ref / recorder / _dmvr_rows_* / _submit) on two frame
    objects taking the pictures in turn; references come out of the device DPB by key; eager DMVR rows before the submit"""
    P = pipe_cases.Pipe(name)
    dpb = engine.Dpb((0,))
    frames = [engine.Frame(dpb, 0, P.w, P.h), engine.Frame(dpb, 0, P.w, P.h)]
    key = lambda k: 0xF000 + 16 * k
    for k in range(P.n):
        f = frames[k & 1]
        wl = P.workload(k, {i: None for i in range(k)})
        f.begin(key(k))
        for slot, i in enumerate(P.ref_indices(k)):
            assert f.ref(key(i)) == slot
        rec = f.recorder()
        _load(rec, f.lib, wl)
        if len(wl.mcx_units):
            assert f.dmvr_rows_begin(P.log2_ctu) == len(wl.mcx_units)
            assert f.dmvr_rows_collect() == len(wl.mcx_units)
        keep = _Keep()
        out = capi.FrameOutput()
        y, cb, cr = np.zeros((P.h, P.w), np.uint16), np.zeros((P.h // 2, P.w // 2), np.uint16), np.zeros((P.h // 2, P.w // 2), np.uint16)
        out.mode, out.y, out.cb, out.cr, out.stride_y, out.stride_c = capi.OUT_PLANES, y.ctypes.data, cb.ctypes.data, cr.ctypes.data, P.w, P.w // 2
        f.submit(engine.Job.make_params(keep, wl), out=out)
        _check(name, k, (y, cb, cr), P, f.job().refined_mvs(), wl)
    for f in frames:
        f.close()
    dpb.close()


@pytest.mark.parametrize("name", STREAMS)
def test_replay_of_the_shims_device_mode_call_log(built_lib, name):
    """tests/golden/shim_pipe*_dev.ovg: the ovhip_frame_* calls the shim's device half made under the reference's slice decoder
    (dry, in the build container; tests/test_shim_device_cpu.py), replayed here call for call on the real frame layer -- the eager
    DMVR passes with exactly the units that had been recorded when the shim started them -- and the pictures must be the
    reference's."""
    import golden_io
    P = pipe_cases.Pipe(name)
    ev = np.frombuffer(golden_io.load(f"shim_{name}_dev.ovg")["events"].tobytes(), capi.FRAME_EVENT_DTYPE)
    dpb = engine.Dpb((0,))
    f = engine.Frame(dpb, 0, P.w, P.h)
    key = lambda k: 0xE000 + 16 * k
    cur, wl, rec, n_mcx_in, n_done, cells = None, None, None, 0, 0, None
    for e in ev:
        op = int(e["op"])
        if op == capi.FE_BEGIN:
            cur = int(e["key"])
            wl = P.workload(cur, {i: None for i in range(cur)})
            f.begin(key(cur), int(e["tag"]))
            rec, n_mcx_in = f.recorder(), 0
        elif op == capi.FE_REF:
            assert f.ref(key(int(e["key"])), int(e["tag"])) == int(e["result"])
        elif op in (capi.FE_DMVR_BEGIN, capi.FE_DMVR_COLLECT, capi.FE_SUBMIT):
            upto = int(e["a"])                                   # refined units the shim had recorded at this call
            if upto > n_mcx_in:
                rec.append_raw(capi.REC_MCX, wl.mcx_units[n_mcx_in:upto]); n_mcx_in = upto
            if op == capi.FE_DMVR_BEGIN:
                assert f.dmvr_rows_begin(P.log2_ctu) == int(e["result"])
            elif op == capi.FE_DMVR_COLLECT:
                n_done = f.dmvr_rows_collect()
                cells = f.job().tmvp_cells()
                assert n_done <= n_mcx_in                        # (a dry frame completes a pass at once; the device when it does)
            else:
                assert n_mcx_in == len(wl.mcx_units)
                rest = pipe_cases.Workload(**{**wl.__dict__, "mcx_units": wl.mcx_units[:0]})
                _load(rec, f.lib, rest)
                keep = _Keep()
                out = capi.FrameOutput()
                y, cb, cr = np.zeros((P.h, P.w), np.uint16), np.zeros((P.h // 2, P.w // 2), np.uint16), np.zeros((P.h // 2, P.w // 2), np.uint16)
                out.mode, out.y, out.cb, out.cr, out.stride_y, out.stride_c = capi.OUT_PLANES, y.ctypes.data, cb.ctypes.data, cr.ctypes.data, P.w, P.w // 2
                f.submit(engine.Job.make_params(keep, wl), out=out)
                _check(name, cur, (y, cb, cr), P, f.job().refined_mvs(), wl)
                # the plane entries the eager passes delivered (read BEFORE the submit, as the shim's dmvr_rows_step does) = the
                # oracle's for the refined vectors: what gets patched into the picture's collocated motion field before a row is reported
                if len(wl.mcx_units):
                    import oracle_lib
                    assert n_done == len(wl.mcx_units) and len(cells) == 4 * n_done
                    want = oracle_lib.tmvp_cells(wl.mcx_units, f.job().refined_mvs(), P.log2_ctu, (P.w + 127) // 128)
                    used = want["cell"] != capi.TMVP_NONE
                    assert np.array_equal(cells["cell"], want["cell"]) and np.array_equal(cells[used], want[used])
                    assert used.sum() >= ((wl.mcx_units["flags"] & 64) != 0).sum()
    assert cur == P.n - 1
    f.close(); dpb.close()


def test_full_size_stream_of_the_reference_slice_decoder(built_lib):
    """BASELINE's full size: the prebuilt harness (oracle/_ref/gen_pipe: the reference's slicedec.c + rcn slots, compiled in the build
    container; it travels with the snapshot, /root/reference does not) decodes nine chained 3840x2160 pictures HERE and records the same
    parse through the installed shim slots; the HIP engine must reproduce all nine frames and every DMVR vector (bench.py's
    config.reference_stream leg, with one repetition)."""
    import importlib.util
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    if not (root / "oracle" / "_ref" / "gen_pipe").exists():
        pytest.fail("oracle/_ref/gen_pipe is missing: the prebuilt harness (make -C oracle, in the build container) must travel with the snapshot -- a GPU run without it would silently drop the live / full-size parity tests")
    spec = importlib.util.spec_from_file_location("bench", root / "bench.py")
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    ctx = engine.Context(0)
    r = bench.reference_stream_on_device(engine, capi, ctx, 3840, 2160, 9, 1)
    ctx.close()
    assert r is not None and r["pictures"] == 9
    assert r["samples_differing_from_the_reference"] == 0 and r["refined_vectors_differing"] == 0
    assert r["units"]["dmvr_calls"] > 20000 and r["units"]["ordered_tasks"] > 20000 and r["units"]["affine"] > 5000


@pytest.mark.parametrize("w,h,extra", [(832, 480, ("tiles", 3, 2, "seed", 4242)), (416, 240, ("variant", 1, "tiles", 4, 1, "seed", 77)),
                                       (1920, 1080, ("seed", 31))])
def test_fresh_streams_of_the_reference_slice_decoder(built_lib, w, h, extra):
    """streams nobody has looked at (other seeds / tilings / sizes than the fixtures: tools/gpu_pipe_sweep.py runs hundreds): decoded
    HERE by the prebuilt harness and by the HIP engine, compared frame by frame and vector by vector"""
    import importlib.util
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    if not (root / "oracle" / "_ref" / "gen_pipe").exists():
        pytest.fail("oracle/_ref/gen_pipe is missing: the prebuilt harness (make -C oracle, in the build container) must travel with the snapshot -- a GPU run without it would silently drop the live / full-size parity tests")
    spec = importlib.util.spec_from_file_location("bench", root / "bench.py")
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    ctx = engine.Context(0)
    r = bench.reference_stream_on_device(engine, capi, ctx, w, h, 5, 0, extra=extra)
    ctx.close()
    assert r is not None and r["pictures"] == 5
    assert r["samples_differing_from_the_reference"] == 0 and r["refined_vectors_differing"] == 0
    assert r["units"]["dmvr_calls"] > 100 and r["units"]["ordered_tasks"] > 100
