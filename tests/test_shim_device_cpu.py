"""CPU: WHEN the shim's device half makes its frame-level calls (tests/golden/shim_pipe*_dev.ovg).

oracle/ref_harness/gen_pipe.c "device" mode ran the reference's own slice decoder over the INSTALLED shim with its device half live
-- begin_picture, dpb_get, ref_slot -> ovhip_frame_ref_tag, dmvr_rows_step, flush_picture (shim/rcn_hip.c) -- on a device DPB with a
test memory back-end (dry frames: nothing launched).  The fixture interleaves the decoder's events (rcn_attach_frame_buff,
sao.rcn_sao_first_pix_rows, alf.rcn_alf_filter_line: the hooks of slicedec.c:934-956, :1058-1073, and every rcn_dmvr_mv_refine
slot call) with the ovhip_frame_* calls made under them (include/ovvc_hip.h, ovhip_frame_set_trace).  The harness itself asserted
that the device half recorded exactly what record-only mode recorded (shim_pipe*.ovg).  tests/test_gpu_pipe.py replays the
frame-level calls on a GPU."""
import numpy as np
import pytest

import golden_io
import pipe_cases
from openvvc_amd import capi

ATTACH, SAO_FIRST, ALF_LINE, HOOK_END, DMVR_SLOT = 100, 101, 102, 103, 104


def events(name):
    g = golden_io.load(f"shim_{name}_dev.ovg")
    return np.frombuffer(g["events"].tobytes(), capi.FRAME_EVENT_DTYPE)


def pictures(ev):
    """split the log at the attach events: one list per picture"""
    starts = [i for i, e in enumerate(ev) if e["op"] == ATTACH] + [len(ev)]
    return [ev[a:b] for a, b in zip(starts[:-1], starts[1:])]


@pytest.mark.parametrize("name", ("pipe", "pipe_b"))
def test_frame_level_calls_follow_the_decoders_events(built_lib, name):
    P = pipe_cases.Pipe(name)
    pics = pictures(events(name))
    assert len(pics) == P.n
    n_rows = (P.h + 127) // 128
    for k, ev in enumerate(pics):
        ops = [int(e["op"]) for e in ev]
        # ---- begin: inside rcn_attach_frame_buff, keyed by this picture, tagged with its POC (pic_tag: cvs 0 -> poc + 1)
        assert ops[0] == ATTACH and ops[1] == capi.FE_BEGIN and ops[2] == HOOK_END
        assert int(ev[1]["key"]) == k and int(ev[1]["tag"]) == int(P.info[k][0]) + 1 and int(ev[1]["result"]) == 0
        # ---- references: named on first use, in the order of the table the recorded units index; each with its POC
        refs = ev[[o == capi.FE_REF for o in ops]]
        assert [int(r["key"]) for r in refs] == P.ref_indices(k)
        assert [int(r["result"]) for r in refs] == list(range(len(refs)))
        for r in refs:
            assert int(r["tag"]) == int(P.info[int(r["key"])][0]) + 1
        # ---- the decoder's row-end hooks in slicedec.c's order: first_pix_rows(0), then alf line y after row y + 1 was parsed, the
        #      last two lines together
        hooks = [(int(e["op"]), int(e["a"])) for e in ev if e["op"] in (SAO_FIRST, ALF_LINE)]
        assert hooks == [(SAO_FIRST, 0)] + [(ALF_LINE, y) for y in range(n_rows)]
        # ---- every frame-level call except REF (made while a CU is parsed) happens inside a hook
        depth, last_hook = 0, None
        n_slot = n_units_begun = n_collected = 0
        submits = []
        for e in ev:
            op = int(e["op"])
            if op in (ATTACH, SAO_FIRST, ALF_LINE):
                depth += 1; last_hook = (op, int(e["a"]))
            elif op == HOOK_END:
                depth -= 1
            elif op == DMVR_SLOT:
                assert depth == 0
                n_slot += 1
            elif op == capi.FE_REF:
                pass
            else:
                assert depth == 1, f"picture {k}: frame-level call {op} outside the decoder's hooks"
                if op == capi.FE_DMVR_BEGIN:
                    # the eager pass covers every refined unit recorded so far; it waits for the references only if one of them is
                    # a DMVR unit (rcn_inter_synchronization waits per block, rcn_inter.c:131-146)
                    assert int(e["result"]) == int(e["a"]) >= n_units_begun
                    n_units_begun = int(e["result"])
                elif op == capi.FE_DMVR_COLLECT:
                    assert n_collected <= int(e["result"]) <= n_units_begun
                    n_collected = int(e["result"])
                elif op == capi.FE_SUBMIT:
                    submits.append((last_hook, int(e["a"]), int(e["b"]), int(e["result"])))
                elif op == capi.FE_FAIL:
                    raise AssertionError("a picture of the stream failed")
        n_mcx = len(P.s.case(k)["mcx"])
        # ---- one submit, in the hook of the picture's LAST row, after every refined unit went through the eager pass
        assert submits == [((ALF_LINE, n_rows - 1), n_mcx, len(refs), 0)]
        assert n_collected == n_units_begun == n_mcx or n_mcx == 0
        assert n_slot == len(P.dmvr_calls(k))
        if n_mcx and n_rows > 2:
            # a row's units are begun at the hook that follows the row and collected at the next one: the search of row y runs
            # while row y + 1 is parsed -- more than one asynchronous pass per picture
            assert sum(1 for o in ops if o == capi.FE_DMVR_BEGIN) >= 2


def test_dry_frames_trace_through_the_c_abi(built_lib):
    """the dry back-end by hand: a DPB on test memory, two frames; the second picture references the first and is only submitted
    after it; the trace sees the calls in order"""
    import ctypes as C
    from test_dpb_cpu import FakeMem
    lib = built_lib
    mem, h = FakeMem(), C.c_void_p()
    assert lib.ovhip_dpb_create_ex(C.byref(h), 2, C.byref(mem.ops)) == 0
    log = []
    sink = capi.FRAME_TRACE_FN(lambda user, ev: log.append(np.frombuffer(C.string_at(ev, 48), capi.FRAME_EVENT_DTYPE)[0].copy()))
    lib.ovhip_frame_set_trace(sink, None)
    f0, f1 = C.c_void_p(), C.c_void_p()
    assert lib.ovhip_frame_create(h, 0, 64, 64, C.byref(f0)) == 0 and lib.ovhip_frame_create(h, 1, 64, 64, C.byref(f1)) == 0
    assert not lib.ovhip_frame_job(f0) and lib.ovhip_frame_recorder(f0)
    assert lib.ovhip_frame_begin_tag(f0, C.c_void_p(0x10), 5) == 0
    assert lib.ovhip_frame_begin_tag(f1, C.c_void_p(0x20), 6) == 0
    assert lib.ovhip_frame_ref_tag(f1, C.c_void_p(0x10), 5) == 0
    assert lib.ovhip_frame_dmvr_rows_begin(f1, 7) == 0 and lib.ovhip_frame_dmvr_rows_collect(f1) == 0
    p = capi.JobParams()
    assert lib.ovhip_frame_submit(f0, None, None, C.byref(p), None) == 0
    assert lib.ovhip_frame_submit(f1, None, None, C.byref(p), None) == 0          # its reference is DONE: the wait returns
    lib.ovhip_frame_set_trace(None, None)
    lib.ovhip_frame_destroy(f0); lib.ovhip_frame_destroy(f1)
    lib.ovhip_dpb_destroy(h)
    ops = [(int(e["op"]), int(e["key"])) for e in log]
    assert ops == [(capi.FE_BEGIN, 0x10), (capi.FE_BEGIN, 0x20), (capi.FE_REF, 0x10), (capi.FE_DMVR_BEGIN, 0x20), (capi.FE_DMVR_COLLECT, 0x20),
                   (capi.FE_SUBMIT, 0x10), (capi.FE_SUBMIT, 0x20)]
    assert log[1]["frame"] == log[0]["frame"] + 1 and any(e[0] == "copy" for e in mem.log), "the reference went to device 1"


@pytest.mark.parametrize("name", ("tiles", "tiles_b"))
def test_rect_entries_of_a_picture_share_one_device_job(built_lib, name):
    """Tiles (slicedec.c:636-657; one entry after the other on one OVCTUDec, :649-653): rcn_attach_frame_buff runs once per rect
    entry, the device picture is begun ONCE (in the first entry's attach) and submitted ONCE -- in the alf line hook of the last row
    of the picture's LAST entry (ovthreads.c:93-114: the last entry to finish ends the picture); the eager DMVR passes run under the
    row hooks of every entry."""
    P = pipe_cases.Pipe(name)
    ev = events(name)
    begins = [i for i, e in enumerate(ev) if e["op"] == capi.FE_BEGIN]
    assert len(begins) == P.n
    bounds = [b - 1 for b in begins] + [len(ev)]                           # (the BEGIN sits inside its ATTACH hook)
    rows_of_entry = (2, 2, 2, 2) if name == "tiles" else (1, 1, 1, 1)
    for k in range(P.n):
        pe = ev[bounds[k]:bounds[k + 1]]
        ops = [int(e["op"]) for e in pe]
        assert ops.count(ATTACH) == 4 and ops[0] == ATTACH and ops[1] == capi.FE_BEGIN and int(pe[1]["key"]) == k
        assert ops.count(capi.FE_BEGIN) == 1 and ops.count(capi.FE_SUBMIT) == 1 and ops.count(capi.FE_FAIL) == 0
        # per entry: attach, first_pix_rows(0), alf lines 0 .. rows - 1 (entry-local rows)
        hooks = [(int(e["op"]), int(e["a"])) for e in pe if e["op"] in (ATTACH, SAO_FIRST, ALF_LINE)]
        want = []
        for r in rows_of_entry:
            want += [(ATTACH, hooks[len(want)][1]), (SAO_FIRST, 0)] + [(ALF_LINE, y) for y in range(r)]
        assert hooks == want
        # the submit: inside the very last hook of the picture, with every refined unit through the eager passes
        depth, last_hook, n_hook = 0, None, 0
        for e in pe:
            op = int(e["op"])
            if op in (ATTACH, SAO_FIRST, ALF_LINE):
                depth += 1; n_hook += 1; last_hook = n_hook
            elif op == HOOK_END:
                depth -= 1
            elif op == capi.FE_SUBMIT:
                assert depth == 1 and last_hook == len(hooks) and int(e["result"]) == 0
                assert int(e["a"]) == len(P.s.case(k)["mcx"])
            elif op in (capi.FE_DMVR_BEGIN, capi.FE_DMVR_COLLECT):
                assert depth == 1
        refs = pe[[o == capi.FE_REF for o in ops]]
        assert [int(r["key"]) for r in refs] == P.ref_indices(k)
