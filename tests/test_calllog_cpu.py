"""CPU: the call log (ovvc_calllog.c) -- the recorder calls of a picture, serialised and replayed -- reproduces the recorder's
arrays byte for byte: what lets the stream driver put the recorder (the parse thread's share of the device path) into the timed
region without Python."""
import numpy as np
import pytest

from openvvc_amd import capi, synth


@pytest.mark.parametrize("w,h,seed,tools,frac", [(416, 240, 0x266, synth.INTRA_TOOLS, 0.12), (832, 480, 7, synth.ALL_TOOLS, 0.0),
                                                 (416, 240, 3, synth.INTRA_TOOLS, 1.0)])
def test_replay_reproduces_every_recorder_array(built_lib, w, h, seed, tools, frac):
    wl = synth.make_workload(w, h, seed, tools=tools, intra_frac=frac, calllog=True)
    assert wl.calllog is not None and wl.calllog.nbytes % 8 == 0
    rec = capi.Recorder(w, h)
    for rep in range(2):                                    # a recorder is replayed into once per picture
        rec.reset()
        assert rec.replay(wl.calllog) > len(wl.cus)
        cmds, classes = rec.tb_cmds_split()
        assert np.array_equal(cmds, wl.tb_cmds) and classes == wl.tb_classes
        assert np.array_equal(rec.coefs(), wl.coefs)
        assert np.array_equal(rec.mc_units(), wl.mc_units) and np.array_equal(rec.mcx_units(), wl.mcx_units)
        assert np.array_equal(rec.aff_units(), wl.aff_units) and np.array_equal(rec.aff_side(), wl.aff_side)
        assert np.array_equal(rec.ciip_units(), wl.ciip_units)
        if wl.itasks is not None:
            assert np.array_equal(rec.itasks(), wl.itasks)
        if wl.lmcs_regions is not None:
            assert np.array_equal(rec.lmcs_regions(), wl.lmcs_regions)
        for d in (0, 1):
            assert np.array_equal(rec.dbf_edges(d)[0], wl.dbf_edges[d])
    rec.close()


def test_truncated_or_foreign_log_is_refused(built_lib):
    wl = synth.make_workload(416, 240, 5, calllog=True)
    rec = capi.Recorder(416, 240)
    with pytest.raises(ValueError):
        rec.replay(wl.calllog[:len(wl.calllog) // 2 + 4].copy())            # not a multiple of 8 / cut inside a record
    bad = wl.calllog.copy()
    bad[0:4] = np.frombuffer(np.uint32(77).tobytes(), np.uint8)              # unknown record type
    with pytest.raises(ValueError):
        rec.replay(bad)
    rec.close()


def _records(log):
    """(offset, type, payload bytes) of every record"""
    out, o = [], 0
    while o < len(log):
        t, n = (int(v) for v in np.frombuffer(log[o:o + 8].tobytes(), np.uint32))
        out.append((o, t, n)); o += 8 + n
    return out


def test_a_record_shorter_than_what_its_fields_imply_is_refused(built_lib):
    """A log comes from a file or another process: a record whose payload is shorter than its fixed part, or than the coefficient
    blocks / sub-block vectors its fields call for, must stop the replay with an error -- not be read through."""
    wl = synth.make_workload(416, 240, 9, tools=synth.INTRA_TOOLS, intra_frac=0.3, calllog=True)
    log = wl.calllog
    recs = _records(log)
    u32 = lambda v: np.frombuffer(np.uint32(v).tobytes(), np.uint8)
    first = {}
    for o, t, n in recs:
        first.setdefault(t, (o, n))
    assert {2, 4, 5} <= set(first), sorted(first)                              # TU, PU, affine came up
    rec = capi.Recorder(416, 240)

    def refused(buf):
        rec.reset()
        with pytest.raises(ValueError):
            rec.replay(buf)

    for t, (o, n) in sorted(first.items()):
        if t == 1:
            continue
        cut = log[:o + 16].copy()                                              # the log ends 8 payload bytes into this record:
        cut[o + 4:o + 8] = u32(8)                                              # below every type's fixed part
        refused(cut)
        odd = log[:o + 8 + n].copy()                                           # a payload length that is not a multiple of 8
        odd[o + 4:o + 8] = u32(n - 4)
        refused(odd)
    o, n = max(((o, n) for o, t, n in recs if t == 2), key=lambda v: v[1])     # the TU carrying the most coefficients,
    short = log[:o + 8 + n - 64].copy()                                        # its last block cut
    short[o + 4:o + 8] = u32(n - 64)
    refused(short)
    o, n = max(((o, n) for o, t, n in recs if t == 5), key=lambda v: v[1])     # an affine CU without its last sub-block vectors
    short = log[:o + 8 + n - 16].copy()
    short[o + 4:o + 8] = u32(n - 16)
    refused(short)
    big = log[:o + 8 + n].copy()                                               # ... and one claiming a 256 x 256 CU
    big[o + 8 + 4] = 8
    refused(big)
    rec.reset()
    assert rec.replay(log) == len(recs)                                        # the intact log still replays
    rec.close()
