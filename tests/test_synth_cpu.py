"""CPU: the synthetic recorded-picture generator and the oracle pipeline (no GPU): every coding tool is
present, the pipeline is deterministic, DMVR moves motion vectors, LMCS chroma scales vary."""
import hashlib

import numpy as np

import oracle_lib
import oracle_pipeline
from oracle_lib import HostPic
from openvvc_amd import capi, synth


def test_workload_has_every_tool_and_oracle_is_deterministic(built_lib):
    wl = synth.make_workload(416, 240, 0x266)
    assert all(v > 0 for v in wl.stats["cu_modes"].values()), wl.stats["cu_modes"]
    assert wl.n_luma_cmds and (wl.tb_cmds["plane"][:wl.n_luma_cmds] == 0).all() and (wl.tb_cmds["plane"][wl.n_luma_cmds:] != 0).all()
    a, mv_a = oracle_pipeline.decode(wl, want_mvs=True)
    b, mv_b = oracle_pipeline.decode(wl, want_mvs=True)
    md5 = lambda p: hashlib.md5(p.y.tobytes() + p.cb.tobytes() + p.cr.tobytes()).hexdigest()
    assert md5(a) == md5(b) and np.array_equal(mv_a, mv_b)
    # DMVR refined at least some motion vectors; BDOF-only units report their input vectors
    ux = wl.mcx_units
    moved = (mv_a != np.stack([ux["mv0x"], ux["mv0y"], ux["mv1x"], ux["mv1y"]], axis=1)).any(axis=1)
    dm = (ux["flags"] & capi.MC_DMVR) != 0
    assert moved[dm].sum() > 0 and not moved[~dm & ((ux["flags"] & capi.MC_BDOF) != 0)].any()
    # the same picture without the new tools differs (they are not no-ops)
    base = oracle_pipeline.decode(synth.make_workload(416, 240, 0x266, tools=()))
    assert md5(base) != md5(a)


def test_deblocking_goes_through_the_ctu_recorder(built_lib):
    """The synthetic picture hands ovhip_rec_dbf_ctu the CTU-local maps df.rcn_dbf_ctu receives (synth.make_dbf_ctus); the edge
    lists the engine uploads are what that call emitted, the planes the oracle filters from hold exactly those segments, and
    the CTU seams lose nothing (horizontal edges of a CTU's last two unit columns come with its right neighbour)."""
    for w, h in ((416, 240), (832, 480), (200, 136)):                  # the last one: truncated CTUs in both directions
        wl = synth.make_workload(w, h, 0x77)
        assert len(wl.dbf_ctus) == ((w + 127) // 128) * ((h + 127) // 128)
        key = ["comp", "uy", "ux"]
        for d in (0, 1):
            emitted, compact = np.sort(wl.dbf_edges[d], order=key), np.sort(capi.dbf_compact(wl.dbf_planes, d), order=key)
            assert len(emitted) and np.array_equal(emitted, compact), (w, h, d)
        # every CU boundary inside the picture whose two sides differ in a strength-giving property is an edge somewhere:
        # at least the 64-sample grid lines and all eight phases of the unit columns / rows carry edges
        on_v, on_h = (wl.dbf_planes["luma_v"] & 3) > 0, (wl.dbf_planes["luma_h"] & 3) > 0
        assert not on_v[:, 0].any() and not on_h[0, :].any()            # never on the picture border
        if w >= 416:
            assert all(on_h[:, c::32].any() for c in range(32)) and all(on_v[r::32, :].any() for r in range(32))
        lp = (wl.dbf_planes["luma_h"] >> 2) & 7
        assert (lp[::32][on_h[::32]] <= 3).all()                        # CTU-row tops: the P side keeps 3 lines (one line buffer)
        assert set(np.unique(lp[on_h])) <= {1, 2, 3, 5, 7} and (lp[on_h] == 7).any()


def test_lmcs_chroma_scales_follow_the_luma(built_lib):
    wl = synth.make_workload(416, 240, 3)
    refs = [HostPic(wl.w, wl.h, *r) for r in wl.refs]
    dst = HostPic(wl.w, wl.h)
    oracle_lib.mc(dst, refs, wl.mc_units, wl.lmcs_fwd)
    scales = oracle_lib.lmcs_scale(dst, wl.lmcs_regions, wl.lmcs)
    assert len(scales) == wl.stats["n_lmcs_regions"] > 8
    assert len(np.unique(scales)) > 2 and scales.min() > 1000 and scales.max() < 4500
    idx = (wl.tb_cmds["res_mode"] & 8) != 0
    assert idx.any() and wl.tb_cmds["c_scale"][idx].max() < len(scales)


def test_level_order_equals_decoding_order():
    """The recorder's levels (ovhip_itask.level) are a valid schedule: executing the ordered tasks level by level -- inside a
    level in REVERSED decoding order, the worst case for a missed dependency -- gives the picture the decoding-order execution
    gives, for B-like pictures and an I picture."""
    import copy
    import numpy as np
    import oracle_pipeline
    from openvvc_amd import synth
    for seed, frac in ((1, 0.12), (2, 0.4), (3, 1.0)):
        wl = synth.make_workload(416, 240, seed, tools=synth.INTRA_TOOLS, intra_frac=frac)
        t = wl.itasks
        assert len(t) > 100 and set(np.unique(t["kind"])) >= {0, 1}
        a = oracle_pipeline.decode(wl, stages=("mc", "itx"))
        order = np.argsort(t["level"], kind="stable")
        lv = t["level"][order]
        rev = np.concatenate([order[lv == l][::-1] for l in np.unique(lv)])
        wl2 = copy.copy(wl)
        wl2.itasks = t[rev]
        b = oracle_pipeline.decode(wl2, stages=("mc", "itx"))
        for x, y in zip(a.planes(), b.planes()):
            assert np.array_equal(x, y), f"seed {seed} intra_frac {frac}: level order differs from decoding order"


def test_by_ctu_grouping_is_a_valid_order():
    """ovhip_rec_itasks_by_ctu: every task once, a CTU's tasks contiguous and in level order; the per-CTU dependency masks
    (from the recorder's per-task ctu_deps) only name neighbours that hold tasks, and are SUFFICIENT: executing the CTUs in the
    most adversarial order the masks allow (always the last CTU in raster order whose neighbours are done) gives the
    decoding-order picture -- what k_intra_ctu's flag waits rely on."""
    import copy
    import numpy as np
    import oracle_pipeline
    from openvvc_amd import capi, synth
    for seed, frac in ((4, 0.15), (5, 1.0), (6, 0.4)):
        wl = synth.make_workload(416, 240, seed, tools=synth.INTRA_TOOLS, intra_frac=frac)
        assert (wl.itasks["ctu_deps"] & 0x8000).all() and not (wl.itasks["ctu_deps"] & 0x10).any()
        rec = capi.Recorder(416, 240)
        rec.append_raw(capi.REC_ITASK, wl.itasks)
        t, c = rec.itasks_by_ctu(7)
        assert len(t) == len(wl.itasks) and int(c["n"].sum()) == len(t)
        ncx = (416 + 127) // 128
        where = {(int(d["cx"]), int(d["cy"])): i for i, d in enumerate(c)}
        prev = -1
        nbr = ((1, (-1, 0)), (2, (-1, -1)), (4, (0, -1)), (8, (1, -1)))
        for d in c:
            idx = int(d["cy"]) * ncx + int(d["cx"])
            assert idx > prev
            prev = idx
            tt = t[int(d["first"]):int(d["first"]) + int(d["n"])]
            assert np.all(np.diff(tt["level"].astype(np.int64)) >= 0)
            sh = np.where((tt["kind"] == capi.IT_CHROMA) | (tt["kind"] == capi.IT_RES_C), 1, 0)
            assert np.all((tt["x"].astype(np.int64) << sh) >> 7 == d["cx"]) and np.all((tt["y"].astype(np.int64) << sh) >> 7 == d["cy"])
            cx, cy = int(d["cx"]), int(d["cy"])
            allowed = sum(bit for bit, (dx, dy) in nbr if (cx + dx, cy + dy) in where)
            assert int(d["deps"]) & ~allowed == 0
        # adversarial topological order
        done, order = set(), []
        while len(order) < len(c):
            for i in range(len(c) - 1, -1, -1):
                if i in done:
                    continue
                cx, cy, deps = int(c[i]["cx"]), int(c[i]["cy"]), int(c[i]["deps"])
                if all(where[(cx + dx, cy + dy)] in done for bit, (dx, dy) in nbr if deps & bit):
                    done.add(i); order.append(i)
                    break
            else:
                raise AssertionError("dependency cycle")
        if frac < 1.0:
            assert order != sorted(order)           # the masks leave freedom (else the test shows nothing)
        a = oracle_pipeline.decode(wl, stages=("mc", "itx"))
        wl2 = copy.copy(wl)
        wl2.itasks = np.concatenate([t[int(c[i]["first"]):int(c[i]["first"]) + int(c[i]["n"])] for i in order])
        b = oracle_pipeline.decode(wl2, stages=("mc", "itx"))
        for x, y in zip(a.planes(), b.planes()):
            assert np.array_equal(x, y), f"seed {seed} intra_frac {frac}"
        rec.close()


def test_level_table_grows_between_pictures():
    """A recorder reused for a picture with more levels than the one before (the level table is reallocated)."""
    import numpy as np
    from openvvc_amd import capi
    rec = capi.Recorder(256, 256)
    for nlev in (5, 11, 12, 40, 3):
        rec.reset()
        t = np.zeros(nlev * 3, capi.ITASK_DTYPE)
        t["level"] = np.repeat(np.arange(1, nlev + 1), 3)[::-1]
        t["log2_w"] = t["log2_h"] = 2
        t["x"] = (np.arange(len(t)) % 32) * 4
        rec.append_raw(capi.REC_ITASK, t)
        ts, ls = rec.itasks_sorted()
        assert len(ls) == nlev + 1 and ls[-1] == len(t) and np.all(np.diff(ts["level"].astype(int)) >= 0)
        tc, cs = rec.itasks_by_ctu(7)
        assert int(cs["n"].sum()) == len(t)
    rec.close()
