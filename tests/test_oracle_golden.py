"""CPU: the oracle restatement + the host recorder reproduce the compiled reference bit-exactly
on the committed golden fixtures (tests/golden/*.ovg, made by oracle/ref_harness/gen_golden.c from
the reference's own rcn_tu_st / rcn_tu_c / rcn_mcp_b* slots)."""
import numpy as np

import golden_cases
import oracle_lib
from oracle_lib import HostPic
from openvvc_amd import capi


def test_itx_oracle_matches_reference(built_lib):
    pic, cmds, coefs, rects, exp = golden_cases.itx_cases()
    assert len(cmds) > 400
    oracle_lib.itx(pic, cmds, coefs)
    golden_cases.check_rects(pic, rects, exp, "itx oracle vs reference")


def test_transform_tree_oracle_matches_reference(built_lib):
    """tmp.rcn_transform_tree: CUs up to 128x128 cut at the maximum transform size, one TUInfo per leaf."""
    pic, cmds, coefs, rects, exp = golden_cases.tt_cases()
    assert len(rects) == 36 and len(cmds) > 60
    oracle_lib.itx(pic, cmds, coefs)
    golden_cases.check_rects(pic, rects, exp, "transform tree oracle vs reference")


def test_mc_oracle_matches_reference(built_lib):
    refs, descs, exp_off, exp = golden_cases.mc_cases()
    rw, rh = refs[0].w, refs[0].h
    rec = capi.Recorder(rw, rh)
    for i, d in enumerate(descs):
        rec.reset()
        rec.pu(d)
        dst = HostPic(rw, rh)
        dst.y[:] = 0xABAB; dst.cb[:] = 0xABAB; dst.cr[:] = 0xABAB
        oracle_lib.mc(dst, refs, rec.mc_units())
        w, h = 1 << d.log2_w, 1 << d.log2_h
        rects = [(0, d.x0, d.y0, w, h, int(exp_off[i, 0])),
                 (1, d.x0 >> 1, d.y0 >> 1, w >> 1, h >> 1, int(exp_off[i, 1])),
                 (2, d.x0 >> 1, d.y0 >> 1, w >> 1, h >> 1, int(exp_off[i, 2]))]
        golden_cases.check_rects(dst, rects, exp, f"mc case {i} dir={d.inter_dir} planes={d.planes}")


def test_mcx_oracle_matches_reference(built_lib):
    """BDOF (rcn_bdof_mcp_l + rcn_mcp_b_c) and DMVR (rcn_dmvr_mv_refine, incl. the MV write-back)."""
    refs, descs, exp_off, exp, exp_mv = golden_cases.mcx_cases()
    rw, rh = refs[0].w, refs[0].h
    rec = capi.Recorder(rw, rh)
    n_moved = n_bdof = 0
    for i, d in enumerate(descs):
        rec.reset()
        rec.pu(d)
        dst = HostPic(rw, rh)
        dst.y[:] = 0xABAB; dst.cb[:] = 0xABAB; dst.cr[:] = 0xABAB
        ux = rec.mcx_units()
        oracle_lib.mc(dst, refs, rec.mc_units())
        mv = oracle_lib.mc_ex(dst, refs, ux)
        w, h = 1 << d.log2_w, 1 << d.log2_h
        rects = [(0, d.x0, d.y0, w, h, int(exp_off[i, 0])),
                 (1, d.x0 >> 1, d.y0 >> 1, w >> 1, h >> 1, int(exp_off[i, 1])),
                 (2, d.x0 >> 1, d.y0 >> 1, w >> 1, h >> 1, int(exp_off[i, 2]))]
        golden_cases.check_rects(dst, rects, exp, f"mcx case {i} refine={d.refine} {w}x{h} @({d.x0},{d.y0})")
        if d.refine & capi.PU_DMVR:
            want = exp_mv[int(exp_off[i, 3]) // 4:int(exp_off[i, 3]) // 4 + len(ux)]
            assert np.array_equal(mv, want), f"mcx case {i}: refined MVs differ {mv.tolist()} vs {want.tolist()}"
            n_moved += int((mv != np.array([d.mv0x, d.mv0y, d.mv1x, d.mv1y])).any(axis=1).sum())
        else:
            n_bdof += len(ux)
    assert n_moved > 100 and n_bdof > 100, "fixture does not exercise DMVR / BDOF"


def test_mca_oracle_matches_reference(built_lib):
    """Affine sub-block MC + PROF (rcn_mcp_b_l(2,2) / rcn_prof_mcp_b_l / rcn_mcp_b_c(3,3))."""
    refs, cases, exp_off, exp = golden_cases.mca_cases()
    rw, rh = refs[0].w, refs[0].h
    rec = capi.Recorder(rw, rh)
    n_prof = 0
    for i, (d, mv0, mv1) in enumerate(cases):
        rec.reset()
        rec.affine_cu(d, mv0, mv1)
        dst = HostPic(rw, rh)
        dst.y[:] = 0xABAB; dst.cb[:] = 0xABAB; dst.cr[:] = 0xABAB
        oracle_lib.mca(dst, refs, rec.aff_units(), rec.aff_side())
        w, h = 1 << d.log2_w, 1 << d.log2_h
        rects = [(0, d.x0, d.y0, w, h, int(exp_off[i, 0])),
                 (1, d.x0 >> 1, d.y0 >> 1, w >> 1, h >> 1, int(exp_off[i, 1])),
                 (2, d.x0 >> 1, d.y0 >> 1, w >> 1, h >> 1, int(exp_off[i, 2]))]
        golden_cases.check_rects(dst, rects, exp, f"mca case {i} dir={d.inter_dir} prof={d.prof_dir} {w}x{h} @({d.x0},{d.y0})")
        n_prof += d.prof_dir != 0
    assert n_prof > 100


def test_lmcs_host_tables_and_oracle_match_reference(built_lib):
    """K11: ovhip_lmcs_build (host) == rcn_init_lmcs; oracle chroma scale == rcn_lmcs_compute_chroma_scale;
    oracle inverse map == lmcs_reshape_backward."""
    pic_y, sets, regions, inverse = golden_cases.lmcs_cases()
    h, w = pic_y.shape
    rec = capi.Recorder(w, h)
    n_inv = 0
    for si, (data, want) in enumerate(sets):
        got = capi.lmcs_build(data)
        assert bytes(got) == bytes(want), f"LMCS tables of set {si} differ"
        rows = regions[regions[:, 0] == si]
        rec.reset()
        for r in rows:
            rec.lmcs_region(int(r[1]), int(r[2]), int(r[3]), int(r[4]))
        pic = HostPic(w, h, pic_y.copy())
        scales = oracle_lib.lmcs_scale(pic, rec.lmcs_regions(), got)
        assert np.array_equal(scales, rows[:, 5].astype(np.int16)), f"chroma scales of set {si}: {scales.tolist()} vs {rows[:, 5].tolist()}"
        if si % 6 == 1:
            oracle_lib.lmcs_inverse(pic, np.frombuffer(bytes(got), np.uint16)[1024:2048])
            exp = np.concatenate([inverse[n_inv, 0], inverse[n_inv, 1]], axis=1)
            assert np.array_equal(pic.y, exp), f"inverse map of set {si} differs"
            n_inv += 1
    assert n_inv == len(inverse) and len(np.unique(regions[:, 5])) > 10


def test_gpm_ciip_oracle_matches_reference(built_lib):
    """K10: geometric partitioning (rcn_gpm_b, all 64 partition indices) and the CIIP blend (rcn_ciip / rcn_ciip_b)."""
    refs, intra, descs, modes, n_gpm, exp_off, exp = golden_cases.gpm_cases()
    _, first_planar = golden_cases.ciip_planar_cases()        # from there on: CIIP through ordered planar tasks (test_shim_cpu)
    descs = descs[:first_planar]
    rw, rh = refs[0].w, refs[0].h
    rec = capi.Recorder(rw, rh)
    for i, d in enumerate(descs):
        rec.reset()
        rec.pu(d)
        dst = HostPic(rw, rh)
        dst.y[:] = 0xABAB; dst.cb[:] = 0xABAB; dst.cr[:] = 0xABAB
        oracle_lib.mc(dst, refs, rec.mc_units())
        if i >= n_gpm:
            rec.ciip(d.x0, d.y0, d.log2_w, d.log2_h, int(modes[i, 0]), int(modes[i, 1]))
            oracle_lib.ciip(dst, intra, rec.ciip_units())
        w, h = 1 << d.log2_w, 1 << d.log2_h
        rects = [(0, d.x0, d.y0, w, h, int(exp_off[i, 0])),
                 (1, d.x0 >> 1, d.y0 >> 1, w >> 1, h >> 1, int(exp_off[i, 1])),
                 (2, d.x0 >> 1, d.y0 >> 1, w >> 1, h >> 1, int(exp_off[i, 2]))]
        what = f"GPM split {d.gpm_split_dir}" if i < n_gpm else f"CIIP modes {modes[i].tolist()} dir {d.inter_dir}"
        golden_cases.check_rects(dst, rects, exp, f"case {i} {what} {w}x{h} @({d.x0},{d.y0})")
        if i >= n_gpm:
            # second route: the blend fused into the prediction units (ovhip_pu_desc.ciip_wt)
            d.ciip_wt = capi.load().ovhip_ciip_weight(int(modes[i, 0]), int(modes[i, 1]))
            rec.reset()
            rec.pu(d)
            assert len(rec.ciip_units()) == 0 and (rec.mc_units()["aux"] != 0).all()
            dst = HostPic(rw, rh)
            oracle_lib.mc(dst, refs, rec.mc_units(), intra=intra)
            golden_cases.check_rects(dst, rects, exp, f"case {i} fused {what} {w}x{h}")
            d.ciip_wt = 0
    assert n_gpm >= 192 and len(descs) - n_gpm >= 100


def test_dbf_oracle_matches_reference(built_lib):
    cases = golden_cases.dbf_cases()
    assert len(cases) == 3
    for i, (pic, planes, exp) in enumerate(cases):
        work = pic.copy()
        oracle_lib.dbf(work, planes)
        for name, a, b, c in (("Y", work.y, exp.y, pic.y), ("Cb", work.cb, exp.cb, pic.cb), ("Cr", work.cr, exp.cr, pic.cr)):
            assert (b != c).sum() > 500, "fixture does not exercise the filter"
            bad = np.argwhere(a != b)
            assert len(bad) == 0, f"dbf picture {i} plane {name}: {len(bad)} samples differ, first at (y,x) {bad[:6].tolist()}"


def test_dbf_edge_lists_equal_compacted_planes(built_lib):
    """The edge lists ovhip_rec_dbf_ctu emits CTU by CTU (what ovhip_job_flush uploads) hold exactly the segments
    ovhip_dbf_compact extracts from the dense planes the reference-pinned oracle test consumes."""
    for i, (_, planes, _) in enumerate(golden_cases.dbf_cases()):
        for d in (0, 1):
            direct, offs = planes["edges"][d]
            compact = capi.dbf_compact(planes, d)
            key = lambda a: sorted(zip(a["comp"].tolist(), a["uy"].tolist(), a["ux"].tolist(), a["word"].tolist()))
            assert key(direct) == key(compact), f"picture {i} dir {d}"
            assert len(direct) > 100 and (direct["pad"] == 0).all()
            assert offs.beta[0] == planes["beta_offset"] and offs.tc[0] == planes["tc_offset"]


def test_sao_oracle_matches_reference(built_lib):
    cases = golden_cases.sao_cases()
    assert len(cases) == 3
    for i, (pic, prm, exp) in enumerate(cases):
        out = HostPic(pic.w, pic.h)
        oracle_lib.sao(out, pic, prm)
        for name, a, b, c in (("Y", out.y, exp.y, pic.y), ("Cb", out.cb, exp.cb, pic.cb), ("Cr", out.cr, exp.cr, pic.cr)):
            assert (b != c).sum() > 20, "fixture does not exercise SAO"
            bad = np.argwhere(a != b)
            assert len(bad) == 0, f"sao picture {i} plane {name}: {len(bad)} samples differ, first at (y,x) {bad[:6].tolist()}"


def test_alf_oracle_matches_reference(built_lib):
    cases = golden_cases.alf_cases()
    assert len(cases) == 3
    for i, (pic, alf, exp) in enumerate(cases):
        out = HostPic(pic.w, pic.h)
        oracle_lib.alf(out, pic, alf)
        for name, a, b, c in (("Y", out.y, exp.y, pic.y), ("Cb", out.cb, exp.cb, pic.cb), ("Cr", out.cr, exp.cr, pic.cr)):
            assert (b != c).sum() > 100, "fixture does not exercise ALF"
            bad = np.argwhere(a != b)
            assert len(bad) == 0, f"alf picture {i} plane {name}: {len(bad)} samples differ, first at (y,x) {bad[:6].tolist()}"


def test_fixtures_regenerate_from_the_compiled_reference(tmp_path):
    """Where the compiled reference is available (this container: oracle/_ref built from /root/reference),
    re-running the harness reproduces every committed fixture byte for byte."""
    import subprocess
    from pathlib import Path
    import pytest
    root = Path(__file__).resolve().parent.parent
    gen = root / "oracle" / "_ref" / "gen_golden"
    if not gen.exists() or not (root / "oracle" / "_ref" / "libovvcref.so").exists():
        pytest.skip("compiled reference not present")
    subprocess.check_call([str(gen), str(tmp_path)], stderr=subprocess.DEVNULL)
    # shim_*: test_shim_cpu.py; pipe* / tiles*: the chained streams of gen_pipe, test_pipe_cpu.py
    names = sorted(p.name for p in (root / "tests" / "golden").glob("*.ovg") if not p.name.startswith(("shim_", "pipe", "tiles")))
    assert len(names) >= 9
    for n in names:
        assert (tmp_path / n).read_bytes() == (root / "tests" / "golden" / n).read_bytes(), f"{n} differs from a fresh run of the reference"


def test_intra_oracle_matches_reference(built_lib):
    """Intra prediction: the reference's intra_pred / intra_pred_mrl / mip.rcn_intra_mip / intra_pred_c (+ cclm.*) slots on
    every mode, block shape and neighbour-availability state vs the oracle's ordered-task executor."""
    import golden_io
    g = golden_io.load("intra.ovg")
    tasks = np.frombuffer(g["task"].tobytes(), dtype=capi.ITASK_DTYPE)
    H, W = g["pic_y"].shape
    kinds = {"luma": 0, "mrl": 0, "mip": 0, "chroma": 0, "lm": 0, "bdpcm": 0}
    bad = []
    for i, t in enumerate(tasks):
        pic = HostPic(W, H, g["pic_y"].copy(), g["pic_cb"].copy(), g["pic_cr"].copy())
        oracle_lib.intra_tasks(pic, tasks[i:i + 1])
        w, h, x, y = 1 << int(t["log2_w"]), 1 << int(t["log2_h"]), int(t["x"]), int(t["y"])
        eo = g["exp_off"][i]
        if t["kind"] == capi.IT_LUMA:
            ok = np.array_equal(pic.y[y:y + h, x:x + w], g["exp"][eo[0]:eo[0] + w * h].reshape(h, w))
        else:
            ok = (np.array_equal(pic.cb[y:y + h, x:x + w], g["exp"][eo[0]:eo[0] + w * h].reshape(h, w))
                  and np.array_equal(pic.cr[y:y + h, x:x + w], g["exp"][eo[1]:eo[1] + w * h].reshape(h, w)))
        fl = int(t["flags"])
        k = ("mip" if fl & capi.IF_MIP else "bdpcm" if fl & capi.IF_BDPCM else "mrl" if t["mrl_idx"] else
             "lm" if t["mode"] >= 67 else "luma" if t["kind"] == capi.IT_LUMA else "chroma")
        kinds[k] += 1
        if not ok:
            bad.append((i, k, int(t["mode"]), w, h, x, y, fl, int(t["avl_lft"]), int(t["avl_abv"])))
    assert not bad, f"{len(bad)} / {len(tasks)} intra cases differ from the reference, first: {bad[:6]}"
    assert kinds["luma"] > 2500 and kinds["mrl"] > 500 and kinds["mip"] > 200 and kinds["chroma"] > 1000 and kinds["lm"] > 200 and kinds["bdpcm"] > 50
