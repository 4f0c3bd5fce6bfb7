"""GPU: the HIP engine (through the C ABI) reproduces the compiled reference bit-exactly on the
committed golden fixtures."""
import numpy as np
import pytest

import golden_cases
from oracle_lib import HostPic
from openvvc_amd import capi, engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built_lib):
    c = engine.Context(0)
    yield c
    c.close()


def test_itx_gpu_matches_reference(ctx):
    pic, cmds, coefs, rects, exp = golden_cases.itx_cases()
    d = ctx.upload_pic(pic.y, pic.cb, pic.cr)
    ctx.itx(d, ctx.upload(cmds), ctx.upload(coefs))
    ctx.sync()
    y, cb, cr = d.download()
    golden_cases.check_rects(HostPic(pic.w, pic.h, y, cb, cr), rects, exp, "itx HIP vs reference")


def test_transform_tree_gpu_matches_reference(ctx):
    pic, cmds, coefs, rects, exp = golden_cases.tt_cases()
    d = ctx.upload_pic(pic.y, pic.cb, pic.cr)
    ctx.itx(d, ctx.upload(cmds), ctx.upload(coefs))
    ctx.sync()
    y, cb, cr = d.download()
    golden_cases.check_rects(HostPic(pic.w, pic.h, y, cb, cr), rects, exp, "transform tree HIP vs reference")


def test_mc_gpu_matches_reference(ctx):
    refs, descs, exp_off, exp = golden_cases.mc_cases()
    rw, rh = refs[0].w, refs[0].h
    n = len(descs)
    drefs = [ctx.upload_pic(r.y, r.cb, r.cr) for r in refs]
    fill = np.full((rh * n, rw), 0xABAB, np.uint16)
    tall = ctx.upload_pic(fill, fill[: rh * n // 2, : rw // 2], fill[: rh * n // 2, : rw // 2])
    rec = capi.Recorder(rw, rh)
    rects = []
    for i, d in enumerate(descs):
        rec.reset()
        rec.pu(d)
        ctx.mc(tall.band(i * rh, rh), drefs, ctx.upload(rec.mc_units()))
        w, h = 1 << d.log2_w, 1 << d.log2_h
        rects += [(0, d.x0, d.y0 + i * rh, w, h, int(exp_off[i, 0])),
                  (1, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 1])),
                  (2, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 2]))]
    ctx.sync()
    y, cb, cr = tall.download()
    golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, "mc HIP vs reference")


def test_mcx_gpu_matches_reference(ctx):
    """BDOF / DMVR units through ovhip_mcx_launch vs the reference's rcn_bdof_mcp_l / rcn_dmvr_mv_refine,
    samples and the refined motion vectors."""
    refs, descs, exp_off, exp, exp_mv = golden_cases.mcx_cases()
    rw, rh = refs[0].w, refs[0].h
    n = len(descs)
    drefs = [ctx.upload_pic(r.y, r.cb, r.cr) for r in refs]
    fill = np.full((rh * n, rw), 0xABAB, np.uint16)
    tall = ctx.upload_pic(fill, fill[: rh * n // 2, : rw // 2], fill[: rh * n // 2, : rw // 2])
    rec = capi.Recorder(rw, rh)
    rects, mv_checks = [], []
    for i, d in enumerate(descs):
        rec.reset()
        rec.pu(d)
        band = tall.band(i * rh, rh)
        if len(rec.mc_units()):
            ctx.mc(band, drefs, ctx.upload(rec.mc_units()))
        ux = rec.mcx_units()
        mv = ctx.alloc(len(ux) * 16)
        ctx.mcx(band, drefs, ctx.upload(ux), mv_out=mv)
        if d.refine & capi.PU_DMVR:
            mv_checks.append((i, mv, exp_mv[int(exp_off[i, 3]) // 4:int(exp_off[i, 3]) // 4 + len(ux)]))
        w, h = 1 << d.log2_w, 1 << d.log2_h
        rects += [(0, d.x0, d.y0 + i * rh, w, h, int(exp_off[i, 0])),
                  (1, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 1])),
                  (2, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 2]))]
    ctx.sync()
    bad_mv = []
    for i, buf, want in mv_checks:
        got = buf.download(np.int32).reshape(-1, 4)
        if not np.array_equal(got, want):
            bad_mv.append((i, got.tolist(), want.tolist()))
    assert not bad_mv, f"refined MVs differ in {len(bad_mv)} / {len(mv_checks)} DMVR cases, first: {bad_mv[:3]}"
    y, cb, cr = tall.download()
    golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, "mcx HIP vs reference")


def test_mca_gpu_matches_reference(ctx):
    """Affine sub-block prediction + PROF through ovhip_mca_launch vs the reference's rcn_mcp_b_l(2,2) /
    rcn_prof_mcp_b_l / rcn_mcp_b_c(3,3) call sequence."""
    refs, cases, exp_off, exp = golden_cases.mca_cases()
    rw, rh = refs[0].w, refs[0].h
    n = len(cases)
    drefs = [ctx.upload_pic(r.y, r.cb, r.cr) for r in refs]
    fill = np.full((rh * n, rw), 0xABAB, np.uint16)
    tall = ctx.upload_pic(fill, fill[: rh * n // 2, : rw // 2], fill[: rh * n // 2, : rw // 2])
    rec = capi.Recorder(rw, rh)
    rects = []
    for i, (d, mv0, mv1) in enumerate(cases):
        rec.reset()
        rec.affine_cu(d, mv0, mv1)
        ctx.mca(tall.band(i * rh, rh), drefs, ctx.upload(rec.aff_units()), ctx.upload(rec.aff_side()))
        w, h = 1 << d.log2_w, 1 << d.log2_h
        rects += [(0, d.x0, d.y0 + i * rh, w, h, int(exp_off[i, 0])),
                  (1, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 1])),
                  (2, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 2]))]
    ctx.sync()
    y, cb, cr = tall.download()
    golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, "mca HIP vs reference")


def test_lmcs_gpu_matches_reference(ctx):
    """K11: device chroma-scale derivation vs rcn_lmcs_compute_chroma_scale, device inverse map vs lmcs_reshape_backward."""
    pic_y, sets, regions, inverse = golden_cases.lmcs_cases()
    h, w = pic_y.shape
    zc = np.zeros((h // 2, w // 2), np.uint16)
    rec = capi.Recorder(w, h)
    n_inv = 0
    for si, (data, want) in enumerate(sets):
        luts = capi.lmcs_build(data)
        rows = regions[regions[:, 0] == si]
        rec.reset()
        for r in rows:
            rec.lmcs_region(int(r[1]), int(r[2]), int(r[3]), int(r[4]))
        d = ctx.upload_pic(pic_y, zc, zc)
        scales = ctx.alloc(2 * len(rows))
        ctx.lmcs_scale(d, ctx.upload(rec.lmcs_regions()), luts, scales)
        got = scales.download(np.int16)
        assert np.array_equal(got, rows[:, 5].astype(np.int16)), f"chroma scales of set {si}: {got.tolist()} vs {rows[:, 5].tolist()}"
        if si % 6 == 1:
            ctx.lmcs_inverse(d, ctx.upload(np.frombuffer(bytes(luts), np.uint16)[1024:2048].copy()))
            ctx.sync()
            y, _, _ = d.download()
            exp = np.concatenate([inverse[n_inv, 0], inverse[n_inv, 1]], axis=1)
            assert np.array_equal(y, exp), f"inverse map of set {si} differs"
            n_inv += 1
    assert n_inv == len(inverse)


def test_gpm_ciip_gpu_matches_reference(ctx):
    """K10: GPM units through ovhip_mc_launch, CIIP = plain MC + ovhip_ciip_launch, vs rcn_gpm_b / rcn_ciip(_b)."""
    refs, intra, descs, modes, n_gpm, exp_off, exp = golden_cases.gpm_cases()
    descs = descs[:golden_cases.ciip_planar_cases()[1]]        # the rest: CIIP through ordered tasks (test_gpu_shim_replay)
    rw, rh = refs[0].w, refs[0].h
    n = len(descs)
    drefs = [ctx.upload_pic(r.y, r.cb, r.cr) for r in refs]
    dintra = ctx.upload_pic(intra.y, intra.cb, intra.cr)
    fill = np.full((rh * n, rw), 0xABAB, np.uint16)
    tall = ctx.upload_pic(fill, fill[: rh * n // 2, : rw // 2], fill[: rh * n // 2, : rw // 2])
    rec = capi.Recorder(rw, rh)
    rects = []
    for i, d in enumerate(descs):
        rec.reset()
        rec.pu(d)
        band = tall.band(i * rh, rh)
        if i >= n_gpm and i % 2:
            # CIIP, route 2: the blend fused into the prediction units (ovhip_pu_desc.ciip_wt + intra picture)
            d.ciip_wt = ctx.lib.ovhip_ciip_weight(int(modes[i, 0]), int(modes[i, 1]))
            rec.reset()
            rec.pu(d)
            ctx.mc(band, drefs, ctx.upload(rec.mc_units()), intra=dintra)
            d.ciip_wt = 0
        else:
            ctx.mc(band, drefs, ctx.upload(rec.mc_units()))
            if i >= n_gpm:                      # CIIP, route 1: separate blend launch
                rec.ciip(d.x0, d.y0, d.log2_w, d.log2_h, int(modes[i, 0]), int(modes[i, 1]))
                ctx.ciip(band, dintra, ctx.upload(rec.ciip_units()))
        w, h = 1 << d.log2_w, 1 << d.log2_h
        rects += [(0, d.x0, d.y0 + i * rh, w, h, int(exp_off[i, 0])),
                  (1, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 1])),
                  (2, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 2]))]
    ctx.sync()
    y, cb, cr = tall.download()
    golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, "gpm/ciip HIP vs reference")


def test_dbf_gpu_matches_reference(ctx):
    """Both drivers of the deblocking kernel: dense edge planes and the compact edge lists."""
    for i, (pic, planes, exp) in enumerate(golden_cases.dbf_cases()):
        for mode in ("planes", "edges"):
            d = ctx.upload_pic(pic.y, pic.cb, pic.cr)
            if mode == "planes":
                ctx.dbf(d, engine.DevDbfPlanes(ctx, planes))
            else:
                ev, eh = capi.dbf_compact(planes, 0), capi.dbf_compact(planes, 1)
                assert 0 < len(ev) < planes["w4"] * planes["h4"] and len(eh) > 0
                ctx.dbf_edges(d, ctx.upload(ev), ctx.upload(eh), planes["beta_offset"], planes["tc_offset"])
            ctx.sync()
            y, cb, cr = d.download()
            for name, a, b in (("Y", y, exp.y), ("Cb", cb, exp.cb), ("Cr", cr, exp.cr)):
                bad = np.argwhere(a != b)
                assert len(bad) == 0, f"dbf HIP ({mode}) vs reference, picture {i} plane {name}: {len(bad)} differ, first {bad[:6].tolist()}"


def test_sao_gpu_matches_reference(ctx):
    for i, (pic, prm, exp) in enumerate(golden_cases.sao_cases()):
        src = ctx.upload_pic(pic.y, pic.cb, pic.cr)
        dst = ctx.new_pic(pic.w, pic.h)
        ctx.sao(dst, src, ctx.upload(prm))
        ctx.sync()
        y, cb, cr = dst.download()
        for name, a, b in (("Y", y, exp.y), ("Cb", cb, exp.cb), ("Cr", cr, exp.cr)):
            bad = np.argwhere(a != b)
            assert len(bad) == 0, f"sao HIP vs reference, picture {i} plane {name}: {len(bad)} differ, first {bad[:6].tolist()}"


def test_alf_gpu_matches_reference(ctx):
    for i, (pic, alf, exp) in enumerate(golden_cases.alf_cases()):
        src = ctx.upload_pic(pic.y, pic.cb, pic.cr)
        dst = ctx.new_pic(pic.w, pic.h)
        ctx.alf(dst, src, engine.DevAlf(ctx, alf, pic.w, pic.h))
        ctx.sync()
        y, cb, cr = dst.download()
        for name, a, b in (("Y", y, exp.y), ("Cb", cb, exp.cb), ("Cr", cr, exp.cr)):
            bad = np.argwhere(a != b)
            assert len(bad) == 0, f"alf HIP vs reference, picture {i} plane {name}: {len(bad)} differ, first {bad[:6].tolist()}"
