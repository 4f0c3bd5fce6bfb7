"""GPU: the command streams the installed slots of the reference-side override block recorded (tests/golden/shim_*.ovg,
shim_cases.py) replayed through the HIP engine reproduce the bytes the reference's scalar slots produced."""
import numpy as np
import pytest

import golden_cases
import golden_io
from oracle_lib import HostPic
from openvvc_amd import capi, engine
from shim_cases import ShimStream

pytestmark = pytest.mark.gpu
BAND = 128


@pytest.fixture(scope="module")
def ctx(built_lib):
    c = engine.Context(0)
    yield c
    c.close()


def test_shim_tu_streams_gpu(ctx):
    """All 425 rcn_tu_st / rcn_tu_c cases + the 12 transform trees, each on its own 128-row band, ONE launch per stream."""
    g = golden_io.load("itx.ovg")
    for stream, key_n, tree_of in (("shim_itx.ovg", "desc", None), ("shim_itx_tree.ovg", "tt_desc", 0)):
        s = ShimStream(stream)
        n = s.n
        tall = ctx.upload_pic(np.tile(g["pred_y"], (n, 1)), np.tile(g["pred_cb"], (n, 1)), np.tile(g["pred_cr"], (n, 1)))
        cmds, coefs, rects, coef_base, tasks = [], [], [], 0, []
        for i in range(n):
            c = s.case(i)
            tb = c["tb"]
            tb["y"] += np.where(tb["plane"] == 0, i * BAND, i * (BAND // 2)).astype(np.uint16)
            tb["coef_off"] += coef_base
            coef_base += len(c["coef"])
            cmds.append(tb); coefs.append(c["coef"])
            # TUs of intra CUs: the chroma prediction task keeps its residual half only (the reference run behind itx.ovg had
            # intra_pred_c stubbed out, test_shim_cpu.run_tu_stream)
            t = c["itask"]
            t["kind"] = capi.IT_RES_C
            t["y"] += np.uint16(i * (BAND // 2))
            tasks.append(t)
            if tree_of is None:
                d = capi.TuDesc.from_buffer_copy(g["desc"][i].tobytes())
                x0, y0, w, h, eo = d.x0, d.y0, 1 << d.log2_tb_w, 1 << d.log2_tb_h, g["exp_off"][i]
                if d.tree == 0:
                    rects += [(0, x0, y0 + i * BAND, w, h, int(eo[0])), (1, x0 >> 1, (y0 >> 1) + i * 64, w >> 1, h >> 1, int(eo[1])),
                              (2, x0 >> 1, (y0 >> 1) + i * 64, w >> 1, h >> 1, int(eo[2]))]
                else:
                    rects += [(1, x0, y0 + i * 64, w, h, int(eo[1])), (2, x0, y0 + i * 64, w, h, int(eo[2]))]
            else:
                d = capi.TtDesc.from_buffer_copy(g["tt_desc"][i].tobytes())
                w, h, eo = 1 << d.log2_w, 1 << d.log2_h, g["tt_exp_off"][i]
                rects += [(0, 0, i * BAND, w, h, int(eo[0])), (1, 0, i * 64, w >> 1, h >> 1, int(eo[1])), (2, 0, i * 64, w >> 1, h >> 1, int(eo[2]))]
        # through the C flush (residual launches, STORE mode for the blocks of ordered tasks, then the ordered pass)
        job = engine.Job(ctx, 128, BAND * n)
        job.begin()
        job.rec.append_raw(capi.REC_COEF, np.concatenate(coefs))
        job.rec.append_raw(capi.REC_TB, np.concatenate(cmds))
        job.rec.append_raw(capi.REC_ITASK, np.concatenate(tasks))
        p = capi.JobParams()
        p.log2_ctu_s = 7
        p.stages = capi.STAGE_ITX | capi.STAGE_INTRA
        job.flush(tall, [], None, params=p)
        job.wait()
        job.close()
        y, cb, cr = tall.download()
        golden_cases.check_rects(HostPic(128, BAND * n, y, cb, cr), rects, g["exp"], f"{stream} HIP vs reference")


def _tall(ctx, rw, rh, n):
    fill = np.full((rh * n, rw), 0xABAB, np.uint16)
    return ctx.upload_pic(fill, fill[: rh * n // 2, : rw // 2], fill[: rh * n // 2, : rw // 2])


def _rects(d, exp_off, i, rh):
    w, h = 1 << d.log2_w, 1 << d.log2_h
    return [(0, d.x0, d.y0 + i * rh, w, h, int(exp_off[i, 0])), (1, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 1])),
            (2, d.x0 >> 1, (d.y0 >> 1) + i * (rh // 2), w >> 1, h >> 1, int(exp_off[i, 2]))]


def test_shim_prediction_streams_gpu(ctx):
    """rcn_mcp_b*, GPM / CIIP, BDOF / DMVR (with the refined vectors) and affine streams."""
    # ---- plain + GPM / CIIP
    for stream, loader in (("shim_mc.ovg", "mc"), ("shim_gpm.ovg", "gpm")):
        if loader == "mc":
            refs, descs, exp_off, exp = golden_cases.mc_cases(); intra = None; n_gpm = len(descs)
        else:
            refs, intra_h, descs, _, n_gpm, exp_off, exp = golden_cases.gpm_cases()
            intra = ctx.upload_pic(intra_h.y, intra_h.cb, intra_h.cr)
        s = ShimStream(stream)
        rw, rh, n = refs[0].w, refs[0].h, len(descs)
        drefs = s.refs([ctx.upload_pic(r.y, r.cb, r.cr) for r in refs])
        tall = _tall(ctx, rw, rh, n)
        first_planar = n
        if loader == "gpm":
            # the CIIP cases with the real planar prediction start from the current picture
            cur, first_planar = golden_cases.ciip_planar_cases()
            fy, fcb, fcr = tall.download()
            for i in range(first_planar, n):
                fy[i * rh:(i + 1) * rh] = cur.y; fcb[i * rh // 2:(i + 1) * rh // 2] = cur.cb; fcr[i * rh // 2:(i + 1) * rh // 2] = cur.cr
            tall.upload(fy, fcb, fcr)
            res = ctx.new_pic(rw, rh)
        rects = []
        for i, d in enumerate(descs):
            c = s.case(i)
            band = tall.band(i * rh, rh)
            ctx.mc(band, drefs, ctx.upload(c["mc"]))
            t = c["itask"]
            if len(t) and i < first_planar:
                # planar stubbed in the reference run: the task's geometry and weight through the fused blend
                u = np.zeros(1, capi.CIIP_UNIT_DTYPE)
                u["x"], u["y"], u["log2_w"], u["log2_h"], u["wt"], u["chroma_inter"] = d.x0, d.y0, d.log2_w, d.log2_h, t[0]["ciip_wt"], d.log2_w <= 2
                ctx.ciip(band, intra, ctx.upload(u))
            elif len(t):
                lv = t["level"]
                for l in np.unique(lv):
                    tl = t[lv == l]
                    ctx.intra_level(band, res, ctx.upload(tl), 0, len(tl))
            rects += _rects(d, exp_off, i, rh)
        ctx.sync()
        y, cb, cr = tall.download()
        golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, f"{stream} HIP vs reference")
    # ---- refined
    refs, descs, exp_off, exp, exp_mv = golden_cases.mcx_cases()
    s = ShimStream("shim_mcx.ovg")
    rw, rh, n = refs[0].w, refs[0].h, len(descs)
    drefs = s.refs([ctx.upload_pic(r.y, r.cb, r.cr) for r in refs])
    tall = _tall(ctx, rw, rh, n)
    rects, mv_checks = [], []
    for i, d in enumerate(descs):
        c = s.case(i)
        band = tall.band(i * rh, rh)
        if len(c["mc"]):
            ctx.mc(band, drefs, ctx.upload(c["mc"]))
        mv = ctx.alloc(len(c["mcx"]) * 16)
        ctx.mcx(band, drefs, ctx.upload(c["mcx"]), mv_out=mv)
        if d.refine & capi.PU_DMVR:
            mv_checks.append((i, mv, exp_mv[int(exp_off[i, 3]) // 4:int(exp_off[i, 3]) // 4 + len(c["mcx"])]))
        rects += _rects(d, exp_off, i, rh)
    ctx.sync()
    for i, buf, want in mv_checks:
        assert np.array_equal(buf.download(np.int32).reshape(-1, 4), want), f"shim mcx case {i}: refined MVs differ"
    y, cb, cr = tall.download()
    golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, "shim_mcx.ovg HIP vs reference")
    # ---- affine
    refs, cases, exp_off, exp = golden_cases.mca_cases()
    s = ShimStream("shim_mca.ovg")
    n = len(cases)
    drefs = s.refs([ctx.upload_pic(r.y, r.cb, r.cr) for r in refs])
    tall = _tall(ctx, rw, rh, n)
    rects = []
    for i, (d, _, _) in enumerate(cases):
        c = s.case(i)
        ctx.mca(tall.band(i * rh, rh), drefs, ctx.upload(c["aff"]), ctx.upload(c["side"]))
        rects += _rects(d, exp_off, i, rh)
    ctx.sync()
    y, cb, cr = tall.download()
    golden_cases.check_rects(HostPic(rw, rh * n, y, cb, cr), rects, exp, "shim_mca.ovg HIP vs reference")


def test_shim_dbf_edge_lists_gpu(ctx):
    """df.rcn_dbf_ctu(_truncated) through the installed table -> edge lists -> k_dbf_list with the per-slice offset table."""
    import ctypes as C
    gd = golden_io.load("shim_dbf.ovg")
    for i, (pic, _, exp) in enumerate(golden_cases.dbf_cases()):
        d = ctx.upload_pic(pic.y, pic.cb, pic.cr)
        ev = np.frombuffer(gd[f"p{i}_edges_v"].tobytes(), dtype=capi.DBF_EDGE_DTYPE)
        eh = np.frombuffer(gd[f"p{i}_edges_h"].tobytes(), dtype=capi.DBF_EDGE_DTYPE)
        offs = capi.DbfOffsets.from_buffer_copy(gd[f"p{i}_offsets"].tobytes())
        dv, dh = ctx.upload(ev), ctx.upload(eh)
        ctx._chk(ctx.lib.ovhip_dbf_launch_edges_ex(ctx.h, C.byref(d.s), dv.ptr, len(ev), dh.ptr, len(eh), C.byref(offs)), "dbf_launch_edges_ex")
        ctx.sync()
        y, cb, cr = d.download()
        for name, a, b in (("Y", y, exp.y), ("Cb", cb, exp.cb), ("Cr", cr, exp.cr)):
            assert np.array_equal(a, b), f"dbf picture {i} plane {name}: {int((a != b).sum())} samples differ"


def test_shim_intra_ctus_gpu(ctx):
    """The 36 intra CTUs the installed slots recorded (shim_intra_ctu.ovg) through the C flush (prediction-free picture, residual
    launches in STORE mode, ordered pass by level and as the one-launch CTU pass): the CTU the reference's slots left."""
    from test_shim_cpu import intra_ctu_cases
    base, cases = intra_ctu_cases()
    s = ShimStream("shim_intra_ctu.ovg")
    h, w = base[0].shape
    job = engine.Job(ctx, w, h)
    dst = ctx.new_pic(w, h)
    for one_launch in (0, capi.STAGE_INTRA_CTU, capi.STAGE_INTRA_LEVELS):
        for i, (dual, n_intra, ey, ecb, ecr) in enumerate(cases):
            c = s.case(i)
            dst.upload(*base)
            job.begin()
            job.rec.append_raw(capi.REC_COEF, c["coef"])
            job.rec.append_raw(capi.REC_TB, c["tb"])
            job.rec.append_raw(capi.REC_ITASK, c["itask"])
            p = capi.JobParams()
            p.log2_ctu_s = 7
            p.stages = capi.STAGE_ITX | capi.STAGE_INTRA | one_launch
            job.flush(dst, [], None, params=p)
            job.wait()
            y, cb, cr = dst.download()
            for name, got, exp in (("Y", y[128:256, 128:256], ey), ("Cb", cb[64:128, 64:128], ecb), ("Cr", cr[64:128, 64:128], ecr)):
                assert np.array_equal(got, exp), f"intra CTU {i} (dual={dual}, one_launch={one_launch}) plane {name}: {int((got != exp).sum())} samples differ"
            # nothing outside the CTU changes
            y[128:256, 128:256] = base[0][128:256, 128:256]
            assert np.array_equal(y, base[0])
    job.close()


def test_shim_isp_cus_gpu(ctx):
    """The 394 intra-sub-partition CUs the installed recon_isp_subtree_v / _h slots recorded (shim_isp.ovg) through the C flush:
    thin (1xN, 2xN, Nx1, Nx2) and regular transform blocks in STORE mode, the partitions' prediction calls as chained ordered tasks."""
    from test_shim_cpu import isp_cases
    base, info, exp = isp_cases()
    s = ShimStream("shim_isp.ovg")
    h, w = base.shape
    cb = np.full((h // 2, w // 2), 512, np.uint16)
    job = engine.Job(ctx, w, h)
    dst = ctx.new_pic(w, h)
    bad = []
    thin_on_flow = 0
    for i, (x, y, l2w, l2h, vertical, mode, bits, off) in enumerate(info):
        c = s.case(i)
        dst.upload(base, cb, cb)
        job.begin()
        job.rec.append_raw(capi.REC_COEF, c["coef"])
        job.rec.append_raw(capi.REC_TB, c["tb"])
        job.rec.append_raw(capi.REC_ITASK, c["itask"])
        p = capi.JobParams()
        p.log2_ctu_s = 7
        p.stages = capi.STAGE_ITX | capi.STAGE_INTRA | (0, capi.STAGE_INTRA_CTU, capi.STAGE_INTRA_LEVELS)[i % 3]
        if i % 3 == 0:
            # the default path is the flow launch for EVERY one of them -- also the partitions less than a unit high (8x2, 16x1 ...: until
            # round 5 one of those sent the whole picture to the per-level launches)
            t, _ = job.rec.itasks_sorted()
            items = np.zeros(4 * len(t) + 16, np.uint32)
            assert ctx.lib.ovhip_intra_flow_items(t.ctypes.data, len(t), items.ctypes.data, len(items)) > 0
            thin_on_flow += int(((t["kind"] == capi.IT_LUMA) & (t["log2_h"] < 2)).any())
        job.flush(dst, [], None, params=p)
        job.wait()
        yy = dst.download()[0]
        bw, bh = 1 << int(l2w), 1 << int(l2h)
        if not np.array_equal(yy[y:y + bh, x:x + bw], exp[int(off):int(off) + bw * bh].reshape(bh, bw)):
            bad.append((i, bw, bh, int(vertical), int(mode), hex(int(bits))))
    job.close()
    assert not bad, f"{len(bad)} / {len(info)} ISP CUs differ from the reference on the GPU, first: {bad[:10]}"
    assert thin_on_flow >= 10, thin_on_flow


@pytest.mark.gpu
def test_tmvp_plane_cells_gpu(ctx):
    """the device's plane-cell derivation for the refined units == the entries the reference's flow wrote (fixture
    tmvp_expected) and == the oracle, for every DMVR case; then the same through the job (ovhip_job_params.tmvp_cells) on a
    synthetic 1080p picture, where the cells must carry the vectors the flush refined."""
    import oracle_lib
    from openvvc_amd import synth
    refs, descs, exp_off, exp, exp_mv = golden_cases.mcx_cases()
    s = ShimStream("shim_mcx.ovg")
    nb_ctb_w = (refs[0].w + 127) // 128
    e = np.asarray(s.g["tmvp_expected"]).reshape(-1, 3)
    n = 0
    for i, d in enumerate(descs):
        units = s.case(i)["mcx"]
        if not len(units):
            continue
        fake = (1000 + 8 * np.arange(len(units))[:, None] + np.arange(4)[None, :]).astype(np.int32)
        got = ctx.tmvp_cells(ctx.upload(units), len(units), ctx.upload(fake), 7, nb_ctb_w)
        assert np.array_equal(got, oracle_lib.tmvp_cells(units, fake, 7, nb_ctb_w)), f"case {i}"
        have = {(int(c["cell"]), int((c["mv0x"] - 1000) // 8)) for c in got if c["cell"] != capi.TMVP_NONE}
        assert have == {(int(c), int(u)) for k, c, u in e if k == i}, f"case {i}"
        n += len(have)
    assert n > 600
    # through the job
    w, h = 1920, 1080
    wl = synth.make_workload(w, h, 0x515)
    job = engine.Job(ctx, w, h)
    drefs = [ctx.upload_pic(*r) for r in wl.refs]
    dst = ctx.new_pic(w, h)
    job.load_workload(wl)
    job.params.tmvp_cells = 1
    job.flush(dst, drefs, None)
    job.wait()
    mv = job.refined_mvs()
    cells = job.tmvp_cells()
    assert len(cells) == 4 * len(mv) and len(mv) == len(wl.mcx_units)
    assert np.array_equal(cells, oracle_lib.tmvp_cells(wl.mcx_units, mv, 7, (w + 127) // 128))
    used = cells[cells["cell"] != capi.TMVP_NONE]
    assert len(used) > 100 and used["cell"].max() < 16 * ((w + 127) // 128) * 16 * ((h + 127) // 128)
