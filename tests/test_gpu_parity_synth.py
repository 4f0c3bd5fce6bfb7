"""GPU: full recorded pictures (synthetic, seeded) through the HIP engine vs the CPU oracle,
bit-exact, plus size-independent properties at BASELINE.json's full size."""
import hashlib

import numpy as np
import pytest

import oracle_pipeline
from openvvc_amd import engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(built_lib):
    c = engine.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("w,h,seed", [(416, 240, 0x266), (416, 240, 7), (1920, 1080, 0x266)])
def test_picture_matches_oracle(ctx, w, h, seed):
    wl = synth.make_workload(w, h, seed)
    rp = engine.ResidentPicture(ctx, wl)
    rp.decode()
    y, cb, cr = rp.result()
    ref, mvs = oracle_pipeline.decode(wl, want_mvs=True)
    for name, a, b in (("Y", y, ref.y), ("Cb", cb, ref.cb), ("Cr", cr, ref.cr)):
        assert np.array_equal(a, b), f"{w}x{h} seed {seed}: plane {name}: {int((a != b).sum())} samples differ"
    # every coding tool is present in the picture, and DMVR's motion-vector write-back agrees
    assert all(v > 0 for v in wl.stats["cu_modes"].values()), wl.stats["cu_modes"]
    assert np.array_equal(rp.refined_mvs(), mvs)
    rp.free()


def test_picture_matches_oracle_base_tools(ctx):
    """Translational uni / bi / BCW prediction only, LMCS off (the round-1 workload)."""
    wl = synth.make_workload(832, 480, 5, tools=())
    rp = engine.ResidentPicture(ctx, wl)
    rp.decode()
    y, cb, cr = rp.result()
    ref = oracle_pipeline.decode(wl)
    for name, a, b in (("Y", y, ref.y), ("Cb", cb, ref.cb), ("Cr", cr, ref.cr)):
        assert np.array_equal(a, b), f"plane {name}: {int((a != b).sum())} samples differ"
    rp.free()


def test_4k_full_pipeline(ctx):
    """3840x2160 (BASELINE configs[3]): the whole rcn path (MC, inverse transform, deblocking, SAO,
    ALF/CC-ALF) equals the oracle bit for bit; decoding twice gives the same MD5 (every stage rewrites
    its whole output); MC+ITX alone equal the oracle on the top two CTU rows."""
    wl = synth.make_workload(3840, 2160, 0x266)
    rp = engine.ResidentPicture(ctx, wl)
    rp.decode(("mc", "itx"))
    a = rp.result()
    ref = oracle_pipeline.decode(wl, rows=(0, 256), stages=("mc", "itx"))
    assert np.array_equal(a[0][:256], ref.y[:256])
    assert np.array_equal(a[1][:128], ref.cb[:128]) and np.array_equal(a[2][:128], ref.cr[:128])
    rp.decode()
    a = rp.result()
    rp.decode()
    b = rp.result()
    md5 = lambda planes: hashlib.md5(b"".join(p.tobytes() for p in planes)).hexdigest()
    assert md5(a) == md5(b)
    full = oracle_pipeline.decode(wl)
    for name, x, y in (("Y", a[0], full.y), ("Cb", a[1], full.cb), ("Cr", a[2], full.cr)):
        assert np.array_equal(x, y), f"4K plane {name}: {int((x != y).sum())} samples differ from the oracle"
    rp.free()


@pytest.mark.parametrize("stage", ["dbf", "sao", "alf"])
def test_stage_isolated(ctx, stage):
    """Each in-loop filter alone (1080p) on the oracle's input for that stage."""
    wl = synth.make_workload(1920, 1080, 11)
    order = list(oracle_pipeline.STAGES)
    upto = order[:order.index(stage)]
    before = oracle_pipeline.decode(wl, stages=tuple(upto))
    after = oracle_pipeline.decode(wl, stages=tuple(upto + [stage]))
    src = ctx.upload_pic(before.y, before.cb, before.cr)
    dst = ctx.new_pic(wl.w, wl.h)
    if stage == "dbf":
        ctx.dbf(src, engine.DevDbfPlanes(ctx, wl.dbf_planes)); out = src
    elif stage == "sao":
        ctx.sao(dst, src, ctx.upload(wl.sao_params)); out = dst
    else:
        # oracle_pipeline keeps SAO output in a temporary when ALF follows: rebuild ALF input = SAO output
        sao_out = oracle_pipeline.decode(wl, stages=("mc", "itx", "dbf", "sao"))
        src = ctx.upload_pic(sao_out.y, sao_out.cb, sao_out.cr)
        ctx.alf(dst, src, engine.DevAlf(ctx, wl.alf, wl.w, wl.h)); out = dst
    ctx.sync()
    y, cb, cr = out.download()
    for name, a, b in (("Y", y, after.y), ("Cb", cb, after.cb), ("Cr", cr, after.cr)):
        assert np.array_equal(a, b), f"{stage} plane {name}: {int((a != b).sum())} samples differ"


def test_merged_launches_equal_standalone(ctx):
    """ovhip_mcxa_launch == ovhip_mcx_launch + ovhip_mca_launch, and ovhip_itx_launch_chroma_lmcs ==
    ovhip_itx_launch_classes(chroma) + ovhip_lmcs_inverse_launch: same picture, same refined MVs."""
    wl = synth.make_workload(832, 480, 21)
    rp = engine.ResidentPicture(ctx, wl)          # merged path (what decode() uses when both kinds of unit are present)
    assert rp.mcx_units and rp.aff_units and rp.lmcs is not None and wl.tb_classes[3]
    rp.decode(("mc", "itx"))
    merged = rp.result()
    mv_merged = rp.refined_mvs().copy()
    # the same stages through the stand-alone entry points
    c = rp.ctx
    c.mc(rp.dst, rp.refs, rp.mc_units, rp.lmcs_fwd, intra=rp.intra)
    c.mcx(rp.dst, rp.refs, rp.mcx_units, rp.lmcs_fwd, rp.mv_out)
    c.mca(rp.dst, rp.refs, rp.aff_units, rp.aff_side, rp.lmcs_fwd)
    if rp.ciip_units:
        c.ciip(rp.dst, rp.intra, rp.ciip_units)
    k = wl.tb_classes
    c.itx_classes(rp.dst, rp.tb_cmds, rp.coefs, 0, k[0], k[1])
    c.lmcs_scale(rp.dst, rp.lmcs_regions, rp.lmcs, rp.lmcs_scales)
    c.itx_classes(rp.dst, rp.tb_cmds, rp.coefs, k[0] + k[1], k[2], k[3], rp.lmcs_scales)
    c.lmcs_inverse(rp.dst, rp.lmcs_bwd)
    alone = rp.result()
    for name, a, b in zip("Y Cb Cr".split(), merged, alone):
        assert np.array_equal(a, b), f"plane {name}: {int((a != b).sum())} samples differ"
    assert np.array_equal(mv_merged, rp.refined_mvs())
    ref = oracle_pipeline.decode(wl, stages=("mc", "itx"))
    assert np.array_equal(alone[0], ref.y) and np.array_equal(alone[1], ref.cb) and np.array_equal(alone[2], ref.cr)
    rp.free()


def test_two_pictures_in_flight(built_lib):
    """Two contexts on two streams, launches interleaved picture by picture (bench.py's default mode): both pictures
    equal the oracle -- the engine keeps no state outside the context and the picture's own buffers."""
    import torch
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    ctxs = [engine.Context(0, stream=s.cuda_stream) for s in streams]
    wls = [synth.make_workload(832, 480, 31), synth.make_workload(832, 480, 32)]
    rps = [engine.ResidentPicture(c, w) for c, w in zip(ctxs, wls)]
    for _ in range(3):                                    # several rounds back to back, no synchronisation in between
        for rp in rps:
            rp.decode()
    torch.cuda.synchronize()
    for rp, wl in zip(rps, wls):
        y, cb, cr = rp.result()
        ref = oracle_pipeline.decode(wl)
        assert np.array_equal(y, ref.y) and np.array_equal(cb, ref.cb) and np.array_equal(cr, ref.cr)
        rp.free()
    for c in ctxs:
        c.close()
