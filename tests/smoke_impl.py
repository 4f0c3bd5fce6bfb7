"""smoke(): one small recorded picture through the C-side flush (ovhip_job_flush) on cuda:0, checked against the oracle."""
import numpy as np


def run():
    import torch
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    from openvvc_amd import engine, synth
    import oracle_pipeline
    wl = synth.make_workload(416, 240)
    ctx = engine.Context(0)
    job = engine.Job(ctx, wl.w, wl.h)
    refs = [ctx.upload_pic(*r) for r in wl.refs]
    intra = ctx.upload_pic(*wl.intra) if wl.intra is not None else None
    dst = ctx.new_pic(wl.w, wl.h)
    job.load_workload(wl)
    job.flush(dst, refs, intra)
    job.wait()
    y, cb, cr = dst.download()
    ref, mvs = oracle_pipeline.decode(wl, want_mvs=True)
    for name, a, b in (("Y", y, ref.y), ("Cb", cb, ref.cb), ("Cr", cr, ref.cr)):
        assert np.array_equal(a, b), f"smoke: plane {name} differs from the oracle ({int((a != b).sum())} samples)"
    assert np.array_equal(job.refined_mvs(), mvs), "smoke: refined motion vectors differ from the oracle"
    st = job.stats()
    print(f"smoke ok: 416x240 recorded picture through ovhip_job_flush: {st.n_launches} launches, {st.h2d_bytes} B H2D, "
          f"{wl.stats['n_mc_units']} MC units, {wl.stats['n_tb_cmds']} TB commands, bit-exact vs oracle incl. refined MVs")
    job.close()
    ctx.close()
