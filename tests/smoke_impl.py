"""smoke(): one small recorded picture through the C-side flush (ovhip_job_flush) on cuda:0, checked against the oracle."""
import numpy as np


def run():
    import torch
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    from openvvc_amd import engine, synth
    import oracle_pipeline
    wl = synth.make_workload(416, 240, 0x266, tools=synth.INTRA_TOOLS, intra_frac=0.2)       # every tool incl. intra CUs (ordered pass)
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
    # output path: the device's digest of the cropped frame against the restatement of dectest's writer
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "oracle"))
    import ovvc_oracle_output as oo
    win = (1, 2, 0, 3)
    assert dst.digest(win) == oo.picture_digest(ref.y, ref.cb, ref.cr, win), "smoke: output digest differs"
    assert dst.output(win).tobytes() == oo.packed_frame(ref.y, ref.cb, ref.cr, win), "smoke: packed output frame differs"
    st = job.stats()
    print(f"smoke ok: 416x240 recorded picture through ovhip_job_flush: {st.n_launches} launches, {st.h2d_bytes} B H2D, "
          f"{wl.stats['n_mc_units']} MC units, {wl.stats['n_tb_cmds']} TB commands, {wl.stats['n_itasks']} ordered tasks in {wl.stats['n_ilevels']} levels, bit-exact vs oracle incl. refined MVs, cropped output frame and its digest")
    job.close()
    ctx.close()
