"""smoke(): one small recorded picture through the HIP engine on cuda:0, checked against the oracle."""
import numpy as np


def run():
    import torch
    assert torch.cuda.is_available(), "smoke() needs the MI355X"
    from openvvc_amd import engine, synth
    import oracle_pipeline
    wl = synth.make_workload(416, 240)
    ctx = engine.Context(0)
    rp = engine.ResidentPicture(ctx, wl)
    rp.decode()
    y, cb, cr = rp.result()
    ref = oracle_pipeline.decode(wl)
    for name, a, b in (("Y", y, ref.y), ("Cb", cb, ref.cb), ("Cr", cr, ref.cr)):
        assert np.array_equal(a, b), f"smoke: plane {name} differs from the oracle ({int((a != b).sum())} samples)"
    print(f"smoke ok: 416x240 recorded picture, {wl.stats['n_mc_units']} MC units, "
          f"{wl.stats['n_tb_cmds']} TB commands, stages {'+'.join(rp.STAGES)}, bit-exact vs oracle")
    rp.free()
    ctx.close()
