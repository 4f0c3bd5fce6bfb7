import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
from pathlib import Path
import golden_io; golden_io.GOLDEN = Path(sys.argv[1])
import pipe_cases, oracle_pipeline
chain = len(sys.argv) > 2 and sys.argv[2] == 'chain'
P = pipe_cases.Pipe()
decoded = {}
for k in range(P.n):
    wl = P.workload(k, decoded)
    dst, mvs = oracle_pipeline.decode(wl, want_mvs=True)
    exp = P.frames[k]
    out = []
    for nm, a, b in (('Y', dst.y, exp[0]), ('Cb', dst.cb, exp[1]), ('Cr', dst.cr, exp[2])):
        d = (a != b)
        out.append('%s %d%s' % (nm, int(d.sum()), (' first %s' % (tuple(int(v) for v in np.argwhere(d)[0]),)) if d.any() else ''))
    calls = P.dmvr_calls(k)
    mvok = ''
    if len(calls):
        ux = wl.mcx_units[(wl.mcx_units['flags'] & 64) != 0]
        got = mvs[(wl.mcx_units['flags'] & 64) != 0]
        mvok = ' | dmvr units %d calls %d mv equal %s' % (len(ux), len(calls), np.array_equal(got, calls[:, 8:12]) if len(ux) == len(calls) else 'n/a')
    print(k, ' | '.join(out) + mvok)
    decoded[k] = (dst.y, dst.cb, dst.cr) if chain else exp
