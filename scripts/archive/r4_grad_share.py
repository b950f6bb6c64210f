"""share of the kernel-parameter sums (gram_grad: table fast path, generated kernel or interpreter) in one dlogp evaluation.
usage: python scripts/r4_grad_share.py [N ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import g3py_amd as g3
Ns = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]
rng = np.random.default_rng(0)
for N in Ns:
    d = 4
    X = rng.uniform(0, N ** (1 / d), (N, d))
    y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    for name, kern in (('SE (table)', lambda: g3.SE(X)), ('MAT52+COS (table)', lambda: g3.MAT52(X) + g3.COS(X)),
                       ('(SE+OU)*(MAT32+0.5) (generated)', lambda: (g3.SE(X) + g3.OU(X)) * (g3.MAT32(X) + 0.5)),
                       ('SE*SINC (generated)', lambda: g3.SE(X) * g3.SINC(X))):
        gp = g3.GaussianProcess(space=X[:8], location=g3.Zero(), kernel=kern())
        gp.observed(X, y)
        p = gp.active.dict_to_array(gp.params_default)
        gp.dlogp(p, array=True)
        dev = gp.device
        ts = []
        for r in range(3):
            q = p + 1e-3 * (r + 1)
            dev.sync(); t0 = time.perf_counter(); gp.dlogp(q, array=True); dev.sync(); ts.append(time.perf_counter() - t0)
        dev.prof_enable(True); dev.prof_reset()
        gp.dlogp(p + 5e-3, array=True)
        prof = dev.prof_collect(); dev.prof_enable(False)
        print('N=%d %-36s dlogp %.2f ms; device phases: %s' % (N, name, min(ts) * 1e3, {k: round(v['ms'], 2) for k, v in prof.items() if v['ms'] > 0.005}), flush=True)
