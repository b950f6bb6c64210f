"""dlogp_chain end to end at small N: rows/s through the public API, and where the time goes (cProfile of one call).
usage: python scripts/r4_dchain_host.py [B]"""
import sys, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, '.')
import g3py_amd as g3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
for N in (64, 128, 256):
    d = 3
    X = rng.uniform(0, N ** (1 / d), (N, d))
    y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    gp = g3.GaussianProcess(space=X, location=g3.Bias(), kernel=g3.MAT52(X))
    gp.observed(X, y)
    a0 = gp.active.dict_to_array(gp.params_default)
    chain = a0 + 0.15 * rng.standard_normal((B, len(a0)))
    gp.dlogp_chain(chain[:64])
    gp.dlogp_chain(chain)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); g = gp.dlogp_chain(chain); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    want = np.array([gp.dlogp(r, array=True) for r in chain[:16]])
    err = np.max(np.abs(g[:16] - want) / (np.abs(want) + 1e-12))
    print('dlogp_chain N=%d B=%d: %.2f ms = %.1f k rows/s   max rel diff vs one-at-a-time (16 rows) %.1e' % (N, B, t * 1e3, B / t / 1e3, err), flush=True)
