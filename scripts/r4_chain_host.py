"""logp_chain end to end (host packing + PCIe + device) at small N: rows/s through the public API.
usage: python scripts/r4_chain_host.py [B]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import g3py_amd as g3

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(0)
for N in (64, 128, 256):
    for dtype in (np.float64, np.float32):
        d = 3
        X = rng.uniform(0, N ** (1 / d), (N, d))
        y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
        gp = g3.GaussianProcess(space=X, location=g3.Bias(), kernel=g3.MAT52(X), dtype=dtype)
        gp.observed(X, y)
        a0 = gp.active.dict_to_array(gp.params_default)
        chain = a0 + 0.15 * rng.standard_normal((B, len(a0)))
        gp.logp_chain(chain[:64])
        gp.logp_chain(chain)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); lp = gp.logp_chain(chain); ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        want = np.array([gp.logp(r, array=True) for r in chain[:32]])
        err = np.max(np.abs(np.asarray(lp[:32], dtype=np.float64) - want) / np.abs(want))
        print('logp_chain N=%d %s B=%d: %.2f ms = %.0f k rows/s   max rel diff vs one-at-a-time (32 rows) %.1e'
              % (N, np.dtype(dtype).name, B, t * 1e3, B / t / 1e3, err), flush=True)
