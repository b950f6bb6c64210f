"""Where the host time of GaussianProcess.logp_chain goes (4096 rows, N = 128, SE d = 4 and MAT52 d = 3 + Bias): cProfile of one
call after warm-up, and the rows/s of logp_chain / dlogp_chain.  usage: python scripts/r5_chain_host.py [B]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import g3py_amd as g3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name, N, d, mk in (('SE d=4, Zero mean', 128, 4, lambda X: (g3.Zero(), g3.SE(X))), ('MAT52 d=3 + Bias', 128, 3, lambda X: (g3.Bias(), g3.MAT52(X))),
                       ('MAT52 d=3 + Bias', 64, 3, lambda X: (g3.Bias(), g3.MAT52(X))), ('MAT52 d=3 + Bias', 256, 3, lambda X: (g3.Bias(), g3.MAT52(X)))):
    rng = np.random.default_rng(1)
    X = rng.uniform(0, N ** (1 / d), (N, d)); y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    loc, ker = mk(X)
    gp = g3.GaussianProcess(space=X, location=loc, kernel=ker)
    gp.observed(X, y)
    p0 = gp.active.dict_to_array(gp.params)
    chain = p0[None, :] + 0.05 * rng.standard_normal((B, len(p0)))
    gp.logp_chain(chain[:64]); gp.logp_chain(chain)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); lp = gp.logp_chain(chain); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    gp.dlogp_chain(chain[:64]); gp.dlogp_chain(chain)
    tg = []
    for _ in range(3):
        t0 = time.perf_counter(); gp.dlogp_chain(chain); tg.append(time.perf_counter() - t0)
    print('%s, N=%d, %d rows: logp_chain %.2f ms = %.0f k rows/s; dlogp_chain %.2f ms = %.0f k rows/s'
          % (name, N, B, t * 1e3, B / t / 1e3, min(tg) * 1e3, B / min(tg) / 1e3), flush=True)
    if N == 128:
        pr = cProfile.Profile(); pr.enable(); gp.logp_chain(chain); pr.disable()
        st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(18)
