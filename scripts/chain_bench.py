"""Throughput of many independent logp evaluations on the same (X, y) with different
hyper-parameters (the logp_chain / find_MAP caller pattern, stochastic.py:515-564):
one Device context per worker thread, ctypes releases the GIL inside the C calls, so the
workers' streams overlap on the GPU."""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, '.')
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec


def main():
    B = 96
    for N in (256, 512, 1024, 2048, 4096):
        d = 4
        rng = np.random.default_rng(N)
        X = rng.uniform(0, N ** (1 / d), (N, d))
        y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
        hyp = [(1.0 + 0.01 * i, 1.0 + 0.005 * i, 0.1) for i in range(B)]
        Np = _lib.roundup(N)
        res = {}
        for S in (1, 2, 4, 8, 16):
            devs = [g3.Device(0) for _ in range(S)]
            bufs = []
            for dv in devs:
                bufs.append((dv.upload(X), dv.upload(y), dv.alloc(Np + 128, Np, np.float64), dv.alloc(1, Np, np.float64),
                             dv.alloc_inverses(Np, np.float64)))

            def work(w):
                dv = devs[w]
                Xd, yd, K, a, W = bufs[w]
                out = []
                for i in range(w, B, S):
                    var, rate, noise = hyp[i]
                    prog = compile_spec(('sum', ('SE', var, np.full(d, rate), None), ('NOISE', noise)), d)
                    st = dv.gp_factor(prog, Xd, N, d, yd, K, W, a)
                    out.append((i, -0.5 * N * np.log(2 * np.pi) - 0.5 * st['quad'] - st['logdet']))
                return out
            with ThreadPoolExecutor(S) as ex:
                list(ex.map(work, range(S)))          # warm-up
                t0 = time.perf_counter()
                r = list(ex.map(work, range(S)))
                t = time.perf_counter() - t0
            lp = dict(sum(r, []))
            res[S] = (t, lp)
            del bufs, devs
        # one batched sweep (g3_gp_factor_batched)
        dv = g3.Device(0)
        progs = [compile_spec(('sum', ('SE', v, np.full(d, r), None), ('NOISE', nz)), d) for v, r, nz in hyp]
        kstride = (Np + 128) * Np
        K = dv.alloc(B * (Np + 128), Np, np.float64)
        W = dv.alloc(B * Np, 128, np.float64)
        a = dv.alloc(B, Np, np.float64)
        Xd, dd = dv.upload(X), dv.upload(np.tile(y, (B, 1)))
        dv.gp_factor_batched(progs, Xd, N, d, dd, K, kstride, W, a)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            st = dv.gp_factor_batched(progs, Xd, N, d, dd, K, kstride, W, a)
        tb = (time.perf_counter() - t0) / reps
        lpb = {i: -0.5 * N * np.log(2 * np.pi) - 0.5 * s_['quad'] - s_['logdet'] for i, s_ in enumerate(st)}
        res['batched'] = (tb, lpb)
        base = res[1][1]
        for S, (t, lp) in res.items():
            assert all(abs(lp[i] - base[i]) <= 1e-9 * abs(base[i]) for i in range(B))
        print('N=%5d  ' % N + '  '.join('S=%s: %.3f ms/eval (%.0f eval/s, %.1f TF)' % (S, t / B * 1e3, B / t, B * N ** 3 / 3 / t / 1e12) for S, (t, _) in res.items()), flush=True)


if __name__ == '__main__':
    main()
