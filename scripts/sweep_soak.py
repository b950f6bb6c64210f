"""Soak of the two-stream sweep (development aid): the same logp + predict evaluation repeated many times must
give bit-identical statistics (an ordering bug between the chain and bulk streams would show as differences).
usage: python scripts/sweep_soak.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec
dev = g3.Device(0)
for N in [int(a) for a in sys.argv[1:]] or [3072, 6144, 8192, 12288]:
    d, M = 4, 300
    rng = np.random.default_rng(N)
    X = rng.uniform(0, N ** (1 / d), (N, d)); Xs = rng.uniform(0, N ** (1 / d), (M, d))
    y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    spec_f = ('SE', 1.0, np.ones(d), None)
    spec_n = ('sum', spec_f, ('NOISE', 0.1))
    Np, Mp = _lib.roundup(N), _lib.roundup(M, 128)
    K = dev.alloc(Np + 128 + Mp, Np, np.float64)
    W, a = dev.alloc_inverses(Np, np.float64), dev.alloc(1, Np, np.float64)
    mu, ss = dev.alloc(1, Mp, np.float64), dev.alloc(1, Mp, np.float64)
    Xd, yd, Xsd = dev.upload(X), dev.upload(y), dev.upload(Xs)
    pn, pf = compile_spec(spec_n, d), compile_spec(spec_f, d)
    ref, diffs = None, 0
    reps = 60 if N <= 8192 else 30
    for r in range(reps):
        st = dev.gp_factor_predict(pn, pf, Xd, N, d, yd, Xsd, M, K, W, a, mu, ss)
        cur = (st['quad'], st['logdet'], dev.download(mu, 1, M).tobytes(), dev.download(ss, 1, M).tobytes())
        if ref is None:
            ref = cur
        elif cur != ref:
            diffs += 1
    print('N %6d: %d evaluations, differing from the first: %d  (logdet %.12f)' % (N, reps, diffs, ref[1]))
g3.Device.close_all()
