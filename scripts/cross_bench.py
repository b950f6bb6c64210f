"""Development: time g3_gp_cross (posterior mean / variance of M new points against a cached factor)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec
from bench import synth
dev = g3.Device(0)
N, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 4
X, y, _ = synth(N, d, 8, 1004)
Np = _lib.roundup(N)
Xd, yd = dev.upload(X), dev.upload(y)
K, a, W = dev.alloc(Np + 128, Np, np.float64), dev.alloc(1, Np, np.float64), dev.alloc_inverses(Np, np.float64)
spec_f = ('SE', 1.0, np.ones(d), None)
dev.gp_factor(compile_spec(('sum', spec_f, ('NOISE', 0.1)), d), Xd, N, d, yd, K, W, a)
for M in (128, 1024, 4096):
    Xs = np.random.default_rng(M).uniform(0, N ** (1 / d), (M, d))
    Mp = _lib.roundup(M, 128)
    V, mu, ss = dev.alloc(Mp, Np, np.float64), dev.alloc(1, Mp, np.float64), dev.alloc(1, Mp, np.float64)
    Sd = dev.upload(Xs)
    prog = compile_spec(spec_f, d)
    dev.gp_cross(prog, Sd, M, Xd, N, d, K, W, a, V, mu, ss); dev.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        dev.gp_cross(prog, Sd, M, Xd, N, d, K, W, a, V, mu, ss)
    dev.sync()
    t = (time.perf_counter() - t0) / 3
    print('g3_gp_cross N=%d M=%d: %.2f ms  (N^2 M = %.1f TFLOP/s)' % (N, M, t * 1e3, N * N * M / t / 1e12))
