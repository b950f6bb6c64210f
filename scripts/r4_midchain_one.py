"""one batched sweep of B medium problems (the chain pattern at N ~ 1000 - 4000): for a kernel-level profile"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 96
d = 4
rng = np.random.default_rng(N)
X = rng.uniform(0, N ** (1 / d), (N, d))
y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
hyp = [(1.0 + 0.01 * i, 1.0 + 0.005 * i, 0.1) for i in range(B)]
Np = _lib.roundup(N)
dv = g3.Device(0)
progs = [compile_spec(('sum', ('SE', v, np.full(d, r), None), ('NOISE', nz)), d) for v, r, nz in hyp]
kstride = (Np + 128) * Np
K = dv.alloc(B * (Np + 128), Np, np.float64); W = dv.alloc(B * Np, 128, np.float64); a = dv.alloc(B, Np, np.float64)
Xd, dd = dv.upload(X), dv.upload(np.tile(y, (B, 1)))
arr = (_lib.KernelProg * B)(*progs)
for _ in range(2):
    dv.gp_factor_batched(arr, Xd, N, d, dd, K, kstride, W, a, raw=True)
ts = []
for _ in range(5):
    dv.sync(); t0 = time.perf_counter(); dv.gp_factor_batched(arr, Xd, N, d, dd, K, kstride, W, a, raw=True); dv.sync(); ts.append(time.perf_counter() - t0)
t = min(ts)
print('N=%d B=%d: %.3f ms per sweep = %.1f k eval/s, %.1f TFLOP/s' % (N, B, t * 1e3, B / t / 1e3, B * N ** 3 / 3 / t / 1e12))
