"""Soak of the factorisation kernels (development aid): many random SPD matrices of random sizes, each factored
several times; every result must equal scipy's to rounding and repeat bit for bit (a race in the wave-level
hand-over of the diagonal kernels would show as run-to-run differences).
usage: python scripts/potrf_soak.py [cases] [repeats] [f32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg
import g3py_amd as g3
dev = g3.Device(0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
DT = np.float32 if len(sys.argv) > 3 and sys.argv[3] == 'f32' else np.float64
TOL = 3e-6 if DT == np.float32 else 1e-13
rng = np.random.default_rng(12345)
worst, nondet = 0.0, 0
for c in range(cases):
    n = 128 * int(rng.integers(1, 25))
    cond = 10.0 ** rng.uniform(0, 3 if DT == np.float32 else 8)
    B = rng.standard_normal((n, n))
    Q, _ = np.linalg.qr(B)
    K = (Q * np.geomspace(1.0, 1.0 / cond, n)) @ Q.T
    K = (0.5 * (K + K.T)).astype(DT)
    first = None
    for r in range(reps):
        Kd = dev.upload(K)
        info = dev.potrf(Kd, n)
        assert info == 0, (n, cond, info)
        L = np.tril(dev.download(Kd)).astype(np.float64)
        if first is None:
            first = L
            err = np.abs(L @ L.T - K.astype(np.float64)).max() / np.abs(K).max()
            worst = max(worst, err)
            assert err < TOL * n, (n, cond, err)
        elif not np.array_equal(L, first):
            nondet += 1
            print('run-to-run difference: n', n, 'cond %.1e' % cond, 'max |dL|', np.abs(L - first).max())
print('cases %d x %d repeats: worst residual |LL^T - K| / |K| = %.2e, run-to-run differences: %d' % (cases, reps, worst, nondet))
g3.Device.close_all()
