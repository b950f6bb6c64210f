"""Quick on-GPU sanity run of the low-level kernels against NumPy (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import g3py_amd as g3
from g3py_amd import _lib

dev = g3.Device.default()
rng = np.random.default_rng(0)

def gemm_case(m, n, k, dt, lower):
    A = rng.standard_normal((m, k)).astype(dt); B = rng.standard_normal((n, k)).astype(dt)
    Cm = rng.standard_normal((m, n)).astype(dt)
    Ad, Bd, Cd = dev.upload(A), dev.upload(B), dev.upload(Cm)
    dev.gemm_nt(Cd, Ad, Bd, m, n, k, alpha=-1.0, beta=1.0, lower_only=lower)
    got = dev.download(Cd)
    ref = Cm - A.astype(np.float64) @ B.astype(np.float64).T
    if lower:
        mask = np.tril(np.ones((m, n), bool))
        ref = np.where(mask, ref, Cm)
    err = np.abs(got - ref).max() / np.abs(ref).max()
    print('gemm', m, n, k, np.dtype(dt).name, 'lower' if lower else 'full', 'relerr %.2e' % err)
    return err

for dt, tol in ((np.float64, 1e-13), (np.float32, 1e-4)):
    for (m, n, k, lo) in [(64, 64, 64, False), (128, 64, 128, False), (192, 192, 64, True), (1536, 1536, 256, True),
                          (2048, 1536, 128, False), (8192, 8192, 64, True)]:
        assert gemm_case(m, n, k, dt, lo) < tol

for n in (128, 256, 384, 512, 768, 1024, 4096, 6144, 8192):
    B = rng.standard_normal((n, n)); K = B @ B.T / n + np.eye(n)
    Kd = dev.upload(K)
    t0 = time.time(); info = dev.potrf(Kd, n); t1 = time.time()
    L = np.tril(dev.download(Kd))
    Lr = np.linalg.cholesky(K)
    print('potrf', n, 'info', info, 'err %.2e' % (np.abs(L - Lr).max()), 'time %.1f ms' % ((t1 - t0) * 1e3))
    assert info == 0 and np.abs(L - Lr).max() < 1e-11
    Bm = rng.standard_normal((128, n))
    Ld = dev.upload(Lr); Bd = dev.upload(Bm)
    dev.trsm_rlt(Ld, n, Bd, 128)
    Xs = dev.download(Bd)
    print('trsm', n, 'err %.2e' % np.abs(Xs @ Lr.T - Bm).max())
    assert np.abs(Xs @ Lr.T - Bm).max() < 1e-10

K = -np.eye(256); Kd = dev.upload(K); print('potrf nonPD info', dev.potrf(Kd, 256))
import __graft_entry__ as ge
ge.smoke()
print('ALL OK')
