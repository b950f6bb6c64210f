"""Chain-server sweep against LAPACK, and its time against the launch-per-kernel sweep (G3_CHAIN=0), development aid.
usage: python scripts/r4_chain_check.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
dev = g3.Device(0)
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or [384, 512, 1024, 2048, 4096, 8192]:
    B = rng.standard_normal((n, n // 4)); K = B @ B.T / n + np.eye(n)
    ts = []
    for rep in range(6):
        t = torch.from_numpy(K).cuda()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        info = dev.potrf(dev.wrap(t.data_ptr(), n, n, n, np.float64), n)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert info == 0, info
    L = torch.tril(t).cpu().numpy()
    Lr = np.linalg.cholesky(K)
    print('n %5d  G3_CHAIN=%s wgs=%s: %.3f ms (first %.2f)  max|L-Lref| %.2e  resid %.2e' % (
        n, os.environ.get('G3_CHAIN', '1'), os.environ.get('G3_CHAIN_WGS', '16'), float(np.median(ts[2:])), ts[0],
        np.abs(L - Lr).max(), np.abs(L @ L.T - K).max()), flush=True)
