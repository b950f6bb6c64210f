"""Latency of g3_potrf on one diagonal-block-sized matrix (development aid): the multi-GPU driver factors
nb x nb blocks with it.  Every call is timed on its own (host clock around call + info read-back); the median is
printed.  usage: [G3_NB=...] python scripts/potrf_small.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
dev = g3.Device(0)
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 1536, 2048]:
    B = rng.standard_normal((n, n // 2)); K = B @ B.T / n + np.eye(n)
    Ks = [torch.from_numpy(K).cuda() for _ in range(24)]
    ts = []
    for t in Ks:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert dev.potrf(dev.wrap(t.data_ptr(), n, n, n, np.float64), n) == 0
        ts.append((time.perf_counter() - t0) * 1e6)
    L = torch.tril(Ks[-1]).cpu().numpy()
    print('n %5d  G3_NB=%s : median %.1f us per factorisation (first call %.0f), residual %.1e'
          % (n, os.environ.get('G3_NB', 'auto'), float(np.median(ts[4:])), ts[0], np.abs(L @ L.T - K).max()))
