"""Time the 128 x 128 diagonal kernel alone (development aid): potrf of 128x128 blocks, fp64 and fp32,
per-call host clock incl. the info read-back (median)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
dev = g3.Device(0)
rng = np.random.default_rng(0)
for dt, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
    for n in (128, 256):
        B = rng.standard_normal((n, 64)); K = (B @ B.T / 64 + np.eye(n)).astype(dt)
        Ks = [torch.from_numpy(K).cuda() for _ in range(60)]
        ts = []
        for t in Ks:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            dev.potrf(dev.wrap(t.data_ptr(), n, n, n, dt), n)
            ts.append((time.perf_counter() - t0) * 1e6)
        print('%s potrf(%d) incl. host round trip: median %.1f us per call' % (np.dtype(dt).name, n, float(np.median(ts[5:]))))
