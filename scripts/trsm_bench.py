"""Time the stripe-local triangular panel solve alone (development aid): X (m x n) <- X L^-T.
usage: [G3_TRSM_TALL_MIN=...] python scripts/trsm_bench.py [m n ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
dev = g3.Device(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); dev.set_stream(st.cuda_stream)
args = [int(a) for a in sys.argv[1:]] or [30720, 1024, 16384, 1024, 8192, 1024, 16384, 512, 8192, 256]
rng = np.random.default_rng(0)
for m, n in zip(args[::2], args[1::2]):
    B = rng.standard_normal((n, n)); K = B @ B.T / n + np.eye(n)
    Lt = torch.from_numpy(np.linalg.cholesky(K)).cuda()
    Ld = dev.wrap(Lt.data_ptr(), n, n, n, np.float64)
    Xt = torch.rand((m, n), dtype=torch.float64, device='cuda')
    X0 = Xt.clone()
    Xd = dev.wrap(Xt.data_ptr(), m, n, n, np.float64)
    dev.trsm_rlt(Ld, n, Xd, m)
    torch.cuda.synchronize()
    err = float((Xt[:256] @ Lt.T - X0[:256]).abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dev.trsm_rlt(Ld, n, Xd, m)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('m %6d n %5d: %.3f ms  %.1f TFLOP/s (m n^2 flops, incl. the inverse-block kernel)  residual %.1e' % (m, n, ms, m * n * n / ms / 1e9, err))
