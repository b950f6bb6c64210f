"""Latency of g3_potrf on one diagonal-block-sized matrix (development aid): the multi-GPU driver factors
nb x nb blocks with it.  usage: [G3_NB=...] python scripts/potrf_small.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
dev = g3.Device(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); dev.set_stream(st.cuda_stream)
rng = np.random.default_rng(0)
for n in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]:
    B = rng.standard_normal((n, n // 2)); K = B @ B.T / n + np.eye(n)
    Ks = [torch.from_numpy(K).cuda() for _ in range(40)]
    W = torch.empty((n, 128), dtype=torch.float64, device='cuda')
    for t in Ks[:5]:
        dev.potrf(dev.wrap(t.data_ptr(), n, n, n, np.float64), n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in Ks[5:]:
        dev.lib.g3_potrf_nowait(dev.ctx, t.data_ptr(), n, n, 1, W.data_ptr(), dev._info_scratch.data_ptr()) if hasattr(dev, '_info_scratch') else dev.potrf(dev.wrap(t.data_ptr(), n, n, n, np.float64), n)
    e1.record(); torch.cuda.synchronize()
    L = torch.tril(Ks[-1]).cpu().numpy()
    print('n %5d  G3_NB=%s : %.1f us per factorisation (incl. the info read-back), residual %.1e'
          % (n, os.environ.get('G3_NB', 'auto'), e0.elapsed_time(e1) / 35 * 1e3, np.abs(L @ L.T - K).max()))
