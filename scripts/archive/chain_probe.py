"""Uncontended latency of the critical-path pieces of the blocked Cholesky (development aid):
recursive potrf of one NB x NB block (diag128m leaves + in-place leaf solves + small SYRKs, all
serial on one stream) and the tiny GEMMs between them, on an otherwise idle GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import g3py_amd as g3

os.environ.setdefault('G3_NB', '65536')      # no look-ahead: potrf_rec only
dev = g3.Device(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
dev.set_stream(st.cuda_stream)
rng = np.random.default_rng(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (128, 256, 512, 1024, 2048):
    B = rng.standard_normal((n, 2 * n))
    K = torch.from_numpy(B @ B.T / n + np.eye(n)).cuda()
    work = torch.empty_like(K)
    W = torch.empty((n, 128), dtype=torch.float64, device='cuda')
    Kd = dev.wrap(work.data_ptr(), n, n, n, np.float64)
    import ctypes as C
    info = C.c_int(0)

    def f():
        work.copy_(K)
        dev.lib.g3_potrf_nowait(dev.ctx, work.data_ptr(), n, n, 0, W.data_ptr(), dev_info.data_ptr())
    dev_info = torch.zeros(1, dtype=torch.int32, device='cuda')
    t_copy = timeit(lambda: work.copy_(K))
    t = timeit(f)
    print('potrf n=%5d: %8.1f us (copy %5.1f us) -> %6.1f us per 128 columns' % (n, t - t_copy, t_copy, (t - t_copy) / (n / 128)))

for (m, n, k, lower) in [(128, 128, 128, 1), (128, 128, 128, 0), (256, 256, 256, 1), (256, 128, 128, 0), (1024, 128, 128, 0),
                         (8192, 128, 128, 0), (8192, 256, 256, 0), (512, 512, 512, 1)]:
    A = torch.rand((max(m, n), k), dtype=torch.float64, device='cuda') - 0.5
    Cm = torch.rand((m, n), dtype=torch.float64, device='cuda')
    Ad = dev.wrap(A.data_ptr(), max(m, n), k, k, np.float64)
    Cd = dev.wrap(Cm.data_ptr(), m, n, n, np.float64)
    t = timeit(lambda: dev.gemm_nt(Cd, Ad, Ad, m, n, k, alpha=-1e-6, beta=1.0, lower_only=bool(lower)))
    print('gemm m=%5d n=%4d k=%4d lower=%d: %7.1f us (dependent launches on one stream)' % (m, n, k, lower, t))
