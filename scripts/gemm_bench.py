"""Time g3_gemm_nt on MI355X for given shapes (development aid).
usage: python scripts/gemm_bench.py  [m n k lower] ...   (G3_GEMM_CFG=1..4 forces a tile config)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import g3py_amd as g3

shapes = [(16384, 16384, 2048, 1), (16384, 16384, 16384, 1), (8192, 8192, 2048, 1), (16384, 2048, 2048, 0),
          (4096, 4096, 4096, 1), (16384, 128, 256, 0), (1024, 8192, 8192, 0)]
if len(sys.argv) > 4:
    a = list(map(int, sys.argv[1:]))
    shapes = [tuple(a[i:i + 4]) for i in range(0, len(a), 4)]
DT = np.float32 if os.environ.get('G3_BENCH_F32') else np.float64
TDT = torch.float32 if DT == np.float32 else torch.float64
dev = g3.Device(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
dev.set_stream(st.cuda_stream)
for (m, n, k, lower) in shapes:
    lda = max(k, int(os.environ.get('G3_LDA', '0')))      # G3_LDA: the operand is a column block of a wider matrix
    A = torch.rand((max(m, n), lda), dtype=TDT, device='cuda') - 0.5
    ldc = max(n, int(os.environ.get('G3_LDC', '0')))      # G3_LDC: C is a block of a wider matrix
    C = torch.rand((m, ldc), dtype=TDT, device='cuda')
    Ad = dev.wrap(A.data_ptr(), max(m, n), k, lda, DT)
    Cd = dev.wrap(C.data_ptr(), m, n, ldc, DT)
    for _ in range(2):
        dev.gemm_nt(Cd, Ad, Ad, m, n, k, alpha=-1e-6, beta=float(os.environ.get('G3_BETA', '1.0')), lower_only=bool(lower))
    torch.cuda.synchronize()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dev.gemm_nt(Cd, Ad, Ad, m, n, k, alpha=-1e-6, beta=float(os.environ.get('G3_BETA', '1.0')), lower_only=bool(lower))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * k * ((m * n - n * (n - 1) / 2) if lower else m * n)
    print(np.dtype(DT).name, 'm %6d n %6d k %6d lower %d lda %d ldc %d cfg %s: %8.3f ms  %6.2f TFLOP/s' % (m, n, k, lower, lda, ldc, os.environ.get('G3_GEMM_CFG', 'auto'), ms, fl / ms / 1e9))
