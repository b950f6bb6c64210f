"""Gram-kernel throughput on MI355X (development aid): algorithmic GB/s = (N d s + bytes written) / time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec
dev = g3.Device(0); st = torch.cuda.Stream(); torch.cuda.set_stream(st); dev.set_stream(st.cuda_stream)
cases = [('SE+noise d=4 N=32768 lower', ('sum', ('SE', 1.0, np.ones(4), None), ('NOISE', 0.1)), 32768, 4, True),
         ('SE+noise d=4 N=32768 full', ('sum', ('SE', 1.0, np.ones(4), None), ('NOISE', 0.1)), 32768, 4, False),
         ('MAT52+COS+noise d=8 N=16384 lower (config 3)', ('sum', ('sum', ('MAT52', 1.0, np.ones(8), None), ('COS', 0.5, np.full(8, 0.125), None)), ('NOISE', 0.1)), 16384, 8, True),
         ('MAT52+SIN+noise d=8 N=16384 lower', ('sum', ('sum', ('MAT52', 1.0, np.ones(8), None), ('SIN', 0.5, np.full(8, 0.125), np.full(8, 0.25), None)), ('NOISE', 0.1)), 16384, 8, True),
         ('MAT52+noise d=8 N=16384 lower (fast path)', ('sum', ('MAT52', 1.0, np.ones(8), None), ('NOISE', 0.1)), 16384, 8, True),
         ('RQ+noise d=4 N=16384 lower (fast path)', ('sum', ('RQ', 1.0, np.ones(4), 1.5, None), ('NOISE', 0.1)), 16384, 4, True),
         ('OU+noise d=4 N=16384 lower (fast path)', ('sum', ('OU', 1.0, np.ones(4), None), ('NOISE', 0.1)), 16384, 4, True),
         ('SE+noise d=16 N=16384 lower', ('sum', ('SE', 1.0, np.ones(16), None), ('NOISE', 0.1)), 16384, 16, True),
         ('SE+noise d=4 N=8192 lower (config 2)', ('sum', ('SE', 1.0, np.ones(4), None), ('NOISE', 0.1)), 8192, 4, True),
         ('SE+noise d=16 N=65536 lower fp32 (config 5 shape)', ('sum', ('SE', 1.0, np.ones(16), None), ('NOISE', 0.1)), 65536, 16, True, np.float32)]
for case in cases:
    name, spec, N, d, lower = case[:5]
    npdt = case[5] if len(case) > 5 else np.float64
    tdt = torch.float32 if npdt == np.float32 else torch.float64
    es = 4 if npdt == np.float32 else 8
    X = torch.rand((N, d), dtype=tdt, device='cuda') * N ** (1 / d)
    K = torch.empty((N, N), dtype=tdt, device='cuda')
    Xd, Kd = dev.wrap(X.data_ptr(), N, d, d, npdt), dev.wrap(K.data_ptr(), N, N, N, npdt)
    prog = compile_spec(spec, d)
    flags = _lib.G3_GRAM_SCRUB | (_lib.G3_GRAM_LOWER if lower else 0)
    for _ in range(2):
        dev.gram(prog, Xd, None, d, Kd, N, N, flags)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dev.gram(prog, Xd, None, d, Kd, N, N, flags)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    by = N * d * es + (N * (N + 1) / 2 if lower else N * N) * es
    del K
    print('%-50s %7.3f ms  %7.1f GB/s algorithmic' % (name, ms, by / ms / 1e6))
