"""One generic-path Gram launch set (config 3's kernel) for counter collection (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec
dev = g3.Device(0)
N, d = 16384, 8
spec = ('sum', ('sum', ('MAT52', 1.0, np.ones(8), None), ('COS', 0.5, np.full(8, 0.125), None)), ('NOISE', 0.1))
X = torch.rand((N, d), dtype=torch.float64, device='cuda') * N ** (1 / d)
K = torch.empty((N, N), dtype=torch.float64, device='cuda')
Xd, Kd = dev.wrap(X.data_ptr(), N, d, d, np.float64), dev.wrap(K.data_ptr(), N, N, N, np.float64)
prog = compile_spec(spec, d)
for _ in range(3):
    dev.gram(prog, Xd, None, d, Kd, N, N, _lib.G3_GRAM_SCRUB | _lib.G3_GRAM_LOWER)
dev.sync()
g3.Device.close_all()
