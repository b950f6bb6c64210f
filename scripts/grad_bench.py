"""time g3_gram_grad (one pass over the lower triangle of K^-1) for the stationary kinds: fast path vs the interpreter
(G3_GRAD_GENERIC=1 in a second process).  usage: python scripts/grad_bench.py [N] [d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import g3py_amd as g3
from g3py_amd.device import compile_spec
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = g3.Device(0)
rng = np.random.default_rng(0)
X = dev.upload(rng.uniform(0, N ** (1 / d), (N, d)))
G = dev.alloc(N, N, np.float64, zero=True)
al = dev.upload(rng.standard_normal(N))
for kind in ['SE', 'OU', 'MAT32', 'MAT52', 'RQ', 'MAT52+COS', 'MAT52*SM', 'SE+SIN']:
    if kind == 'MAT52+COS':
        spec = ('sum', ('sum', ('MAT52', 1.0, np.ones(d), None), ('COS', 0.5, np.full(d, 0.125), None)), ('NOISE', 0.1))
    elif kind == 'MAT52*SM':
        spec = ('sum', ('prod', ('MAT52', 1.0, np.ones(d), None), ('SM', 0.5, np.full(d, 0.125), np.full(d, 0.1), None)), ('NOISE', 0.1))
    elif kind == 'SE+SIN':
        spec = ('sum', ('sum', ('SE', 1.0, np.ones(d), None), ('SIN', 0.5, np.full(d, 0.125), np.full(d, 0.1), None)), ('NOISE', 0.1))
    else:
        spec = ('sum', (kind, 1.0, np.ones(d), 2.0, None) if kind == 'RQ' else (kind, 1.0, np.ones(d), None), ('NOISE', 0.1))
    prog = compile_spec(spec, d)
    gmap = dev.grad_layout(prog)
    dev.gram_grad(prog, gmap, X, N, d, G, al)
    t0 = time.perf_counter()
    for _ in range(5):
        dev.gram_grad(prog, gmap, X, N, d, G, al)
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print('%s N=%d d=%d generic=%s: %.3f ms  (%.2f TB/s of the %.2f GB lower triangle)' % (kind, N, d, os.environ.get('G3_GRAD_GENERIC', '0'), ms, N * (N + 1) / 2 * 8 / ms / 1e9, N * (N + 1) / 2 * 8 / 1e9))
