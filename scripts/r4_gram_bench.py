"""Gram: interpreted vs generated-at-first-use vs compile-time table (profiles/r04_gram.md).  usage: python scripts/r4_gram_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
from g3py_amd.device import compile_spec
from oracle.gen_golden import kernel_zoo
N = int(os.environ.get('R4_N', 16384))
rows = []
for d in (4, 8):
    z = kernel_zoo(d)
    r = np.linspace(0.6, 1.4, d)
    cases = {k: z[k] for k in ('SE', 'MAT52+COS', 'SINC', 'SM', '(SE+OU)*(MAT32+0.5)', '2*SE+0.1', 'SE*COS', 'COS', 'WN')}
    cases['MAT52*SM'] = ('prod', ('MAT52', 1.0, r, None), ('SM', 0.9, np.linspace(0.11, 0.23, d), 0.3 * r, None))
    cases['(SE+RQ)*OU+COS'] = ('sum', ('prod', ('sum', ('SE', 1.0, r, None), ('RQ', 1.2, r, 1.7, None)), ('OU', 0.5, r, None)), ('COS', 0.5, np.linspace(0.11, 0.23, d), None))
    X = np.random.default_rng(d).uniform(0, N ** (1 / d), (N, d))
    for name, spec in cases.items():
        res = {}
        for mode, env in (('interpreted', {'G3_GRAM_JIT': '0', 'G3_GRAM_NOFAST': '1'}), ('default', {})):
            for k in ('G3_GRAM_JIT', 'G3_GRAM_NOFAST'):
                os.environ.pop(k, None)
            os.environ.update(env)
            dev = g3.Device(0)
            Xd = dev.upload(X)
            K = dev.alloc(N, N, np.float64)
            prog = compile_spec(('sum', spec, ('NOISE', 0.1)), d)
            for _ in range(2):
                dev.gram(prog, Xd, None, d, K, N, N, 1 | 2)
            dev.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                dev.gram(prog, Xd, None, d, K, N, N, 1 | 2)
            dev.sync()
            res[mode] = ((time.perf_counter() - t0) / 5 * 1e3, dev.gram_path_stats())
            dev.close()
        st = res['default'][1]
        path = 'table' if st['table'] else ('generated' if st['generated'] else 'interpreted')
        gb = (N * d * 8 + 0.5 * N * (N + 1) * 8) / 1e9
        rows.append('| %s + noise | %d | %.3f | %.3f (%s) | %.2fx | %.2f |' % (name, d, res['interpreted'][0], res['default'][0], path,
                    res['interpreted'][0] / res['default'][0], gb / (res['default'][0] * 1e-3) / 1e3))
        print(rows[-1], flush=True)
print('\n| expression | d | interpreted ms | default ms (path) | speed-up | TB/s of algorithmic bytes |\n|---|---|---|---|---|---|')
print('\n'.join(rows))
