"""Soak of the native multi-GPU driver's three-stream schedule (development aid): the same evaluation repeated many times
on one rank through RCCL (fully asynchronous: collectives, look-ahead and bulk stream overlap) must give bit-identical
statistics; an ordering bug between the streams or a buffer re-used too early would show as run-to-run differences.
usage: python scripts/dist_soak.py [N nb reps] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import g3py_amd as g3
from g3py_amd.distributed import NativeDistributedGP
a = [int(v) for v in sys.argv[1:]] or [4096, 256, 40, 8192, 512, 40, 8192, 128, 20, 12288, 1024, 20]
dev = g3.Device(0)
for i in range(0, len(a), 3):
    N, nb, reps = a[i:i + 3]
    d, M = 4, 300
    rng = np.random.default_rng(N + nb)
    X = rng.uniform(0, N ** (1 / d), (N, d)); Xs = rng.uniform(0, N ** (1 / d), (M, d))
    y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    spec_f = ('SE', 1.0, np.ones(d), None)
    spec_n = ('sum', spec_f, ('NOISE', 0.1))
    dgp = NativeDistributedGP(dev, None, 0, 1, N, d, M, nb=nb, transport='rccl')
    Xd, Xsd, yd = dev.upload(X), dev.upload(Xs), dev.upload(y)
    ref, diffs = None, 0
    for r in range(reps):
        lp = dgp.step(spec_n, spec_f, Xd, Xsd, yd)
        cur = (lp, dgp.last['logdet'], dgp.last['mean'].tobytes(), dgp.last['ss'].tobytes())
        if ref is None:
            ref = cur
        elif cur != ref:
            diffs += 1
    # the same numbers as the one-GPU fused sweep
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec
    Np, Mp = _lib.roundup(N), _lib.roundup(M, 128)
    K = dev.alloc(Np + 128 + Mp, Np, np.float64)
    W, av = dev.alloc_inverses(Np, np.float64), dev.alloc(1, Np, np.float64)
    mu, ss = dev.alloc(1, Mp, np.float64), dev.alloc(1, Mp, np.float64)
    st = dev.gp_factor_predict(compile_spec(spec_n, d), compile_spec(spec_f, d), Xd, N, d, yd, Xsd, M, K, W, av, mu, ss)
    lp1 = -0.5 * N * np.log(2 * np.pi) - 0.5 * st['quad'] - st['logdet']
    print('N %6d nb %5d: %d evaluations through RCCL, differing from the first: %d; logp vs one-GPU sweep rel %.1e'
          % (N, nb, reps, diffs, abs(ref[0] - lp1) / abs(lp1)))
    dgp.close()
    for b in (K, W, av, mu, ss):
        b.free()
g3.Device.close_all()
