"""Soak of the distributed gradient (development aid): gradient-mode step + g3_dist_gp_dlogp repeated on one rank through RCCL
(asynchronous: the gathers of the L^-T panels on the chain stream overlap the staircase products on the bulk stream, three
gather buffers in rotation) must give bit-identical parameter sums, equal the one-GPU g3_gp_dlogp, and leave the device
memory where it was.  usage: python scripts/dist_grad_soak.py [N nb reps] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec
from g3py_amd.distributed import NativeDistributedGP
a = [int(v) for v in sys.argv[1:]] or [4096, 256, 30, 8192, 512, 20, 8192, 128, 10, 12288, 1024, 10]
dev = g3.Device(0)
for i in range(0, len(a), 3):
    N, nb, reps = a[i:i + 3]
    d, M = 4, 300
    rng = np.random.default_rng(N + nb)
    X = rng.uniform(0, N ** (1 / d), (N, d)); Xs = rng.uniform(0, N ** (1 / d), (M, d))
    y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    spec_f = ('sum', ('MAT52', 1.0, np.ones(d), None), ('COS', 0.3, np.full(d, 0.1), None))
    spec_n = ('sum', spec_f, ('NOISE', 0.1))
    dgp = NativeDistributedGP(dev, None, 0, 1, N, d, M, nb=nb, transport='rccl')
    dgp.set_grad(True)
    Xd, Xsd, yd = dev.upload(X), dev.upload(Xs), dev.upload(y)
    ref, diffs, free0 = None, 0, None
    for r in range(reps):
        lp = dgp.step(spec_n, spec_f, Xd, Xsd, yd)
        _, gmap, slots, alpha = dgp.dlogp(spec_n, Xd)
        cur = (lp, slots.tobytes(), alpha.tobytes())
        if ref is None:
            ref = cur
        elif cur != ref:
            diffs += 1
        if r == 1:
            free0 = torch.cuda.mem_get_info()[0]
    leak = free0 - torch.cuda.mem_get_info()[0]
    # the one-GPU gradient of the same problem
    Np = _lib.roundup(N)
    K = dev.alloc(Np + 128, Np, np.float64)
    W, av = dev.alloc_inverses(Np, np.float64), dev.alloc(1, Np, np.float64)
    prog = compile_spec(spec_n, d)
    dev.gp_factor(prog, Xd, N, d, yd, K, W, av)
    Y, Kinv, al = dev.alloc(Np, Np, np.float64), dev.alloc(Np, Np, np.float64), dev.alloc(1, Np, np.float64)
    s1 = dev.gp_dlogp(prog, dev.grad_layout(prog), Xd, N, d, K, W, av, Y, Kinv, al)
    s0 = np.frombuffer(ref[1])
    print('N %6d nb %5d: %d gradient evaluations through RCCL, differing from the first: %d; device memory drift %d bytes; '
          'max rel diff vs one-GPU g3_gp_dlogp %.1e' % (N, nb, reps, diffs, leak, np.max(np.abs(s0 - s1) / np.abs(s1))))
    dgp.close()
    for b in (K, W, av, Y, Kinv, al):
        b.free()
g3.Device.close_all()
