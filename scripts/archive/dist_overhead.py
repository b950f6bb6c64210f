"""Host-side cost per step of the multi-GPU driver (development aid): one rank, tiny blocks, so GPU
work is negligible and the wall time per step is Python + launch overhead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import g3py_amd as g3
from g3py_amd.distributed import DistributedGP

torch.cuda.set_device(0)
dev = g3.Device(0)
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
dev.set_stream(st.cuda_stream)
tdev = torch.device('cuda', 0)
for N, nb in ((2048, 128), (4096, 128), (8192, 512), (16384, 512)):
    d, M = 4, 256
    rng = np.random.default_rng(0)
    X = rng.uniform(0, N ** 0.25, (N, d)); Xs = rng.uniform(0, N ** 0.25, (M, d)); y = np.sin(X.sum(1))
    gp = DistributedGP(dev, dist, 0, 1, N, d, M, nb=nb, torch_device=tdev)
    o = gp.ops
    Xt, Xst, yt = o.from_host(X), o.from_host(Xs), o.from_host(y)
    spec_f = ('SE', 1.0, np.ones(d), None); spec_n = ('sum', spec_f, ('NOISE', 0.1))
    for _ in range(2):
        gp.step(spec_n, spec_f, Xt, Xst, yt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        gp.step(spec_n, spec_f, Xt, Xst, yt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('N=%6d nb=%4d: %7.2f ms per evaluation, %d steps -> %6.1f us per step' % (N, nb, dt * 1e3, gp.nblk, dt / gp.nblk * 1e6))
