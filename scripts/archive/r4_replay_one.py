"""One replayed rank of a P-rank evaluation, for a kernel trace (development aid).
usage: python scripts/r4_replay_one.py P rank [nb]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import g3py_amd as g3
from g3py_amd.distributed import NativeDistributedGP
import bench
P, r = int(sys.argv[1]), int(sys.argv[2])
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
N, d, M = int(os.environ.get('R4_N', 32768)), 4, 1024
X, y, Xs = bench.synth(N, d, M, 1004)
dev = g3.Device(0)
spec_f = ('SE', 1.0, np.ones(d), None)
spec_n = ('sum', spec_f, ('NOISE', 0.1))
Xd, Xsd, dd = dev.upload(X), dev.upload(Xs), dev.upload(y)
ref = NativeDistributedGP(dev, None, 0, 1, N, d, M, nb=nb, transport='callbacks', keep=True)
ref.step(spec_n, spec_f, Xd, Xsd, dd)
rp = NativeDistributedGP(dev, None, r, P, N, d, M, nb=nb, transport='replay', reference=ref)
for i in range(3):
    dev.sync(); t0 = time.perf_counter(); rp.step(spec_n, spec_f, Xd, Xsd, dd); dev.sync()
    print('MARK step %d %.2f ms' % (i, (time.perf_counter() - t0) * 1e3), flush=True)
