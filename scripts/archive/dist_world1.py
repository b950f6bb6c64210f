"""Development: time the multi-GPU driver with one rank (no communication) against the in-library sweep."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import g3py_amd as g3
from g3py_amd.distributed import DistributedGP
from bench import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
d, M = 4, 1024
X, y, Xs = synth(N, d, M, 1004)
dev = g3.Device(0)
st = torch.cuda.Stream(priority=-1); torch.cuda.set_stream(st); dev.set_stream(st.cuda_stream)
tdev = torch.device("cuda", 0)
for nb in (512, 1024, 2048):
    dgp = DistributedGP(dev, None, 0, 1, N, d, M, nb=nb, torch_device=tdev)
    o = dgp.ops
    Xt, Xst, yt = o.from_host(X), o.from_host(Xs), o.from_host(y)
    spec_f = ("SE", 1.0, np.ones(d), None); spec_n = ("sum", spec_f, ("NOISE", 0.1))
    dgp.step(spec_n, spec_f, Xt, Xst, yt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lp = dgp.step(spec_n, spec_f, Xt, Xst, yt)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("N", N, "nb", nb, "world=1 row-cyclic driver: %.1f ms" % (t * 1e3), "logp", lp, flush=True)
    del dgp
