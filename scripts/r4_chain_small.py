"""Small-N chains: evaluations per second of g3_gp_factor_batched with 4096 hyper-parameter sets at
N = 64 / 128 / 256 -- one workgroup per member since round 4 (g3_potrf.hip::small_factor_kernel) -- at the C ABI (programs
packed once) and through GaussianProcess.logp_chain; every member checked against the one-at-a-time path.
usage: python scripts/r4_chain_small.py [B]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = g3.Device(0)
d = 4
for N in (64, 128, 256, 512):
    rng = np.random.default_rng(N)
    X = rng.uniform(0, N ** (1 / d), (N, d))
    y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    hyp = [(1.0 + 0.3 * (i % 97) / 97, 0.7 + 0.6 * (i % 89) / 89, 0.05 + 0.1 * (i % 13) / 13) for i in range(B)]
    progs = [compile_spec(('sum', ('SE', v, np.full(d, r), None), ('NOISE', nz)), d) for v, r, nz in hyp]
    arr = (_lib.KernelProg * B)(*progs)
    Np = _lib.roundup(N)
    kstride = (Np + 128) * Np
    K = dev.alloc(B * (Np + 128), Np, np.float64); W = dev.alloc(B * Np, 128, np.float64); a = dev.alloc(B, Np, np.float64)
    Xd, dd = dev.upload(X), dev.upload(np.tile(y, (B, 1)))
    st = dev.gp_factor_batched(arr, Xd, N, d, dd, K, kstride, W, a, raw=True)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); st = dev.gp_factor_batched(arr, Xd, N, d, dd, K, kstride, W, a, raw=True); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    # device side only: HIP events through the library's profiler
    dev.prof_enable(1); dev.prof_reset()
    dev.gp_factor_batched(arr, Xd, N, d, dd, K, kstride, W, a, raw=True)
    pr = dev.prof_collect(); dev.prof_enable(False)
    dev_ms = pr['gram']['ms'] + pr['potrf']['ms']
    # every 37th member against the one-at-a-time path
    K1, a1, W1 = dev.alloc(Np + 128, Np, np.float64), dev.alloc(1, Np, np.float64), dev.alloc_inverses(Np, np.float64)
    yd = dev.upload(y)
    worst = 0.0
    for i in range(0, B, 37):
        s1 = dev.gp_factor(progs[i], Xd, N, d, yd, K1, W1, a1)
        lp1 = -0.5 * s1['quad'] - s1['logdet']; lpb = -0.5 * st[i, 1] - st[i, 0]
        worst = max(worst, abs(lp1 - lpb) / abs(lp1))
    print('N=%4d B=%d: %.3f ms per batch at the C ABI = %.0f k eval/s (device: Gram %.3f ms + factor/solve/reduce %.3f ms = %.0f k eval/s); '
          'max rel. difference to the one-at-a-time path %.1e' % (N, B, t * 1e3, B / t / 1e3, pr['gram']['ms'], pr['potrf']['ms'], B / (dev_ms * 1e-3) / 1e3, worst), flush=True)
    del K, W, a
# through the public API
rng = np.random.default_rng(1)
N = 128
X = rng.uniform(0, N ** (1 / d), (N, d)); y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
gp = g3.GaussianProcess(space=X, location=g3.Zero(), kernel=g3.SE(X))
gp.observed(X, y)
p0 = gp.active.dict_to_array(gp.params)
chain = p0[None, :] + 0.05 * rng.standard_normal((B, len(p0)))
gp.logp_chain(chain[:64])
t0 = time.perf_counter(); lp = gp.logp_chain(chain); t = time.perf_counter() - t0
print('GaussianProcess.logp_chain, N=128, %d rows: %.1f ms = %.0f k eval/s (block host path, see scripts/r4_chain_host.py)' % (B, t * 1e3, B / t / 1e3))
