"""Medium-N chains: evaluations / s and TFLOP/s of g3_gp_factor_batched for 256 < N <= 1024 with the
cooperative kernel (g3_chainb.hip: a group of workgroups per member, the batch in one launch) against the batched large-N sweep
(G3_COOP_MAX_N=0), and every member against the one-at-a-time path.  usage: python scripts/r5_chain_medium.py [case ...]   case = N:B[:G]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
cases = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]] or [(512, 4096), (512, 512), (384, 1024), (640, 512), (768, 256), (1024, 96), (1024, 512), (896, 64)]
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec
d = 4
for case in cases:
    N, B = case[0], case[1]
    rng = np.random.default_rng(N)
    X = rng.uniform(0, N ** (1 / d), (N, d)); y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    hyp = [(1.0 + 0.3 * (i % 97) / 97, 0.7 + 0.6 * (i % 89) / 89, 0.05 + 0.1 * (i % 13) / 13) for i in range(B)]
    progs = [compile_spec(('sum', ('SE', v, np.full(d, r), None), ('NOISE', nz)), d) for v, r, nz in hyp]
    arr = (_lib.KernelProg * B)(*progs)
    Np = _lib.roundup(N); kstride = (Np + 128) * Np
    res = {}
    for mode in ('coop', 'sweep'):
        os.environ['G3_COOP_MAX_N'] = '1024' if mode == 'coop' else '0'
        if len(case) > 2:
            os.environ['G3_COOP_GROUP'] = str(case[2])
        dev = g3.Device(0)                                    # knobs are read when a context is created
        K = dev.alloc(B * (Np + 128), Np, np.float64); W = dev.alloc(B * Np, 128, np.float64); a = dev.alloc(B, Np, np.float64)
        Xd, dd = dev.upload(X), dev.upload(np.tile(y, (B, 1)))
        st = dev.gp_factor_batched(arr, Xd, N, d, dd, K, kstride, W, a, raw=True)
        ts = []
        for _ in range(4):
            dev.sync(); t0 = time.perf_counter(); st = dev.gp_factor_batched(arr, Xd, N, d, dd, K, kstride, W, a, raw=True); dev.sync(); ts.append(time.perf_counter() - t0)
        dev.prof_enable(1); dev.prof_reset(); dev.gp_factor_batched(arr, Xd, N, d, dd, K, kstride, W, a, raw=True); pr = dev.prof_collect(); dev.prof_enable(False)
        res[mode] = (min(ts), st.copy(), dev.download(a, B, N).copy(), pr['gram']['ms'], pr['potrf']['ms'])
        if mode == 'coop':      # every 29th member against the one-at-a-time path
            K1, a1, W1 = dev.alloc(Np + 128, Np, np.float64), dev.alloc(1, Np, np.float64), dev.alloc_inverses(Np, np.float64)
            yd = dev.upload(y); worst = 0.0; worst_a = 0.0
            for i in range(0, B, 29):
                s1 = dev.gp_factor(progs[i], Xd, N, d, yd, K1, W1, a1)
                lp1 = -0.5 * s1['quad'] - s1['logdet']; lpb = -0.5 * st[i, 1] - st[i, 0]
                worst = max(worst, abs(lp1 - lpb) / abs(lp1))
                worst_a = max(worst_a, float(np.abs(dev.download(a1, 1, N)[0] - res[mode][2][i]).max()))
            for b_ in (K1, a1, W1, yd):
                b_.free()
        for b_ in (K, W, a, Xd, dd):
            b_.free()
        dev.close()
    tc, ts_ = res['coop'][0], res['sweep'][0]
    same = float(np.abs(res['coop'][1][:, :2] - res['sweep'][1][:, :2]).max() / np.abs(res['sweep'][1][:, :2]).max())
    fl = B * N ** 3 / 3.0
    print('N=%4d B=%4d%s: cooperative %.3f ms = %.0f k eval/s, %.1f TFLOP/s (device: Gram %.3f + factor %.3f ms) | batched sweep %.3f ms = %.0f k eval/s, %.1f TFLOP/s '
          '(Gram %.3f + factor %.3f) | x%.2f; logdet/quad max rel diff coop vs sweep %.1e, logp vs one-at-a-time %.1e, |a - a1| %.1e, failed members %d'
          % (N, B, (' G=%d' % case[2]) if len(case) > 2 else '', tc * 1e3, B / tc / 1e3, fl / tc / 1e12, res['coop'][3], res['coop'][4], ts_ * 1e3, B / ts_ / 1e3, fl / ts_ / 1e12,
             res['sweep'][3], res['sweep'][4], ts_ / tc, same, worst, worst_a, int((res['coop'][1][:, 3] > 0).sum())), flush=True)
