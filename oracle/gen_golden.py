"""Generate the committed golden fixtures under tests/golden/.  TEST INFRASTRUCTURE ONLY.

Run in the BUILD container only (needs /root/reference):

    python oracle/gen_golden.py

Two families of fixtures:
  gpmm_*.npz   -- outputs of the reference's own executable NumPy GP prototype
                  `/root/reference/sandbox/gpmm.py` (CovCholesky :74-93, NLLFromCholAndObs
                  :121-125, MeanPredFromCho/VarPredFromCho/CovPredFromCho :107-118,
                  GaussianKernel/LaplaceKernel :295-302), imported as-is with the
                  styling-only `seaborn` module stubbed.  These PIN the oracle.
                  Mapping to the main path:
                  H = log var, W_k = log(0.5*rate_k**2) (SE) or log(rate_k) (OU),
                  noise = log sigma^2, zero mean, identity mapping => logp = -NLL.
  oracle_*.npz -- outputs of oracle/g3_oracle.py on seeded inputs for every kernel family
                  and for the warped / jitter paths; regression pins for the HIP build
                  (the reference has no vectors for these: "parity unpinned").
Only data (inputs and expected outputs) is written; no reference source is copied.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')

from oracle import g3_oracle as orc  # noqa: E402


def import_gpmm():
    sb = types.ModuleType('seaborn')
    sb.set = lambda *a, **k: None
    sys.modules['seaborn'] = sb
    import matplotlib
    matplotlib.use('Agg')
    sys.path.insert(0, '/root/reference/sandbox')
    import gpmm
    return gpmm


def synth(seed, N, d, M):
    """Synthetic inputs as in SURVEY.md section 8(d)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Lbox = N ** (1.0 / d)
    X = rng.uniform(0, Lbox, (N, d))
    Xs = rng.uniform(0, Lbox, (M, d))
    y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(N)
    return X, y, Xs


def gen_gpmm():
    gpmm = import_gpmm()
    cases = [
        ('se_d1', 'SE', 2001, 128, 1, 24, 1.3, [0.7], 0.1),
        ('se_d3', 'SE', 2002, 300, 3, 40, 1.0, [1.0, 0.6, 1.7], 0.1),
        ('se_d4', 'SE', 2003, 512, 4, 64, 0.8, [1.0, 1.0, 1.0, 1.0], 0.05),
        ('ou_d2', 'OU', 2004, 200, 2, 32, 1.5, [0.8, 1.2], 0.2),
    ]
    for name, kind, seed, N, d, M, var, rate, noise in cases:
        X, y, Xs = synth(seed, N, d, M)
        rate = np.array(rate)
        if kind == 'SE':
            W = np.log(0.5 * rate ** 2)
            Kf = gpmm.GaussianKernel(np.log(noise), np.log(var), W)
        else:
            W = np.log(rate)
            Kf = gpmm.LaplaceKernel(np.log(noise), np.log(var), W)
        K = Kf.cov(X)                       # includes noise on the diagonal
        L = gpmm.CovCholesky(K)
        nll = gpmm.NLLFromCholAndObs(L, y)
        Kyx = Kf.cov(Xs, X)                 # cross covariance: no noise
        Kyy = Kf.cov(Xs) - noise * np.eye(M)  # prior f-kernel on the test points
        mean = gpmm.MeanPredFromCho(Kyx, L, y)
        varp = gpmm.VarPredFromCho(Kyx, L, Kyy)
        cov = gpmm.CovPredFromCho(Kyx, L, Kyy)
        # gradient of logp = -NLL w.r.t. (log noise, log var, log rate_k): 5-point central
        # differences of gpmm's OWN NLL(params, Kernel, X, Fx) (:128-130) -- pins dlogp for SE / OU.
        # gpmm's parameters are (log noise, H = log var, W_k) with W_k = log(rate_k^2 / 2) (SE) or
        # log rate_k (OU), so d/dlog rate_k = 2 d/dW_k (SE) or d/dW_k (OU).
        p0 = Kf.params().astype(float)
        h = 1e-3
        dl = np.zeros(len(p0))
        for i in range(len(p0)):
            e = np.zeros(len(p0)); e[i] = h
            f = [gpmm.NLL(p0 + c * e, Kf, X, y) for c in (-2, -1, 1, 2)]
            dl[i] = -(f[0] - 8 * f[1] + 8 * f[2] - f[3]) / (12 * h)
        Kf.setParameters(p0)
        dl[2:] *= 2.0 if kind == 'SE' else 1.0
        np.savez_compressed(os.path.join(OUT, 'gpmm_%s.npz' % name), kind=kind, X=X, y=y, Xs=Xs,
                            var=var, rate=rate, noise=noise, K=K, L=np.tril(L), logp=-nll,
                            mean=mean, variance=varp, covariance=cov, dlogp_log=dl)
        print('gpmm', name, 'logp', -nll)


def kernel_zoo(d):
    r = np.linspace(0.6, 1.4, d)
    f = np.linspace(0.11, 0.23, d)
    return {
        'SE': ('SE', 1.3, r, None),
        'OU': ('OU', 0.9, r, None),
        'MAT32': ('MAT32', 1.1, r, None),
        'MAT52': ('MAT52', 0.7, r, None),
        'RQ': ('RQ', 1.2, r, 1.7, None),
        'COS': ('COS', 0.8, f, None),
        'SIN': ('SIN', 0.5, f, 0.25 * r, None),
        'SINC': ('SINC', 1.4, f, None),
        'SM': ('SM', 0.9, f, 0.3 * r, None),
        'WN': ('WN', 0.4, None),
        'MAT52+COS': ('sum', ('MAT52', 1.0, r, None), ('COS', 0.5, f, None)),
        'SE*COS': ('prod', ('SE', 1.0, r, None), ('COS', 1.0, f, None)),
        '2*SE+0.1': ('shift', 0.1, ('scale', 2.0, ('SE', 1.0, r, None))),
        '(SE+OU)*(MAT32+0.5)': ('prod', ('sum', ('SE', 1.0, r, None), ('OU', 0.5, r, None)),
                                ('shift', 0.5, ('MAT32', 0.7, r, None))),
        'SE[dims]': ('SE', 1.0, r[:2], np.array([0, d - 1])) if d > 1 else ('SE', 1.0, r, None),
    }


def gen_oracle():
    out = {}
    for d in (1, 3, 8):
        X, y, Xs = synth(3000 + d, 96, d, 40)
        if d == 3:          # exact coincidences exercise SINC's d==0 branch and WN cross
            Xs[:5] = X[:5]
            Xs[5, 0] = X[5, 0]
        out['d%d_X' % d], out['d%d_y' % d], out['d%d_Xs' % d] = X, y, Xs
        for name, spec in kernel_zoo(d).items():
            out['d%d_%s_sym' % (d, name)] = orc.kernel_cov(spec, X)
            out['d%d_%s_cross' % (d, name)] = orc.kernel_cov(spec, Xs, X)
    np.savez_compressed(os.path.join(OUT, 'oracle_kernels.npz'), **out)

    # full process fixtures: GP and warped GP, fp64
    res = {}
    X, y, Xs = synth(4001, 160, 2, 48)
    rng = np.random.Generator(np.random.PCG64(4002))
    Z = rng.standard_normal((48, 5))
    res.update(X=X, y=y, Xs=Xs, Z=Z)
    r = np.array([0.9, 1.2])
    f = np.array([0.125, 0.125])
    procs = {
        'gp_se_bias': dict(kernel_f=('SE', 1.1, r, None), noise_var=0.1, mean=('Bias', 0.3)),
        'gp_mat52cos_zero': dict(kernel_f=('sum', ('MAT52', 1.0, r, None), ('COS', 0.5, f, None)),
                                 noise_var=0.1, mean=('Zero',)),
        'gp_se_linear': dict(kernel_f=('SE', 0.7, r, None), noise_var=0.05,
                             mean=('Linear', 0.1, np.array([0.02, -0.03]), None)),
        'wgp_boxcox': dict(kernel_f=('SE', 1.0, r, None), noise_var=0.1, mean=('Bias', 0.2),
                           mapping=('BoxCoxLinear', 1.0, 1.0, 1.2)),
        'wgp_arcsinh': dict(kernel_f=('SE', 1.0, r, None), noise_var=0.1, mean=('Zero',),
                            mapping=('ArcsinhLinear', 0.1, 0.8)),
        'wgp_logshift': dict(kernel_f=('OU', 1.0, r, None), noise_var=0.1, mean=('Zero',),
                             mapping=('LogShifted', -0.5)),
        'wgp_linear': dict(kernel_f=('RQ', 1.0, r, 1.5, None), noise_var=0.1, mean=('Zero',),
                           mapping=('LinearMapping', 0.2, 1.5)),
    }
    for name, kw in procs.items():
        gp = orc.GP(**kw)
        yy = y
        if kw.get('mapping', ('Identity',))[0] in ('BoxCoxLinear', 'LogShifted'):
            yy = y - y.min() + 1.0
        res[name + '_y'] = yy
        res[name + '_logp'] = gp.logp(X, yy)
        for noise in (False, True):
            s = '_n%d' % noise
            res[name + '_mean' + s] = gp.mean(Xs, X, yy, noise=noise)
            res[name + '_median' + s] = gp.median(Xs, X, yy, noise=noise)
            res[name + '_var' + s] = gp.variance(Xs, X, yy, noise=noise)
            res[name + '_cov' + s] = gp.kernel(Xs, X, noise=noise)
            res[name + '_q975' + s] = gp.quantiler(Xs, X, yy, 0.975, noise=noise)
            res[name + '_samples' + s] = gp.sampler(Xs, X, yy, Z, noise=noise)
        res[name + '_prior_mean'] = gp.mean(Xs, prior=True)
        res[name + '_prior_var_n1'] = gp.variance(Xs, prior=True, noise=True)
        res[name + '_logpred'] = gp.logpredictive(gp.median(Xs, X, yy), Xs, X, yy)
        dg = gp.dlogp_natural(X, yy)     # natural-space gradient, oracle ordering (see GP.dlogp_natural)
        res[name + '_dlogp_kernel'] = np.array([v for _, _, _, v in dg['kernel']])
        res[name + '_dlogp_mean'] = np.array([v for _, _, v in dg['mean']])
        res[name + '_dlogp_mapping'] = np.array([v for _, v in dg['mapping']])
    np.savez_compressed(os.path.join(OUT, 'oracle_process.npz'), **res)

    # jitter-path cases for CholeskyRobust (tensors.py:197-222)
    jit = {}
    rng = np.random.Generator(np.random.PCG64(5001))
    B = rng.standard_normal((48, 6))
    K1 = B.dot(B.T)                                  # rank-6 PSD: dpotrf fails, jitter rescues
    K2 = K1.copy(); K2[7, 7] = -0.3                  # negative diagonal: lifted first
    K3 = K1 + 1e-3 * np.eye(48)                      # PD: plain path
    K4 = -np.eye(48) - 3.0 * np.ones((48, 48))       # hopeless: negative jitter, ends in the 1e-10*I fallback
    # (a NaN entry is not a fixture: whether dpotrf reports info != 0 for NaN pivots is
    #  BLAS-vendor specific -- OpenBLAS returns info == 0 -- and the process path scrubs
    #  NaN with tt_to_cov before factorising, tensors.py:95-98)
    for i, K in enumerate((K1, K2, K3, K4), 1):
        L, tries, fb = orc.cholesky_robust(K, return_info=True)
        jit['K%d' % i], jit['L%d' % i] = K, L
        jit['tries%d' % i], jit['fallback%d' % i] = tries, fb
        print('jitter case', i, 'tries', tries, 'fallback', fb)
    np.savez_compressed(os.path.join(OUT, 'oracle_jitter.npz'), **jit)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    gen_gpmm()
    gen_oracle()
    print('fixtures written to', OUT)
