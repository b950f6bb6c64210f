"""Where the error on the jitter path comes from (development aid / evidence for DESIGN.md section 2):
near-singular covariances (duplicated inputs, no noise), factor and solve on the device vs LAPACK."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sl
import g3py_amd as g3
from g3py_amd import _lib
from g3py_amd.device import compile_spec
from oracle import g3_oracle as orc

dev = g3.Device.default()
for nu, d in ((40, 1), (150, 2), (600, 2), (1500, 3)):
    rng = np.random.default_rng(0)
    X = np.repeat(rng.uniform(0, 3, (nu, d)), 2, axis=0)
    y = np.sin(X.sum(1))
    N = len(X)
    spec = ('SE', 1.0, np.ones(d), None)
    K = orc.tt_to_cov(orc.tt_to_num(orc.kernel_cov(spec, X)))
    L, tries, fb = orc.cholesky_robust(K, return_info=True)
    a_ref = sl.solve_triangular(L, y, lower=True)
    Np = _lib.roundup(N)
    Kd, W, ad = dev.alloc(Np + 128, Np, np.float64), dev.alloc_inverses(Np, np.float64), dev.alloc(1, Np, np.float64)
    st = dev.gp_factor(compile_spec(spec, d), dev.upload(X), N, d, dev.upload(y), Kd, W, ad)
    Lg = np.tril(dev.download(Kd, N, N))
    ag = dev.download(ad, 1, N)[0]
    a_sub = sl.solve_triangular(Lg, y, lower=True)          # substitution with the DEVICE factor
    lp = lambda a_, L_: -0.5 * N * np.log(2 * np.pi) - 0.5 * a_.dot(a_) - np.sum(np.log(np.diag(L_)))
    print('N=%5d tries %d/%d cond(L) %.1e | factor rel err %.1e | a: device-inverse vs LAPACK %.1e, substitution on device factor vs LAPACK %.1e'
          ' | logp rel err %.1e (with substitution: %.1e)'
          % (N, st['tries'], tries, np.linalg.cond(L), np.abs(Lg - L).max() / np.abs(L).max(),
             np.abs(ag - a_ref).max() / np.abs(a_ref).max(), np.abs(a_sub - a_ref).max() / np.abs(a_ref).max(),
             abs(lp(ag, Lg) - lp(a_ref, L)) / abs(lp(a_ref, L)), abs(lp(a_sub, Lg) - lp(a_ref, L)) / abs(lp(a_ref, L))))
