"""CPU tests: pin oracle/g3_oracle.py against the reference's own gpmm.py outputs (committed
fixtures) and against analytic known answers."""
import glob
import os

import numpy as np
import pytest
import scipy.linalg

from oracle import g3_oracle as orc


def _gpmm_cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, 'gpmm_*.npz')))


def test_gpmm_fixtures_exist(golden_dir):
    assert len(_gpmm_cases(golden_dir)) == 4


@pytest.mark.parametrize('name', ['se_d1', 'se_d3', 'se_d4', 'ou_d2'])
def test_oracle_matches_gpmm(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'gpmm_%s.npz' % name))
    kind = str(g['kind'])
    spec = (kind, float(g['var']), g['rate'], None)
    gp = orc.GP(kernel_f=spec, noise_var=float(g['noise']))
    X, y, Xs = g['X'], g['y'], g['Xs']
    K = gp.prior_kernel(X, noise=True)
    np.testing.assert_allclose(K, g['K'], rtol=1e-13, atol=1e-15)
    L = orc.cholesky_robust(K)
    np.testing.assert_allclose(L, g['L'], rtol=1e-10, atol=1e-13)
    lp = gp.logp(X, y)
    assert abs(lp - float(g['logp'])) <= 1e-11 * abs(float(g['logp']))
    # the reference main path uses an LU solve (elliptical.py:81-91); gpmm uses Cholesky solves
    np.testing.assert_allclose(gp.mean(Xs, X, y), g['mean'], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(gp.kernel(Xs, X), g['covariance'], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(gp.variance(Xs, X, y), np.maximum(g['variance'], 0), rtol=1e-8, atol=1e-11)


def test_kat_n1_n2_closed_form():
    # N=1: logp = -0.5 log(2 pi (v+s)) - y^2 / (2 (v+s))
    v, s, y = 1.7, 0.3, 0.8
    gp = orc.GP(('SE', v, np.array([1.0]), None), noise_var=s)
    lp = gp.logp(np.array([[0.2]]), np.array([y]))
    assert abs(lp - (-0.5 * np.log(2 * np.pi * (v + s)) - y * y / (2 * (v + s)))) < 1e-14
    # N=2 closed form through the 2x2 determinant / inverse
    x = np.array([[0.0], [0.9]])
    yy = np.array([0.3, -0.4])
    rate = 1.3
    k = v * np.exp(-0.5 * rate ** 2 * 0.81)
    S = np.array([[v + s, k], [k, v + s]])
    ref = -np.log(2 * np.pi) - 0.5 * np.log(np.linalg.det(S)) - 0.5 * yy.dot(np.linalg.solve(S, yy))
    gp = orc.GP(('SE', v, np.array([rate]), None), noise_var=s)
    assert abs(gp.logp(x, yy) - ref) < 1e-13


def test_kat_noise_only():
    rng = np.random.default_rng(0)
    n, s = 37, 0.45
    y = rng.standard_normal(n)
    gp = orc.GP(('scale', 0.0, ('SE', 1.0, np.array([1.0]), None)), noise_var=s)
    ref = -0.5 * n * np.log(2 * np.pi * s) - y.dot(y) / (2 * s)
    assert abs(gp.logp(rng.standard_normal((n, 1)), y) - ref) < 1e-12


def test_kat_interpolation_and_limits():
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 5, (20, 2))
    y = np.sin(X.sum(1))
    gp = orc.GP(('SE', 1.0, np.array([1.0, 1.0]), None), noise_var=1e-10)
    np.testing.assert_allclose(gp.mean(X[:5], X, y), y[:5], atol=1e-5)
    # RQ -> SE as alpha -> inf ; MAT kernels at d=0 equal var
    a = orc.kernel_cov(('RQ', 1.2, np.array([0.7, 0.9]), 1e7, None), X)
    b = orc.kernel_cov(('SE', 1.2, np.array([0.7, 0.9]), None), X)
    np.testing.assert_allclose(a, b, rtol=1e-4)
    for k in ('MAT32', 'MAT52'):
        np.testing.assert_allclose(np.diag(orc.kernel_cov((k, 0.9, np.array([1.0, 1.0]), None), X)), 0.9)
    # COS in d=1 has rank 2 (cos(a-b) = cos a cos b + sin a sin b)
    x1 = rng.uniform(0, 5, (30, 1))
    C = orc.kernel_cov(('COS', 1.0, np.array([0.3]), None), x1)
    assert np.linalg.matrix_rank(C, tol=1e-10) == 2


def test_half_factor_in_ard_l2():
    # metrics.py:102: the 1/2 lives inside the metric, so MAT32 sees sqrt(3 * 0.5 * r^2 * dx^2)
    x = np.array([[0.0], [2.0]])
    r = 0.8
    d = 0.5 * r * r * 4.0
    k = orc.kernel_cov(('MAT32', 1.0, np.array([r]), None), x)[0, 1]
    assert abs(k - (1 + np.sqrt(3 * d)) * np.exp(-np.sqrt(3 * d))) < 1e-15


def test_noise_excluded_from_cross(golden_dir):
    X = np.arange(6.0)[:, None]
    spec = orc.with_noise(('SE', 1.0, np.array([1.0]), None), 0.5)
    sq = orc.kernel_cov(spec, X)
    cr = orc.kernel_cov(spec, X, X)       # coincident points, but cross => no noise (kernels.py:367-371)
    np.testing.assert_allclose(np.diag(sq) - np.diag(cr), 0.5)


def test_scrubs():
    a = np.array([[np.nan, np.inf], [-np.inf, 2.0]])
    r = orc.tt_to_num(a)
    assert r[0, 0] == 0 and r[0, 1] == np.float32(1e10) and r[1, 0] == np.float32(1e10) and r[1, 1] == 2
    c = np.array([[-0.5, 0.1], [0.1, 1.0]])
    rc = orc.tt_to_cov(c)
    np.testing.assert_allclose(np.diag(rc), [np.float32(1e-6), 1.0 + 0.5 + np.float32(1e-6)])
    np.testing.assert_allclose(orc.tt_to_bounded(np.array([-1.0, 2.0]), 0.0), [0.0, 2.0])


def test_jitter_schedule(golden_dir):
    g = np.load(os.path.join(golden_dir, 'oracle_jitter.npz'))
    for i in (1, 2, 3, 4):
        L, tries, fb = orc.cholesky_robust(g['K%d' % i], return_info=True)
        np.testing.assert_allclose(L, g['L%d' % i], rtol=1e-12, atol=1e-14)
        assert tries == int(g['tries%d' % i]) and fb == bool(g['fallback%d' % i])
    # case 1: one jitter step mean(diag)*1e-6 rescues a rank-deficient PSD matrix
    K = g['K1']
    L = g['L1']
    jit = np.diag(K).mean() * np.float32(1e-6)
    np.testing.assert_allclose(L.dot(L.T), K + jit * np.eye(len(K)), atol=1e-10)
    # case 4: final fallback is 1e-10 * I (tensors.py:221)
    np.testing.assert_allclose(g['L4'], np.float32(1e-10) * np.eye(len(K)))


def test_logp_sentinel():
    gp = orc.GP(('SE', 1.0, np.array([1.0]), None), noise_var=0.1)
    X = np.arange(5.0)[:, None]
    y = np.array([0.1, np.inf, 0.2, 0.3, 0.4])
    assert gp.logp(X, y) == np.float32(-1e30)
    # log-transform Jacobian: exp(v) <= 1e-6 => -inf (hypers/__init__.py:199-200)
    gp2 = orc.GP(('SE', 1.0, np.array([1.0]), None), noise_var=0.1, log_positive_hypers=[np.log(1e-7)])
    assert gp2.logp(X, np.zeros(5)) == -np.inf


def test_mappings_inverse_and_logdet():
    y = np.linspace(0.5, 3.0, 11)
    for spec in [('Identity',), ('LinearMapping', 0.2, 1.5), ('LogShifted', -0.5),
                 ('BoxCoxLinear', 1.0, 0.7, 1.2), ('ArcsinhLinear', 0.1, 0.8)]:
        m = orc.Mapping(spec)
        np.testing.assert_allclose(m(m.inv(y)), y, rtol=1e-12)
        # logdet_dinv == sum log |d inv / dy| by central differences
        h = 1e-6
        num = np.sum(np.log(np.abs((m.inv(y + h) - m.inv(y - h)) / (2 * h))))
        assert abs(num - m.logdet_dinv(y)) < 1e-6


def test_gauss_hermite_lognormal():
    # E[exp(f)] for f ~ N(mu, s^2) is exp(mu + s^2/2); 10 nodes are ample for s = 0.3
    gp = orc.GP(('SE', 1.0, np.array([1.0]), None), noise_var=0.1, mapping=('LogShifted', 0.0))
    mu, s = np.array([0.2, -0.1]), np.array([0.3, 0.2])
    got = gp.gauss_hermite(lambda v: gp.map(v), mu, s)
    np.testing.assert_allclose(got, np.exp(mu + s ** 2 / 2), rtol=1e-9)


def test_fp32_graph_mode_close_to_fp64():
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 4, (64, 2))
    y = np.sin(X.sum(1))
    a = orc.GP(('SE', 1.0, np.array([1.0, 1.0]), None), 0.1).logp(X, y)
    b = orc.GP(('SE', 1.0, np.array([1.0, 1.0]), None), 0.1, dtype=np.float32).logp(
        X.astype(np.float32), y.astype(np.float32))
    assert b.dtype == np.float32 and abs(a - b) < 1e-3 * abs(a)


def test_cpu_hot_path_matches_oracle():
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 4, (200, 4))
    Xs = rng.uniform(0, 4, (16, 4))
    y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(200)
    lp, mean, var, _ = orc.cpu_hot_path(X, y, Xs)
    gp = orc.GP(('SE', 1.0, np.ones(4), None), 0.1)
    assert abs(lp - gp.logp(X, y)) < 1e-9 * abs(lp)
    np.testing.assert_allclose(mean, gp.mean(Xs, X, y), atol=1e-9)
    np.testing.assert_allclose(var, gp.variance(Xs, X, y), atol=1e-9)


# ------------------------------------------------------------------ gradient of logp
def _bump(spec, leaf, pname, k, delta, cnt=None):
    """copy of a kernel spec tree with one leaf parameter moved by delta"""
    cnt = [0] if cnt is None else cnt
    op = spec[0]
    if op in ('sum', 'prod'):
        a = _bump(spec[1], leaf, pname, k, delta, cnt)
        return (op, a, _bump(spec[2], leaf, pname, k, delta, cnt))
    if op in ('scale', 'shift'):
        return (op, spec[1], _bump(spec[2], leaf, pname, k, delta, cnt))
    me = cnt[0]
    cnt[0] += 1
    if me != leaf:
        return spec
    idx = dict(var=1)
    idx.update({'SE': dict(rate=2), 'OU': dict(rate=2), 'MAT32': dict(rate=2), 'MAT52': dict(rate=2),
                'RQ': dict(rate=2, alpha=3), 'COS': dict(freq=2), 'SINC': dict(freq=2),
                'SIN': dict(freq=2, rate=3), 'SM': dict(freq=2, rate=3)}.get(op, {}))
    s = list(spec)
    if k is None:
        s[idx[pname]] = s[idx[pname]] + delta
    else:
        v = np.array(s[idx[pname]], dtype=float)
        v[k] += delta
        s[idx[pname]] = v
    return tuple(s)


def _grad_specs(d):
    from oracle.gen_golden import kernel_zoo
    z = kernel_zoo(d)
    z['SE+NOISE'] = ('sum', z['SE'], ('NOISE', 0.3))
    return z


@pytest.mark.parametrize('name', sorted(_grad_specs(3)))
def test_kernel_cov_grads_match_finite_differences(name):
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 3, (30, 3))
    spec = _grad_specs(3)[name]
    K, grads = orc.kernel_cov_grads(spec, X)
    np.testing.assert_allclose(K, orc.kernel_cov(spec, X), rtol=1e-14)
    h = 1e-6
    for leaf, pname, k, dK in grads:
        fd = (orc.kernel_cov(_bump(spec, leaf, pname, k, h), X) - orc.kernel_cov(_bump(spec, leaf, pname, k, -h), X)) / (2 * h)
        np.testing.assert_allclose(dK, fd, atol=2e-8 * max(1.0, np.abs(fd).max()), err_msg=str((name, leaf, pname, k)))


def test_cholesky_grad_is_the_adjoint_of_the_factorisation():
    """CholeskyRobust.grad (tensors.py:224-260): sum(Kbar * dK) == sum(Lbar * dL) for symmetric dK"""
    rng = np.random.default_rng(4)
    A = rng.standard_normal((12, 12))
    K = A.dot(A.T) + 12 * np.eye(12)
    L = np.linalg.cholesky(K)
    Lbar = np.tril(rng.standard_normal((12, 12)))
    Kbar = orc.cholesky_grad(L, Lbar)
    S = rng.standard_normal((12, 12))
    dK = (S + S.T) / 2
    h = 1e-6
    dL = (np.linalg.cholesky(K + h * dK) - np.linalg.cholesky(K - h * dK)) / (2 * h)
    assert abs(np.sum(Kbar * dK) - np.sum(Lbar * dL)) < 1e-7


@pytest.mark.parametrize('kname', ['SE', 'OU', 'MAT32', 'MAT52', 'RQ', 'SM', 'SE*COS', '(SE+OU)*(MAT32+0.5)', 'SE[dims]'])
@pytest.mark.parametrize('mname', ['bias_identity', 'linear_boxcox', 'zero_arcsinh', 'zero_logshift', 'zero_linearmap'])
def test_dlogp_matches_finite_differences_of_logp(kname, mname):
    """the reverse-mode restatement (logp_cho -> CholeskyRobust.grad -> kernel / mean / mapping) against
    central differences of the oracle's own logp, which the gpmm fixtures pin"""
    rng = np.random.default_rng(0)
    N, d = 40, 3
    X = rng.uniform(0, 3, (N, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(N) + 2.5
    kf = _grad_specs(d)[kname]
    mean, mapping = {'bias_identity': (('Bias', 0.3), ('Identity',)),
                     'linear_boxcox': (('Linear', 0.1, np.array([0.2, -0.1, 0.05]), None), ('BoxCoxLinear', 1.0, 1.1, 1.2)),
                     'zero_arcsinh': (('Zero',), ('ArcsinhLinear', 0.2, 1.5)),
                     'zero_logshift': (('Zero',), ('LogShifted', -0.5)),
                     'zero_linearmap': (('Zero',), ('LinearMapping', 0.3, 1.7))}[mname]
    gp = orc.GP(kf, 0.1, mean, mapping)
    g = gp.dlogp_natural(X, y)
    h = 1e-5

    def check(val, lp, what):
        fd = (lp(h) - lp(-h)) / (2 * h)
        assert abs(fd - val) <= 2e-6 * max(1.0, abs(val)), (what, val, fd)

    for leaf, pname, k, val in g['kernel']:
        def lp(dl):
            g2 = orc.GP(kf, 0.1, mean, mapping)
            g2.kn = _bump(gp.kn, leaf, pname, k, dl)
            return g2.loglike(X, y)
        check(val, lp, (leaf, pname, k))
    for pname, k, val in g['mean']:
        def lp(dl):
            m = list(mean)
            if pname in ('bias', 'constant'):
                m[1] = m[1] + dl
            else:
                v = np.array(m[2], dtype=float)
                v[k] += dl
                m[2] = v
            return orc.GP(kf, 0.1, tuple(m), mapping).loglike(X, y)
        check(val, lp, (pname, k))
    for i, (pname, val) in enumerate(g['mapping']):
        def lp(dl):
            m = list(mapping)
            m[1 + i] = m[1 + i] + dl
            return orc.GP(kf, 0.1, mean, tuple(m)).loglike(X, y)
        check(val, lp, pname)


@pytest.mark.parametrize('name', ['se_d1', 'se_d3', 'se_d4', 'ou_d2'])
def test_dlogp_matches_reference_gpmm(golden_dir, name):
    """dlogp against finite differences of the reference prototype's own NLL (sandbox/gpmm.py:128-130)"""
    g = np.load(os.path.join(golden_dir, 'gpmm_%s.npz' % name))
    kind, rate = str(g['kind']), g['rate']
    gp = orc.GP((kind, float(g['var']), rate, None), float(g['noise']))
    dg = {(l, p, k): v for l, p, k, v in gp.dlogp_natural(g['X'], g['y'])['kernel']}
    got = np.array([dg[(1, 'var', None)] * float(g['noise']), dg[(0, 'var', None)] * float(g['var'])]
                   + [dg[(0, 'rate', k)] * rate[k] for k in range(len(rate))])
    np.testing.assert_allclose(got, g['dlogp_log'], rtol=1e-7, atol=1e-7)


def test_dlogp_sentinel_branch_is_flat():
    X = np.linspace(0, 1, 8)[:, None]
    y = np.ones(8)
    y[2] = np.inf
    g = orc.GP(('SE', 1.0, np.ones(1), None), 0.1, ('Bias', 0.0)).dlogp_natural(X, y)
    assert all(v == 0 for *_, v in g['kernel']) and all(v == 0 for *_, v in g['mean'])


# ------------------------------------------------------------------ Student-t process
def test_student_t_known_answers():
    """n = 1: the density is scipy's univariate t with scale^2 = K (nu - 2) / nu; nu -> infinity: the
    Gaussian logp; posterior scaling = 1 when beta = n (studentT.py:36-44,114-135)"""
    from scipy import stats
    X = np.array([[0.3]])
    y = np.array([0.7])
    tp = orc.TP(('SE', 1.3, np.ones(1), None), degree=3.0, noise_var=0.2)
    nu, K = 5.0, 1.5
    want = stats.t(df=nu, scale=np.sqrt(K * (nu - 2) / nu)).logpdf(0.7)
    assert abs(tp.logp(X, y) - want) < 1e-7       # the reference's normaliser uses float32 pi (studentT.py:122)
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 3, (30, 2))
    y = np.sin(X.sum(1))
    kf = ('SE', 1.0, np.ones(2), None)
    big = orc.TP(kf, degree=1e7, noise_var=0.1)
    # nu >= 1e6 switches to the Gaussian normaliser in float32 pi (:125): compare to 1e-6
    assert abs(big.logp(X, y) - orc.GP(kf, 0.1).logp(X, y)) < 1e-5 * abs(orc.GP(kf, 0.1).logp(X, y))
    tp = orc.TP(kf, degree=4.0, noise_var=0.1)
    s = tp.scaling(X, y)
    L = np.linalg.cholesky(orc.kernel_cov(orc.with_noise(kf, 0.1), X))
    beta = np.sum(np.linalg.solve(L, y) ** 2)
    assert abs(s - (6.0 + beta - 2) / (6.0 + 30 - 2)) < 1e-12
    Xs = rng.uniform(0, 3, (5, 2))
    np.testing.assert_allclose(tp.variance(Xs, X, y), orc.GP(kf, 0.1).variance(Xs, X, y) * s, rtol=1e-12)


def test_student_t_dlogp_matches_finite_differences():
    rng = np.random.default_rng(2)
    N, d = 35, 2
    X = rng.uniform(0, 3, (N, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(N) + 2.0
    r = np.array([0.8, 1.2])
    mk = lambda var, rate, noise, bias, deg, shift: orc.TP(('SE', var, rate, None), deg, noise, ('Bias', bias), ('ArcsinhLinear', shift, 1.3))
    base = dict(var=1.1, rate=r, noise=0.1, bias=0.3, deg=3.0, shift=0.2)
    tp = mk(**base)
    g = tp.dlogp_natural(X, y)
    h = 1e-5

    def fd(key, k=None):
        def lp(dl):
            q = dict(base)
            if k is None:
                q[key] = q[key] + dl
            else:
                v = np.array(q[key], dtype=float)
                v[k] += dl
                q[key] = v
            return mk(**q).loglike(X, y)
        return (lp(h) - lp(-h)) / (2 * h)
    kern = {(l, p, k): v for l, p, k, v in g['kernel']}
    for val, f in [(kern[(0, 'var', None)], fd('var')), (kern[(0, 'rate', 0)], fd('rate', 0)), (kern[(0, 'rate', 1)], fd('rate', 1)),
                   (kern[(1, 'var', None)], fd('noise')), (g['mean'][0][2], fd('bias')), (g['mapping'][0][1], fd('shift')),
                   (tp.dlogp_degree(X, y), fd('deg'))]:
        assert abs(val - f) <= 2e-6 * max(1.0, abs(val)), (val, f)


def test_tkernel_oracle_posterior_is_the_gp_posterior():
    """transports.py:239-257 with pred = 0 is the GP posterior mean; with pred = z a posterior draw"""
    rng = np.random.default_rng(8)
    X, Xs = rng.uniform(0, 3, (50, 2)), rng.uniform(0, 3, (7, 2))
    y = np.sin(X.sum(1))
    kf = ('SE', 1.2, np.array([0.8, 1.1]), None)
    t = orc.TKernelOracle(kf, 0.1)
    gp = orc.GP(kf, 0.1)
    np.testing.assert_allclose(t.posterior(Xs, np.zeros(7), X, y), gp.mean(Xs, X, y), atol=1e-10)
    z = rng.standard_normal(7)
    Lp = np.linalg.cholesky(gp.kernel(Xs, X, noise=False))
    np.testing.assert_allclose(t.posterior(Xs, z, X, y), gp.mean(Xs, X, y) + Lp.dot(z), atol=1e-8)
    np.testing.assert_allclose(t.inv(X, t(X, y, noise=True), noise=True), y, atol=1e-10)
    assert abs(t.logdet_dinv(X, y) + 0.5 * np.linalg.slogdet(orc.kernel_cov(t.kn, X))[1]) < 1e-10


def test_fullsize_generator_equals_the_literal_oracle_at_c5mini(golden_dir):
    """oracle/gen_fullsize.py factors in blocks and shares solves so that N=65536 fits in memory; at `c5mini`
    (the same warped-GP path, N=2048) its committed numbers must equal the oracle's literal restatement -- GP.logp,
    location, kernel_diag, sampler with the same normals (gaussian.py:75-97,208-232; elliptical.py:81-97)"""
    import json
    from oracle import g3_oracle as orc
    g = json.load(open(os.path.join(golden_dir, 'fullsize.json')))['c5mini']
    N, d, M, seed, S = g['N'], g['d'], g['M'], g['seed'], g['draws']
    rng = np.random.Generator(np.random.PCG64(seed))
    Lbox = N ** (1.0 / d)
    X = rng.uniform(0, Lbox, (N, d))
    Xs = rng.uniform(0, Lbox, (M, d))
    y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(N)
    yw = y - y.min() + 1.0
    gp = orc.GP(('SE', 1.0, np.ones(d), None), g['noise'], mapping=tuple(g['mapping']))
    lp = gp.logp(X, yw)
    assert abs(lp - g['logp']) <= 1e-10 * abs(g['logp'])
    nq = len(g['mean'])
    np.testing.assert_allclose(gp.location(Xs[:nq], X, yw), g['mean'], atol=1e-9)
    np.testing.assert_allclose(gp.kernel_diag(Xs[:nq], X), g['variance'], atol=1e-9)
    np.testing.assert_allclose(gp.mean(Xs[:nq], X, yw), g['gh_mean'], atol=1e-9)
    np.testing.assert_allclose(gp.variance(Xs[:nq], X, yw), g['gh_variance'], atol=1e-9)
    Z = np.random.Generator(np.random.PCG64(seed + 100)).standard_normal((M, S))
    draws = gp.sampler(Xs, X, yw, rand=Z)
    np.testing.assert_allclose(draws[g['draw_rows']], np.asarray(g['draw_values']), atol=1e-8)
