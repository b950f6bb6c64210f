"""GPU parity of the g3py-compatible process API (GaussianProcess / WarpedGaussianProcess)
against the oracle's golden fixtures and the reference's own gpmm.py outputs.
Tolerances: fp64 logp 1e-8 relative (BASELINE.json), means/variances 1e-8 absolute*scale."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _params(gp, **nat):
    """natural-space keyword values -> the reference's transformed-space params dict"""
    p = gp.params_test
    by = {v.name: v for v in gp.model.vars}
    for k, val in nat.items():
        v = by[gp.name + '_' + k]
        p[v.key] = (np.log(val) if v.positive else np.asarray(val, dtype=float)) * np.ones(v.shape)
    return p


@pytest.mark.parametrize('name', ['se_d1', 'se_d3', 'se_d4', 'ou_d2'])
def test_gp_matches_reference_gpmm(golden_dir, name):
    import g3py_amd as g3
    g = np.load(os.path.join(golden_dir, 'gpmm_%s.npz' % name))
    X, y, Xs = g['X'], g['y'], g['Xs']
    kern = g3.SE(X) if str(g['kind']) == 'SE' else g3.OU(X)
    gp = g3.GaussianProcess(space=Xs, location=g3.Zero(), kernel=kern)
    gp.observed(X, y)
    kn = kern.name
    p = _params(gp, **{kn + '_var': float(g['var']), kn + '_rate': g['rate'], 'Noise_var': float(g['noise'])})
    lp = gp.logp(p)
    assert abs(lp - float(g['logp'])) <= 1e-9 * abs(float(g['logp']))
    pred = gp.predict(p, var=True, cov=True)
    np.testing.assert_allclose(pred.mean, g['mean'], atol=1e-8)
    np.testing.assert_allclose(pred.variance, np.maximum(g['variance'], 0), atol=1e-8)
    np.testing.assert_allclose(pred.covariance, g['covariance'], atol=1e-8)
    np.testing.assert_allclose(pred.std, np.sqrt(np.maximum(g['variance'], 0)), atol=1e-7)
    # the factor itself
    np.testing.assert_allclose(gp.cholesky(p, space=X, prior=True, noise=True), g['L'], atol=1e-9)


PROCS = {
    'gp_se_bias': lambda g3, X: (g3.GaussianProcess, dict(location=g3.Bias(), kernel=g3.SE(X)),
                                 dict(SE_var=1.1, SE_rate=[0.9, 1.2], Noise_var=0.1, Bias_Bias=0.3)),
    'gp_mat52cos_zero': lambda g3, X: (g3.GaussianProcess, dict(location=g3.Zero(), kernel=g3.MAT52(X) + g3.COS(X)),
                                       dict(MAT52_var=1.0, MAT52_rate=[0.9, 1.2], COS_var=0.5,
                                            COS_freq=[0.125, 0.125], Noise_var=0.1)),
    'gp_se_linear': lambda g3, X: (g3.GaussianProcess, dict(location=g3.Linear(), kernel=g3.SE(X)),
                                   dict(SE_var=0.7, SE_rate=[0.9, 1.2], Noise_var=0.05, Linear_Constant=0.1,
                                        Linear_Coeff=[0.02, -0.03])),
    'wgp_boxcox': lambda g3, X: (g3.WarpedGaussianProcess,
                                 dict(location=g3.Bias(), kernel=g3.SE(X), mapping=g3.BoxCoxLinear()),
                                 dict(SE_var=1.0, SE_rate=[0.9, 1.2], Noise_var=0.1, Bias_Bias=0.2,
                                      BoxCoxLinear_shift=1.0, BoxCoxLinear_scale=1.0, BoxCoxLinear_power=1.2)),
    'wgp_arcsinh': lambda g3, X: (g3.WarpedGaussianProcess,
                                  dict(location=g3.Zero(), kernel=g3.SE(X), mapping=g3.ArcsinhLinear()),
                                  dict(SE_var=1.0, SE_rate=[0.9, 1.2], Noise_var=0.1, ArcsinhLinear_shift=0.1,
                                       ArcsinhLinear_scale=0.8)),
    'wgp_logshift': lambda g3, X: (g3.WarpedGaussianProcess,
                                   dict(location=g3.Zero(), kernel=g3.OU(X), mapping=g3.LogShifted()),
                                   dict(OU_var=1.0, OU_rate=[0.9, 1.2], Noise_var=0.1, LogShifted_shift=-0.5)),
    'wgp_linear': lambda g3, X: (g3.WarpedGaussianProcess,
                                 dict(location=g3.Zero(), kernel=g3.RQ(X), mapping=g3.LinearMapping()),
                                 dict(RQ_var=1.0, RQ_rate=[0.9, 1.2], RQ_alpha=1.5, Noise_var=0.1,
                                      LinearMapping_shift=0.2, LinearMapping_scale=1.5)),
}


@pytest.mark.parametrize('name', sorted(PROCS))
def test_process_matches_oracle_fixture(golden_dir, name):
    import g3py_amd as g3
    g = np.load(os.path.join(golden_dir, 'oracle_process.npz'))
    X, Xs, Z, y = g['X'], g['Xs'], g['Z'], g[name + '_y']
    cls, kw, nat = PROCS[name](g3, X)
    gp = cls(space=Xs, **kw)
    gp.observed(X, y)
    p = _params(gp, **nat)
    ref = float(g[name + '_logp'])
    assert abs(gp.logp(p) - ref) <= 1e-8 * abs(ref)
    assert abs(gp.loglike(p) - ref) <= 1e-8 * abs(ref)
    scale = max(1.0, np.abs(g[name + '_mean_n0']).max())
    for noise in (False, True):
        s = '_n%d' % noise
        np.testing.assert_allclose(gp.mean(p, noise=noise), g[name + '_mean' + s], atol=1e-8 * scale)
        np.testing.assert_allclose(gp.median(p, noise=noise), g[name + '_median' + s], atol=1e-8 * scale)
        np.testing.assert_allclose(gp.variance(p, noise=noise), g[name + '_var' + s], atol=1e-7 * scale ** 2)
        np.testing.assert_allclose(gp.kernel(p, noise=noise), g[name + '_cov' + s], atol=1e-8)
        np.testing.assert_allclose(gp.quantiler(p, q=0.975, noise=noise), g[name + '_q975' + s], atol=1e-7 * scale)
        got = gp.sampler(p, samples=Z.shape[1], noise=noise, rand=Z)
        np.testing.assert_allclose(got, g[name + '_samples' + s], atol=2e-6 * scale)
    np.testing.assert_allclose(gp.mean(p, prior=True), g[name + '_prior_mean'], atol=1e-9 * scale)
    np.testing.assert_allclose(gp.variance(p, prior=True, noise=True), g[name + '_prior_var_n1'], atol=1e-8 * scale ** 2)
    lpred = gp.logpredictive(p, vector=g[name + '_median_n0'])
    assert abs(lpred - float(g[name + '_logpred'])) <= 1e-7 * abs(float(g[name + '_logpred']))


def test_api_surface_and_dispatch(golden_dir):
    """names, defaults, method registry and predict() keys of stochastic.py:385-513"""
    import g3py_amd as g3
    g = np.load(os.path.join(golden_dir, 'gpmm_se_d1.npz'))
    X, y, Xs = g['X'], g['y'], g['Xs']
    gp = g3.GP(space=Xs[:, 0], location=g3.Bias(), kernel=g3.SE(X))     # 1-D space is reshaped to a column
    assert gp.space.shape == (len(Xs), 1) and not gp.is_observed
    assert [v.key for v in gp.model.vars] == ['GP_Bias_Bias', 'GP_SE_var_log_', 'GP_SE_rate_log_', 'GP_Noise_var_log_']
    pr = gp.predict()                                 # unobserved: falls back to the prior (stochastic.py:395-396,475-476)
    assert set(pr) == {'mean', 'std'} and 'prior_mean' in gp.compiles and 'prior_std' in gp.compiles
    gp.observed(X, y)
    d = gp.params_default                            # defaults of kernels.py:33-40, metrics.py:104-108, means.py:133-134
    assert np.isclose(d['GP_SE_var_log_'], np.log(y.var())) and np.isclose(d['GP_Bias_Bias'], y.mean())
    assert np.isclose(d['GP_SE_rate_log_'][0], np.log(0.5 / np.abs(X[1:] - X[:-1]).mean()))
    p = gp.params
    out = gp.predict(p, var=True, cov=True, median=True, quantiles=True, quantiles_noise=True, samples=3, distribution=True)
    assert set(out) == {'mean', 'variance', 'std', 'covariance', 'median', 'quantile_up', 'quantile_down',
                        'noise_std', 'noise_up', 'noise_down', 'samples', 'logpredictive'}
    assert out.samples.shape == (len(Xs), 3) and np.all(out.quantile_up >= out.quantile_down)
    assert np.all(out.noise_std >= out.std - 1e-12) and np.isfinite(out.logpredictive(out.mean))
    for k in ('posterior_mean', 'posterior_std', 'posterior_kernel_sd_noise', 'posterior_location', 'posterior_mapping'):
        assert k in gp.compiles, k
    # array=True goes through the dict<->array bijection (models.py:143-155)
    a = gp.active.dict_to_array(p)
    assert np.isclose(gp.logp(a, array=True), gp.logp(p)) and 'array_posterior_logp' in gp.compiles
    assert gp.executed['posterior_logp'] >= 1
    np.testing.assert_allclose(gp.logp_chain(np.stack([a, a])), [gp.logp(p)] * 2)
    with pytest.raises(KeyError):
        gp.logp({'GP_SE_var_log_': 0.0})             # filter_params needs every model variable
    # prior logp: Flat priors -> 0; FlatExp variables below 1e-6 -> -inf (hypers/__init__.py:199-200)
    assert gp.logp(p, prior=True) == 0
    q = dict(p); q['GP_Noise_var_log_'] = np.log(1e-7)
    assert gp.logp(q) == -np.inf
    # non-finite observation -> the -1e30 sentinel (gaussian.py:238-241)
    y2 = y.copy(); y2[3] = np.inf
    assert gp.logp(p, outputs=y2) == np.float32(-1e30)
    assert not hasattr(g3.WGP(space=Xs, location=g3.Zero(), kernel=g3.SE(X), mapping=g3.BoxCoxLinear()), 'covariance')


def test_jitter_path_through_process():
    """duplicate inputs without noise make K singular: the reference's jitter schedule rescues it"""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(0)
    X = np.repeat(rng.uniform(0, 3, (40, 1)), 2, axis=0)
    y = np.sin(X[:, 0])
    gp = g3.GaussianProcess(space=X, location=g3.Zero(), kernel=g3.SE(X), noisy=False)
    gp.observed(X, y)
    p = _params(gp, SE_var=1.0, SE_rate=[1.0])
    ref = orc.GP(('SE', 1.0, np.array([1.0]), None), None).logp(X, y)
    got = gp.logp(p)
    assert gp._cache['stats']['tries'] >= 1
    # identical jitter schedule; measured 2e-12 here and <= 1e-11 up to N=3000 (profiles/r02_jitter_accuracy.txt:
    # the difference to LAPACK comes from the factor's rounding, not from the inverse-based panel solves)
    assert abs(got - ref) <= 1e-8 * abs(ref)


def test_fp32_process_close_to_fp64(golden_dir):
    import g3py_amd as g3
    g = np.load(os.path.join(golden_dir, 'gpmm_se_d3.npz'))
    X, y, Xs = g['X'], g['y'], g['Xs']
    gp = g3.GaussianProcess(space=Xs, location=g3.Zero(), kernel=g3.SE(X), dtype=np.float32)
    gp.observed(X, y)
    p = _params(gp, SE_var=float(g['var']), SE_rate=g['rate'], Noise_var=float(g['noise']))
    lp = gp.logp(p)
    assert lp.dtype == np.float32 and abs(lp - float(g['logp'])) <= 1e-4 * abs(float(g['logp']))
    np.testing.assert_allclose(gp.mean(p), g['mean'], atol=2e-3)


def test_cross_mean_and_wgp_fp32_draws():
    """th_cross_mean (gaussian.py:99-112) and a float32 warped GP with posterior draws (the shape of
    BASELINE config 5, at a size the CPU oracle handles)"""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(12)
    N, d, M = 1500, 16, 96
    X = rng.uniform(0, N ** (1 / d), (N, d))
    Xs = rng.uniform(0, N ** (1 / d), (M, d))
    f = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(N)
    y = f - f.min() + 1.0
    r = np.ones(d)
    gp = g3.GaussianProcess(space=Xs, location=g3.Zero(), kernel=g3.SE(X))
    gp.observed(X, y)
    p = _params(gp, SE_var=1.0, SE_rate=r, Noise_var=0.1)
    ref = orc.GP(('SE', 1.0, r, None), 0.1)
    np.testing.assert_allclose(gp.cross_mean(p), ref.mean(Xs, X, y), atol=1e-8)
    k2 = g3.OU(X, var=0.7, metric=g3.ARD_L1(X, rate=r))
    got = gp.cross_mean(p, cross_kernel=k2)
    K = ref.prior_kernel(X, True)
    want = orc.kernel_cov(('OU', 0.7, r, None), Xs, X).dot(np.linalg.solve(K, y))
    np.testing.assert_allclose(got, want, atol=1e-8)
    # float32 warped GP with draws
    wgp = g3.WGP(space=Xs, location=g3.Zero(), kernel=g3.SE(X), mapping=g3.BoxCoxLinear(), dtype=np.float32)
    wgp.observed(X, y)
    pw = _params(wgp, SE_var=1.0, SE_rate=r, Noise_var=0.1, BoxCoxLinear_shift=1.0, BoxCoxLinear_scale=1.0,
                 BoxCoxLinear_power=1.2)
    o = orc.GP(('SE', 1.0, r, None), 0.1, mapping=('BoxCoxLinear', 1.0, 1.0, 1.2))
    lp = wgp.logp(pw)
    assert lp.dtype == np.float32 and abs(lp - o.logp(X, y)) <= 1e-4 * abs(o.logp(X, y))
    np.testing.assert_allclose(wgp.mean(pw), o.mean(Xs, X, y), rtol=2e-3, atol=2e-3)
    Z = rng.standard_normal((M, 4))
    draws = wgp.sampler(pw, samples=4, rand=Z)
    np.testing.assert_allclose(draws, o.sampler(Xs, X, y, Z), rtol=5e-3, atol=5e-3)


# ------------------------------------------------------------------ dlogp (stochastic.py:308-309)
@pytest.mark.parametrize('name', ['se_d1', 'se_d3', 'se_d4', 'ou_d2'])
def test_dlogp_matches_reference_gpmm(golden_dir, name):
    """gradient in transformed (log) space against finite differences of the reference
    prototype's own NLL (fixture made by oracle/gen_golden.py from sandbox/gpmm.py:128-130)"""
    import g3py_amd as g3
    g = np.load(os.path.join(golden_dir, 'gpmm_%s.npz' % name))
    X, y, Xs = g['X'], g['y'], g['Xs']
    kern = g3.SE(X) if str(g['kind']) == 'SE' else g3.OU(X)
    gp = g3.GaussianProcess(space=Xs, location=g3.Zero(), kernel=kern)
    gp.observed(X, y)
    kn = kern.name
    p = _params(gp, **{kn + '_var': float(g['var']), kn + '_rate': g['rate'], 'Noise_var': float(g['noise'])})
    got = gp.dlogp(p)
    assert [v.key for v in gp.model.vars] == ['GP_%s_var_log_' % kn, 'GP_%s_rate_log_' % kn, 'GP_Noise_var_log_']
    d = len(g['rate'])
    want = np.concatenate([g['dlogp_log'][1:2], g['dlogp_log'][2:2 + d], g['dlogp_log'][0:1]])
    np.testing.assert_allclose(got, want, rtol=1e-7, atol=1e-7)
    np.testing.assert_array_equal(gp.dlogp(gp.active.dict_to_array(p), array=True), got)
    assert 'posterior_dlogp' in gp.compiles and 'array_posterior_dlogp' in gp.compiles


def _oracle_grad(gp, g, name, nat):
    """the oracle's natural-space gradient pieces -> the product's flat transformed-space vector"""
    leaves = [k.split('_')[0] for k in nat if k.split('_')[0] not in
              ('Noise', 'Bias', 'Linear', 'BoxCoxLinear', 'ArcsinhLinear', 'LogShifted', 'LinearMapping')]
    leaves = list(dict.fromkeys(leaves)) + ['Noise']
    kern, mean, mapp = list(g[name + '_dlogp_kernel']), list(g[name + '_dlogp_mean']), list(g[name + '_dlogp_mapping'])
    by_key = {}
    order = {'SE': ['var', 'rate'], 'OU': ['var', 'rate'], 'MAT52': ['var', 'rate'], 'RQ': ['var', 'alpha', 'rate'],
             'COS': ['var', 'freq'], 'Noise': ['var']}
    d = gp.inputs.shape[1]
    for lf in leaves:
        for pn in order[lf]:
            n = d if pn in ('rate', 'freq') else 1
            by_key[lf + '_' + pn] = np.array([kern.pop(0) for _ in range(n)])
    for k in nat:
        if k.startswith(('Bias_', 'Linear_')):
            n = d if k == 'Linear_Coeff' else 1
            by_key[k] = np.array([mean.pop(0) for _ in range(n)])
        elif k.split('_')[0] in ('BoxCoxLinear', 'ArcsinhLinear', 'LogShifted', 'LinearMapping'):
            by_key[k] = np.array([mapp.pop(0)])
    assert not kern and not mean and not mapp
    out = []
    for v in gp.model.vars:
        k = v.name[len(gp.name) + 1:]
        val = np.atleast_1d(np.asarray(nat[k], dtype=float)) * np.ones(max(v.size, 1))
        out.append(by_key[k] * (val if v.positive else 1.0))
    return np.concatenate(out)


@pytest.mark.parametrize('name', sorted(PROCS))
def test_dlogp_matches_oracle_fixture(golden_dir, name):
    import g3py_amd as g3
    g = np.load(os.path.join(golden_dir, 'oracle_process.npz'))
    X, Xs, y = g['X'], g['Xs'], g[name + '_y']
    cls, kw, nat = PROCS[name](g3, X)
    gp = cls(space=Xs, **kw)
    gp.observed(X, y)
    p = _params(gp, **nat)
    got = gp.dlogp(p)
    want = _oracle_grad(gp, g, name, nat)
    np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-8 * max(1.0, np.abs(want).max()))
    # directional finite difference of the device logp itself
    rng = np.random.default_rng(1)
    v = rng.standard_normal(len(got))
    a = gp.active.dict_to_array(p)
    h = 1e-5
    fd = (float(gp.logp(a + h * v, array=True)) - float(gp.logp(a - h * v, array=True))) / (2 * h)
    assert abs(fd - got.dot(v)) <= 1e-5 * max(1.0, abs(fd))


def test_dlogp_edge_branches():
    """prior=True, the -1e30 sentinel, the -inf Jacobian, an L2 potential and the jitter path"""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(2)
    X = rng.uniform(0, 3, (60, 2))
    y = np.sin(X.sum(1))
    k = g3.SE(X)
    k.set_potential('rate', 'L2', 0.5)
    gp = g3.GaussianProcess(space=X, location=g3.Bias(), kernel=k)
    gp.observed(X, y)
    p = _params(gp, SE_var=1.2, SE_rate=[0.8, 1.1], Noise_var=0.1, Bias_Bias=0.1)
    names = [v.key for v in gp.model.vars]
    rate = np.array([0.8, 1.1])
    pot = np.zeros(len(gp.dlogp(p)))
    i0 = gp.active.dict_to_array({n: (np.arange(len(np.atleast_1d(p[n]))) + 100 * j) for j, n in enumerate(names)})
    ridx = [i for i, t in enumerate(i0) if 100 * names.index('GP_SE_rate_log_') <= t < 100 * names.index('GP_SE_rate_log_') + 50]
    pot[ridx] = -2 * 0.5 * rate * rate               # d(-c sum h^2)/d log h
    np.testing.assert_allclose(gp.dlogp(p, prior=True), pot, atol=1e-12)
    ref = orc.GP(('SE', 1.2, rate, None), 0.1, ('Bias', 0.1)).dlogp_natural(X, y)
    kern = {(l, pn, kk): v for l, pn, kk, v in ref['kernel']}
    want = np.concatenate([[ref['mean'][0][2]], [kern[(0, 'var', None)] * 1.2],
                           [kern[(0, 'rate', i)] * rate[i] for i in range(2)], [kern[(1, 'var', None)] * 0.1]]) + pot
    np.testing.assert_allclose(gp.dlogp(p), want, rtol=1e-8, atol=1e-8)
    y2 = y.copy(); y2[5] = np.nan
    np.testing.assert_allclose(gp.dlogp(p, outputs=y2), pot, atol=1e-12)     # constant -1e30 branch
    q = dict(p); q['GP_Noise_var_log_'] = np.log(1e-7)
    assert np.all(np.isfinite(gp.dlogp(q)))                                   # tt_to_num on the result
    # jitter path: duplicate inputs, no noise; CholeskyRobust.grad re-uses the jittered factor
    Xd = np.repeat(rng.uniform(0, 3, (30, 1)), 2, axis=0)
    yd = np.sin(Xd[:, 0])
    gj = g3.GaussianProcess(space=Xd, location=g3.Zero(), kernel=g3.SE(Xd), noisy=False)
    gj.observed(Xd, yd)
    pj = _params(gj, SE_var=1.0, SE_rate=[1.0])
    got = gj.dlogp(pj)
    assert gj._cache['stats']['tries'] >= 1
    rj = orc.GP(('SE', 1.0, np.array([1.0]), None), None).dlogp_natural(Xd, yd)
    wantj = np.array([v for *_, v in rj['kernel']])
    np.testing.assert_allclose(got, wantj, rtol=1e-4, atol=1e-4 * np.abs(wantj).max())


# ------------------------------------------------------------------ batched chains (stochastic.py:515-520)
@pytest.mark.parametrize('name', ['gp_se_bias', 'gp_mat52cos_zero', 'wgp_boxcox', 'wgp_linear'])
def test_logp_chain_batched_equals_row_by_row(golden_dir, name):
    """g3_gp_factor_batched: every member of the batch equals the single evaluation of the same row
    (and therefore the oracle fixture), including a row that needs the jitter schedule's neighbours"""
    import g3py_amd as g3
    g = np.load(os.path.join(golden_dir, 'oracle_process.npz'))
    X, Xs, y = g['X'], g['Xs'], g[name + '_y']
    cls, kw, nat = PROCS[name](g3, X)
    gp = cls(space=Xs, **kw)
    gp.observed(X, y)
    a0 = gp.active.dict_to_array(_params(gp, **nat))
    rng = np.random.default_rng(3)
    chain = a0 + 0.2 * rng.standard_normal((37, len(a0)))
    chain[0] = a0
    want = np.array([gp.logp(r, array=True) for r in chain])
    got = gp.logp_chain(chain, batch=16)                       # 3 sweeps: 16 + 16 + 5 members
    np.testing.assert_allclose(got, want, rtol=1e-10)
    assert abs(got[0] - float(g[name + '_logp'])) <= 1e-8 * abs(float(g[name + '_logp']))
    np.testing.assert_allclose(gp.logp_chain(chain), want, rtol=1e-10)      # one sweep


def test_logp_chain_batched_edge_members():
    """members that hit the -1e30 sentinel, the -inf Jacobian and the jitter schedule inside one batch"""
    import g3py_amd as g3
    rng = np.random.default_rng(4)
    X = np.repeat(rng.uniform(0, 3, (40, 1)), 2, axis=0)      # duplicated inputs: singular without noise
    y = np.sin(X[:, 0]) + 2.0
    gp = g3.WarpedGaussianProcess(space=X, location=g3.Zero(), kernel=g3.SE(X), mapping=g3.LogShifted())
    gp.observed(X, y)
    base = _params(gp, SE_var=1.0, SE_rate=[1.0], Noise_var=0.1, LogShifted_shift=0.0)
    rows = []
    for noise, shift in [(0.1, 0.0), (1e-14, 0.0), (0.1, 5.0), (1e-7, 0.0), (0.3, -1.0)]:
        p = dict(base)
        p['WGP_Noise_var_log_'] = np.log(noise)
        p['WGP_LogShifted_shift'] = np.asarray(shift)
        rows.append(gp.active.dict_to_array(p))
    chain = np.stack(rows)
    want = np.array([gp.logp(r, array=True) for r in chain])
    got = gp.logp_chain(chain)
    assert want[2] == np.float32(-1e30) and want[3] == -np.inf     # log of a negative number; exp(x) <= 1e-6
    np.testing.assert_array_equal(got[[2, 3]], want[[2, 3]])
    np.testing.assert_allclose(got[[0, 1, 4]], want[[0, 1, 4]], rtol=1e-9)


# ------------------------------------------------------------------ Student-t process (studentT.py)
@pytest.mark.parametrize('warped', [False, True])
def test_student_t_process_matches_oracle(golden_dir, warped):
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    g = np.load(os.path.join(golden_dir, 'oracle_process.npz'))
    X, Xs = g['X'], g['Xs']
    y = g['wgp_boxcox_y'] if warped else g['gp_se_bias_y']
    r = np.array([0.9, 1.2])
    if warped:
        tp = g3.WTP(space=Xs, location=g3.Bias(), kernel=g3.SE(X), mapping=g3.BoxCoxLinear())
        nat = dict(SE_var=1.0, SE_rate=r, Noise_var=0.1, Bias_Bias=0.2, Freedom_degree=3.5,
                   BoxCoxLinear_shift=1.0, BoxCoxLinear_scale=1.0, BoxCoxLinear_power=1.2)
        ref = orc.TP(('SE', 1.0, r, None), 3.5, 0.1, ('Bias', 0.2), ('BoxCoxLinear', 1.0, 1.0, 1.2))
    else:
        tp = g3.TP(space=Xs, location=g3.Bias(), kernel=g3.SE(X))
        nat = dict(SE_var=1.1, SE_rate=r, Noise_var=0.1, Bias_Bias=0.3, Freedom_degree=3.5)
        ref = orc.TP(('SE', 1.1, r, None), 3.5, 0.1, ('Bias', 0.3))
    tp.observed(X, y)
    assert tp.name == ('WTP' if warped else 'TP') and tp.name + '_Freedom_degree_log_' in [v.key for v in tp.model.vars]
    assert np.isclose(np.exp(tp.params_default[tp.name + '_Freedom_degree_log_']), len(y))
    p = _params(tp, **nat)
    lp = ref.logp(X, y)
    assert abs(tp.logp(p) - lp) <= 1e-8 * abs(lp)
    assert tp.freedom(p) == 5.5 + len(y) and tp.freedom(p, prior=True) == 5.5
    np.testing.assert_allclose(tp.mean(p), ref.mean(Xs, X, y), atol=1e-8)
    np.testing.assert_allclose(tp.variance(p), ref.variance(Xs, X, y), atol=1e-8)
    np.testing.assert_allclose(tp.variance(p, noise=True), ref.variance(Xs, X, y, noise=True), atol=1e-8)
    np.testing.assert_allclose(tp.quantiler(p, q=0.9), ref.quantiler(Xs, X, y, 0.9), atol=1e-7)
    if not warped:
        np.testing.assert_allclose(tp.covariance(p), ref.covariance(Xs, X, y), atol=1e-8)
        Z = g['Z']
        np.testing.assert_allclose(tp.sampler(p, samples=Z.shape[1], rand=Z), ref.sampler(Xs, X, y, Z), atol=2e-6)
    else:
        assert not hasattr(tp, 'covariance')
    assert tp.sampler(p, samples=3).shape == (len(Xs), 3)
    out = tp.predict(p, var=True, quantiles=True)
    assert np.all(out.quantile_up >= out.quantile_down) and np.all(np.isfinite(out.std))


def test_scores_harness_matches_oracle(golden_dir):
    """StochasticProcess.scores (models.py:449-469): the caller of mean / variance / median / logpredictive"""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    g = np.load(os.path.join(golden_dir, 'oracle_process.npz'))
    X, Xs, y = g['X'], g['Xs'], g['gp_se_bias_y']
    r = np.array([0.9, 1.2])
    hidden = np.sin(Xs.sum(1) / np.sqrt(2))
    gp = g3.GaussianProcess(space=Xs, location=g3.Bias(), kernel=g3.SE(X), hidden=hidden)
    gp.observed(X, y)
    p = _params(gp, SE_var=1.1, SE_rate=r, Noise_var=0.1, Bias_Bias=0.3)
    sc = gp.scores(p, logp=True, logpred=True, variance=True, median=True)
    ref = orc.GP(('SE', 1.1, r, None), 0.1, ('Bias', 0.3))
    m, v = ref.mean(Xs, X, y), ref.variance(Xs, X, y)
    assert set(sc) == {'_l1', '_l2', '_mse', '_rmse', '_median_l1', '_median_l2', '_logp', '_loglike', '_logprior', '_nlpd'}
    np.testing.assert_allclose(sc['_l1'], np.mean(np.abs(m - hidden)), rtol=1e-8)
    np.testing.assert_allclose(sc['_l2'], np.mean((m - hidden) ** 2), rtol=1e-8)
    np.testing.assert_allclose(sc['_mse'], np.mean((m - hidden) ** 2 + v), rtol=1e-8)
    np.testing.assert_allclose(sc['_rmse'], np.sqrt(np.mean((m - hidden) ** 2 + v)), rtol=1e-8)
    np.testing.assert_allclose(sc['_logp'], ref.logp(X, y), rtol=1e-9)
    np.testing.assert_allclose(sc['_nlpd'], -ref.logpredictive(hidden, Xs, X, y) / len(hidden), rtol=1e-7)
    assert sc['_logprior'] == 0 and gp.sample(p, samples=2).shape == (len(Xs), 2)
    # error statistics against a supplied vector (stochastic.py:315-326)
    np.testing.assert_allclose(gp.error_l1(p, vector=hidden), sc['_l1'], rtol=1e-12)
    np.testing.assert_allclose(gp.error_l2(p, vector=hidden), sc['_l2'], rtol=1e-12)
    dv = np.abs(y[:len(y)] - y[::-1])
    np.testing.assert_allclose(gp.error_mse(p, vector=y[::-1]), np.mean(dv) ** 2 + np.var(dv), rtol=1e-12)


def test_student_t_dlogp_matches_oracle(golden_dir):
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    g = np.load(os.path.join(golden_dir, 'oracle_process.npz'))
    X, Xs, y = g['X'], g['Xs'], g['gp_se_bias_y']
    r = np.array([0.9, 1.2])
    tp = g3.TP(space=Xs, location=g3.Bias(), kernel=g3.SE(X))
    tp.observed(X, y)
    nat = dict(SE_var=1.1, SE_rate=r, Noise_var=0.1, Bias_Bias=0.3, Freedom_degree=3.5)
    p = _params(tp, **nat)
    got = tp.dlogp(p)
    ref = orc.TP(('SE', 1.1, r, None), 3.5, 0.1, ('Bias', 0.3))
    gr = ref.dlogp_natural(X, y)
    kern = {(l, pn, k): v for l, pn, k, v in gr['kernel']}
    want = {'TP_Bias_Bias': [gr['mean'][0][2]], 'TP_SE_var_log_': [kern[(0, 'var', None)] * 1.1],
            'TP_SE_rate_log_': [kern[(0, 'rate', i)] * r[i] for i in range(2)],
            'TP_Noise_var_log_': [kern[(1, 'var', None)] * 0.1],
            'TP_Freedom_degree_log_': [ref.dlogp_degree(X, y) * 3.5]}
    flat = np.concatenate([want[v.key] for v in tp.model.vars])
    np.testing.assert_allclose(got, flat, rtol=1e-8, atol=1e-8)
    a = tp.active.dict_to_array(p)
    v = np.random.default_rng(0).standard_normal(len(a))
    h = 1e-5
    fd = (float(tp.logp(a + h * v, array=True)) - float(tp.logp(a - h * v, array=True))) / (2 * h)
    assert abs(fd - got.dot(v)) <= 1e-5 * max(1.0, abs(fd))


def test_fixed_chain_averages(golden_dir):
    """fixed_logp / fixed_loglike / fixed_logprior / fixed_dlogp (stochastic.py:522-564) over a chain whose
    noise column is pinned: the batched sweep equals the row-by-row evaluation"""
    import g3py_amd as g3
    g = np.load(os.path.join(golden_dir, 'oracle_process.npz'))
    X, Xs, y = g['X'], g['Xs'], g['gp_se_bias_y']
    gp = g3.GaussianProcess(space=Xs, location=g3.Bias(), kernel=g3.SE(X))
    gp.observed(X, y)
    a0 = gp.active.dict_to_array(_params(gp, SE_var=1.1, SE_rate=[0.9, 1.2], Noise_var=0.1, Bias_Bias=0.3))
    rng = np.random.default_rng(6)
    chain = a0 + 0.1 * rng.standard_normal((9, len(a0)))
    gp.active.fix_vars(chain, ['GP_Noise_var_log_'])
    assert gp.active.fixed_dims == [4] and gp.active.sampling_dims == [0, 1, 2, 3]
    sp = gp.active.sampling_params(a0) + 0.05
    rows = chain.copy()
    rows[:, :4] = sp
    want = np.array([gp.logp(r, array=True) for r in rows])
    np.testing.assert_allclose(gp.fixed_logp(sp, return_array=True), want, rtol=1e-10)
    assert abs(gp.fixed_logp(sp) - want.mean()) <= 1e-10 * abs(want.mean())
    np.testing.assert_allclose(gp.fixed_loglike(sp, return_array=True), want, rtol=1e-10)      # flat priors
    assert gp.fixed_logprior(sp) == 0
    gd = gp.fixed_dlogp(sp)
    assert gd.shape == (4,)
    np.testing.assert_allclose(gd, np.mean([gp.dlogp(r, array=True)[:4] for r in rows], axis=0), rtol=1e-10)
    d = gp.active.dict_from_sampling_array(sp)
    np.testing.assert_allclose(gp.active.dict_to_array(d)[:4], sp)
    gp.active.fix_vars()
    assert gp.active.fixed_chain is None and gp.active.sampling_dims == list(range(5))


# ------------------------------------------------------------------ transports (hypers/transports.py)
def test_tkernel_transport_matches_oracle():
    """TKernel on the device (block schedule: V a + chol(Kss - V V^T) z) against the literal joint-covariance
    restatement, plus composition with a location and a warping"""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(8)
    X, Xs = rng.uniform(0, 3, (300, 2)), rng.uniform(0, 3, (40, 2))
    y = np.sin(X.sum(1)) + 0.05 * rng.standard_normal(300)
    r = np.array([0.8, 1.1])
    with g3.Model('t') as model:
        tk = g3.TKernel(g3.SE(X), noisy=True)
        tl = g3.TLocation(g3.Bias(X))
        tm = g3.TMapping(g3.ArcsinhLinear(y))
        for t in (tk, tl, tm):
            t.check_dims(X)
            t.check_hypers('T_')
    names = [v.name for v in model.vars]
    assert names == ['T_SE_var', 'T_SE_rate', 'T_NoiseSE_var', 'T_Bias_Bias', 'T_ArcsinhLinear_shift', 'T_ArcsinhLinear_scale']
    values = {'T_SE_var': 1.2, 'T_SE_rate': r, 'T_NoiseSE_var': 0.1, 'T_Bias_Bias': 0.3, 'T_ArcsinhLinear_shift': 0.1,
              'T_ArcsinhLinear_scale': 0.9}
    ref = orc.TKernelOracle(('SE', 1.2, r, None), 0.1)
    z = rng.standard_normal(40)
    v = rng.standard_normal(300)
    np.testing.assert_allclose(tk(X, v, noise=True, values=values), ref(X, v, noise=True), atol=1e-9)
    np.testing.assert_allclose(tk.inv(X, y, noise=True, values=values), ref.inv(X, y, noise=True), atol=1e-8)
    np.testing.assert_allclose(tk.diag(X, v, noise=True, values=values), ref.diag(X, v, noise=True), atol=1e-12)
    assert abs(tk.logdet_dinv(X, y, values=values) - ref.logdet_dinv(X, y)) <= 1e-9 * abs(ref.logdet_dinv(X, y))
    for noise_pred in (False, True):
        got = tk.posterior(Xs, z, X, y, noise_pred=noise_pred, noise_obs=True, values=values)
        np.testing.assert_allclose(got, ref.posterior(Xs, z, X, y, noise_pred=noise_pred, noise_obs=True), atol=1e-8)
    # composition: y = arcsinh-warp(bias + chol(K) x)
    T = tm @ (tl @ tk)
    yy = np.sinh((y - 0.1) / 0.9)
    mp = orc.Mapping(('ArcsinhLinear', 0.1, 0.9))
    x_ref = ref.inv(X, mp.inv(yy) - 0.3, noise=True)
    np.testing.assert_allclose(T.inv(X, yy, noise=True, values=values), x_ref, atol=1e-8)
    np.testing.assert_allclose(T(X, x_ref, noise=True, values=values), yy, atol=1e-8)
    ld = T.logdet_dinv(X, yy, values=values)
    assert abs(ld - (ref.logdet_dinv(X, y) + mp.logdet_dinv(yy))) <= 1e-9 * abs(ld)
    want = mp(0.3 + ref.posterior(Xs, z, X, mp.inv(yy) - 0.3))
    np.testing.assert_allclose(T.posterior(Xs, z, X, yy, values=values), want, atol=1e-8)


def test_transport_gaussian_process_is_the_warped_gp(golden_dir):
    """TGP with T = TMapping @ TLocation @ TKernel(noisy) is the warped GP as a push-forward
    (transport.py:17-246): same logp, median = transport of 0, draws = transport of normal vectors"""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    g = np.load(os.path.join(golden_dir, 'oracle_process.npz'))
    X, Xs, Z, y = g['X'], g['Xs'], g['Z'], g['wgp_arcsinh_y']
    r = np.array([0.9, 1.2])
    T = g3.TMapping(g3.ArcsinhLinear(y)) @ (g3.TLocation(g3.Bias(X)) @ g3.TKernel(g3.SE(X), noisy=True))
    tgp = g3.TGP(space=Xs, transport=T)
    tgp.observed(X, y)
    p = _params(tgp, SE_var=1.0, SE_rate=r, NoiseSE_var=0.1, Bias_Bias=0.0, ArcsinhLinear_shift=0.1, ArcsinhLinear_scale=0.8)
    ref = float(g['wgp_arcsinh_logp'])                      # the same model as oracle fixture `wgp_arcsinh` (Zero mean)
    assert abs(tgp.logp(p) - ref) <= 1e-8 * abs(ref) and abs(tgp.loglike(p) - ref) <= 1e-8 * abs(ref)
    med = tgp.transport(p, vector=np.zeros(len(Xs)))
    np.testing.assert_allclose(med, g['wgp_arcsinh_median_n0'], atol=1e-8)
    draws = tgp.sampler(p, rand=Z)
    np.testing.assert_allclose(draws, g['wgp_arcsinh_samples_n0'], atol=2e-6)
    o = orc.GP(('SE', 1.0, r, None), 0.1, ('Zero',), ('ArcsinhLinear', 0.1, 0.8))
    np.testing.assert_allclose(tgp.transport(p, vector=Z[:, 0], prior=True),
                               o.map(np.linalg.cholesky(o.prior_kernel(Xs, False)).dot(Z[:, 0])), atol=1e-7)
    q = tgp.quantiler(p, q=0.9, simulations=draws)
    assert q.shape == (len(Xs),) and np.all(q >= np.nanpercentile(draws, 10, axis=1))
    assert tgp.mean(p, simulations=draws).shape == (len(Xs),) and tgp.std(p, simulations=4).shape == (len(Xs),)
    y2 = y.copy(); y2[0] = np.nan
    assert tgp.logp(p, outputs=y2) == np.float32(-1e30)
    assert 'posterior_transport' in tgp.compiles and 'prior_transport' in tgp.compiles


def test_tscale_and_transport_density_match_oracle():
    """TScale (transports.py:165-181) composed with a noisy TKernel: map, inverse, log-determinant and
    the transport log-density TransportGaussianDistribution.logp_t (transport.py:220-243) against the
    literal restatement; TTriangular is the reference's constructor-only shell"""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(21)
    X = rng.uniform(0, 3, (260, 2))
    y = 1.7 * (np.sin(X.sum(1)) + 0.05 * rng.standard_normal(260))
    r = np.array([0.7, 1.3])
    T = g3.TScale(g3.Bias(X)) @ g3.TKernel(g3.SE(X), noisy=True)
    tgp = g3.TGP(space=X[:5], transport=T)
    tgp.observed(X, y)
    p = _params(tgp, SE_var=1.1, SE_rate=r, NoiseSE_var=0.05, Bias_Bias=1.7)
    values = {'TGP_SE_var': 1.1, 'TGP_SE_rate': r, 'TGP_NoiseSE_var': 0.05, 'TGP_Bias_Bias': 1.7}
    ts, tk = orc.TScaleOracle(1.7), orc.TKernelOracle(('SE', 1.1, r, None), 0.05)
    v = rng.standard_normal(260)
    np.testing.assert_allclose(T.t1(X, v, values=values), ts(X, v), rtol=1e-15)
    np.testing.assert_allclose(T.t1.inv(X, v, values=values), ts.inv(X, v), rtol=1e-15)
    assert abs(T.t1.logdet_dinv(X, v, values=values) - ts.logdet_dinv(X, v)) <= 1e-12
    np.testing.assert_allclose(T(X, v, noise=True, values=values), ts(X, tk(X, v, noise=True)), atol=1e-9)
    ref = orc.transport_logp(y, ts, tk, X)
    assert abs(tgp.loglike(p) - ref) <= 1e-8 * abs(ref)
    dist = g3.TransportGaussianDistribution(T, X, values)
    assert abs(dist.logp(y) - ref) <= 1e-8 * abs(ref)
    assert g3.TransportGaussianDistribution.logp_t(np.where(np.arange(260) == 3, np.nan, y), T, X, values) == np.float32(-1e30)
    tri = g3.TTriangular(g3.Bias(X))
    assert isinstance(tri, g3.TNoLinear) and tri.generator is tri.parametrics[0]
    with pytest.raises(NotImplementedError):
        tri(X, v)


def test_transport_process_includes_potentials():
    """a potential registered on a transport's hypers enters logp exactly as for the elliptical
    processes (hypers/__init__.py:94-109; stochastic.py:300-306)"""
    import g3py_amd as g3
    rng = np.random.default_rng(22)
    X = rng.uniform(0, 3, (150, 1))
    y = np.sin(X[:, 0]) + 0.05 * rng.standard_normal(150)

    def build(potential):
        T = g3.TLocation(g3.Bias(X)) @ g3.TKernel(g3.SE(X), noisy=True)
        if potential:
            T.set_potential(hypers='rate', reg='L2', c=0.5)
        t = g3.TGP(space=X[:4], transport=T)
        t.observed(X, y)
        return t
    plain, pot = build(False), build(True)
    p = _params(plain, SE_var=0.9, SE_rate=np.array([1.4]), NoiseSE_var=0.1, Bias_Bias=0.2)
    assert len(pot.model.potentials) == 1 and len(plain.model.potentials) == 0
    expect = plain.logp(p) + 0.5 * -(1.4 ** 2)
    assert abs(pot.logp(p) - expect) <= 1e-10 * abs(expect)
    assert abs(pot.logp(p, prior=True) - (plain.logp(p, prior=True) - 0.5 * 1.4 ** 2)) <= 1e-12


def test_batched_dlogp_chain_matches_single_gradients():
    """dlogp_chain / fixed_dlogp: one batched factor + K^-1 sweep (g3_gp_dlogp_batched) equals the loop
    of single dlogp evaluations (stochastic.py:554-564), warped GP with a Bias location, ragged N"""
    import g3py_amd as g3
    rng = np.random.default_rng(31)
    N, d = 330, 3
    X = rng.uniform(0, 4, (N, d))
    y = np.exp(0.4 * np.sin(X.sum(1))) + 0.02 * rng.standard_normal(N)
    gp = g3.WGP(space=X[:5], location=g3.Bias(X), kernel=g3.SE(X) + g3.MAT32(X, name='M'), mapping=g3.LogShifted(y))
    gp.observed(X, y)
    base = gp.active.dict_to_array(gp.params)
    chain = base[None, :] + 0.15 * rng.standard_normal((7, len(base)))
    got = gp.dlogp_chain(chain, batch=4)
    ref = np.array([gp.dlogp(c, array=True) for c in chain])
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)
    # fixed_dlogp over a fixed chain uses the batched sweep
    gp.active.fix_vars(chain, keys=[gp.model.vars[0].key])
    sp = gp.active.sampling_params(base)
    r = gp.fixed_dlogp(sp, return_array=True)
    rows = gp.active.fixed_chain.copy()
    ref2 = np.array([gp.dlogp(c, array=True)[gp.active.sampling_dims] for c in rows])
    np.testing.assert_allclose(r, ref2, rtol=1e-9, atol=1e-9)


def test_sampler_single_call_and_vectorised_mapping(golden_dir):
    """g3_gp_sample: loc + L Z in one C-ABI call, mapping applied to the whole M x S matrix;
    same draws as the oracle's per-sample loop (gaussian.py:89-97)"""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(33)
    X, Xs = rng.uniform(0, 3, (200, 2)), rng.uniform(0, 3, (37, 2))
    y = np.sinh(np.sin(X.sum(1))) + 0.03 * rng.standard_normal(200)
    Z = rng.standard_normal((37, 5))
    r = np.array([0.9, 1.2])
    gp = g3.WGP(space=Xs, location=g3.Zero(), kernel=g3.SE(X), mapping=g3.ArcsinhLinear(y))
    gp.observed(X, y)
    p = _params(gp, SE_var=1.0, SE_rate=r, Noise_var=0.1, ArcsinhLinear_shift=0.1, ArcsinhLinear_scale=0.8)
    o = orc.GP(('SE', 1.0, r, None), 0.1, ('Zero',), ('ArcsinhLinear', 0.1, 0.8))
    for prior in (False, True):
        got = gp.sampler(p, rand=Z, prior=prior)
        ref = o.sampler(Xs, X, y, rand=Z, prior=prior)
        assert got.shape == (37, 5)
        np.testing.assert_allclose(got, ref, atol=2e-6)


def test_experiment_harness_scores_and_timings():
    """Experiment (selection.py:43-338): split -> observe -> scores on observations / hold-out / test with the
    time_* columns; the test scores equal a by-hand predict on the same split"""
    import g3py_amd as g3
    np.random.seed(4)
    x = np.linspace(0, 4, 160)[:, None]
    y = np.sin(2 * x[:, 0]) + 0.05 * np.random.randn(160)
    models = [g3.GP(space=x, location=g3.Zero(), kernel=g3.SE(x), name='A'),
              g3.GP(space=x, location=g3.Bias(x), kernel=g3.MAT32(x), name='B')]
    ex = g3.Experiment(models)
    ex.data(x, y, p=0.4, method='random', include_min=True)
    ex.scores(logpred=True, mean=True, median=False, variance=True)
    ex.model_selection(holdout='_l2', holdout_p=0.25)
    ex.run(n_simulations=2)
    ex.run(n_simulations=0, repeat=[0])                       # a stored split again
    res = ex.results()
    assert len(res) == 6 and set(res.model) == {'A', 'B'} and len(ex.simulations()) == 2
    for c in ('time_params', 'time_obs', 'time_valid', 'time_test', 'obs_l1', 'valid_l2', 'test_l2', 'test_mse', 'test_nlpd'):
        assert c in res.columns and np.all(np.isfinite(res[c].astype(float)))
    assert (res.time_obs > 0).all() and (res.time_test > 0).all()
    # by hand: model A on split 0, conditioned on observations + hold-out, scored on the test points
    sim = ex.simulations().loc[0]
    tr = np.concatenate([sim['obs'], sim['valid']])
    gp = g3.GP(space=x[sim['test']], location=g3.Zero(), kernel=g3.SE(x), name='A')
    gp.observed(x[sim['obs']], y[sim['obs']])
    p = gp.params_default                                       # what the experiment used (defaults from the observations)
    gp.observed(x[tr], y[tr])
    want = np.mean((gp.mean(p) - y[sim['test']]) ** 2)
    got = float(res[(res.n_sim == 0) & (res.model == 'A')].iloc[0].test_l2)
    assert abs(got - want) <= 1e-9 * max(1.0, abs(want))
    rep = res[(res.n_sim == 0) & (res.model == 'A')].test_l2.astype(float).values
    assert len(rep) == 2 and abs(rep[0] - rep[1]) <= 1e-12
    with pytest.raises(NotImplementedError):
        ex.model_selection(find_MAP=True)


@pytest.mark.parametrize('N,dtype', [(64, np.float64), (100, np.float64), (128, np.float64), (200, np.float64), (256, np.float64),
                                     (128, np.float32), (256, np.float32)])
def test_small_chain_one_workgroup_per_member(N, dtype):
    """N <= 256 members of a chain are evaluated by ONE workgroup each in one launch
    (g3_potrf.hip::small_factor_kernel: factor, block inverses, a = L^-1 delta, log-determinant and quadratic form).
    Every member equals the one-at-a-time evaluation at 1e-12 (fp32: 1e-5) and the oracle at 1e-8 (fp32: 1e-4); the
    factor, the inverses and a left behind are what the batched gradient needs (checked through dlogp_chain)."""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(N)
    d = 3
    X = rng.uniform(0, N ** (1 / d), (N, d))
    y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    gp = g3.GaussianProcess(space=X, location=g3.Bias(), kernel=g3.MAT52(X), dtype=dtype)
    gp.observed(X, y)
    base = _params(gp, MAT52_var=1.2, MAT52_rate=[0.8, 1.0, 1.3], Noise_var=0.1, Bias_Bias=0.1)
    a0 = gp.active.dict_to_array(base)
    chain = a0 + 0.15 * rng.standard_normal((70, len(a0)))
    chain[0] = a0
    want = np.array([gp.logp(r, array=True) for r in chain], dtype=np.float64)
    got = np.asarray(gp.logp_chain(chain), dtype=np.float64)
    f32 = np.dtype(dtype) == np.float32
    np.testing.assert_allclose(got, want, rtol=1e-5 if f32 else 1e-12)
    ref = orc.GP(('MAT52', 1.2, np.array([0.8, 1.0, 1.3]), None), 0.1, ('Bias', 0.1)).logp(X, y)
    assert abs(got[0] - ref) <= (1e-4 if f32 else 1e-8) * abs(ref)
    if not f32:
        g1 = np.array([gp.dlogp(r, array=True) for r in chain[:5]])
        np.testing.assert_allclose(gp.dlogp_chain(chain[:5]), g1, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize('N,B,group,dtype', [(300, 210, 0, np.float64), (512, 260, 0, np.float64), (700, 24, 3, np.float64), (1024, 12, 5, np.float64),
                                             (896, 10, 2, np.float32), (384, 33, 8, np.float64)])
def test_medium_chain_cooperative_kernel_equals_the_sweep_and_the_one_at_a_time_path(monkeypatch, N, B, group, dtype):
    """VERDICT r4 item 5 (stochastic.py:515-531,740-771): members of 256 < N <= 1024 are factored by a GROUP of
    workgroups each, the whole batch in one launch (g3_chainb.hip) -- one workgroup per member for long chains (group 0 = by
    batch size), several (G3_COOP_GROUP) when forced, incl. an odd number of 128-blocks and fp32.  Every member equals the
    batched lock-step sweep (G3_COOP_MAX_N=0) at 1e-12 and the one-at-a-time evaluation; a member whose first factorisation
    fails (no noise, duplicated inputs) takes the jitter schedule alone, as before."""
    import g3py_amd as g3
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec
    rng = np.random.default_rng(N + B)
    d = 3
    X = rng.uniform(0, N ** (1 / d), (N, d))
    X[1] = X[0]                                    # a duplicated input: singular without noise
    y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
    hyp = [(1.0 + 0.3 * (i % 7) / 7, 0.7 + 0.6 * (i % 5) / 5, 0.05 + 0.1 * (i % 3) / 3) for i in range(B)]
    hyp[B // 2] = (1.0, 1.0, 0.0)                  # this member needs the jitter schedule
    progs = [compile_spec(('sum', ('MAT52', v, np.full(d, r), None), ('NOISE', nz)), d) for v, r, nz in hyp]
    arr = (_lib.KernelProg * B)(*progs)
    Np = _lib.roundup(N)
    kstride = (Np + _lib.G3_RHS_PAD) * Np
    f32 = np.dtype(dtype) == np.float32
    res = {}
    for mode in ('coop', 'sweep'):
        monkeypatch.setenv('G3_COOP_MAX_N', '1024' if mode == 'coop' else '0')
        monkeypatch.setenv('G3_COOP_MIN_BATCH', '2')
        monkeypatch.setenv('G3_COOP_GROUP', str(group))
        dev = g3.Device(0)                         # the knobs are read when a context is created
        K = dev.alloc(B * (Np + _lib.G3_RHS_PAD), Np, dtype)
        W = dev.alloc(B * Np, _lib.G3_PAD, dtype)
        a = dev.alloc(B, Np, dtype)
        Xd, dd = dev.upload(X.astype(dtype)), dev.upload(np.tile(y, (B, 1)).astype(dtype))
        st = dev.gp_factor_batched(arr, Xd, N, d, dd, K, kstride, W, a, raw=True)
        res[mode] = (st.copy(), dev.download(a, B, N).astype(np.float64))
        if mode == 'coop':
            K1, a1, W1 = dev.alloc(Np + _lib.G3_RHS_PAD, Np, dtype), dev.alloc(1, Np, dtype), dev.alloc_inverses(Np, dtype)
            yd = dev.upload(y.astype(dtype))
            for i in (0, B // 2, B - 1):
                s1 = dev.gp_factor(progs[i], Xd, N, d, yd, K1, W1, a1)
                assert abs(s1['logdet'] - st[i, 0]) <= (2e-4 if f32 else 1e-11) * abs(s1['logdet']), (i, s1['logdet'], st[i, 0])
                assert abs(s1['quad'] - st[i, 1]) <= (2e-3 if f32 else 1e-9) * abs(s1['quad'])
                assert int(s1['tries']) == int(st[i, 3])
        dev.close()
    (sc, ac), (ss, as_) = res['coop'], res['sweep']
    assert sc[B // 2, 3] >= 1 and np.all(np.delete(sc[:, 3], B // 2) == 0)          # only that member retried
    np.testing.assert_array_equal(sc[:, 3:], ss[:, 3:])
    np.testing.assert_allclose(sc[:, :2], ss[:, :2], rtol=2e-4 if f32 else 1e-11)
    np.testing.assert_allclose(ac, as_, rtol=0, atol=(2e-3 if f32 else 1e-9) * np.abs(as_).max())


@pytest.mark.gpu
@pytest.mark.parametrize('N', [96, 700])
def test_chain_fields_abi_equals_programs_abi(N):
    """g3_gp_factor_batched_fields (one template + the doubles that differ, expanded on the device) returns what
    g3_gp_factor_batched returns for the same members packed as whole programs -- bit for bit, both below and above
    the one-workgroup-per-member size, including a member that needs the jitter schedule; bad offsets are refused."""
    import g3py_amd as g3
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec, compile_spec_rows
    rng = np.random.default_rng(N)
    d, B = 2, 9
    X = np.repeat(rng.uniform(0, 6, (N // 2, d)), 2, axis=0)          # every input twice: singular without noise
    y = np.sin(X.sum(1))
    gp = g3.GaussianProcess(space=X, location=g3.Zero(), kernel=g3.SE(X) * g3.COS(X) + 0.5 * g3.RQ(X))
    gp.observed(X, y)
    a0 = gp.active.dict_to_array(gp.params_default)
    chain = a0 + 0.2 * rng.standard_normal((B, len(a0)))
    sizes = [(v.key, v.size) for v in gp.model.vars]
    at = sum(n for k, n in sizes[:[k for k, _ in sizes].index('GP_Noise_var_log_')])
    chain[4, at] = -800.0                  # exp underflows: no noise, the first factorisation fails
    values_b, _ = gp._values_rows(chain)
    tmpl, offs, fields = compile_spec_rows(gp.f_kernel_noise.spec(values_b, d),
                                           gp.f_kernel_noise.spec(gp._values_row(values_b, 0), d), d, B)
    progs = [compile_spec(gp.f_kernel_noise.spec(gp._values_row(values_b, j), d), d) for j in range(B)]
    dev = gp.device
    Np = _lib.roundup(N)
    kstride = (Np + _lib.G3_RHS_PAD) * Np
    Xd = dev.upload(X)
    dd = dev.upload(np.tile(y, (B, 1)))
    res = []
    for call in ('progs', 'fields'):
        K = dev.alloc(B * (Np + _lib.G3_RHS_PAD), Np, np.float64)
        W = dev.alloc(B * Np, _lib.G3_PAD, np.float64)
        a = dev.alloc(B, Np, np.float64)
        if call == 'progs':
            st = dev.gp_factor_batched(progs, Xd, N, d, dd, K, kstride, W, a, raw=True)
        else:
            st = dev.gp_factor_batched_fields(tmpl, offs, fields, Xd, N, d, dd, K, kstride, W, a)
        res.append((st.copy(), dev.download(a, B, N).copy()))
        for b in (K, W, a):
            b.free()
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][0][4, 3] >= 1 and np.all(np.delete(res[0][0][:, 3], 4) == 0), res[0][0]   # only member 4 retried
    K = dev.alloc(B * (Np + _lib.G3_RHS_PAD), Np, np.float64)
    W = dev.alloc(B * Np, _lib.G3_PAD, np.float64)
    a = dev.alloc(B, Np, np.float64)
    for bad in (4, 0, _lib.KernelProg.leaf.offset + 8, 10 ** 6):        # not 8-aligned / nleaf / a leaf's dims / outside
        o2 = offs.copy()
        o2[0] = bad
        with pytest.raises(g3.G3Error, match='status -5'):
            dev.gp_factor_batched_fields(tmpl, o2, fields, Xd, N, d, dd, K, kstride, W, a)


@pytest.mark.parametrize('case', ['wgp_boxcox_linear_l2', 'gp_periodic_product', 'wgp_linearmap_bias_small'])
def test_dlogp_chain_block_path_equals_row_by_row(case):
    """dlogp_chain's block path -- programs as template + fields (g3_gp_dlogp_batched_fields), the members' alpha and
    kernel-parameter sums in launches that carry the member in grid.y, the chain rule as array arithmetic over the rows --
    against one dlogp per row, including rows on the -1e30 branch, the -inf Jacobian and the jitter schedule."""
    import g3py_amd as g3
    rng = np.random.default_rng(41)
    if case == 'wgp_boxcox_linear_l2':
        N, d = 200, 2
        X = rng.uniform(0, 4, (N, d))
        y = np.exp(0.3 * np.sin(X.sum(1))) + 0.02 * rng.standard_normal(N) + 1.0
        k = g3.SE(X)
        k.set_potential('var', 'L2', c=0.3)
        gp = g3.WGP(space=X[:4], location=g3.Linear(X), kernel=k + g3.RQ(X), mapping=g3.BoxCoxLinear(y))
    elif case == 'gp_periodic_product':
        N, d = 150, 2
        X = rng.uniform(0, 4, (N, d))
        y = np.sin(X.sum(1)) + 0.05 * rng.standard_normal(N)
        gp = g3.GP(space=X[:4], location=g3.Zero(), kernel=g3.MAT52(X) * g3.COS(X) + 0.5 * g3.OU(X))
    else:
        N, d = 96, 3
        X = np.repeat(rng.uniform(0, 3, (N // 2, d)), 2, axis=0)          # duplicated inputs: singular without noise
        y = np.sin(X.sum(1)) + 2.0
        gp = g3.WGP(space=X[:4], location=g3.Bias(X), kernel=g3.MAT32(X), mapping=g3.LinearMapping(y))
    gp.observed(X, y)
    base = gp.active.dict_to_array(gp.params)
    chain = base[None, :] + 0.1 * rng.standard_normal((9, len(base)))
    sizes = [(v.key, v.size) for v in gp.model.vars]

    def at(key):
        i = [k for k, _ in sizes].index(key)
        return sum(n for _, n in sizes[:i])
    noise = at(gp.name + '_Noise_var_log_')
    jit = case == 'wgp_linearmap_bias_small'
    chain[3, noise] = -800.0 if jit else np.log(1e-3)     # exp underflows: no noise at all -> singular K, the jitter schedule
    if case == 'wgp_boxcox_linear_l2':
        chain[5, at('WGP_BoxCoxLinear_shift')] = -50.0                      # log of negative numbers: the -1e30 branch
    got = gp.dlogp_chain(chain, batch=5)                                    # two blocks: 5 + 4 rows
    ref = np.array([gp.dlogp(c, array=True) for c in chain])
    rest = np.arange(9) != 3 if jit else np.ones(9, dtype=bool)
    for g in (got, gp.dlogp_chain(chain)):
        np.testing.assert_allclose(g[rest], ref[rest], rtol=1e-8, atol=1e-8)
        if jit:       # the jittered factor has condition ~ 1 / jitter: the two summation orders of K^-1 agree to that
            scale = np.abs(ref[3]).max()
            np.testing.assert_allclose(g[3], ref[3], rtol=1e-4, atol=1e-6 * scale)


@pytest.mark.parametrize('warped', [False, True])
def test_student_t_chains_use_the_block_path_and_equal_row_by_row(warped):
    """StudentTProcess.logp_chain / dlogp_chain: the Gaussian block path with the t density and the per-row scale
    s = (nu + n) / (nu - 2 + beta) (studentT.py:114-146) against one evaluation per row; prior=True is the free
    variables' terms alone"""
    import g3py_amd as g3
    rng = np.random.default_rng(53)
    N, d = 140, 2
    X = rng.uniform(0, 4, (N, d))
    y = np.exp(0.3 * np.sin(X.sum(1))) + 0.03 * rng.standard_normal(N) + 1.0
    if warped:
        tp = g3.WarpedStudentTProcess(space=X[:4], location=g3.Bias(X), kernel=g3.SE(X), mapping=g3.LinearMapping(y))
    else:
        tp = g3.StudentTProcess(space=X[:4], location=g3.Bias(X), kernel=g3.MAT52(X) + g3.COS(X))
    tp.observed(X, y)
    base = tp.active.dict_to_array(tp.params)
    chain = base[None, :] + 0.1 * rng.standard_normal((7, len(base)))
    chain[2, :] = base - 40.0 * (np.arange(len(base)) == len(base) - 1)      # (harmless shift of the last variable)
    want = np.array([tp.logp(c, array=True) for c in chain])
    np.testing.assert_allclose(tp.logp_chain(chain, batch=4), want, rtol=1e-10)
    np.testing.assert_allclose(tp.logp_chain(chain, prior=True), [tp.logp(c, array=True, prior=True) for c in chain], rtol=1e-12)
    ref = np.array([tp.dlogp(c, array=True) for c in chain])
    np.testing.assert_allclose(tp.dlogp_chain(chain, batch=4), ref, rtol=1e-8, atol=1e-8)
    assert type(tp).dlogp_chain is g3.GaussianProcess.dlogp_chain
