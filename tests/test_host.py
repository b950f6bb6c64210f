"""CPU tests of the host logic: parameter naming / ordering / defaults, the dict<->array
bijection, set_space rules, kernel algebra -> g3_kernel_prog lowering.  No GPU calls."""
import numpy as np
import pytest

import g3py_amd as g3
from g3py_amd.device import compile_spec, _expand
from oracle import g3_oracle as orc


def _eval_prog(prog, X1, X2=None):
    """evaluate a g3_kernel_prog on the CPU with the oracle's leaf formulas (test-only)"""
    from g3py_amd._lib import KINDS
    inv = {v: k for k, v in KINDS.items()}
    n1 = len(X1)
    n2 = n1 if X2 is None else len(X2)
    leaf_vals = []
    for i in range(prog.nleaf):
        L = prog.leaf[i]
        kind = inv[L.kind]
        dims = np.array([L.dims[k] for k in range(L.ndims)])
        rate = np.array([L.rate[k] for k in range(L.ndims)])
        freq = np.array([L.freq[k] for k in range(L.ndims)])
        spec = {'SE': (kind, L.var, rate, dims), 'OU': (kind, L.var, rate, dims), 'MAT32': (kind, L.var, rate, dims),
                'MAT52': (kind, L.var, rate, dims), 'RQ': (kind, L.var, rate, L.alpha, dims),
                'COS': (kind, L.var, freq, dims), 'SINC': (kind, L.var, freq, dims),
                'SIN': (kind, L.var, freq, rate, dims), 'SM': (kind, L.var, freq, rate, dims),
                'NOISE': (kind, L.var), 'WN': (kind, L.var, dims)}[kind]
        leaf_vals.append(orc.kernel_cov(spec, X1, X2))
    out = np.full((n1, n2), prog.shift)
    for p in range(prog.nprod):
        t = np.full((n1, n2), prog.prod[p].coef)
        for f in range(prog.prod[p].nfac):
            t = t * leaf_vals[prog.prod[p].fac[f]]
        out = out + t
    return out


def test_sum_of_products_lowering_matches_tree():
    rng = np.random.default_rng(0)
    X, X2 = rng.uniform(0, 3, (30, 3)), rng.uniform(0, 3, (11, 3))
    r, f = np.array([0.6, 1.0, 1.4]), np.array([0.1, 0.2, 0.3])
    trees = [
        ('sum', ('SE', 1.3, r, None), ('NOISE', 0.2)),
        ('prod', ('sum', ('SE', 1.0, r, None), ('OU', 0.5, r, None)), ('shift', 0.5, ('MAT32', 0.7, r, None))),
        ('shift', 0.1, ('scale', 2.0, ('prod', ('COS', 1.0, f, None), ('SE', 1.0, r[:2], np.array([0, 2]))))),
        ('prod', ('shift', 1.0, ('SM', 0.4, f, r, None)), ('shift', 2.0, ('RQ', 0.9, r, 1.5, None))),
    ]
    for tree in trees:
        prog = compile_spec(tree, 3)
        np.testing.assert_allclose(_eval_prog(prog, X), orc.kernel_cov(tree, X), rtol=1e-13)
        np.testing.assert_allclose(_eval_prog(prog, X2, X), orc.kernel_cov(tree, X2, X), rtol=1e-13)


def test_lowering_limits():
    r = np.ones(2)
    big = ('SE', 1.0, r, None)
    for _ in range(9):
        big = ('sum', big, ('SE', 1.0, r, None))
    with pytest.raises(g3.G3Error):
        compile_spec(big, 2)
    with pytest.raises(g3.G3Error):
        compile_spec(('SE', 1.0, np.ones(3), np.array([0, 1, 5])), 3)      # column out of range
    with pytest.raises(g3.G3Error):
        compile_spec(('SE', 1.0, np.ones(70), None), 70)                   # too many columns


def test_parameter_names_order_and_defaults():
    rng = np.random.default_rng(1)
    X = rng.uniform(0, 5, (25, 2))
    y = np.sin(X.sum(1)) + 3.0
    gp = g3.WGP(space=X, location=g3.Linear(), kernel=g3.MAT52(X) * g3.SIN(X) + g3.RQ(X),
                mapping=g3.BoxCoxLinear())
    keys = [v.key for v in gp.model.vars]
    assert keys == ['WGP_Linear_Constant', 'WGP_Linear_Coeff', 'WGP_MAT52_var_log_', 'WGP_MAT52_rate_log_',
                    'WGP_SIN_freq_log_', 'WGP_SIN_rate_log_', 'WGP_RQ_var_log_',   # KernelProd fixes k2.var = 1 (kernels.py:217-219)
                    'WGP_RQ_rate_log_', 'WGP_RQ_alpha_log_', 'WGP_Noise_var_log_', 'WGP_BoxCoxLinear_shift',
                    'WGP_BoxCoxLinear_scale_log_', 'WGP_BoxCoxLinear_power_log_'], keys
    assert gp.active.ndim == 1 + 2 + 1 + 2 + 2 + 2 + 1 + 2 + 1 + 1 + 3
    gp.observed(X, y)
    d = gp.params_default
    assert np.isclose(d['WGP_Noise_var_log_'], np.log(y.var()))
    np.testing.assert_allclose(d['WGP_Linear_Coeff'], y.mean() / X.mean(axis=0))
    np.testing.assert_allclose(np.exp(d['WGP_SIN_freq_log_']), 1 / (X.max(0) - X.min(0)))
    assert d['WGP_RQ_alpha_log_'] == 0 and d['WGP_BoxCoxLinear_shift'] == 1
    # bijection round trip in creation order (models.py:143-155)
    a = gp.active.dict_to_array(d)
    assert a.shape == (gp.active.ndim,)
    back = gp.active.array_to_dict(a)
    for k in d:
        np.testing.assert_allclose(back[k], d[k])
    # natural <-> transformed names
    nat = gp.transform_params(d, to_transformed=False)
    assert np.isclose(nat['WGP_Noise_var'], y.var()) and 'WGP_Noise_var_log_' not in nat
    # set_params / params
    gp.set_params(d)
    assert gp.params == d
    gp.set_params(None)
    assert set(gp.params_test) == set(keys) and all(np.all(v == 0) for v in gp.params_test.values())


def test_set_space_rules_and_copy_semantics():
    x = np.linspace(0, 1, 7)
    gp = g3.GP(space=x, location=g3.Zero(), kernel=g3.SE(x[:, None]))
    assert gp.space.shape == (7, 1) and gp.nspace == 1
    np.testing.assert_array_equal(gp.order, x)
    gp.observed(x[:3], np.zeros((3, 1)))
    assert gp.inputs.shape == (3, 1) and gp.outputs.shape == (3,) and gp.is_observed
    np.testing.assert_array_equal(gp.index, x[:3])
    s = gp.space
    s[0, 0] = 99.0                      # getters return copies (borrow=False, stochastic.py:219-259)
    assert gp.space[0, 0] == 0.0
    gp.observed()
    assert not gp.is_observed
    x2 = np.random.default_rng(0).uniform(size=(9, 3))
    gp2 = g3.GP(space=x2, location=g3.Zero(), kernel=g3.SE(x2))
    assert gp2.nspace == 3 and len(gp2.order) == 9


def test_method_registry_names():
    x = np.linspace(0, 1, 5)[:, None]
    gp = g3.GP(space=x, location=g3.Zero(), kernel=g3.SE(x))
    assert set(gp.compiles) == {'posterior_logp', 'array_posterior_logp', 'prior_logp', 'array_prior_logp'}
    m = gp._compiled('th_kernel_sd', False, True, False, (), {})
    assert m is gp.compiles['posterior_kernel_sd_noise'] and m.executed == 0
    for name in ('mean', 'median', 'variance', 'std', 'covariance', 'logpredictive', 'logp', 'loglike', 'location',
                 'kernel', 'cholesky', 'kernel_diag', 'kernel_sd', 'cholesky_diag', 'mapping', 'mapping_inv',
                 'quantiler', 'sampler', 'predict'):
        assert callable(getattr(gp, name)), name
    assert g3.GP is g3.GaussianProcess and g3.WGP is g3.WarpedGaussianProcess


def test_makefn_protocol():
    calls = []
    fn = g3.makefn(['v'], lambda s, i, o, v, p: calls.append((s, i, o, v, p)) or 7)
    assert fn({'a': 1}, 's', 'i', 'o') == 7 and fn.executed == 1
    c = fn.clone(lambda arr: {'a': arr[0]})
    assert c({0: 5}.get(0) and [5], 's', 'i', 'o') == 7 and calls[-1][4] == {'a': 5}
    assert c.executed == 2 and fn.executed == 1          # the clone copies the counter, then counts on its own


def test_mappings_and_means_match_oracle():
    y = np.linspace(0.6, 3.0, 9)
    vals = {'m_shift': 1.0, 'm_scale': 0.7, 'm_power': 1.2}
    with g3.Model('t'):
        m = g3.BoxCoxLinear(name='m')
        m.check_hypers('')
    o = orc.Mapping(('BoxCoxLinear', 1.0, 0.7, 1.2))
    np.testing.assert_allclose(m.inv(y, vals), o.inv(y))
    np.testing.assert_allclose(m(o.inv(y), vals), y)
    assert np.isclose(m.logdet_dinv(y, vals), o.logdet_dinv(y))
    x = np.random.default_rng(0).uniform(size=(6, 2))
    with g3.Model('t2'):
        lin = g3.Linear(x, name='L')
        lin.check_hypers('')
    np.testing.assert_allclose(lin(x, {'L_Constant': 0.5, 'L_Coeff': np.array([1.0, -2.0])}),
                               orc.mean_eval(('Linear', 0.5, np.array([1.0, -2.0]), None), x))


def test_potentials_enter_logp_prior():
    """optional L1 / L2 regularisers (hypers/__init__.py:94-109) are added to th_logp like pm.Potential"""
    x = np.linspace(0, 1, 6)[:, None]
    k = g3.SE(x)
    k.set_potential('var', 'L1', c=2.0)
    gp = g3.GP(space=x, location=g3.Zero(), kernel=k)
    assert len(gp.model.potentials) == 1
    p = gp.params_test
    p['GP_SE_var_log_'] = np.log(3.0)
    values, extra = gp._values(p)
    assert np.isclose(extra, -2.0 * 3.0)
    assert np.isclose(gp.th_logp(None, None, None, [], p, prior=True), -6.0)


def test_hyper_slots_positional_keyword_and_supplied_values():
    """the declarative slot table behind every hyper-parametric function: slots are filled by position or keyword
    (the reference's constructor signatures), unknown names are refused, a supplied number stays a constant and is
    not registered, and the registered names are `<parent><Owner><suffix>` (ARD rates: `<parent>rate`)"""
    import g3py_amd as g3
    from g3py_amd.processes.hypers import Hypers, HyperVar, Model
    x = np.zeros((4, 3))
    with Model('slots') as m:
        a = g3.LinearMapping(x[:, 0], 'T', None, 2.5)        # positional: shift free, scale = 2.5
        a.check_hypers('P_')
        b = g3.BoxCoxLinear(name='B', power=1.2)
        b.check_hypers('P_')
        k = g3.SE(x)
        k.check_hypers('P_')
        per = g3.COS(x)
        per.check_hypers('P_')
        ou = g3.OU(x)
    assert isinstance(a.shift, HyperVar) and a.shift.name == 'P_T_shift' and not a.shift.positive
    assert a.scale == 2.5 and [h.name for h in a.hypers if isinstance(h, HyperVar)] == ['P_T_shift']
    assert b.power == 1.2 and b.scale.key == 'P_B_scale_log_'
    assert [v.key for v in m.vars if v.name.startswith('P_SE')] == ['P_SE_var_log_', 'P_SE_rate_log_']
    assert k.metric.rate.shape == (3,) and type(k.metric).__name__ == 'ARD_L2' and type(ou.metric).__name__ == 'ARD_L1'
    assert per.rate == 1.0 and per.freq.shape == (3,) and type(per.metric).__name__ == 'Difference'
    with pytest.raises(TypeError):
        g3.LinearMapping(x[:, 0], 'T', colour=1)
    with pytest.raises(TypeError):
        g3.Bias(x, 'b', 1.0, 2.0)                             # one slot, two values


def test_generated_gram_kernel_compiles_for_every_zoo_expression():
    """Round 4: a Gram kernel is generated for a kernel expression's STRUCTURE at first use (g3_gram_jit.hip, hipRTC).
    hipRTC cross-compiles without a GPU, so the build host can check that the source generated for every expression of
    the test zoo (all d) compiles for gfx950, fp64 and fp32"""
    import ctypes as C
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec
    from oracle.gen_golden import kernel_zoo
    lib = _lib.load()
    n = 0
    for d in (1, 3, 8):
        for name, spec in kernel_zoo(d).items():
            for dt in ((0, 1) if name in ('SINC', '(SE+OU)*(MAT32+0.5)', 'SM') else (0,)):
                prog = compile_spec(spec, d)
                cb, log = C.c_int64(0), C.create_string_buffer(8000)
                rc = lib.g3_gram_jit_check(C.byref(prog), d, dt, C.byref(cb), log, 8000)
                if rc == -1:
                    pytest.skip('libhiprtc is not available on this host')
                assert rc == 0 and cb.value > 1000, (d, name, dt, rc, log.value.decode()[:2000])
                n += 1
    assert n >= 45


def test_generated_gradient_kernel_compiles_for_every_zoo_expression():
    """the kernel of the gradient's kernel-parameter sums is generated per expression structure too
    (g3_gram_jit.hip::g3_grad_jit: register accumulators at compile-time slot indices): every zoo expression compiles for
    gfx950 on the build host"""
    import ctypes as C
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec
    from oracle.gen_golden import kernel_zoo
    lib = _lib.load()
    n = 0
    for d in (1, 3, 8):
        for name, spec in kernel_zoo(d).items():
            for dt in ((0, 1) if name in ('SINC', '(SE+OU)*(MAT32+0.5)') else (0,)):
                prog = compile_spec(spec, d)
                cb, log = C.c_int64(0), C.create_string_buffer(8000)
                rc = lib.g3_grad_jit_check(C.byref(prog), d, dt, C.byref(cb), log, 8000)
                if rc == -1:
                    pytest.skip('libhiprtc is not available on this host')
                assert rc == 0 and cb.value > 1000, (d, name, dt, rc, log.value.decode()[:3000])
                n += 1
    assert n >= 40


def test_chain_rows_packing_equals_row_by_row():
    """logp_chain's block path: one template program + per-row fields (compile_spec_rows, _values_rows, the `rows`
    forms of the mean and the warp) against the one-row-at-a-time host path, byte for byte where that is defined."""
    import ctypes as C
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec_rows
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 5, (25, 3))
    y = np.sin(X.sum(1)) + 3.0
    k1 = g3.SE(X)
    k1.set_potential('var', 'L2', c=0.5)
    cases = [
        g3.GP(space=X, location=g3.Bias(), kernel=2.5 * k1 * g3.SIN(X) + g3.RQ(X[:, :2], name='RQ2') + 0.25),
        g3.WGP(space=X, location=g3.Linear(), kernel=g3.MAT52(X) + g3.COS(X) * g3.OU(X), mapping=g3.BoxCoxLinear()),
        g3.WGP(space=X, location=g3.Zero(), kernel=g3.SM(X) + g3.WN(X), mapping=g3.LinearMapping()),
    ]
    for gp in cases:
        gp.observed(X, y)
        B, d = 7, X.shape[1]
        base = gp.active.dict_to_array(gp.params_default)
        chain = base[None, :] + 0.3 * rng.standard_normal((B, gp.active.ndim))
        chain[2, :] = -40.0                       # exp() underflows below 1e-6: -inf in the Jacobian term
        values_b, logjac = gp._values_rows(chain)
        tmpl, offs, fields = compile_spec_rows(gp.f_kernel_noise.spec(values_b, d),
                                               gp.f_kernel_noise.spec(gp._values_row(values_b, 0), d), d, B)
        assert fields.shape == (B, len(offs)) and len(set(offs.tolist())) == len(offs)
        yv = np.asarray(y, dtype=gp.dtype)
        with np.errstate(all='ignore'):
            inv_b = gp.f_mapping.inv_rows(yv, values_b, B)
            det_b = gp.f_mapping.logdet_dinv_rows(yv, values_b, B)
            loc_b = gp.f_location.rows(X, values_b, B)
        for j in range(B):
            gp._values_memo = None
            values, lj = gp._values(gp.active.array_to_dict(chain[j]))
            assert (lj == logjac[j]) or np.isclose(lj, logjac[j], rtol=1e-15), (lj, logjac[j])
            want = compile_spec(gp.f_kernel_noise.spec(values, d), d)
            got = _lib.KernelProg.from_buffer_copy(bytes(tmpl))
            raw = (C.c_char * C.sizeof(got)).from_buffer(got)
            for o, v in zip(offs, fields[j]):
                raw[o:o + 8] = np.float64(v).tobytes()
            assert bytes(got) == bytes(want), j
            with np.errstate(all='ignore'):
                np.testing.assert_array_equal(inv_b[j], np.asarray(gp.f_mapping.inv(yv, values), dtype=gp.dtype))
                np.testing.assert_array_equal(loc_b[j], gp.f_location(X, values))
                assert np.array_equal(det_b[j], gp.dtype.type(gp.f_mapping.logdet_dinv(yv, values)), equal_nan=True)
