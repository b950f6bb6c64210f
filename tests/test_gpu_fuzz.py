"""Seeded random shapes against the CPU oracle: ragged N and M around every blocking boundary of the HIP path
(128-wide diagonal blocks, 256-wide fused blocks, 32-row solve stripes, the panel widths of the sweep and its
tail), several kernel expressions, with and without the fused predict rows.  What the fixed-size tests cannot
see: an off-by-one in a padding rule or a raster table for a size nobody wrote a case for."""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    import g3py_amd as g3
    return g3.Device.default()


def _specs(rng, d):
    rate = np.exp(rng.uniform(-0.5, 0.5, d))
    freq = rng.uniform(0.05, 0.3, d)
    pool = [
        ('SE', 1.3, rate, None),
        ('MAT32', 0.8, rate, None),
        ('MAT52', 1.1, rate, None),
        ('OU', 0.9, rate, None),
        ('sum', ('SE', 0.7, rate, None), ('COS', 0.4, freq, None)),
        ('prod', ('MAT52', 1.2, rate, None), ('RQ', 1.0, rate, 1.7, None)),
        ('sum', ('MAT32', 0.6, rate, None), ('SM', 0.3, freq, rate * 0.2, None)),
        ('sum', ('SE', 1.0, rate[:1], [0]), ('WN', 0.05, None)) if d > 1 else ('SE', 1.0, rate, None),
    ]
    return pool[int(rng.integers(len(pool)))]


def _oracle(orc, spec_f, noise, X, y, Xs):
    K = orc.tt_to_num(orc.kernel_cov(orc.with_noise(spec_f, noise), X))
    L = scipy.linalg.cholesky(K, lower=True)
    a = scipy.linalg.solve_triangular(L, y, lower=True)
    lp = -0.5 * len(y) * np.log(2 * np.pi) - 0.5 * a.dot(a) - np.log(np.diag(L)).sum()
    Ks = orc.tt_to_num(orc.kernel_cov(spec_f, Xs, X))
    V = scipy.linalg.solve_triangular(L, Ks.T, lower=True)
    prior = np.diag(orc.kernel_cov(spec_f, Xs[:1], Xs[:1]))[0]
    return lp, V.T.dot(a), np.maximum(prior - (V ** 2).sum(0), 0.0), prior


SIZES = [1, 2, 31, 127, 128, 129, 255, 256, 257, 383, 385, 511, 640, 767, 769, 1000, 1023, 1025, 1279, 1537,
         2047, 2049, 2305, 3071, 3200, 4095, 4097, 4500,
         5000, 6145, 7169, 9001]      # the wider panel rules of the sweep (256 / 512 columns) and their tails; d <= 2 (oracle memory)


@pytest.mark.parametrize('case', range(len(SIZES)))
def test_random_shape_matches_oracle(dev, case):
    from oracle import g3_oracle as orc
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec
    rng = np.random.default_rng(9000 + case)
    N = SIZES[case]
    d = int(rng.integers(1, 6)) if N < 4600 else int(rng.integers(1, 3))
    M = int(rng.choice([1, 5, 127, 128, 129, 300]))
    noise = float(rng.choice([0.05, 0.2, 1.0]))
    X = rng.uniform(0, max(N, 2) ** (1.0 / d), (N, d))
    Xs = rng.uniform(0, max(N, 2) ** (1.0 / d), (M, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(N)
    spec_f = _specs(rng, d) if N < 4600 else [('SE', 1.3, np.ones(d), None), ('MAT52', 1.1, np.ones(d) * 0.8, None)][case % 2]
    lp_ref, mean_ref, var_ref, prior = _oracle(orc, spec_f, noise, X, y, Xs)
    Np, Mp = _lib.roundup(N), _lib.roundup(M, 128)
    K = dev.alloc(Np + 128 + Mp, Np, np.float64)
    W, a = dev.alloc_inverses(Np, np.float64), dev.alloc(1, Np, np.float64)
    mu, ss = dev.alloc(1, Mp, np.float64), dev.alloc(1, Mp, np.float64)
    st = dev.gp_factor_predict(compile_spec(orc.with_noise(spec_f, noise), d), compile_spec(spec_f, d), dev.upload(X), N, d,
                               dev.upload(y), dev.upload(Xs), M, K, W, a, mu, ss)
    lp = -0.5 * N * np.log(2 * np.pi) - 0.5 * st['quad'] - st['logdet']
    assert st['info'] == 0 and st['tries'] == 0
    assert abs(lp - lp_ref) <= 1e-8 * max(1.0, abs(lp_ref)), (N, d, M, spec_f[0])
    np.testing.assert_allclose(dev.download(mu, 1, M)[0], mean_ref, atol=1e-8, rtol=1e-8)
    np.testing.assert_allclose(np.maximum(prior - dev.download(ss, 1, M)[0], 0), var_ref, atol=1e-8, rtol=1e-8)


GRAD_SIZES = [1, 2, 63, 64, 65, 127, 129, 255, 257, 511, 513, 700, 1023, 1025, 1300, 1537]


@pytest.mark.parametrize('case', range(len(GRAD_SIZES)))
def test_random_shape_gradient_matches_oracle(dev, case):
    """g3_gp_dlogp (g3_potri + alpha + g3_gram_grad after g3_gp_factor) at ragged sizes around the 64-wide pair tiles of
    the gradient kernels and the 128 / 256-wide blocks of the inverse, random kernel expressions (fast paths and the
    interpreter): every parameter sum 1/2 sum_ij (alpha alpha^T - K^-1)_ij dK_ij/dtheta and alpha against the oracle"""
    from oracle import g3_oracle as orc
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec
    rng = np.random.default_rng(7000 + case)
    N = GRAD_SIZES[case]
    d = int(rng.integers(1, 5))
    noise = float(rng.choice([0.1, 0.5]))
    X = rng.uniform(0, max(N, 2) ** (1.0 / d), (N, d))
    y = np.sin(X.sum(1)) + 0.1 * rng.standard_normal(N)
    spec_n = orc.with_noise(_specs(rng, d), noise)
    K, grads = orc.kernel_cov_grads(spec_n, X)
    Kinv = np.linalg.inv(K)
    alpha_ref = Kinv @ y
    G = np.outer(alpha_ref, alpha_ref) - Kinv
    prog = compile_spec(spec_n, d)
    gmap = dev.grad_layout(prog)
    assert gmap.nslots == len(grads)
    Np = _lib.roundup(N)
    Kd = dev.alloc(Np + 128, Np, np.float64)
    W, a = dev.alloc_inverses(Np, np.float64), dev.alloc(1, Np, np.float64)
    Xd = dev.upload(X)
    st = dev.gp_factor(prog, Xd, N, d, dev.upload(y), Kd, W, a)
    assert st['info'] == 0
    Y, Ki, al = dev.alloc(Np, Np, np.float64), dev.alloc(Np, Np, np.float64), dev.alloc(1, Np, np.float64)
    slots = dev.gp_dlogp(prog, gmap, Xd, N, d, Kd, W, a, Y, Ki, al)
    np.testing.assert_allclose(dev.download(al, 1, N)[0], alpha_ref, rtol=1e-7, atol=1e-8 * max(np.abs(alpha_ref).max(), 1e-30))
    for (leaf, pname, k, dK) in grads:
        want, scale = 0.5 * np.sum(G * dK), 0.5 * np.sum(np.abs(G * dK)) + 1e-30
        got = slots[getattr(gmap, pname)[leaf] + (0 if k is None else k)]
        assert abs(got - want) < 1e-8 * scale, (N, d, spec_n, leaf, pname, k, got, want)


def _random_tree(rng, d, depth=0):
    """a random kernel expression: leaves of every family, sums / products / scales / shifts up to depth 3, with column
    subsets -- within one g3_kernel_prog's limits (checked by the caller)"""
    rate = np.exp(rng.uniform(-0.6, 0.4, d))
    freq = rng.uniform(0.05, 0.3, d)

    def leaf():
        kind = str(rng.choice(['SE', 'OU', 'MAT32', 'MAT52', 'RQ', 'COS', 'SIN', 'SINC', 'SM', 'WN']))
        nd = int(rng.integers(1, d + 1))
        dims = None if nd == d else np.sort(rng.choice(d, nd, replace=False))
        r, f = (rate, freq) if dims is None else (rate[dims], freq[dims])
        var = float(rng.uniform(0.3, 1.5))
        if kind in ('SE', 'OU', 'MAT32', 'MAT52'):
            return (kind, var, r, dims)
        if kind == 'RQ':
            return (kind, var, r, float(rng.uniform(0.8, 3.0)), dims)
        if kind in ('COS', 'SINC'):
            return (kind, var, f, dims)
        if kind in ('SIN', 'SM'):
            return (kind, var, f, 0.3 * r, dims)
        return ('WN', var, dims)
    if depth >= 3 or rng.uniform() < 0.25 + 0.2 * depth:
        return leaf()
    op = str(rng.choice(['sum', 'prod', 'scale', 'shift'], p=[0.4, 0.35, 0.15, 0.1]))
    if op in ('sum', 'prod'):
        return (op, _random_tree(rng, d, depth + 1), _random_tree(rng, d, depth + 1))
    return (op, float(rng.uniform(0.3, 2.0)), _random_tree(rng, d, depth + 1))


@pytest.mark.parametrize('seed', [20260, 7, 424242])
def test_random_expressions_generated_kernels_equal_the_interpreter(monkeypatch, seed):
    """random kernel expression trees (every leaf family, column subsets, sums / products / scales / shifts): the Gram
    kernel and the gradient kernel GENERATED for the expression's structure (g3_gram_jit.hip) against the interpreter
    on a second context (G3_GRAM_JIT=0, G3_GRAM_NOFAST=1, G3_GRAD_GENERIC=1), square and cross blocks"""
    import g3py_amd as g3
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec, G3Error
    dev1 = g3.Device(0)
    monkeypatch.setenv('G3_GRAM_JIT', '0')
    monkeypatch.setenv('G3_GRAM_NOFAST', '1')
    monkeypatch.setenv('G3_GRAD_GENERIC', '1')
    dev0 = g3.Device(0)
    rng = np.random.default_rng(seed)
    done = 0
    for trial in range(60):
        if done >= 10:
            break
        d = int(rng.integers(1, 5))
        spec = _random_tree(rng, d)
        try:
            prog = compile_spec(spec, d)
        except G3Error:
            continue                      # more leaves / products than one program holds
        if prog.nleaf < 2:
            continue
        done += 1
        N, M = int(rng.integers(70, 260)), int(rng.integers(3, 90))
        X, Xs = rng.uniform(0, 4, (N, d)), rng.uniform(0, 4, (M, d))
        X[5] = X[1]
        Np, Mp = _lib.roundup(N), _lib.roundup(M)
        got = []
        for dev in (dev1, dev0):
            Xd, Xsd = dev.upload(X), dev.upload(Xs)
            K = dev.alloc(Np, Np, np.float64)
            dev.gram(prog, Xd, None, d, K, Np, Np, 0)
            C = dev.alloc(Mp, Np, np.float64)
            dev.gram(prog, Xsd, Xd, d, C, Mp, Np, 0)
            A = rng.standard_normal((N, N)) if dev is dev1 else A
            alpha = rng.standard_normal(N) if dev is dev1 else alpha
            gmap = dev.grad_layout(prog)
            g = dev.gram_grad(prog, gmap, Xd, N, d, dev.upload(np.tril((A + A.T) / 2)), dev.upload(alpha))
            got.append((dev.download(K, N, N), dev.download(C, M, N), g))
        (K1, C1, g1), (K0, C0, g0) = got
        np.testing.assert_allclose(K1, K0, rtol=1e-12, atol=1e-13 * max(1.0, np.abs(K0).max()), err_msg=repr(spec))
        np.testing.assert_allclose(C1, C0, rtol=1e-12, atol=1e-13 * max(1.0, np.abs(C0).max()), err_msg=repr(spec))
        np.testing.assert_allclose(g1, g0, rtol=1e-8, atol=1e-10 * (np.abs(g0).max() + 1e-30), err_msg=repr(spec))
    assert done >= 8
    s1, s0 = dev1.gram_path_stats(), dev0.gram_path_stats()
    assert s1['interpreted'] == 0 and s0['generated'] == 0 and s0['table'] == 0, (s1, s0)
    q1, q0 = dev1.grad_path_stats(), dev0.grad_path_stats()
    assert q1['generated'] > 0 and q0['generated'] == 0 and q0['table'] == 0 and q0['interpreted'] > 0, (q1, q0)
    dev0.close()
    dev1.close()


@pytest.mark.parametrize('seed', list(range(1, 13)))
def test_random_processes_chain_block_paths_equal_row_by_row(seed):
    """random processes through the public API -- kernel objects combined with + and *, every mean and warp on the path,
    Gaussian and Student-t -- logp_chain / dlogp_chain (block host path, template + fields, members in grid.y) against
    one logp / dlogp per row"""
    import g3py_amd as g3
    rng = np.random.default_rng(seed)
    N, d = int(rng.integers(40, 300)), int(rng.integers(1, 4))
    X = rng.uniform(0, 4, (N, d))
    y = np.exp(0.3 * np.sin(X.sum(1))) + 0.03 * rng.standard_normal(N) + 1.0
    fams = [g3.SE, g3.OU, g3.MAT32, g3.MAT52, g3.RQ, g3.COS, g3.SM]

    def kern(i):
        return fams[int(rng.integers(len(fams)))](X, name='K%d' % i)
    k = kern(0)
    for i in range(1, int(rng.integers(1, 4))):
        k = (k + kern(i)) if rng.uniform() < 0.6 else (k * kern(i))
    if rng.uniform() < 0.3:
        k = float(rng.uniform(0.5, 2.0)) * k
    mean = [g3.Zero, g3.Bias, g3.Linear][int(rng.integers(3))](X)
    warp = [None, g3.LinearMapping, g3.BoxCoxLinear, g3.LogShifted, g3.ArcsinhLinear][int(rng.integers(5))]
    student = rng.uniform() < 0.35
    if warp is None:
        cls = g3.StudentTProcess if student else g3.GaussianProcess
        gp = cls(space=X[:3], location=mean, kernel=k)
    else:
        cls = g3.WarpedStudentTProcess if student else g3.WarpedGaussianProcess
        gp = cls(space=X[:3], location=mean, kernel=k, mapping=warp(y))
    gp.observed(X, y)
    base = gp.active.dict_to_array(gp.params)
    chain = base[None, :] + 0.08 * rng.standard_normal((6, len(base)))
    want = np.array([gp.logp(c, array=True) for c in chain], dtype=np.float64)
    got = np.asarray(gp.logp_chain(chain, batch=4), dtype=np.float64)
    fin = np.isfinite(want)
    assert np.array_equal(fin, np.isfinite(got)) and fin.sum() >= 4, (want, got)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-9)
    ref = np.array([gp.dlogp(c, array=True) for c in chain])
    np.testing.assert_allclose(gp.dlogp_chain(chain, batch=4), ref, rtol=2e-7, atol=1e-7 * (np.abs(ref).max() + 1.0))
