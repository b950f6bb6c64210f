"""GPU parity of the Gram kernels against the oracle's golden fixtures and the live oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ZOO = ['SE', 'OU', 'MAT32', 'MAT52', 'RQ', 'COS', 'SIN', 'SINC', 'SM', 'WN', 'MAT52+COS', 'SE*COS', '2*SE+0.1',
       '(SE+OU)*(MAT32+0.5)', 'SE[dims]']


def _zoo(d):
    from oracle.gen_golden import kernel_zoo
    return kernel_zoo(d)


def _gram(dev, spec, X1, X2=None, dtype=np.float64, flags=0, pad=None):
    from g3py_amd.device import compile_spec
    X1 = np.asarray(X1, dtype=dtype)
    d = X1.shape[1]
    prog = compile_spec(spec, d)
    A = dev.upload(X1)
    B = dev.upload(np.asarray(X2, dtype=dtype)) if X2 is not None else None
    n1 = X1.shape[0]
    n2 = n1 if X2 is None else len(X2)
    p1, p2 = (pad or n1), (pad or n2) if X2 is None else n2
    out = dev.alloc(p1, p2, dtype, zero=True)
    dev.gram(prog, A, B, d, out, p1, p2, flags)
    return dev.download(out)


@pytest.fixture(scope='module')
def dev():
    import g3py_amd as g3
    return g3.Device.default()


@pytest.mark.parametrize('d', [1, 3, 8])
@pytest.mark.parametrize('name', ZOO)
def test_gram_matches_golden(dev, golden_dir, d, name):
    g = np.load(os.path.join(golden_dir, 'oracle_kernels.npz'))
    X, Xs = g['d%d_X' % d], g['d%d_Xs' % d]
    spec = _zoo(d)[name]
    got = _gram(dev, spec, X)
    ref = g['d%d_%s_sym' % (d, name)]
    np.testing.assert_allclose(got, ref, rtol=2e-12, atol=1e-13 * max(1.0, np.abs(ref).max()))
    got = _gram(dev, spec, Xs, X)
    ref = g['d%d_%s_cross' % (d, name)]
    np.testing.assert_allclose(got, ref, rtol=2e-12, atol=1e-13 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize('name', ['SE', 'MAT52+COS', 'SM'])
def test_gram_fp32(dev, name):
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(11)
    X = rng.uniform(0, 3, (70, 3))
    spec = _zoo(3)[name]
    got = _gram(dev, spec, X, dtype=np.float32)
    ref = orc.kernel_cov(spec, X)
    np.testing.assert_allclose(got, ref, rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize('n', [1, 2, 63, 65, 130, 257])
def test_gram_ragged_sizes_and_padding(dev, n):
    import g3py_amd._lib as lib
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(n)
    X = rng.uniform(0, 2, (n, 4))
    spec = orc.with_noise(('SE', 1.0, np.ones(4), None), 0.1)       # the SE fast path
    pad = lib.roundup(n)
    K = _gram(dev, spec, X, flags=lib.G3_GRAM_PAD_EYE | lib.G3_GRAM_SCRUB, pad=pad)
    np.testing.assert_allclose(K[:n, :n], orc.kernel_cov(spec, X), rtol=1e-13)
    np.testing.assert_array_equal(K[n:, n:], np.eye(pad - n))
    assert not K[:n, n:].any() and not K[n:, :n].any()
    # LOWER: tiles strictly above the diagonal are left untouched (zero-initialised here)
    Kl = _gram(dev, spec, X, flags=lib.G3_GRAM_LOWER, pad=pad)
    np.testing.assert_allclose(np.tril(Kl[:n, :n]), np.tril(orc.kernel_cov(spec, X)), rtol=1e-13)


def test_gram_interior_tiles_equal_edge_tiles_and_the_oracle(dev):
    """round 5 (profiles/r05_gram.md): tiles that lie wholly inside the matrix take a loop without per-element control flow
    (noise by select in the tiles the diagonal crosses, tt_to_num as one test per eight values).  The same elements
    computed by the interior loop (even leading dimension, N = 391: six full tile rows, three full tile columns) and by
    the general loop (odd leading dimension: no paired stores, so every tile takes it) agree to rounding -- same
    formulas, but the compiler places its fused multiply-adds per loop: 1e-14 in fp64, the fp32 tolerance in fp32 --
    and both equal the oracle; a NaN input inside an interior tile is scrubbed as tensors.py:90-92 prescribes; the
    row-block entry point of the multi-GPU driver puts the noise on the TRUE diagonal"""
    import ctypes as C
    import g3py_amd._lib as lib
    from g3py_amd.device import compile_spec
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(5)
    n, m = 391, 100
    for d, spec in ((4, orc.with_noise(('SE', 1.3, np.linspace(0.6, 1.4, 4), None), 0.1)),
                    (2, orc.with_noise(('sum', ('MAT52', 1.0, np.array([0.7, 1.1]), None),
                                        ('COS', 0.5, np.array([0.2, 0.3]), None)), 0.2)),
                    (8, orc.with_noise(('sum', ('MAT52', 1.0, np.linspace(0.5, 1.2, 8), None),
                                        ('COS', 0.5, np.full(8, 0.125), None)), 0.1)),
                    (3, ('OU', 0.8, np.array([0.5, 1.0, 1.5]), None))):
        X, Xs = rng.uniform(0, 3, (n, d)), rng.uniform(0, 3, (200, d))
        pad = lib.roundup(n)
        for dtype, tol in ((np.float64, 1e-12), (np.float32, 3e-4)):
            before = dev.gram_path_stats()['table']
            same = dict(rtol=1e-14, atol=1e-15) if dtype == np.float64 else dict(rtol=tol, atol=tol * 0.1)
            big, small = _gram(dev, spec, X, dtype=dtype), _gram(dev, spec, X[:m], dtype=dtype)    # ld = 391: general loop
            np.testing.assert_array_equal(big[:m, :m], small)
            np.testing.assert_allclose(big, orc.kernel_cov(spec, X), rtol=tol, atol=tol * 0.1)
            cross = _gram(dev, spec, Xs, X, dtype=dtype)
            np.testing.assert_allclose(cross, orc.kernel_cov(spec, Xs, X), rtol=tol, atol=tol * 0.1)
            even = _gram(dev, spec, X, dtype=dtype, pad=pad)[:n, :n]                                # ld = 512: interior loop
            np.testing.assert_allclose(even, big, **same)
            np.testing.assert_allclose(even, orc.kernel_cov(spec, X), rtol=tol, atol=tol * 0.1)
            np.testing.assert_array_equal(even[:128, :128], _gram(dev, spec, X[:128], dtype=dtype))  # N = 128: interior loop too
            low = _gram(dev, spec, X, dtype=dtype, flags=lib.G3_GRAM_LOWER | lib.G3_GRAM_PAD_EYE | lib.G3_GRAM_SCRUB, pad=pad)
            np.testing.assert_array_equal(np.tril(low[:n, :n]), np.tril(even))
            np.testing.assert_array_equal(low[n:, n:], np.eye(pad - n))
            assert dev.gram_path_stats()['table'] == before + 6       # all of it on the compile-time table
        # a NaN input in the middle of an interior tile: row / column 70 of the covariance are NaN -> 0 (tt_to_num)
        Xn = X.copy()
        Xn[70, 0] = np.nan
        with np.errstate(all='ignore'):
            ref = orc.tt_to_num(orc.kernel_cov(spec, Xn))
        got = _gram(dev, spec, Xn, flags=lib.G3_GRAM_SCRUB, pad=pad)[:n, :n]
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)
        assert not got[70, :70].any() and np.isnan(_gram(dev, spec, Xn, pad=pad)[70, 3])
        # rows [128, 256) of the square covariance through g3_gram_rows (noise where i + row0 == j)
        prog = compile_spec(spec, d)
        A = dev.upload(X)
        out = dev.alloc(128, 256, np.float64, zero=True)
        assert dev.lib.g3_gram_rows(dev.ctx, C.byref(prog), A.ptr, n, A.ld, d, 128, 128, lib.dtype_code(np.float64), out.ptr, out.ld,
                                    lib.G3_GRAM_PAD_EYE) == 0
        np.testing.assert_array_equal(dev.download(out), _gram(dev, spec, X, pad=pad)[128:256, :256])


def test_gram_scrub_fuses_tt_to_num(dev):
    import g3py_amd._lib as lib
    from oracle import g3_oracle as orc
    X = np.array([[0.0], [1.0], [np.nan], [np.inf]])
    spec = ('SE', 1.0, np.array([1.0]), None)
    with np.errstate(all='ignore'):
        ref = orc.tt_to_num(orc.kernel_cov(spec, X))
    got = _gram(dev, spec, X, flags=lib.G3_GRAM_SCRUB)
    np.testing.assert_allclose(got, ref)


def test_kernel_objects_cov_and_algebra(dev):
    """Kernel.cov / operators (kernels.py:45-75,106-110) through the product classes"""
    import g3py_amd as g3
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(2)
    X, X2 = rng.uniform(0, 3, (50, 2)), rng.uniform(0, 3, (20, 2))
    r = np.array([0.7, 1.1])
    k = 2.0 * g3.SE(X, var=1.3, metric=g3.ARD_L2(X, rate=r)) + g3.COS(X, var=0.4, freq=np.array([0.2, 0.3])) * \
        g3.OU(X, var=1.0, metric=g3.ARD_L1(X, rate=r)) + 0.25
    spec = ('shift', 0.25, ('sum', ('scale', 2.0, ('SE', 1.3, r, None)),
                            ('prod', ('COS', 0.4, np.array([0.2, 0.3]), None), ('OU', 1.0, r, None))))
    np.testing.assert_allclose(k.cov(X), orc.kernel_cov(spec, X), rtol=1e-12)
    np.testing.assert_allclose(k(X2, X), orc.kernel_cov(spec, X2, X), rtol=1e-12)
    kn = g3.KernelSum(g3.SE(X, var=1.0, metric=g3.ARD_L2(X, rate=r)), g3.KernelNoise(name='Noise', var=0.3))
    np.testing.assert_allclose(np.diag(kn.cov(X)) - np.diag(kn.cov(X, X)), 0.3)     # noise only when square
    assert 'SE' in str(k) and isinstance(3 * g3.SE(X), g3.KernelScale) and isinstance(g3.SE(X) * g3.OU(X), g3.KernelProd)


def test_gram_diag(dev):
    from oracle import g3_oracle as orc
    from g3py_amd.device import compile_spec
    rng = np.random.default_rng(4)
    X = rng.uniform(0, 3, (77, 3))
    spec = orc.with_noise(_zoo(3)['MAT52+COS'], 0.2)
    out = dev.alloc(1, 77, np.float64)
    dev.gram_diag(compile_spec(spec, 3), dev.upload(X), 3, out)
    np.testing.assert_allclose(dev.download(out)[0], np.diag(orc.kernel_cov(spec, X)), rtol=1e-13)


# ------------------------------------------------------------------ kernel-parameter gradient sums
def _grad_zoo(d):
    r = np.linspace(0.6, 1.4, d)
    f = np.linspace(0.11, 0.23, d)
    return {
        'SE+noise': ('sum', ('SE', 1.3, r, None), ('NOISE', 0.1)),
        'OU': ('OU', 0.9, r, None),
        'MAT32': ('MAT32', 1.1, r, None),
        'MAT52+COS': ('sum', ('MAT52', 1.0, r, None), ('COS', 0.5, f, None)),
        'RQ': ('RQ', 1.2, r, 1.7, None),
        'SIN': ('SIN', 0.5, f, 0.25 * r, None),
        'SINC': ('SINC', 1.4, f, None),
        'SM+WN': ('sum', ('SM', 0.9, f, 0.3 * r, None), ('WN', 0.4, None)),
        'SE*COS': ('prod', ('SE', 1.0, r, None), ('COS', 1.0, f, None)),
        '(SE+OU)*(MAT32+0.5)': ('prod', ('sum', ('SE', 1.0, r, None), ('OU', 0.5, r, None)),
                                ('shift', 0.5, ('MAT32', 0.7, r, None))),
        '2*SE[dims]+0.1': ('shift', 0.1, ('scale', 2.0, ('SE', 1.0, r[:2], np.array([0, d - 1])))),
    }


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(_grad_zoo(3)))
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_gram_grad_matches_oracle(dev, name, dtype):
    """g3_gram_grad: out[slot] = 1/2 sum_ij (alpha_i alpha_j - G_ij) dK_ij/dparam against the
    oracle's dK/dparam tensors, for every leaf parameter of every kernel family"""
    from g3py_amd.device import compile_spec
    from oracle import g3_oracle as orc
    N, d = 150, 3
    rng = np.random.default_rng(5)
    X = rng.uniform(0, 4, (N, d))
    X[7] = X[3]                                   # a coincident pair (WN / SINC special cases)
    spec = _grad_zoo(d)[name]
    A = rng.standard_normal((N, N))
    G = (A + A.T) / 2
    alpha = rng.standard_normal(N)
    K, grads = orc.kernel_cov_grads(spec, X)
    Gfull = np.outer(alpha, alpha) - G
    prog = compile_spec(spec, d)
    gmap = dev.grad_layout(prog)
    Xd, Gd, ad = dev.upload(X.astype(dtype)), dev.upload(np.tril(G).astype(dtype)), dev.upload(alpha.astype(dtype))
    out = dev.gram_grad(prog, gmap, Xd, N, d, Gd, ad)
    assert len(out) == len(grads)
    tol = 1e-10 if dtype == np.float64 else 2e-4
    for (leaf, pname, k, dK) in grads:
        slot = getattr(gmap, pname)[leaf] + (0 if k is None else k)
        want = 0.5 * np.sum(Gfull * dK)
        scale = 0.5 * np.sum(np.abs(Gfull * dK)) + 1e-30
        assert abs(out[slot] - want) < tol * scale, (name, leaf, pname, k, out[slot], want)


@pytest.mark.gpu
def test_gram_grad_many_slots_and_ragged(dev):
    """more parameters than one accumulation window (32 slots) and N not a multiple of the tile"""
    from g3py_amd.device import compile_spec
    from oracle import g3_oracle as orc
    N, d = 203, 12
    rng = np.random.default_rng(9)
    X = rng.uniform(0, 2, (N, d))
    r, f = np.linspace(0.3, 0.6, d), np.linspace(0.05, 0.1, d)
    spec = ('sum', ('prod', ('SE', 1.0, r, None), ('SM', 0.7, f, 0.2 * r, None)), ('sum', ('RQ', 0.5, r, 2.0, None), ('NOISE', 0.2)))
    A = rng.standard_normal((N, N))
    G = (A + A.T) / 2
    alpha = rng.standard_normal(N)
    K, grads = orc.kernel_cov_grads(spec, X)
    prog = compile_spec(spec, d)
    gmap = dev.grad_layout(prog)
    assert gmap.nslots == len(grads) > 32
    out = dev.gram_grad(prog, gmap, dev.upload(X), N, d, dev.upload(np.tril(G)), dev.upload(alpha))
    Gfull = np.outer(alpha, alpha) - G
    for (leaf, pname, k, dK) in grads:
        slot = getattr(gmap, pname)[leaf] + (0 if k is None else k)
        want, scale = 0.5 * np.sum(Gfull * dK), 0.5 * np.sum(np.abs(Gfull * dK)) + 1e-30
        assert abs(out[slot] - want) < 1e-10 * scale
    out2 = dev.gram_grad(prog, gmap, dev.upload(X), N, d, dev.upload(np.tril(G)), dev.upload(alpha))
    assert np.array_equal(out, out2)              # fixed-order reduction: bitwise reproducible


@pytest.mark.parametrize('per', ['SIN', 'SM'])
@pytest.mark.parametrize('d', [1, 2, 4, 8])
@pytest.mark.parametrize('stat', ['SE', 'MAT52'])
def test_gram_fast_path_stationary_plus_sin_or_sm(dev, per, d, stat):
    """the other periodic leaves as the second term of the compile-time Gram variants: SIN with the reference's
    POSITIVE exponent (kernels.py:471-472) and SM (kernels.py:486-487), square (with noise) and cross"""
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(7 * d + len(stat) + len(per))
    n, m = 190, 70
    X = rng.uniform(0, n ** (1.0 / d), (n, d))
    Xs = rng.uniform(0, n ** (1.0 / d), (m, d))
    rate, freq, prate = rng.uniform(0.5, 1.5, d), rng.uniform(0.05, 0.4, d), rng.uniform(0.05, 0.3, d)
    spec = orc.with_noise(('sum', (stat, 1.3, rate, None), (per, 0.5, freq, prate, None)), 0.1)
    for dtype, rtol, atol in ((np.float64, 1e-12, 1e-13), (np.float32, 3e-4, 3e-5)):
        np.testing.assert_allclose(_gram(dev, spec, X, dtype=dtype), orc.kernel_cov(spec, X), rtol=rtol, atol=atol)
        np.testing.assert_allclose(_gram(dev, spec, Xs, X, dtype=dtype), orc.kernel_cov(spec, Xs, X), rtol=rtol, atol=atol)


@pytest.mark.parametrize('per', ['COS', 'SIN', 'SM'])
@pytest.mark.parametrize('d', [1, 4, 8])
@pytest.mark.parametrize('order', [0, 1])
def test_gram_fast_path_locally_periodic_product(dev, per, d, order):
    """KernelProd of a stationary and a periodic leaf (kernels.py:225-226) -- the locally periodic form -- in the
    compile-time Gram variants, either factor first, with and without the noise term, square and cross"""
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(5 * d + len(per) + order)
    n, m = 180, 66
    X = rng.uniform(0, n ** (1.0 / d), (n, d))
    Xs = rng.uniform(0, n ** (1.0 / d), (m, d))
    rate, freq, prate = rng.uniform(0.5, 1.5, d), rng.uniform(0.05, 0.4, d), rng.uniform(0.05, 0.3, d)
    a = ('MAT52', 1.3, rate, None)
    b = (per, 0.5, freq, None) if per == 'COS' else (per, 0.5, freq, prate, None)
    core = ('prod', a, b) if order == 0 else ('prod', b, a)
    for spec in (core, orc.with_noise(core, 0.1)):
        np.testing.assert_allclose(_gram(dev, spec, X), orc.kernel_cov(spec, X), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(_gram(dev, spec, Xs, X), orc.kernel_cov(spec, Xs, X), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(_gram(dev, core, X, dtype=np.float32), orc.kernel_cov(core, X), rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize('d', [1, 2, 4, 8])
@pytest.mark.parametrize('stat', ['SE', 'MAT32', 'MAT52'])
@pytest.mark.parametrize('noise', [None, 0.1])
def test_gram_fast_path_stationary_plus_periodic(dev, d, stat, noise):
    """round 3: compile-time Gram variants for  stationary + COS (+ noise)  -- the shape of BASELINE config 3's
    kernel (kernels.py:406-426, 466-467; sum: 240-241; noise on the square diagonal only: 367-371) -- against the
    live oracle, square and cross, at ragged sizes"""
    from oracle import g3_oracle as orc
    rng = np.random.default_rng(100 * d + len(stat))
    n, m = 197, 75
    X = rng.uniform(0, n ** (1.0 / d), (n, d))
    Xs = rng.uniform(0, n ** (1.0 / d), (m, d))
    rate = rng.uniform(0.5, 1.5, d)
    freq = rng.uniform(0.05, 0.4, d)
    spec = ('sum', (stat, 1.3, rate, None), ('COS', 0.5, freq, None))
    if noise is not None:
        spec = orc.with_noise(spec, noise)
    ref = orc.kernel_cov(spec, X)
    got = _gram(dev, spec, X)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)
    got = _gram(dev, spec, Xs, X)
    np.testing.assert_allclose(got, orc.kernel_cov(spec, Xs, X), rtol=1e-12, atol=1e-13)     # cross: no noise term
    got32 = _gram(dev, spec, X, dtype=np.float32)                                            # the fp32 instantiation
    np.testing.assert_allclose(got32, ref, rtol=3e-4, atol=3e-5)


def test_fast_exp_accuracy_and_range(dev):
    """the exp of the compile-time fast paths (the device library's; the own range-reduction + polynomial variant of round 3
    was measured slower and removed in round 4, profiles/r03_gram.md) against NumPy's over the whole range an SE / Matern
    exponent can take: <= 4 ulp down to the subnormal range, exact zeros below it, NaN / Inf inputs scrubbed exactly
    as tt_to_num prescribes (tensors.py:90-92)"""
    import g3py_amd._lib as lib
    from oracle import g3_oracle as orc
    n = 4096
    # d = 1, rate = sqrt(2): D_ij = (x_i - x_j)^2; x_j = 0 for the first column
    t = np.concatenate([np.linspace(0, 27.4, n - 8), [26.0, 27.2, 27.29, 27.3, 27.31, 27.5, 30.0, 40.0]])
    X = t[:, None].copy()
    spec = ('SE', 1.0, np.array([np.sqrt(2.0)]), None)
    got = _gram(dev, spec, X)[:, 0]
    w = 0.5 * np.sqrt(2.0) ** 2
    ref = np.exp(-(t * t) * w)
    big = ref > 1e-300
    rel = np.abs(got[big] - ref[big]) / ref[big]
    assert rel.max() <= 4 * np.finfo(np.float64).eps, rel.max()
    np.testing.assert_allclose(got[~big], ref[~big], rtol=1e-9, atol=5e-324)      # subnormal results: a few units in the last place
    assert got[-1] == 0.0 and got[-2] == 0.0                                       # exp(-900), exp(-1600)
    # MAT52 through the same exp
    spec5 = ('MAT52', 1.0, np.array([1.0]), None)
    np.testing.assert_allclose(_gram(dev, spec5, X)[:, 0], orc.kernel_cov(spec5, X)[:, 0], rtol=2e-15, atol=1e-300)
    # non-finite inputs: NaN -> 0 and an infinite distance -> exp(-inf) = 0 after the scrub
    Xb = X[:130].copy()
    Xb[3, 0] = np.nan
    Xb[5, 0] = np.inf
    K = _gram(dev, spec, Xb, flags=lib.G3_GRAM_SCRUB)
    Kr = orc.tt_to_num(orc.kernel_cov(spec, Xb))
    np.testing.assert_allclose(K, Kr, rtol=1e-14, atol=0)


@pytest.mark.parametrize('kind', ['SE', 'OU', 'MAT32', 'MAT52', 'RQ'])
@pytest.mark.parametrize('d', [1, 3, 4, 8])
@pytest.mark.parametrize('noise', [None, 0.2])
def test_gram_grad_fast_paths_all_stationary_kinds(dev, kind, d, noise):
    """round 3: the register fast path of g3_gram_grad for  var * k (+ noise), k any of the five stationary kinds, all
    columns in order (the shape `find_MAP` differentiates thousands of times, stochastic.py:308-309): every
    parameter slot against the oracle's dK/dparam (kernels.py:388-436 differentiated), ragged N, fp64 and fp32"""
    from g3py_amd.device import compile_spec
    from oracle import g3_oracle as orc
    N = 171
    rng = np.random.default_rng(17 * d + len(kind))
    X = rng.uniform(0, 3, (N, d))
    X[11] = X[5]                                  # a coincident pair: d = 0 (sqrt at 0 for the Matern kinds)
    r = rng.uniform(0.4, 1.3, d)
    spec = (kind, 1.3, r, 1.7, None) if kind == 'RQ' else (kind, 1.3, r, None)
    if noise is not None:
        spec = orc.with_noise(spec, noise)
    A = rng.standard_normal((N, N))
    G = (A + A.T) / 2
    alpha = rng.standard_normal(N)
    K, grads = orc.kernel_cov_grads(spec, X)
    Gfull = np.outer(alpha, alpha) - G
    prog = compile_spec(spec, d)
    gmap = dev.grad_layout(prog)
    assert gmap.nslots == len(grads)
    for dtype, tol in ((np.float64, 1e-11), (np.float32, 3e-4)):
        out = dev.gram_grad(prog, gmap, dev.upload(X.astype(dtype)), N, d, dev.upload(np.tril(G).astype(dtype)),
                            dev.upload(alpha.astype(dtype)))
        for (leaf, pname, k, dK) in grads:
            slot = getattr(gmap, pname)[leaf] + (0 if k is None else k)
            want = 0.5 * np.sum(Gfull * dK)
            scale = 0.5 * np.sum(np.abs(Gfull * dK)) + 1e-30
            assert abs(out[slot] - want) < tol * scale, (kind, d, leaf, pname, k, out[slot], want)


@pytest.mark.parametrize('stat', ['SE', 'MAT32', 'MAT52'])
@pytest.mark.parametrize('d', [1, 2, 4, 8])
@pytest.mark.parametrize('noise', [None, 0.2])
def test_gram_grad_fast_path_stationary_plus_periodic(dev, stat, d, noise):
    """the register fast path of g3_gram_grad for  stationary + COS (+ noise)  -- config 3's expression: every slot
    (var, rate_k of the stationary leaf; var, freq_k of the COS leaf, kernels.py:466-467 differentiated; noise var)
    against the oracle's dK/dparam, fp64 and fp32"""
    from g3py_amd.device import compile_spec
    from oracle import g3_oracle as orc
    N = 150
    rng = np.random.default_rng(31 * d + len(stat))
    X = rng.uniform(0, 3, (N, d))
    X[9] = X[4]
    r, f = rng.uniform(0.4, 1.3, d), rng.uniform(0.05, 0.4, d)
    spec = ('sum', (stat, 1.3, r, None), ('COS', 0.6, f, None))
    if noise is not None:
        spec = orc.with_noise(spec, noise)
    A = rng.standard_normal((N, N))
    G = (A + A.T) / 2
    alpha = rng.standard_normal(N)
    K, grads = orc.kernel_cov_grads(spec, X)
    Gfull = np.outer(alpha, alpha) - G
    prog = compile_spec(spec, d)
    gmap = dev.grad_layout(prog)
    assert gmap.nslots == len(grads)
    for dtype, tol in ((np.float64, 1e-11), (np.float32, 3e-4)):
        out = dev.gram_grad(prog, gmap, dev.upload(X.astype(dtype)), N, d, dev.upload(np.tril(G).astype(dtype)),
                            dev.upload(alpha.astype(dtype)))
        for (leaf, pname, k, dK) in grads:
            slot = getattr(gmap, pname)[leaf] + (0 if k is None else k)
            want = 0.5 * np.sum(Gfull * dK)
            scale = 0.5 * np.sum(np.abs(Gfull * dK)) + 1e-30
            assert abs(out[slot] - want) < tol * scale, (stat, d, leaf, pname, k, out[slot], want)


@pytest.mark.parametrize('per', ['COS', 'SIN', 'SM'])
@pytest.mark.parametrize('form', ['sum', 'prod'])
@pytest.mark.parametrize('d', [1, 2, 4, 8])
@pytest.mark.parametrize('stat', ['SE', 'MAT52'])
def test_gram_grad_fast_path_every_periodic_shape(dev, per, form, d, stat):
    """the gradient twin of every compile-time Gram variant: stationary (+ or *) COS / SIN / SM (+ noise).  In the product
    form each factor's derivatives carry the other factor's value (kernels.py KernelProd); SIN and SM add their own rate
    slots (kernels.py:471-472, 486-487 differentiated).  Every slot against the oracle's dK/dparam, fp64 and fp32"""
    from g3py_amd.device import compile_spec
    from oracle import g3_oracle as orc
    N = 140
    rng = np.random.default_rng(13 * d + len(stat) + 3 * len(per) + len(form))
    X = rng.uniform(0, 3, (N, d))
    X[9] = X[4]
    r, f, pr = rng.uniform(0.4, 1.3, d), rng.uniform(0.05, 0.4, d), rng.uniform(0.05, 0.3, d)
    leaf = ('COS', 0.6, f, None) if per == 'COS' else (per, 0.6, f, pr, None)
    spec = orc.with_noise((form, (stat, 1.3, r, None), leaf), 0.15)
    A = rng.standard_normal((N, N))
    G = (A + A.T) / 2
    alpha = rng.standard_normal(N)
    K, grads = orc.kernel_cov_grads(spec, X)
    Gfull = np.outer(alpha, alpha) - G
    prog = compile_spec(spec, d)
    gmap = dev.grad_layout(prog)
    assert gmap.nslots == len(grads) == 1 + d + 1 + d + (0 if per == 'COS' else d) + 1
    for dtype, tol in ((np.float64, 1e-11), (np.float32, 3e-4)):
        out = dev.gram_grad(prog, gmap, dev.upload(X.astype(dtype)), N, d, dev.upload(np.tril(G).astype(dtype)),
                            dev.upload(alpha.astype(dtype)))
        for (lf, pname, k, dK) in grads:
            slot = getattr(gmap, pname)[lf] + (0 if k is None else k)
            want = 0.5 * np.sum(Gfull * dK)
            scale = 0.5 * np.sum(np.abs(Gfull * dK)) + 1e-30
            assert abs(out[slot] - want) < tol * scale, (stat, per, form, d, lf, pname, k, out[slot], want)


def test_no_zoo_expression_is_interpreted_and_generated_equals_interpreted(golden_dir, monkeypatch):
    """every expression of the zoo runs a compile-time kernel -- the table of g3_gram.hip or one
    generated for the expression at first use (g3_gram_jit.hip) -- and the generated kernel gives what the interpreter
    gives (same formulas, same order: 1e-13)"""
    import g3py_amd as g3
    g = np.load(os.path.join(golden_dir, 'oracle_kernels.npz'))
    dev1 = g3.Device(0)
    monkeypatch.setenv('G3_GRAM_JIT', '0')
    monkeypatch.setenv('G3_GRAM_NOFAST', '1')
    dev0 = g3.Device(0)                      # this context interprets everything
    for d in (1, 3, 8):
        X, Xs = g['d%d_X' % d], g['d%d_Xs' % d]
        for name, spec in _zoo(d).items():
            a, b = _gram(dev1, spec, X), _gram(dev0, spec, X)
            np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-14 * max(1.0, np.abs(b).max()), err_msg='%s d=%d' % (name, d))
            a32 = _gram(dev1, spec, X, dtype=np.float32)
            np.testing.assert_allclose(a32, b, rtol=3e-4, atol=3e-5 * max(1.0, np.abs(b).max()), err_msg='%s d=%d fp32' % (name, d))
            a, b = _gram(dev1, spec, Xs, X), _gram(dev0, spec, Xs, X)
            np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-14 * max(1.0, np.abs(b).max()), err_msg='%s d=%d cross' % (name, d))
    s1, s0 = dev1.gram_path_stats(), dev0.gram_path_stats()
    assert s1['interpreted'] == 0 and s1['generated'] > 0 and s1['table'] > 0, s1
    assert s0['generated'] == 0 and s0['table'] == 0 and s0['interpreted'] > 0, s0
    dev0.close()
    dev1.close()


@pytest.mark.gpu
def test_gradient_sums_generated_equal_interpreted_and_no_zoo_expression_is_interpreted(monkeypatch):
    """the kernel-parameter sums of the gradient run a compile-time kernel for every zoo expression -- the table of
    g3_grad.hip or one generated for the expression's structure (g3_gram_jit.hip::g3_grad_jit) -- and the generated kernel
    gives what the interpreter gives (same formulas; the sums differ only in their order: 1e-11 of the absolute sum);
    a custom map (slots skipped / permuted) is honoured; members of a batch each get their own program"""
    import g3py_amd as g3
    from g3py_amd.device import compile_spec
    dev1 = g3.Device(0)
    monkeypatch.setenv('G3_GRAM_JIT', '0')
    dev0 = g3.Device(0)                      # table + interpreter only
    rng = np.random.default_rng(11)
    for d in (1, 3, 8):
        N = 190
        X = rng.uniform(0, 4, (N, d))
        X[9] = X[2]
        A = rng.standard_normal((N, N))
        G = np.tril((A + A.T) / 2)
        alpha = rng.standard_normal(N)
        for name, spec in _grad_zoo(d).items():
            prog = compile_spec(spec, d)
            outs = []
            for dev in (dev1, dev0):
                gmap = dev.grad_layout(prog)
                outs.append(dev.gram_grad(prog, gmap, dev.upload(X), N, d, dev.upload(G), dev.upload(alpha)))
            scale = np.abs(outs[1]).max() + 1e-30
            np.testing.assert_allclose(outs[0], outs[1], rtol=1e-9, atol=1e-11 * scale, err_msg='%s d=%d' % (name, d))
            if name == '(SE+OU)*(MAT32+0.5)':            # a custom map: only the OU leaf's rates and the MAT32 variance
                gmap = dev1.grad_layout(prog)
                full = outs[0]
                want = np.concatenate([full[gmap.rate[1]:gmap.rate[1] + d], [full[gmap.var[2]]]])
                for l in range(prog.nleaf):
                    gmap.var[l] = gmap.alpha[l] = gmap.rate[l] = gmap.freq[l] = -1
                gmap.rate[1], gmap.var[2], gmap.nslots = 0, d, d + 1
                got = dev1.gram_grad(prog, gmap, dev1.upload(X), N, d, dev1.upload(G), dev1.upload(alpha))
                np.testing.assert_array_equal(got, want)
    s1, s0 = dev1.grad_path_stats(), dev0.grad_path_stats()
    assert s1['interpreted'] == 0 and s1['generated'] > 0 and s1['table'] > 0, s1
    assert s0['generated'] == 0 and s0['interpreted'] > 0, s0
    dev0.close()
    dev1.close()
