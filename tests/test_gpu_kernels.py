"""GPU parity tests of the low-level C-ABI kernels (GEMM, Cholesky, triangular solve, scrubs)
against NumPy/SciPy on the same seeded inputs.  All calls go through libg3hip.so."""
import os

import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    import g3py_amd as g3
    return g3.Device.default()


TOL = {np.float64: 5e-13, np.float32: 2e-4}


@pytest.mark.parametrize('dt', [np.float64, np.float32])
@pytest.mark.parametrize('m,n,k,lower', [(64, 64, 64, False), (128, 64, 128, False), (192, 192, 64, True),
                                         (320, 128, 256, False), (1536, 1536, 256, True),
                                         (2048, 1536, 128, False), (4096, 4096, 96, True),
                                         (2176, 384, 512, True)])
def test_gemm_nt(dev, dt, m, n, k, lower):
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    A = rng.standard_normal((m, k)).astype(dt)
    B = rng.standard_normal((n, k)).astype(dt)      # asymmetric operands: a transposed C would show
    C = rng.standard_normal((m, n)).astype(dt)
    Ad, Bd, Cd = dev.upload(A), dev.upload(B), dev.upload(C)
    dev.gemm_nt(Cd, Ad, Bd, m, n, k, alpha=-0.5, beta=1.0, lower_only=lower)
    got = dev.download(Cd)
    ref = C.astype(np.float64) - 0.5 * A.astype(np.float64) @ B.astype(np.float64).T
    if lower:
        ref = np.where(np.tril(np.ones((m, n), bool)), ref, C)
    assert np.abs(got - ref).max() <= TOL[dt] * np.abs(ref).max() * max(1, k / 64)


@pytest.mark.parametrize('dt', [np.float64, np.float32])
@pytest.mark.parametrize('case', ['two_blocks', 'ragged_rhs', 'perm', 'big', 'empty_segment'])
def test_gemm_nt_stair(dev, dt, case):
    """staircase product (one launch): stacked row segments with their own widths, optional block
    table for B -- the per-rank trailing update of the multi-GPU factorisation"""
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    k = 256
    if case == 'two_blocks':
        seg_rows, seg_cols, nbr, perm = [256, 256], [256, 768], 0, None
    elif case == 'ragged_rhs':
        seg_rows, seg_cols, nbr, perm = [512, 512, 512, 384], [512, 1536, 2048, 2560], 0, None
    elif case == 'perm':
        seg_rows, seg_cols, nbr = [256, 256, 128], [256, 512, 1024], 256
        perm = [2, 0, 3, 1]
    elif case == 'empty_segment':
        seg_rows, seg_cols, nbr, perm = [128, 256, 128], [384, 0, 512], 0, None
    else:   # enough tiles for the 128 x 128 kernel (>= 4096 tiles of 128 x 128)
        seg_rows, seg_cols, nbr, perm = [1024] * 12, [512 * (i + 4) for i in range(12)], 512, None
        perm = list(rng.permutation(max(seg_cols) // 512))
        k = 128
    m, n = sum(seg_rows), max(seg_cols)
    A = rng.standard_normal((m, k)).astype(dt)
    B = rng.standard_normal((n, k)).astype(dt)
    C = rng.standard_normal((m, n + 64)).astype(dt)
    Ad, Bd, Cd = dev.upload(A), dev.upload(B), dev.upload(C)
    dev.gemm_nt_stair(Cd.ptr, Cd.ld, Ad.ptr, Ad.ld, Bd.ptr, Bd.ld, k, seg_rows, seg_cols, dt, alpha=-1.0, beta=1.0,
                      b_block_rows=nbr, b_perm=perm)
    got = dev.download(Cd)
    Bl = B.astype(np.float64)
    if perm is not None:
        Bl = np.concatenate([Bl[p * nbr:(p + 1) * nbr] for p in perm])
    ref = C.astype(np.float64).copy()
    r = 0
    for rows, cols in zip(seg_rows, seg_cols):
        ref[r:r + rows, :cols] -= A[r:r + rows].astype(np.float64) @ Bl[:cols].T
        r += rows
    assert np.abs(got - ref).max() <= TOL[dt] * np.abs(ref).max() * max(1, k / 64)


@pytest.mark.parametrize('dt', [np.float64, np.float32])
@pytest.mark.parametrize('case', ['small', 'big'])
def test_gemm_nt_stair_diagonal_blocks(dev, dt, case):
    """seg_diag: the trailing square block of a flagged segment is a DIAGONAL block -- its lower triangle must be
    exact, tiles entirely above it are not launched (C keeps its old value there), and the strict upper part of the
    block inside launched tiles is scratch (either the old or the updated value)"""
    rng = np.random.default_rng(3)
    if case == 'small':
        seg_rows, seg_cols, diag, k = [256, 512, 384, 128], [256, 1024, 384, 640], [1, 1, 1, 0], 128
    else:
        seg_rows, seg_cols, diag, k = [1024] * 10 + [256], [1024 * (i + 1) for i in range(10)] + [10240], [1] * 10 + [0], 128
    m, n = sum(seg_rows), max(seg_cols)
    A = rng.standard_normal((m, k)).astype(dt)
    B = rng.standard_normal((n, k)).astype(dt)
    C = rng.standard_normal((m, n)).astype(dt)
    Ad, Bd, Cd = dev.upload(A), dev.upload(B), dev.upload(C)
    dev.gemm_nt_stair(Cd.ptr, Cd.ld, Ad.ptr, Ad.ld, Bd.ptr, Bd.ld, k, seg_rows, seg_cols, dt, alpha=-1.0, beta=1.0, seg_diag=diag)
    got = dev.download(Cd).astype(np.float64)
    old = C.astype(np.float64)
    tol = TOL[dt] * np.abs(old).max() * max(1, k / 64) * 8
    r = 0
    for rows, cols, dg in zip(seg_rows, seg_cols, diag):
        new = old[r:r + rows, :cols] - A[r:r + rows].astype(np.float64) @ B[:cols].astype(np.float64).T
        g = got[r:r + rows, :cols]
        if not dg:
            assert np.abs(g - new).max() <= tol
        else:
            c0 = cols - rows
            assert np.abs(g[:, :c0] - new[:, :c0]).max() <= tol if c0 else True
            lower = np.tril(np.ones((rows, rows), bool))
            gd, nd, od = g[:, c0:], new[:, c0:], old[r:r + rows, c0:cols]
            assert np.abs(gd - nd)[lower].max() <= tol
            up = ~lower
            assert np.all((np.abs(gd - nd)[up] <= tol) | (gd[up] == od[up]))       # updated or untouched, nothing else
            if rows >= 512:
                assert np.any(gd[up] == od[up])                                   # some tiles above the diagonal were skipped
        assert np.array_equal(got[r:r + rows, cols:], old[r:r + rows, cols:])      # beyond the segment's width: untouched
        r += rows


@pytest.mark.parametrize('dt', [np.float64, np.float32])
@pytest.mark.parametrize('lower', [False, True])
def test_gemm_unaligned_c_takes_the_scalar_epilogue(dev, dt, lower):
    """C with an odd leading dimension / offset start is not 16-byte aligned row by row: the kernel
    falls back from the vectorised LDS epilogue to element-wise stores; a sub-block of a wider C and
    beta = 0 are covered on the way"""
    m, n, k, ld = 192, 128, 64, 131
    rng = np.random.default_rng(11)
    A = rng.standard_normal((m, k)).astype(dt)
    B = rng.standard_normal((n, k)).astype(dt)
    C = rng.standard_normal((m, ld)).astype(dt)
    Ad, Bd, Cd = dev.upload(A), dev.upload(B), dev.upload(C)
    dev.gemm_nt(Cd, Ad, Bd, m, n, k, alpha=2.0, beta=-1.0, lower_only=lower, c_off=np.dtype(dt).itemsize)   # starts at column 1
    got = dev.download(Cd)
    ref = C.astype(np.float64)
    upd = 2.0 * A.astype(np.float64) @ B.astype(np.float64).T - ref[:, 1:1 + n]
    mask = np.tril(np.ones((m, n), bool)) if lower else np.ones((m, n), bool)
    ref[:, 1:1 + n] = np.where(mask, upd, ref[:, 1:1 + n])
    assert np.abs(got - ref).max() <= TOL[dt] * np.abs(ref).max() * 4
    C0 = dev.upload(np.full((m, n), np.nan, dtype=dt))          # beta = 0 must not read C
    dev.gemm_nt(C0, Ad, Bd, m, n, k, alpha=1.0, beta=0.0)
    assert np.abs(dev.download(C0) - A.astype(np.float64) @ B.astype(np.float64).T).max() <= TOL[dt] * 50


def test_gemm_identity_layout(dev):
    # A = I with asymmetric B catches a row<->col swap of the MFMA accumulator layout
    n = 128
    B = np.arange(n * n, dtype=np.float64).reshape(n, n)
    Cd = dev.alloc(n, n, np.float64, zero=True)
    dev.gemm_nt(Cd, dev.upload(np.eye(n)), dev.upload(B), n, n, n)
    np.testing.assert_array_equal(dev.download(Cd), B.T)


def test_gemm_argument_errors(dev):
    import g3py_amd as g3
    a = dev.alloc(64, 64, np.float64)
    with pytest.raises(g3.G3Error):
        dev.gemm_nt(a, a, a, 60, 64, 64)        # m not a tile multiple
    with pytest.raises(g3.G3Error):
        dev.gemm_nt(a, a, a, 64, 64, 8)         # k not a multiple of 16


def _spd(rng, n, dt=np.float64):
    B = rng.standard_normal((n, max(8, n // 4)))
    return (B @ B.T / B.shape[1] + np.eye(n)).astype(dt)


@pytest.mark.parametrize('n', [128, 256, 384, 640, 1024, 2048, 3072, 4224])
def test_potrf_matches_lapack(dev, n):
    rng = np.random.default_rng(n)
    K = _spd(rng, n)
    Kd = dev.upload(K)
    assert dev.potrf(Kd, n) == 0
    L = np.tril(dev.download(Kd))
    Lr = scipy.linalg.cholesky(K, lower=True)
    assert np.abs(L - Lr).max() < 1e-12 * n ** 0.5
    # strict upper triangle is never touched
    np.testing.assert_array_equal(np.triu(dev.download(Kd), 1), np.triu(K, 1))


def test_potrf_fp32(dev):
    rng = np.random.default_rng(5)
    K = _spd(rng, 512, np.float32)
    Kd = dev.upload(K)
    assert dev.potrf(Kd, 512) == 0
    L = np.tril(dev.download(Kd)).astype(np.float64)
    assert np.abs(L @ L.T - K).max() < 5e-5


def test_potrf_info_on_bad_pivot(dev):
    rng = np.random.default_rng(1)
    K = _spd(rng, 512)
    K[300, 300] = -5.0
    assert dev.potrf(dev.upload(K), 512) == 301          # LAPACK-style 1-based pivot index
    K2 = _spd(rng, 256)
    K2[10, 3] = K2[3, 10] = np.nan
    assert dev.potrf(dev.upload(K2), 256) != 0


@pytest.mark.parametrize('n', [128, 256, 768])
def test_potrf_info_is_the_first_failing_pivot(dev, n):
    """dpotrf's info (tensors.py:198 reads it): the 1-based column of the first non-positive pivot, wherever it
    falls inside the 16 x 16 tile routine's 4-column block steps or the 128 / 256-wide diagonal kernels"""
    rng = np.random.default_rng(n)
    base = _spd(rng, n)
    for col in sorted({0, 1, 3, 4, 6, 15, 16, 21, 127, n // 2 + 5, n - 130 + 3, n - 1} & set(range(n))):
        K = base.copy()
        K[col, col] = -1.0
        ref = scipy.linalg.lapack.dpotrf(K, lower=True)[1]
        assert ref == col + 1
        assert dev.potrf(dev.upload(K), n) == ref, col
    # positive semi-definite: the Schur complement of a duplicated row is exactly zero -> fails at that row
    K = base.copy()
    j, i = 40, min(n - 1, 200)
    K[i, :] = K[j, :]
    K[:, i] = K[:, j]
    K[i, i] = K[j, j]
    info = dev.potrf(dev.upload(K), n)
    assert info == i + 1 or info == 0        # (zero pivot up to rounding: either detected at row i or passed as tiny positive)
    # two bad pivots: the first one is reported
    K = base.copy()
    K[70, 70] = K[9, 9] = -2.0
    assert dev.potrf(dev.upload(K), n) == 10


@pytest.mark.parametrize('n,m', [(128, 128), (384, 256), (1024, 128), (2048, 384)])
def test_trsm_rlt(dev, n, m):
    rng = np.random.default_rng(n + m)
    L = scipy.linalg.cholesky(_spd(rng, n), lower=True)
    B = rng.standard_normal((m, n))
    Bd = dev.upload(B)
    dev.trsm_rlt(dev.upload(L), n, Bd, m)
    X = dev.download(Bd)
    ref = scipy.linalg.solve_triangular(L, B.T, lower=True).T
    assert np.abs(X - ref).max() < 1e-11 * np.abs(ref).max()


@pytest.mark.parametrize('dt', [np.float64, np.float32])
@pytest.mark.parametrize('n,m', [(128, 64), (256, 192), (512, 320), (1024, 1088), (2048, 128)])
def test_full_inverse_of_a_diagonal_factor_and_the_one_product_solve(dev, dt, n, m):
    """g3_trtri_full: V = L^-1 by recursive doubling from the 128-block inverses; g3_trsm_full: X L^T = B as the ONE
    K-triangular product B V^T -- what the multi-GPU driver does with a broadcast diagonal factor (tensors.py:265-270)"""
    import ctypes as C
    from g3py_amd import _lib
    rng = np.random.default_rng(n + m)
    L = scipy.linalg.cholesky(_spd(rng, n), lower=True)
    Ld = dev.upload(L.astype(dt))
    code = _lib.dtype_code(dt)
    # the block inverses: factor K = L L^T on the device (leaves L and inv(L_kk))
    Kd = dev.upload((L @ L.T).astype(dt))
    W = dev.alloc_inverses(n, dt)
    info = C.c_int(0)
    assert dev.lib.g3_potrf(dev.ctx, Kd.ptr, n, Kd.ld, code, W.ptr, C.byref(info)) == 0 and info.value == 0
    V, Vt, U = (dev.alloc(n, n, dt, zero=True) for _ in range(3))
    assert dev.lib.g3_trtri_full(dev.ctx, Kd.ptr, n, W.ptr, V.ptr, Vt.ptr, U.ptr, code) == 0
    Lg = np.tril(dev.download(Kd).astype(np.float64))
    Vh = dev.download(V).astype(np.float64)
    assert np.abs(np.triu(Vh, 1)).max() == 0.0
    ref = np.linalg.inv(Lg)
    tol = (5e-13 if dt == np.float64 else 3e-4) * np.linalg.cond(Lg)
    assert np.abs(Vh - ref).max() <= tol * np.abs(ref).max()
    if n > 128:      # the transpose is carried up to the last level's off-diagonal block
        h = n // 2
        np.testing.assert_array_equal(dev.download(Vt)[:h, :h], Vh[:h, :h].T.astype(dt))
    B = rng.standard_normal((m, n)).astype(dt)
    Bd, Xd = dev.upload(B), dev.alloc(m, n, dt, zero=True)
    assert dev.lib.g3_trsm_full(dev.ctx, V.ptr, n, n, Bd.ptr, m, n, Xd.ptr, n, code) == 0
    X = dev.download(Xd).astype(np.float64)
    want = scipy.linalg.solve_triangular(Lg, B.astype(np.float64).T, lower=True).T
    assert np.abs(X - want).max() <= tol * np.abs(want).max()
    # argument checks: a block size that is not 128 * 2^q, aliasing
    assert dev.lib.g3_trtri_full(dev.ctx, Kd.ptr, 384, W.ptr, V.ptr, Vt.ptr, U.ptr, code) == -3
    assert dev.lib.g3_trsm_full(dev.ctx, V.ptr, n, n, Bd.ptr, m, n, Bd.ptr, n, code) == -8


def test_potrf_robust_follows_reference_schedule(dev, golden_dir):
    """CholeskyRobust (g3py/libs/tensors.py:197-222): jitter schedule, lift of non-positive
    diagonals and the 1e-10*I fallback, against oracle fixtures."""
    g = np.load(os.path.join(golden_dir, 'oracle_jitter.npz'))
    for i in (1, 2, 3, 4):
        K = g['K%d' % i]
        n = K.shape[0]
        Kd, Ld = dev.upload(K), dev.alloc(n, n, np.float64)
        tries, fallback, jitter = dev.potrf_robust(Kd, Ld, n)
        assert tries == int(g['tries%d' % i]) and fallback == bool(g['fallback%d' % i])
        L = dev.download(Ld)
        np.testing.assert_allclose(L, g['L%d' % i], rtol=1e-7, atol=1e-9)
        np.testing.assert_array_equal(dev.download(Kd), K)      # non-destructive


def test_scrub_and_cov_lift(dev):
    a = np.array([[np.nan, np.inf, 1.0], [-np.inf, 2.0, 3.0], [0.5, -1.0, 4.0]])
    d = dev.upload(a)
    dev.scrub(d, 3, 3)
    r = dev.download(d)
    assert r[0, 0] == 0 and r[0, 1] == np.float32(1e10) and r[1, 0] == np.float32(1e10)
    c = np.array([[-0.5, 0.1], [0.1, 1.0]])
    d = dev.upload(c)
    dev.cov_lift(d, 2)
    np.testing.assert_allclose(np.diag(dev.download(d)), [np.float32(1e-6), 1.5 + np.float32(1e-6)])
    d = dev.upload(np.eye(3) * 2)
    dev.cov_lift(d, 3)
    np.testing.assert_array_equal(dev.download(d), np.eye(3) * 2)


def test_reference_named_tensor_helpers():
    """the device-backed equivalents of tensors.py helpers keep the reference's semantics"""
    import g3py_amd as g3
    rng = np.random.default_rng(3)
    K = _spd(rng, 100)
    L = g3.cholesky_robust(K)
    np.testing.assert_allclose(L, scipy.linalg.cholesky(K, lower=True), atol=1e-12)
    b = rng.standard_normal(100)
    np.testing.assert_allclose(g3.solve_lower_triangular(L, b), scipy.linalg.solve_triangular(L, b, lower=True),
                               atol=1e-11)
    with pytest.raises(AssertionError):
        g3.cholesky_robust(np.ones(4))                         # tensors.py:194 (2-D only)
    out = [[None]]
    g3.cholesky_robust.perform(None, [K], out)                 # Theano Op protocol
    np.testing.assert_allclose(out[0][0], L)
    assert g3.cholesky_robust.infer_shape(None, [(100, 100)]) == [(100, 100)]
    np.testing.assert_array_equal(g3.tt_to_num(np.array([np.nan, np.inf, 1.0])), [0, np.float32(1e10), 1.0])


# ------------------------------------------------------------------ K^-1 from the factor (dlogp path)
@pytest.mark.gpu
@pytest.mark.parametrize('n,dtype', [(128, np.float64), (384, np.float64), (1024, np.float64), (2048, np.float32)])
def test_potri_matches_lapack(dev, n, dtype):
    """g3_potri: Y = L^-T and lower(K^-1) = Y Y^T (LAPACK dpotri's result); n >= 768 takes the
    two-stream look-ahead schedule"""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    K = A.dot(A.T) / n + np.eye(n)
    L = np.linalg.cholesky(K)
    Ld = dev.upload(L.astype(dtype))
    Y = dev.alloc(n, n, dtype)
    Ki = dev.alloc(n, n, dtype)
    dev.potri(Ld, n, None, Y, Ki)
    Yh, Kih = dev.download(Y), dev.download(Ki)
    tol = 1e-10 if dtype == np.float64 else 2e-4
    Linv = np.linalg.inv(L)
    assert np.max(np.abs(Yh - Linv.T)) < tol * np.max(np.abs(Linv))
    Kinv = np.linalg.inv(K)
    assert np.max(np.abs(np.tril(Kih) - np.tril(Kinv))) < tol * np.max(np.abs(Kinv))


def test_new_entry_points_reject_bad_arguments(dev):
    """status codes (-i = bad argument i) of the gradient / batched / multi-GPU entry points"""
    import ctypes as C
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec
    lib, ctx = dev.lib, dev.ctx
    prog = compile_spec(('sum', ('SE', 1.0, np.ones(2), None), ('NOISE', 0.1)), 2)
    gmap = dev.grad_layout(prog)
    assert gmap.nslots == 4 and list(gmap.var[:2]) == [0, 3] and gmap.rate[0] == 1
    a = dev.alloc(256, 256, np.float64, zero=True)
    out = (C.c_double * 8)()
    assert lib.g3_potri(ctx, a.ptr, 100, 256, None, 0, a.ptr, 256, a.ptr, 256) == -3            # n not a multiple of 128
    assert lib.g3_potri(ctx, a.ptr, 256, 255, None, 0, a.ptr, 256, a.ptr, 256) == -4            # ld < n
    assert lib.g3_gram_grad(ctx, C.byref(prog), C.byref(gmap), a.ptr, 10, 1, 2, 0, a.ptr, 256, a.ptr, out) == -6   # ldx < d
    bad = _lib.GradMap()
    bad.nslots = 2
    bad.var[0] = 5
    assert lib.g3_gram_grad(ctx, C.byref(prog), C.byref(bad), a.ptr, 10, 2, 2, 0, a.ptr, 256, a.ptr, out) == -3    # slot outside the output
    progs = (_lib.KernelProg * 2)(prog, compile_spec(('SE', 1.0, np.ones(2), None), 2))
    rc = lib.g3_gp_factor_batched(ctx, progs, 2, a.ptr, 100, 2, 2, a.ptr, 128, 0, a.ptr, 128, 256 * 128, a.ptr, a.ptr, out)
    assert rc == -2                                                                               # members of different structure
    assert lib.g3_gp_factor_batched(ctx, progs, 0, a.ptr, 100, 2, 2, a.ptr, 128, 0, a.ptr, 128, 256 * 128, a.ptr, a.ptr, out) == -3
    assert lib.g3_gram_rows(ctx, C.byref(prog), a.ptr, 100, 2, 2, 0, 128, 0, a.ptr, 64, 0) == -11                  # ldk < row0 + nrows
    assert lib.g3_gram_rows(ctx, C.byref(prog), a.ptr, 100, 2, 2, 0, 128, 0, a.ptr, 128, _lib.G3_GRAM_LOWER) == -12
    assert lib.g3_potrf_nowait(ctx, a.ptr, 128, 256, 0, a.ptr, None) == -7


def test_device_close_is_idempotent_and_final():
    """teardown contract (g3py_amd/device.py): close() twice is fine, DeviceArray.free() after it is a no-op,
    every other call raises instead of touching a destroyed context; a second context is unaffected"""
    import g3py_amd as g3
    from g3py_amd._lib import G3Error
    other = g3.Device(0)
    d = g3.Device(0)
    a = d.alloc(256, 256, np.float64, zero=True)
    b = d.upload(np.arange(12.0).reshape(3, 4))
    assert a.ptr and b.ptr
    d.close()
    assert d.ctx is None and a.ptr == 0 and b.ptr == 0      # owned buffers were released with the context
    d.close()                                               # idempotent
    a.free()                                                # no-op, no call into HIP
    b.free()
    for call in (lambda: d.sync(), lambda: d.alloc(8, 8, np.float64), lambda: d.upload(np.zeros((2, 2))),
                 lambda: d.prof_reset()):
        with pytest.raises((G3Error, Exception)):
            call()
    x = other.upload(np.eye(128))                           # the other context still works
    assert other.potrf(x, 128) == 0
    other.close()


def test_interpreter_exit_without_explicit_close(tmp_path):
    """a process that never calls close(): the atexit hook tears the contexts down while HIP is alive (the
    round-1 exit-time SIGSEGV under a profiler); exit status 0, no fault"""
    import subprocess
    import sys
    code = ("import numpy as np, g3py_amd as g3\n"
            "d = g3.Device(0); e = g3.Device(0)\n"
            "a = d.upload(np.eye(256) * 2.0); b = e.alloc(128, 128, np.float32, zero=True)\n"
            "assert d.potrf(a, 256) == 0\n"
            "print('done')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])
    assert 'done' in r.stdout
