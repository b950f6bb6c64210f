"""Device-backed equivalents of the reference's tensor helpers (g3py/libs/tensors.py).

`makefn` is the compiled-method object the reference installs per process
(tensors.py:35-74): same call signature, `.executed` counter and `.clone(bijection)`;
instead of a Theano function it holds a Python closure that drives libg3hip.
`cholesky_robust`, `tt_to_num`, `tt_to_cov`, `tt_to_bounded`, `solve_lower_triangular` take
and return NumPy arrays and run on the GPU (no CPU fallback).
"""
import numpy as np

from . import clone
from ..device import Device
from .._lib import roundup


class makefn:
    """tensors.py:35-74.  `fn(space, inputs, outputs, vector, params)` replaces the compiled
    Theano graph; `th_vars` / `givens` are kept only as labels for introspection."""

    def __init__(self, th_vars, fn, givens=None, bijection=None, precompile=False):
        self.th_vars = th_vars
        self.fn = fn
        self.givens = givens
        self.bijection = bijection
        self.compiled = fn if precompile else None
        self.executed = 0

    def __call__(self, params, space=None, inputs=None, outputs=None, vector=[]):
        self.executed += 1
        if self.compiled is None:
            self.compiled = self.fn
        if self.bijection is not None:
            params = self.bijection(params)
        return self.compiled(space, inputs, outputs, vector, params)

    def clone(self, bijection=None):
        r = clone(self)
        r.bijection = bijection
        return r


def _as2d(a, dtype):
    a = np.asarray(a, dtype=dtype)
    if a.ndim != 2:
        raise AssertionError('x.ndim == 2')          # tensors.py:194
    return a


def tt_to_num(r, device=None):
    """NaN -> 0, +-Inf -> 1e10 (tensors.py:90-92) on the device."""
    r = np.asarray(r)
    dev = device or Device.default()
    shape = r.shape
    a = dev.upload(r.reshape(1, -1) if r.ndim != 2 else r)
    dev.scrub(a, a.rows, a.cols)
    return dev.download(a).reshape(shape)


def tt_to_cov(c, device=None):
    """tt_to_num + diagonal lift when min(diag) <= 0 (tensors.py:95-98)."""
    c = _as2d(c, np.asarray(c).dtype)
    dev = device or Device.default()
    a = dev.upload(c)
    dev.scrub(a, a.rows, a.cols)
    dev.cov_lift(a, a.rows)
    return dev.download(a)


def tt_to_bounded(r, lower=None, upper=None):
    """clamp (tensors.py:101-108); O(n) host arithmetic on vectors the device returned"""
    r = np.asarray(r)
    if lower is None and upper is None:
        return r
    if lower is None:
        return np.where(r > upper, upper, r)
    if upper is None:
        return np.where(r < lower, lower, r)
    return np.where(r < lower, lower, np.where(r > upper, upper, r))


class CholeskyRobust:
    """The Op protocol of tensors.py:174-263 reduced to what the path uses: calling the
    object, `perform(node, inputs, output_storage)` and `infer_shape`.  The reverse-mode `grad`
    (tensors.py:224-260) has no symbolic graph to act on here: the product path evaluates the
    gradient of logp in closed form on the device (`GaussianProcess.th_dlogp`, `g3_gp_dlogp`)."""
    __props__ = ('lower', 'destructive')

    def __init__(self, device=None):
        self.lower = True
        self.destructive = False
        self.maxtries = 20
        self.device = device
        self.last = None

    def infer_shape(self, node, shapes):
        return [shapes[0]]

    def _cholesky(self, K):
        K = _as2d(K, np.asarray(K).dtype if np.asarray(K).dtype in (np.float32, np.float64) else np.float64)
        dev = self.device or Device.default()
        n = K.shape[0]
        if n == 0:
            return K.copy()
        Kd = dev.upload(K)
        Ld = dev.alloc(n, n, K.dtype)
        tries, fallback, jitter = dev.potrf_robust(Kd, Ld, n, self.maxtries)
        self.last = dict(tries=tries, fallback=fallback, jitter=jitter)
        return dev.download(Ld)

    def perform(self, node, inputs, outputs):
        outputs[0][0] = self._cholesky(inputs[0])

    def __call__(self, x):
        return self._cholesky(x)

    def grad(self, inputs, gradients):
        raise NotImplementedError('no symbolic graph: use GaussianProcess.dlogp (g3_gp_dlogp), which evaluates the '
                                  'chain through CholeskyRobust.grad (tensors.py:224-260) in closed form')


cholesky_robust = CholeskyRobust()


def solve_lower_triangular(L, b, device=None):
    """solve L x = b for lower-triangular L (tensors.py:265-270) through g3_trsm_rlt."""
    L = _as2d(L, np.asarray(L).dtype)
    b = np.asarray(b, dtype=L.dtype)
    vec = b.ndim == 1
    B = (b[None, :] if vec else b.T)
    dev = device or Device.default()
    n, m = L.shape[0], B.shape[0]
    npad, mpad = roundup(n), roundup(m, 128)
    Lp = np.eye(npad, dtype=L.dtype)
    Lp[:n, :n] = np.tril(L)
    Ld = dev.upload(Lp)
    Bd = dev.upload(B, pad_rows=mpad, pad_cols=npad)
    dev.trsm_rlt(Ld, npad, Bd, mpad)
    X = dev.download(Bd, m, n)
    return X[0] if vec else X.T
