"""CPU oracle for the g3py GP-inference hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product (`g3py_amd`) never does.

What it is: a NumPy/SciPy restatement of the reference's formulas for the path in
SURVEY.md section 8(a), rows 1-14, using the same LAPACK entry points the reference
calls (`scipy.linalg.lapack.dpotrf`, `scipy.linalg.solve`, `solve_triangular`).
Every function cites the reference file:line it follows (paths relative to
/root/reference).

Pinning status (see DESIGN.md "Oracle"):
  * PINNED against the reference's own executable NumPy prototype `sandbox/gpmm.py`
    (imported in the build container by `oracle/gen_golden.py`; outputs committed under
    `tests/golden/gpmm_*.npz`): SE(+noise) and OU(+noise) Gram, Cholesky factor,
    log marginal likelihood, posterior mean / variance / covariance, and the gradient of logp
    (finite differences of gpmm's own NLL).
  * The main Theano/PyMC3 path cannot be imported here (ModuleNotFoundError: theano,
    pymc3 -- an ordinary error, not a permission denial) and the reference ships no
    tests or golden vectors, so for everything else (MAT32/MAT52/RQ/periodic kernels,
    warped GP, Gauss-Hermite, jitter schedule, Student-t process, transports) the restatement
    is "parity unpinned" by the reference and is pinned by analytic known-answer tests in
    tests/test_oracle.py (closed forms, limits, finite differences of logp for dlogp).

dtype: the reference graph is hard-wired float32 (g3py/config.py:4,9) with the Cholesky
done in float64 and cast back (g3py/libs/tensors.py:198,219).  `dtype=np.float64`
(default) is the fp64 restatement the build is compared with; `dtype=np.float32`
reproduces the float32-graph / fp64-dpotrf behaviour.

Kernel "spec" format (plain nested tuples, natural-space hyper-parameters):
    ('SE', var, rate[d], dims)      ('OU', var, rate[d], dims)
    ('MAT32', var, rate[d], dims)   ('MAT52', var, rate[d], dims)
    ('RQ', var, rate[d], alpha, dims)
    ('COS', var, freq[d], dims)     ('SIN', var, freq[d], rate[d], dims)
    ('SINC', var, freq[d], dims)    ('SM', var, freq[d], rate[d], dims)
    ('NOISE', var)                  ('WN', var, dims)
    ('sum', a, b) ('prod', a, b) ('scale', c, a) ('shift', c, a)
`dims` is None (all columns) or an index array (hypers/__init__.py:55-83).
"""
import numpy as np
import scipy as sp
import scipy.linalg
import scipy.linalg.lapack
from scipy import stats

pi = np.pi
pi2 = np.pi ** 2


# --------------------------------------------------------------------------- metrics
def _cols(x, dims):
    """x[:, self.dims] -- g3py/processes/hypers/metrics.py:11-13."""
    x = np.asarray(x)
    if x.ndim == 1:
        x = x[:, None]
    return x if dims is None else x[:, dims]


def _diff(x1, x2, dims):
    """Broadcast difference x1[:,None,:]-x2[None,:,:] -- metrics.py:11-13, 59-61."""
    return _cols(x1, dims)[:, None, :] - _cols(x2, dims)[None, :, :]


def ard_l2(x1, x2, rate, dims=None):
    """dot((x1-x2)**2, 0.5*rate**2) -- metrics.py:100-102."""
    d = _diff(x1, x2, dims)
    rate = np.asarray(rate, dtype=d.dtype)
    return np.dot(d ** 2, d.dtype.type(0.5) * rate ** 2)


def ard_l1(x1, x2, rate, dims=None):
    """dot(|x1-x2|, rate) -- metrics.py:89-91."""
    d = _diff(x1, x2, dims)
    return np.dot(np.abs(d), np.asarray(rate, dtype=d.dtype))


def delta(x1, x2, dims=None):
    """tt_to_num(eq(x1-x2, 0).sum(axis=2)) -- metrics.py:30-35."""
    d = _diff(x1, x2, dims)
    return tt_to_num((d == 0).sum(axis=2).astype(d.dtype))


# --------------------------------------------------------------------------- scrubs
def tt_to_num(r, nan=0.0, inf=1e10):
    """NaN -> 0, +-Inf -> 1e10 -- g3py/libs/tensors.py:90-92 (both infinities map to +1e10)."""
    r = np.asarray(r)
    t = r.dtype.type
    return np.where(np.isnan(r), t(nan), np.where(np.isinf(r), t(np.float32(inf)), r))


def tt_to_cov(c):
    """tt_to_num then lift the diagonal if min(diag) <= 0 -- tensors.py:95-98."""
    r = tt_to_num(c)
    t = r.dtype.type
    m = np.min(np.diag(r))
    if m > 0:
        return r
    return r + (t(np.float32(1e-6)) - m) * np.eye(r.shape[0], dtype=r.dtype)


def tt_to_bounded(r, lower=None, upper=None):
    """clamp -- tensors.py:101-108."""
    r = np.asarray(r)
    if lower is None and upper is None:
        return r
    if lower is None:
        return np.where(r > upper, upper, r)
    if upper is None:
        return np.where(r < lower, lower, r)
    return np.where(r < lower, lower, np.where(r > upper, upper, r))


# --------------------------------------------------------------------------- kernels
def kernel_cov(spec, x1, x2=None, dtype=np.float64):
    """Kernel.cov(x1, x2=None) for a spec tree -- kernels.py:96-110,192-244,360-487."""
    x1 = np.asarray(x1, dtype=dtype)
    if x1.ndim == 1:
        x1 = x1[:, None]
    sym = x2 is None
    xb = x1 if sym else np.asarray(x2, dtype=dtype)
    if xb.ndim == 1:
        xb = xb[:, None]
    t = np.dtype(dtype).type
    op = spec[0]
    if op == 'sum':      # kernels.py:240-241
        return kernel_cov(spec[1], x1, x2, dtype) + kernel_cov(spec[2], x1, x2, dtype)
    if op == 'prod':     # kernels.py:225-226
        return kernel_cov(spec[1], x1, x2, dtype) * kernel_cov(spec[2], x1, x2, dtype)
    if op == 'scale':    # kernels.py:196-197
        return t(spec[1]) * kernel_cov(spec[2], x1, x2, dtype)
    if op == 'shift':    # kernels.py:207-208
        return t(spec[1]) + kernel_cov(spec[2], x1, x2, dtype)
    var = t(spec[1])
    if op == 'NOISE':    # kernels.py:367-371: var*I when square, zeros for cross
        if sym:
            return var * np.eye(x1.shape[0], dtype=dtype)
        return np.zeros((x1.shape[0], xb.shape[0]), dtype=dtype)
    if op == 'WN':       # kernels.py:381-385: var*I when square, var*Delta.gram for cross
        if sym:
            return var * np.eye(x1.shape[0], dtype=dtype)
        return var * delta(x1, xb, spec[2])
    if op in ('SE', 'MAT32', 'MAT52'):
        d = ard_l2(x1, xb, np.asarray(spec[2], dtype=dtype), spec[3])
        if op == 'SE':       # kernels.py:424-426
            return var * np.exp(-d)
        if op == 'MAT32':    # kernels.py:410-412
            d3 = np.sqrt(t(3) * d)
            return var * ((t(1) + d3) * np.exp(-d3))
        d5 = np.sqrt(t(5) * d)   # kernels.py:419-421
        return var * ((t(1) + d5 + t(5) * d / t(3)) * np.exp(-d5))
    if op == 'OU':       # kernels.py:429-431 (ARD_L1 metric)
        return var * np.exp(-ard_l1(x1, xb, np.asarray(spec[2], dtype=dtype), spec[3]))
    if op == 'RQ':       # kernels.py:402-403
        d = ard_l2(x1, xb, np.asarray(spec[2], dtype=dtype), spec[4])
        alpha = t(spec[3])
        return var * np.power(t(1) + d / alpha, -alpha)
    if op == 'COS':      # kernels.py:466-467
        d = _diff(x1, xb, spec[3])
        f = np.asarray(spec[2], dtype=dtype)
        return var * np.prod(np.cos(t(2 * pi) * d * f), axis=2)
    if op == 'SIN':      # kernels.py:471-472 (positive exponent, as written)
        d = _diff(x1, xb, spec[4])
        f = np.asarray(spec[2], dtype=dtype)
        r = np.asarray(spec[3], dtype=dtype)
        return var * np.exp(t(2) * np.dot(np.sin(t(pi) * d * f) ** 2, r))
    if op == 'SINC':     # kernels.py:479-482
        d = _diff(x1, xb, spec[3])
        f = np.asarray(spec[2], dtype=dtype)
        with np.errstate(divide='ignore', invalid='ignore'):
            sinc = np.sin(t(2 * pi2) * d * f) / (t(2 * pi2) * f * d)
        return var * np.prod(np.where(d != 0, sinc, t(1)), axis=2)
    if op == 'SM':       # kernels.py:486-487
        d = _diff(x1, xb, spec[4])
        f = np.asarray(spec[2], dtype=dtype)
        r = np.asarray(spec[3], dtype=dtype)
        return var * (np.exp(t(-2 * pi2) * np.dot(d ** 2, r ** 2)) *
                      np.prod(np.cos(t(2 * pi) * d * f), axis=2))
    raise ValueError('unknown kernel spec ' + str(op))


def with_noise(spec_f, noise_var):
    """KernelSum(kernel, KernelNoise('Noise')) -- g3py/processes/elliptical.py:26-31."""
    return ('sum', spec_f, ('NOISE', noise_var))


def kernel_cov_grads(spec, x, dtype=np.float64, _leaf=None):
    """Kernel.cov(x) (square) and its derivative with respect to every natural-space parameter of
    every leaf of the spec tree.  The reference gets these by Theano reverse mode through the
    formulas of kernels.py:96-110,192-244,360-487 / metrics.py:89-102; this is the same chain
    rule written out.  Returns (K, grads) with grads a list of (leaf, pname, k, dK):
    leaf = index of the leaf in left-to-right order, pname in {'var','rate','freq','alpha'},
    k = position inside the rate / freq vector (None for scalars)."""
    x = np.asarray(x, dtype=dtype)
    if x.ndim == 1:
        x = x[:, None]
    if _leaf is None:
        _leaf = [0]
    op = spec[0]
    if op in ('sum', 'prod'):
        K1, g1 = kernel_cov_grads(spec[1], x, dtype, _leaf)
        K2, g2 = kernel_cov_grads(spec[2], x, dtype, _leaf)
        if op == 'sum':
            return K1 + K2, g1 + g2
        return K1 * K2, [(l, p, k, dK * K2) for l, p, k, dK in g1] + [(l, p, k, dK * K1) for l, p, k, dK in g2]
    if op == 'scale':
        K, g = kernel_cov_grads(spec[2], x, dtype, _leaf)
        return spec[1] * K, [(l, p, k, spec[1] * dK) for l, p, k, dK in g]
    if op == 'shift':
        K, g = kernel_cov_grads(spec[2], x, dtype, _leaf)
        return spec[1] + K, g
    leaf = _leaf[0]
    _leaf[0] += 1
    var = spec[1]
    unit = (op,) + (1.0,) + tuple(spec[2:])
    k0 = kernel_cov(unit, x, None, dtype)            # unit-variance leaf
    grads = [(leaf, 'var', None, k0)]
    if op in ('NOISE', 'WN'):
        return var * k0, grads
    dims = spec[-1]
    diff = _diff(x, x, dims)                         # n x n x nd
    nd = diff.shape[2]
    if op in ('SE', 'MAT32', 'MAT52', 'RQ'):
        rate = np.broadcast_to(np.asarray(spec[2], dtype=dtype), (nd,))
        D = np.dot(diff ** 2, 0.5 * rate ** 2)
        if op == 'SE':
            dkdD = -np.exp(-D)
        elif op == 'MAT32':
            dkdD = -1.5 * np.exp(-np.sqrt(3 * D))
        elif op == 'MAT52':
            s5 = np.sqrt(5 * D)
            dkdD = -(5.0 / 6.0) * (1 + s5) * np.exp(-s5)
        else:
            al = spec[3]
            b = 1 + D / al
            dkdD = -np.power(b, -al - 1)
            grads.append((leaf, 'alpha', None, var * k0 * (-np.log(b) + D / (al + D))))
        for k in range(nd):
            grads.append((leaf, 'rate', k, var * dkdD * rate[k] * diff[:, :, k] ** 2))
    elif op == 'OU':
        for k in range(nd):
            grads.append((leaf, 'rate', k, -var * k0 * np.abs(diff[:, :, k])))
    elif op == 'SIN':
        f = np.broadcast_to(np.asarray(spec[2], dtype=dtype), (nd,))
        r = np.broadcast_to(np.asarray(spec[3], dtype=dtype), (nd,))
        for k in range(nd):
            grads.append((leaf, 'freq', k, var * k0 * 2 * pi * r[k] * diff[:, :, k] * np.sin(2 * pi * diff[:, :, k] * f[k])))
        for k in range(nd):
            grads.append((leaf, 'rate', k, var * k0 * 2 * np.sin(pi * diff[:, :, k] * f[k]) ** 2))
    elif op in ('COS', 'SINC', 'SM'):
        f = np.broadcast_to(np.asarray(spec[2], dtype=dtype), (nd,))
        cs = 2 * pi2 if op == 'SINC' else 2 * pi
        th = cs * diff * f
        with np.errstate(divide='ignore', invalid='ignore'):
            if op == 'SINC':
                fac = np.where(diff != 0, np.sin(th) / th, 1.0)
                dfac = np.where(diff != 0, (np.cos(th) - fac) / f, 0.0)
            else:
                fac = np.cos(th)
                dfac = -cs * diff * np.sin(th)
        env = 1.0
        if op == 'SM':
            r = np.broadcast_to(np.asarray(spec[3], dtype=dtype), (nd,))
            env = np.exp(-2 * pi2 * np.dot(diff ** 2, r ** 2))
        for k in range(nd):
            others = np.prod(np.delete(fac, k, axis=2), axis=2)
            grads.append((leaf, 'freq', k, var * env * dfac[:, :, k] * others))
        if op == 'SM':
            for k in range(nd):
                grads.append((leaf, 'rate', k, var * k0 * (-4 * pi2) * diff[:, :, k] ** 2 * r[k]))
    else:
        raise ValueError('unknown kernel spec ' + str(op))
    return var * k0, grads


# --------------------------------------------------------------------------- means
def mean_eval(spec, x, dtype=np.float64):
    """Zero / Bias / Linear -- g3py/processes/hypers/means.py:117-159."""
    x = np.asarray(x, dtype=dtype)
    if x.ndim == 1:
        x = x[:, None]
    op = spec[0]
    if op == 'Zero':
        return np.zeros(x.shape[0], dtype=dtype)
    if op == 'Bias':
        return np.dtype(dtype).type(spec[1]) * np.ones(x.shape[0], dtype=dtype)
    if op == 'Linear':
        xs = x if len(spec) < 4 or spec[3] is None else x[:, spec[3]]
        return np.dtype(dtype).type(spec[1]) + np.dot(xs, np.asarray(spec[2], dtype=dtype))
    raise ValueError('unknown mean spec ' + str(op))


def mean_grads(spec, x, dtype=np.float64):
    """d m(x) / d(parameter): list of (pname, k, column) -- means.py:117-159 differentiated."""
    x = np.asarray(x, dtype=dtype)
    if x.ndim == 1:
        x = x[:, None]
    if spec[0] == 'Zero':
        return []
    if spec[0] == 'Bias':
        return [('bias', None, np.ones(x.shape[0], dtype=dtype))]
    xs = x if len(spec) < 4 or spec[3] is None else x[:, spec[3]]
    return [('constant', None, np.ones(x.shape[0], dtype=dtype))] + [('coeff', k, xs[:, k]) for k in range(xs.shape[1])]


# --------------------------------------------------------------------------- mappings
class Mapping:
    """spec: ('Identity',) ('LinearMapping', shift, scale) ('LogShifted', shift)
    ('BoxCoxLinear', shift, scale, power) ('ArcsinhLinear', shift, scale)
    -- g3py/processes/hypers/mappings.py:88-215, 309-333."""

    def __init__(self, spec, dtype=np.float64):
        self.spec = spec
        self.t = np.dtype(dtype).type

    def __call__(self, x):
        s, t = self.spec, self.t
        x = np.asarray(x)
        if s[0] == 'Identity':           # mappings.py:92-93
            return x
        if s[0] == 'LinearMapping':      # mappings.py:119-120
            return t(s[2]) * (x - t(s[1]))
        if s[0] == 'LogShifted':         # mappings.py:142-143
            return np.exp(x) + t(s[1])
        if s[0] == 'BoxCoxLinear':       # mappings.py:204-207
            shift, scale, power = t(s[1]), t(s[2]), t(s[3])
            scaled = power * x + t(1)
            transformed = np.sign(scaled) * np.abs(scaled) ** (t(1) / power)
            return transformed / scale - shift
        if s[0] == 'ArcsinhLinear':      # mappings.py:326-327
            return np.sinh((x - t(s[1])) / t(s[2]))
        raise ValueError(s[0])

    def inv(self, y):
        s, t = self.spec, self.t
        y = np.asarray(y)
        if s[0] == 'Identity':           # mappings.py:95-96
            return y
        if s[0] == 'LinearMapping':      # mappings.py:122-123
            return y / t(s[2]) + t(s[1])
        if s[0] == 'LogShifted':         # mappings.py:145-146
            return np.log(np.maximum(y - t(s[1]), t(np.float32(1e-32))))
        if s[0] == 'BoxCoxLinear':       # mappings.py:209-211
            shift, scale, power = t(s[1]), t(s[2]), t(s[3])
            shifted = scale * (y + shift)
            if power < np.float32(1e-5):
                return np.log(shifted)
            return ((np.sign(shifted) * np.abs(shifted) ** power) - t(1)) / power
        if s[0] == 'ArcsinhLinear':      # mappings.py:329-330
            return np.arcsinh(y) * t(s[2]) + t(s[1])
        raise ValueError(s[0])

    def logdet_dinv(self, y):
        s, t = self.spec, self.t
        y = np.asarray(y)
        n = t(y.shape[0])
        if s[0] == 'Identity':           # mappings.py:98-99
            return t(0)
        if s[0] == 'LinearMapping':      # mappings.py:125-126
            return -n * np.log(t(s[2]))
        if s[0] == 'LogShifted':         # mappings.py:148-149
            return -np.sum(np.log(y - t(s[1])))
        if s[0] == 'BoxCoxLinear':       # mappings.py:213-215
            shift, scale, power = t(s[1]), t(s[2]), t(s[3])
            return (power - t(1)) * np.sum(np.log(np.abs(scale * (y + shift)))) + n * np.log(scale)
        if s[0] == 'ArcsinhLinear':      # mappings.py:332-333
            return n * np.log(t(s[2])) - t(0.5) * np.sum(np.log1p(y ** 2))
        raise ValueError(s[0])


    def dinv(self, y):
        """d inv(y) / d(parameter), in the order of the spec (mappings.py:95-96,122-123,145-146,
        209-211,329-330 differentiated): list of (pname, array)"""
        s, t = self.spec, self.t
        y = np.asarray(y)
        one = np.ones_like(y)
        if s[0] == 'Identity':
            return []
        if s[0] == 'LinearMapping':
            return [('shift', one), ('scale', -y / t(s[2]) ** 2)]
        if s[0] == 'LogShifted':
            z = y - t(s[1])
            with np.errstate(all='ignore'):
                return [('shift', np.where(z > t(np.float32(1e-32)), -one / z, 0 * one))]
        if s[0] == 'BoxCoxLinear':
            shift, scale, power = t(s[1]), t(s[2]), t(s[3])
            sh = scale * (y + shift)
            with np.errstate(all='ignore'):
                if power < np.float32(1e-5):
                    return [('shift', scale / sh), ('scale', (y + shift) / sh), ('power', 0 * one)]
                a = np.abs(sh)
                sp_ = np.sign(sh) * a ** power
                dsh = a ** (power - 1)
                return [('shift', dsh * scale), ('scale', dsh * (y + shift)),
                        ('power', (sp_ * np.log(a) * power - (sp_ - 1)) / power ** 2)]
        if s[0] == 'ArcsinhLinear':
            return [('shift', one), ('scale', np.arcsinh(y))]
        raise ValueError(s[0])

    def dlogdet_dinv(self, y):
        """d logdet_dinv(y) / d(parameter): list of (pname, scalar)"""
        s, t = self.spec, self.t
        y = np.asarray(y)
        n = t(y.shape[0])
        if s[0] == 'Identity':
            return []
        if s[0] == 'LinearMapping':
            return [('shift', t(0)), ('scale', -n / t(s[2]))]
        if s[0] == 'LogShifted':
            with np.errstate(all='ignore'):
                return [('shift', np.sum(1 / (y - t(s[1]))))]
        if s[0] == 'BoxCoxLinear':
            shift, scale, power = t(s[1]), t(s[2]), t(s[3])
            with np.errstate(all='ignore'):
                return [('shift', (power - 1) * np.sum(1 / (y + shift))), ('scale', power * n / scale),
                        ('power', np.sum(np.log(np.abs(scale * (y + shift)))))]
        if s[0] == 'ArcsinhLinear':
            return [('shift', t(0)), ('scale', n / t(s[2]))]
        raise ValueError(s[0])


# --------------------------------------------------------------------------- Cholesky
def cholesky_robust(K, maxtries=20, return_info=False):
    """CholeskyRobust._cholesky + .perform -- g3py/libs/tensors.py:197-222.

    dpotrf in float64 whatever K's dtype, result cast back to K.dtype (tensors.py:198,219).
    Jitter schedule: dK = mean(diag)*1e-6*I; lift non-positive diagonals; retry up to 20
    times with dK *= 10; final fallback 1e-10*I (tensors.py:203-221).
    `return_info` additionally returns (tries, fallback) for tests."""
    x = np.asarray(K)
    tries, fallback = 0, False

    def _cholesky(Kd):
        nonlocal tries
        L, info = sp.linalg.lapack.dpotrf(Kd, lower=True)
        if info == 0:
            return L
        diagK = np.diag(Kd)
        dK = np.eye(Kd.shape[0]) * diagK.mean() * np.float32(1e-6)
        if np.any(diagK <= 0.0):
            Kd = Kd + np.eye(Kd.shape[0]) * (diagK.mean() * np.float32(1e-6) - diagK.min())
        for _ in range(maxtries):
            tries += 1
            try:
                return np.nan_to_num(sp.linalg.cholesky(Kd + dK, lower=True))
            except Exception:
                dK = dK * np.float32(10)
        raise sp.linalg.LinAlgError("not approximate positive-definite")

    try:
        z = _cholesky(x).astype(x.dtype)
    except Exception:
        fallback = True
        z = (0 * x + np.float32(1e-10) * np.eye(len(x))).astype(x.dtype)
    if return_info:
        return z, tries, fallback
    return z


def cholesky_grad(L, Lbar):
    """CholeskyRobust.grad -- g3py/libs/tensors.py:224-260 (Murray 2016, reverse mode):
    given Lbar = d f / d L returns d f / d K for the lower-triangular convention used there:
    s = L^-T Phi(L^T Lbar) L^-1 with Phi = tril with halved diagonal, result tril(s + s^T) - diag(s)."""
    P = np.tril(L.T.dot(Lbar))
    P[np.diag_indices_from(P)] *= 0.5                                   # tril_and_halve_diagonal :245-247
    # conjugate_solve_triangular(outer=L, inner=P) = L^-T P L^-1  :249-252
    s = sp.linalg.solve_triangular(L, sp.linalg.solve_triangular(L, P.T, lower=True, trans='T').T,
                                   lower=True, trans='T')
    return np.tril(s + s.T) - np.diag(np.diag(s))                       # :258


# --------------------------------------------------------------------------- logp
def logp_cho(value, mu, cho, mapping):
    """WarpedGaussianDistribution.logp_cho -- g3py/processes/gaussian.py:192-241."""
    t = cho.dtype.type
    delta_ = mapping.inv(value) - mu                                    # :208
    if np.any(~np.isfinite(delta_)):                                    # cond1 :234
        return t(np.float32(-1e30))
    det_m = mapping.logdet_dinv(value)                                  # :225
    if np.any(~np.isfinite(det_m)):                                     # cond2 :235
        return t(np.float32(-1e30))
    if np.any(~np.isfinite(cho)):                                       # cond3 :236
        return t(np.float32(-1e30))
    with np.errstate(all='ignore'):
        lcho = sp.linalg.solve_triangular(cho, delta_, lower=True, check_finite=False)  # :212
        lcho2 = lcho.T.dot(lcho)                                        # :215
        npi = t(-0.5) * t(cho.shape[0]) * np.log(t(2.0 * np.pi))        # :218
        dot2 = t(-0.5) * lcho2                                          # :219
        det_k = -np.sum(np.log(np.diag(cho)))                           # :224
        r = npi + dot2 + det_k + det_m                                  # :232
    if np.any(~np.isfinite(lcho)):                                      # cond4 :237
        return t(np.float32(-1e30))
    return r


def log_jacobian_positive(log_values):
    """NonTransformLog.jacobian_det: 0 if exp(x) > 1e-6 else -inf -- hypers/__init__.py:199-200."""
    v = np.atleast_1d(np.asarray(log_values))
    return float(np.sum(np.where(np.exp(v) > 1e-6, 0.0, -np.inf)))


# --------------------------------------------------------------------------- process
class GP:
    """EllipticalProcess.th_define_process + GaussianProcess / WarpedGaussianProcess methods.

    kernel_f: spec of f_kernel; noise_var: None for noisy=False (elliptical.py:26-31);
    mean: mean spec; mapping: mapping spec.  Follows elliptical.py:60-204,
    gaussian.py:42-97,127-174, stochastic.py:294-313."""

    def __init__(self, kernel_f, noise_var=None, mean=('Zero',), mapping=('Identity',),
                 dtype=np.float64, log_positive_hypers=()):
        self.kf = kernel_f
        self.kn = kernel_f if noise_var is None else with_noise(kernel_f, noise_var)
        self.mean_spec = mean
        self.map = Mapping(mapping, dtype)
        self.warped = mapping[0] != 'Identity'
        self.dtype = dtype
        self.t = np.dtype(dtype).type
        self.log_positive_hypers = log_positive_hypers

    # ---- basic tensors (elliptical.py:63-79)
    def _x(self, a):
        a = np.asarray(a, dtype=self.dtype)
        return a[:, None] if a.ndim == 1 else a

    def mapping_outputs(self, outputs):
        return tt_to_num(self.map.inv(np.asarray(outputs, dtype=self.dtype)))   # :63

    def prior_location(self, x):
        return mean_eval(self.mean_spec, self._x(x), self.dtype)                # :67-68

    def prior_kernel(self, x, noise):
        if noise:
            return tt_to_cov(kernel_cov(self.kn, self._x(x), None, self.dtype))  # :70-71
        return kernel_cov(self.kf, self._x(x), None, self.dtype)                # :74-75

    def cross_kernel(self, space, inputs, noise):
        k = self.kn if noise else self.kf
        return tt_to_num(kernel_cov(k, self._x(space), self._x(inputs), self.dtype))  # :78-79

    # ---- selectors (elliptical.py:121-188)
    def location(self, space, inputs=None, outputs=None, prior=False, noise=False):
        if prior:
            return self.prior_location(space)
        c = self.cross_kernel(space, inputs, noise)
        kin = self.prior_kernel(inputs, True)
        rhs = self.mapping_outputs(outputs) - self.prior_location(inputs)
        return self.prior_location(space) + c.dot(sp.linalg.solve(kin, rhs))    # :81-84

    def kernel(self, space, inputs=None, prior=False, noise=False):
        if prior:
            return self.prior_kernel(space, noise)
        c = self.cross_kernel(space, inputs, noise)
        kin = self.prior_kernel(inputs, True)
        return self.prior_kernel(space, noise) - c.dot(sp.linalg.solve(kin, c.T))  # :86-91

    def cholesky(self, space, inputs=None, prior=False, noise=False):
        return cholesky_robust(self.kernel(space, inputs, prior, noise))        # :72,76,88,92

    def kernel_diag(self, space, inputs=None, prior=False, noise=False):
        return tt_to_bounded(np.diag(self.kernel(space, inputs, prior, noise)), self.t(0))  # :94-97

    def kernel_sd(self, space, inputs=None, prior=False, noise=False):
        return np.sqrt(self.kernel_diag(space, inputs, prior, noise))           # :99-102

    def cholesky_diag(self, space, inputs=None, prior=False, noise=False):
        return np.diag(self.kernel_sd(space, inputs, prior, noise))             # :104-107

    # ---- statistics
    def gauss_hermite(self, f, mu, sigma, n=10):
        """gaussian.py:162-174."""
        _a, _w = np.polynomial.hermite.hermgauss(n)
        a = _a.astype(self.dtype)[:, None]
        w = _w.astype(self.dtype)
        grille = mu + sigma * self.t(np.sqrt(2)) * a
        return np.dot(w, f(grille.flatten()).reshape(grille.shape)) / self.t(np.sqrt(np.pi))

    def mean(self, space, inputs=None, outputs=None, prior=False, noise=False):
        loc = self.location(space, inputs, outputs, prior, noise)
        if not self.warped:
            return self.map(loc)                                                # elliptical.py:194-196
        sd = self.kernel_sd(space, inputs, prior, noise)
        return self.gauss_hermite(lambda v: self.map(v), loc, sd)               # gaussian.py:127-141

    def median(self, space, inputs=None, outputs=None, prior=False, noise=False):
        return self.map(self.location(space, inputs, outputs, prior, noise))    # elliptical.py:190-192

    def variance(self, space, inputs=None, outputs=None, prior=False, noise=False):
        if not self.warped:
            return self.kernel_diag(space, inputs, prior, noise)                # elliptical.py:198-200
        loc = self.location(space, inputs, outputs, prior, noise)
        sd = self.kernel_sd(space, inputs, prior, noise)
        return (self.gauss_hermite(lambda v: self.map(v) ** 2, loc, sd)
                - self.mean(space, inputs, outputs, prior, noise) ** 2)         # gaussian.py:143-157

    def std(self, *a, **k):
        return np.sqrt(self.variance(*a, **k))                                  # stochastic.py:294-298

    def covariance(self, space, inputs=None, prior=False, noise=False):
        return self.kernel(space, inputs, prior, noise)                         # elliptical.py:202-204

    def quantiler(self, space, inputs=None, outputs=None, q=0.975, prior=False, noise=False):
        p = stats.norm.ppf(q)                                                   # gaussian.py:71-73
        return self.map(self.location(space, inputs, outputs, prior, noise)
                        + p * self.kernel_sd(space, inputs, prior, noise))

    def sampler(self, space, inputs=None, outputs=None, rand=None, prior=False, noise=False):
        """gaussian.py:89-97 with the normal draws `rand` (len(space) x samples) supplied."""
        g = (self.location(space, inputs, outputs, prior, noise)[:, None]
             + self.cholesky(space, inputs, prior, noise).dot(rand))
        return np.array([self.map(k.T) for k in g.T]).T

    def logpredictive(self, vector, space, inputs=None, outputs=None, prior=False, noise=False):
        """gaussian.py:42-54: independent marginals, diagonal 'cholesky' with noise=True."""
        return logp_cho(np.asarray(vector, dtype=self.dtype),
                        self.location(space, inputs, outputs, prior, noise),
                        self.cholesky_diag(space, inputs, prior, True), self.map)

    def loglike(self, inputs, outputs):
        """observed RV term -- gaussian.py:37-40,243-260; stochastic.py:311-313."""
        cho = cholesky_robust(self.prior_kernel(inputs, True))
        return logp_cho(np.asarray(outputs, dtype=self.dtype), self.prior_location(inputs), cho, self.map)

    def logp(self, inputs=None, outputs=None, prior=False):
        """stochastic.py:300-306: Flat priors contribute 0 plus the log-transform Jacobian
        (hypers/__init__.py:199-200); observed term unless prior=True."""
        lp = sum(log_jacobian_positive(v) for v in self.log_positive_hypers)
        if prior:
            return self.t(lp)
        return self.t(lp) + self.loglike(inputs, outputs)


    def _dlogp_scale(self, beta, n):
        """d logp / d beta = -s/2: s = 1 for the Gaussian density (gaussian.py:219)"""
        return 1.0

    def dlogp_natural(self, inputs, outputs):
        """Gradient of `logp` (observed term; the Flat priors and the log-transform Jacobian are
        constants) with respect to the NATURAL-space parameters, as the reference's
        th_dlogp = gradient(th_logp) (stochastic.py:308-309, tensors.py:11-22) computes it by
        reverse mode: logp_cho (gaussian.py:208-232) -> Lbar, delta_bar; CholeskyRobust.grad
        (tensors.py:224-260) -> Kbar; then the kernel / mean / mapping formulas.  The jittered
        factor is re-used as the Cholesky of K exactly as grad() does (`chol_x = self(x)`).
        Returns dict(kernel=[(leaf, pname, k, value)], mean=[(pname, k, value)],
        mapping=[(pname, value)]); all zeros when logp takes the -1e30 branch (a constant)."""
        x = self._x(inputs)
        y = np.asarray(outputs, dtype=self.dtype)
        K, kg = kernel_cov_grads(self.kn, x, self.dtype)
        mg = mean_grads(self.mean_spec, x, self.dtype)
        di, dl = self.map.dinv(y), self.map.dlogdet_dinv(y)
        zero = dict(kernel=[(l, p, k, 0.0) for l, p, k, _ in kg], mean=[(p, k, 0.0) for p, k, _ in mg],
                    mapping=[(p, 0.0) for p, _ in di])
        cho = cholesky_robust(tt_to_cov(K))
        delta_ = self.mapping_outputs(y) - self.prior_location(x)
        if not np.isfinite(self.loglike(inputs, outputs)) or self.loglike(inputs, outputs) == self.t(np.float32(-1e30)):
            return zero
        a = sp.linalg.solve_triangular(cho, delta_, lower=True)
        alpha = sp.linalg.solve_triangular(cho, a, lower=True, trans='T')
        # logp = f(beta) - sum log L_ii + ..., beta = a^T a, f' = -s/2:
        #   Lbar = s tril(alpha a^T) - diag(1/L_ii); delta_bar = -s alpha
        sc = self._dlogp_scale(float(a.dot(a)), len(a))
        alpha = sc * alpha
        Lbar = np.tril(np.outer(alpha, a)) - np.diag(1.0 / np.diag(cho))
        Kbar = cholesky_grad(cho, Lbar)
        out = dict(kernel=[(l, p, k, float(np.sum(Kbar * dK))) for l, p, k, dK in kg],
                   mean=[(p, k, float(alpha.dot(col))) for p, k, col in mg],
                   mapping=[(p, float(-alpha.dot(dcol) + dscalar)) for (p, dcol), (_, dscalar) in zip(di, dl)])
        return out


def logp_cho_t(value, mu, cho, freedom, mapping):
    """WarpedStudentTDistribution.logp_cho -- g3py/processes/studentT.py:114-146."""
    t = cho.dtype.type
    delta_ = mapping.inv(value) - mu                                    # :115
    if np.any(~np.isfinite(delta_)):
        return t(np.float32(-1e30))
    det_m = mapping.logdet_dinv(value)                                  # :130
    if np.any(~np.isfinite(det_m)) or np.any(~np.isfinite(cho)):
        return t(np.float32(-1e30))
    with np.errstate(all='ignore'):
        lcho = sp.linalg.solve_triangular(cho, delta_, lower=True, check_finite=False)   # :117
        beta = lcho.T.dot(lcho)                                         # :118
        n = t(cho.shape[0])
        nu = t(freedom)
        r1 = t(-0.5) * (nu + n) * np.log1p(beta / (nu - t(2)))          # :124
        if t(np.float32(1e6)) <= nu:                                    # :125-126
            r2 = -n * t(0.5) * np.log(t(2.0 * np.float32(np.pi)))
        else:
            from scipy.special import gammaln
            r2 = t(gammaln((nu + n) * 0.5) - gammaln(nu * 0.5)) - t(0.5) * n * np.log((nu - t(2)) * t(np.float32(np.pi)))
        r3 = -np.sum(np.log(np.diag(cho)))                              # :128
    if np.any(~np.isfinite(lcho)):
        return t(np.float32(-1e30))
    return r1 + r2 + r3 + det_m                                         # :135


class TP(GP):
    """StudentTProcess / WarpedStudentTProcess -- g3py/processes/studentT.py:16-102.
    `degree`: the FlatExp hyper; freedom = bound + degree with bound = 2 (hypers/__init__.py:144-160)."""

    def __init__(self, kernel_f, degree, noise_var=None, mean=('Zero',), mapping=('Identity',), dtype=np.float64):
        super().__init__(kernel_f, noise_var, mean, mapping, dtype)
        self.degree = degree

    def freedom(self, inputs=None, prior=False):
        nu = float(np.float32(2.0)) + self.degree                      # Freedom.__call__
        return nu if prior else nu + len(inputs)                       # elliptical.py:109-113

    def scaling(self, inputs, outputs, prior=False):
        """studentT.py:36-44"""
        if prior:
            return self.t(1.0)
        cho = cholesky_robust(self.prior_kernel(inputs, True))
        alpha = sp.linalg.solve_triangular(cho, self.mapping_outputs(outputs) - self.prior_location(inputs), lower=True)
        beta = alpha.T.dot(alpha)
        return (self.freedom(prior=True) + beta - 2.0) / (self.freedom(inputs) - 2.0)

    def variance(self, space, inputs=None, outputs=None, prior=False, noise=False):
        if self.warped:     # WarpedStudentTProcess.th_variance: Gauss-Hermite, no scaling (:88-94)
            return super().variance(space, inputs, outputs, prior, noise)
        return self.kernel_diag(space, inputs, prior, noise) * self.scaling(inputs, outputs, prior)   # :46-47

    def covariance(self, space, inputs=None, outputs=None, prior=False, noise=False):
        return self.kernel(space, inputs, prior, noise) * self.scaling(inputs, outputs, prior)       # :49-50

    def quantiler(self, space, inputs=None, outputs=None, q=0.975, prior=False, noise=False):
        p = stats.t.ppf(q, df=self.freedom(inputs, prior))              # :53
        return self.map(self.location(space, inputs, outputs, prior, noise)
                        + p * self.kernel_sd(space, inputs, prior, noise))

    def _dlogp_scale(self, beta, n):
        """r1 = -1/2 (nu + n) log1p(beta / (nu - 2))  (studentT.py:124)  =>  d r1 / d beta = -s/2"""
        nu = self.freedom(prior=True)
        return (nu + n) / (nu - 2.0 + beta)

    def dlogp_degree(self, inputs, outputs):
        """d logp / d degree (natural space): r1 and r2 of studentT.py:124-126 differentiated"""
        from scipy.special import digamma
        cho = cholesky_robust(self.prior_kernel(inputs, True))
        a = sp.linalg.solve_triangular(cho, self.mapping_outputs(outputs) - self.prior_location(inputs), lower=True)
        beta, n, nu = float(a.dot(a)), float(len(a)), self.freedom(prior=True)
        g = -0.5 * np.log1p(beta / (nu - 2.0)) + 0.5 * (nu + n) * beta / ((nu - 2.0) * (nu - 2.0 + beta))
        if not float(np.float32(1e6)) <= nu:
            g += 0.5 * digamma((nu + n) * 0.5) - 0.5 * digamma(nu * 0.5) - 0.5 * n / (nu - 2.0)
        return g

    def loglike(self, inputs, outputs):
        cho = cholesky_robust(self.prior_kernel(inputs, True))
        return logp_cho_t(np.asarray(outputs, dtype=self.dtype), self.prior_location(inputs), cho,
                          self.freedom(prior=True), self.map)


# --------------------------------------------------------------------------- transports
class TKernelOracle:
    """TKernel -- g3py/processes/hypers/transports.py:200-257, literally (the joint covariance is formed)."""

    def __init__(self, kernel_f, noise_var=None, dtype=np.float64):
        self.kf = kernel_f
        self.kn = kernel_f if noise_var is None else ('sum', kernel_f, ('NOISE', noise_var))
        self.dtype = dtype

    def _cov(self, x, noise, x2=None):
        return kernel_cov(self.kn if noise else self.kf, x, x2, self.dtype)

    def __call__(self, inputs, outputs, noise=False):
        return cholesky_robust(self._cov(inputs, noise)).dot(outputs)                        # :212-218

    def diag(self, inputs, outputs, noise=False):
        return np.sqrt(np.diag(self._cov(inputs, noise))) * outputs                          # :220-227

    def inv(self, inputs, outputs, noise=False):
        return sp.linalg.solve_triangular(cholesky_robust(self._cov(inputs, noise)), outputs, lower=True)   # :229-234

    def logdet_dinv(self, inputs, outputs):
        return -np.sum(np.log(np.diag(cholesky_robust(self._cov(inputs, True)))))            # :236-238

    def posterior(self, space, pred, inputs, outputs, noise_pred=False, noise_obs=True):
        outputs_inv = self.inv(inputs, outputs, noise=noise_obs)                             # :240
        cov_inputs = self._cov(inputs, noise_obs)
        cov_space = self._cov(space, noise_pred)
        cov_space_inputs = kernel_cov(self.kf, inputs, space, self.dtype)                    # :249
        cov = np.concatenate([np.concatenate([cov_inputs, cov_space_inputs], axis=1),
                              np.concatenate([cov_space_inputs.T, cov_space], axis=1)])      # :250-252
        cho = cholesky_robust(cov)
        return cho.dot(np.concatenate([outputs_inv, pred]))[len(inputs):]                    # :253-257


class TScaleOracle:
    """TScale -- g3py/processes/hypers/transports.py:165-181 with a constant scale function."""

    def __init__(self, scale):
        self.scale = scale

    def _s(self, inputs):
        return np.full(len(inputs), self.scale, dtype=np.float64)

    def __call__(self, inputs, outputs, noise=False):
        return outputs * self._s(inputs)                                                     # :171-172

    def inv(self, inputs, outputs, noise=False):
        return outputs / self._s(inputs)                                                     # :174-175

    def logdet_dinv(self, inputs, outputs):
        return -np.sum(np.log(self._s(inputs)))                                              # :177-181


def transport_logp(value, t1, t2, inputs):
    """TransportGaussianDistribution.logp_t for the composition t1 @ t2 -- g3py/processes/transport.py:
    220-243 with TransportComposed.inv / logdet_dinv (hypers/transports.py:103-107)."""
    value = np.asarray(value, dtype=np.float64)
    inner = t1.inv(inputs, value, noise=True)
    delta = t2.inv(inputs, inner, noise=True)                                                # :226 via :103-104
    det_m = t2.logdet_dinv(inputs, inner) + t1.logdet_dinv(inputs, value)                    # :227 via :106-107
    if not np.all(np.isfinite(delta)) or not np.all(np.isfinite(det_m)):
        return np.float32(-1e30)                                                             # :240-243
    return -0.5 * len(value) * np.log(2.0 * np.pi) - 0.5 * delta.dot(delta) + det_m          # :230-238


# --------------------------------------------------------------------------- CPU baseline
def cpu_hot_path(X, y, Xs, var=1.0, rate=1.0, noise=0.1):
    """One pass of the benchmark hot path on the CPU (bench.py `cpu_baseline`, kind "port"):
    SE Gram + dpotrf + triangular solves for logp, posterior mean and variance, using the
    LAPACK entry points the reference calls.  Cholesky-based posterior (same algebra as the
    HIP path) so the two are timed doing the same work.  Returns (logp, mean, var, timings)."""
    import time
    t0 = time.perf_counter()
    N, d = X.shape
    r = np.full(d, rate)
    # tiled Gram (the reference's N x N x d broadcast would not fit at bench sizes)
    w = 0.5 * r ** 2
    Xw = X * np.sqrt(w)
    sq = (Xw ** 2).sum(1)
    K = np.empty((N, N))
    for r0 in range(0, N, 4096):              # row tiles: the same arithmetic without three N x N temporaries
        r1 = min(N, r0 + 4096)
        T = sq[r0:r1, None] + sq[None, :] - 2.0 * Xw[r0:r1].dot(Xw.T)
        np.maximum(T, 0.0, out=T)
        np.negative(T, out=T)
        np.exp(T, out=T)
        T *= var
        K[r0:r1] = T
    K[np.diag_indices(N)] += noise
    t1 = time.perf_counter()
    if N <= 16384:
        L, info = sp.linalg.lapack.dpotrf(K, lower=True, overwrite_a=True)
        assert info == 0
    else:
        # One dpotrf over an 8.6 GB matrix segfaults in the OpenBLAS of this image (as in oracle/gen_fullsize.py): the
        # same factorisation as a right-looking sweep over 8192-wide panels with the same LAPACK / BLAS entry points --
        # dpotrf on the diagonal blocks, dtrsm for the panel, dgemm for the trailing update (lower block rows only)
        b = 8192
        for j in range(0, N, b):
            e = min(N, j + b)
            Ljj, info = sp.linalg.lapack.dpotrf(K[j:e, j:e], lower=True)
            assert info == 0, (j, info)
            K[j:e, j:e] = np.tril(Ljj)
            if e < N:
                K[e:, j:e] = sp.linalg.solve_triangular(Ljj, K[e:, j:e].T, lower=True, check_finite=False).T
                for i in range(e, N, b):
                    ie = min(N, i + b)
                    K[i:ie, e:ie] -= K[i:ie, j:e] @ K[e:ie, j:e].T
        L = K                                              # lower triangle valid (the solves below reference only that)
    t2 = time.perf_counter()
    a = sp.linalg.solve_triangular(L, y, lower=True, check_finite=False)
    logp = -0.5 * N * np.log(2 * np.pi) - 0.5 * a.dot(a) - np.sum(np.log(np.diag(L)))
    t3 = time.perf_counter()
    Xsw = Xs * np.sqrt(w)
    sqs = (Xsw ** 2).sum(1)
    Ks = var * np.exp(-np.maximum(sqs[:, None] + sq[None, :] - 2.0 * Xsw.dot(Xw.T), 0.0))
    V = sp.linalg.solve_triangular(L, Ks.T, lower=True, check_finite=False)
    mean = V.T.dot(a)
    varp = np.maximum(var - (V ** 2).sum(0), 0.0)
    t4 = time.perf_counter()
    return logp, mean, varp, dict(gram=t1 - t0, potrf=t2 - t1, logp=t3 - t2, predict=t4 - t3,
                                  total=t4 - t0)
