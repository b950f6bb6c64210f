"""Transports (triangular push-forwards) of g3py/processes/hypers/transports.py:10-257 on the HIP path.

A transport maps a reference vector to observations, `y = T(x; inputs)`, with an inverse and a
log-determinant; composing them (`T1 @ T2`) composes the maps.  The element-wise ones
(`TLocation`, `TMapping`, `ID`) are O(N) host arithmetic.  `TKernel` is the dense one,
`y = chol(K(inputs)) x`: its inverse, log-determinant and posterior are the same Gram + Cholesky +
triangular-solve kernels as the GP path in a different schedule --

    reference (transports.py:239-257)           here
    chol([[Kxx, Kxs], [Ksx, Kss]]) [Lxx^-1 y; z]  =  V a + chol(Kss - V V^T) z ,
                                                     V = Ksx Lxx^-T,  a = Lxx^-1 y

i.e. the joint (N+M) x (N+M) factorisation is never formed: `g3_gp_factor_predict` carries the
K(space, inputs) rows through the factorisation of Kxx (which yields V and a), the M x M Schur
complement is factored by `g3_potrf_robust`, and `L z` runs in the MFMA GEMM.  Equal to the
reference in exact arithmetic whenever no jitter is needed (the reference applies its jitter
schedule to the joint matrix, this module to the two blocks).

Numeric API: every method takes the natural-space hyper values (`values`, name -> value) that a
process would hand down; constants need none.  `g3py_amd/processes/transport.py` is the
process front end (`TransportProcess`, `TransportGaussianProcess`).
"""
import numpy as np

from . import Hypers
from .kernels import KernelSum, KernelNoise
from ... import _lib


class Transport(Hypers):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.parametrics = []

    def __call__(self, inputs, outputs, noise=False, values=None):
        raise NotImplementedError

    def diag(self, inputs, outputs, noise=False, values=None):
        return self(inputs, outputs, noise=noise, values=values)

    def inv(self, inputs, outputs, noise=False, values=None):
        raise NotImplementedError

    def logdet_dinv(self, inputs, outputs, values=None):
        raise NotImplementedError

    def posterior(self, space, pred, inputs, outputs, noise_pred=False, noise_obs=True, diag=False, inv=False, values=None):
        """transports.py:27-33: push [T^-1(outputs); pred] through the transport of [inputs; space]"""
        outputs_inv = self.inv(inputs, outputs, noise=True, values=values)
        full = self(np.concatenate([inputs, space]), np.concatenate([outputs_inv, pred]), noise=True, values=values)
        return full[len(inputs):]

    def check_hypers(self, parent=''):
        for p in self.parametrics:
            p.check_hypers(parent)
        self.hypers = [h for p in self.parametrics for h in p.hypers]

    def check_dims(self, x=None):
        super().check_dims(x)
        for p in self.parametrics:
            p.check_dims(x)

    def default_hypers_dims(self, x=None, y=None):
        r = dict()
        for p in self.parametrics:
            r.update(p.default_hypers_dims(x, y))
        return r

    def __matmul__(self, other):
        return TransportComposed(self, other)


class TransportComposed(Transport):
    """t1 @ t2: y = t1(t2(x)) -- transports.py:91-118"""

    def __init__(self, t1, t2):
        self.t1, self.t2 = t1, t2
        self.name = t1.name + ' ' + t2.name
        self.hypers, self.parametrics, self.potential = [], [t1, t2], None
        self.shape, self.dims = None, None

    def __call__(self, inputs, outputs, noise=False, values=None):
        return self.t1(inputs, self.t2(inputs, outputs, noise=noise, values=values), noise=noise, values=values)

    def diag(self, inputs, outputs, noise=False, values=None):
        return self.t1.diag(inputs, self.t2(inputs, outputs, noise=noise, values=values), noise=noise, values=values)

    def inv(self, inputs, outputs, noise=False, values=None):
        return self.t2.inv(inputs, self.t1.inv(inputs, outputs, noise=noise, values=values), noise=noise, values=values)

    def logdet_dinv(self, inputs, outputs, values=None):
        return (self.t2.logdet_dinv(inputs, self.t1.inv(inputs, outputs, noise=True, values=values), values=values)
                + self.t1.logdet_dinv(inputs, outputs, values=values))

    def posterior(self, space, pred, inputs, outputs, noise_pred=False, noise_obs=True, diag=False, inv=False, values=None):
        inner = self.t2.posterior(space, pred, inputs, self.t1.inv(inputs, outputs, noise=noise_obs, values=values),
                                  noise_pred=noise_pred, noise_obs=noise_obs, values=values)
        return self.t1.posterior(space, inner, inputs, outputs, noise_pred=noise_pred, noise_obs=noise_obs, values=values)


class ID(Transport):
    def __call__(self, inputs, outputs, noise=False, values=None):
        return outputs

    def inv(self, inputs, outputs, noise=False, values=None):
        return outputs

    def logdet_dinv(self, inputs, outputs, values=None):
        return 1.0          # as written at transports.py:128-129


class TElemwise(Transport):
    def posterior(self, space, pred, inputs=None, outputs=None, noise_pred=False, noise_obs=True, diag=False, inv=False,
                  values=None):
        return self(space, pred, noise=noise_pred, values=values)


class TLocation(TElemwise):
    """y = x + m(inputs) -- transports.py:145-163"""

    def __init__(self, location=None, x=None, name=None):
        super().__init__(x, name)
        self.location = location
        self.parametrics.append(location)

    def __call__(self, inputs, outputs, noise=False, values=None):
        return outputs + self.location(np.asarray(inputs), values)

    def inv(self, inputs, outputs, noise=False, values=None):
        return outputs - self.location(np.asarray(inputs), values)

    def logdet_dinv(self, inputs, outputs, values=None):
        return 0.0


class TScale(TElemwise):
    """y = x * s(inputs) -- transports.py:165-181; `scale` is any parametric function of the inputs
    (called as scale(inputs, values) like the means); log det dT^-1 = -sum log s(inputs)"""

    def __init__(self, scale=None, x=None, name=None):
        super().__init__(x, name)
        self.scale = scale
        self.parametrics.append(scale)

    def __call__(self, inputs, outputs, noise=False, values=None):
        return outputs * self.scale(np.asarray(inputs), values)

    def inv(self, inputs, outputs, noise=False, values=None):
        return outputs / self.scale(np.asarray(inputs), values)

    def logdet_dinv(self, inputs, outputs, values=None):
        with np.errstate(all='ignore'):
            return -np.sum(np.log(self.scale(np.asarray(inputs), values)))


class TMapping(TElemwise):
    """y = mapping(x) -- transports.py:184-197"""

    def __init__(self, mapping=None, x=None, name=None):
        super().__init__(x, name)
        self.mapping = mapping
        self.parametrics.append(mapping)

    def __call__(self, inputs, outputs, noise=False, values=None):
        return self.mapping(np.asarray(outputs), values)

    def inv(self, inputs, outputs, noise=False, values=None):
        return self.mapping.inv(np.asarray(outputs), values)

    def logdet_dinv(self, inputs, outputs, values=None):
        return self.mapping.logdet_dinv(np.asarray(outputs), values)


class TLinear(Transport):
    """marker: transports that are linear in the reference vector (transports.py:137-138)"""


class TNoLinear(Transport):
    """marker: non-linear transports (transports.py:141-142)"""


class TTriangular(TNoLinear):
    """triangular transport from a generator (transports.py:260-263).  The reference defines the
    constructor only (no map, inverse or log-determinant): kept as the same shell -- calling it
    raises NotImplementedError through Transport."""

    def __init__(self, generator, x=None, name=None):
        super().__init__(x, name)
        self.generator = generator
        self.parametrics.append(generator)


class TKernel(TLinear):
    """y = chol(K(inputs)) x -- transports.py:200-257, on the device"""

    def __init__(self, kernel, noisy=False, x=None, name=None, dtype=np.float64, device=None):
        super().__init__(x, name)
        self.kernel = kernel
        self.noisy = KernelSum(kernel, KernelNoise(name='Noise' + kernel.name)) if noisy else kernel
        self.parametrics.append(self.noisy)
        self.dtype = np.dtype(dtype)
        self._device = device

    @property
    def device(self):
        from ...device import Device
        if self._device is None:
            self._device = Device.default()
        return self._device

    def _x(self, a):
        a = np.asarray(a, dtype=self.dtype)
        return a.reshape(len(a), 1) if a.ndim < 2 else np.ascontiguousarray(a)

    def _prog(self, noise, values, d):
        from ...device import compile_spec
        return compile_spec((self.noisy if noise else self.kernel).spec(values or {}, d), d)

    def _factor(self, inputs, outputs, noise, values, space=None):
        """(L in K, block inverses W, a = L^-1 outputs, stats[, V rows, V a]) via g3_gp_factor[_predict]"""
        dev, X = self.device, self._x(inputs)
        N, d = X.shape
        Np = _lib.roundup(N)
        Xd = dev.upload(X)
        dd = dev.upload(np.asarray(outputs, dtype=self.dtype).reshape(-1))
        a, W = dev.alloc(1, Np, self.dtype), dev.alloc_inverses(Np, self.dtype)
        if space is None:
            K = dev.alloc(Np + _lib.G3_RHS_PAD, Np, self.dtype)
            st = dev.gp_factor(self._prog(noise, values, d), Xd, N, d, dd, K, W, a)
            return dict(K=K, W=W, a=a, st=st, N=N, Np=Np, d=d, Xd=Xd)
        S = self._x(space)
        M, Mp = len(S), _lib.roundup(len(S), _lib.G3_RHS_PAD)
        K = dev.alloc(Np + _lib.G3_RHS_PAD + Mp, Np, self.dtype)
        mu, ss = dev.alloc(1, Mp, self.dtype), dev.alloc(1, Mp, self.dtype)
        st = dev.gp_factor_predict(self._prog(noise, values, d), self._prog(False, values, d), Xd, N, d, dd,
                                   dev.upload(S), M, K, W, a, mu, ss)
        return dict(K=K, W=W, a=a, st=st, N=N, Np=Np, d=d, Xd=Xd, M=M, Mp=Mp, mu=dev.download(mu, 1, M)[0], S=S)

    def _chol_apply(self, inputs, vec, noise, values):
        """chol_robust(K(inputs)) @ vec: factor on the device, product in the MFMA GEMM"""
        dev, X = self.device, self._x(inputs)
        M, d = X.shape
        Mp = _lib.roundup(M, _lib.G3_RHS_PAD)
        Kd = dev.alloc(M, M, self.dtype)
        dev.gram(self._prog(noise, values, d), dev.upload(X), None, d, Kd, M, M, 0)
        Ld = dev.alloc(Mp, Mp, self.dtype, zero=True)
        dev.potrf_robust(Kd, Ld, M)
        return self._lower_times(Ld, M, Mp, vec)

    def _lower_times(self, Ld, M, Mp, vec):
        dev = self.device
        z = np.zeros((64, Mp), dtype=self.dtype)
        z[0, :M] = np.asarray(vec, dtype=self.dtype).reshape(-1)
        out = dev.alloc(64, Mp, self.dtype)
        dev.gemm_nt(out, dev.upload(z), Ld, 64, Mp, Mp)            # (L z)^T
        return dev.download(out, 1, M)[0]

    def __call__(self, inputs, outputs, noise=False, values=None):
        return self._chol_apply(inputs, outputs, noise, values)                               # :212-218

    def diag(self, inputs, outputs, noise=False, values=None):
        dev, X = self.device, self._x(inputs)
        out = dev.alloc(1, len(X), self.dtype)
        dev.gram_diag(self._prog(noise, values, X.shape[1]), dev.upload(X), X.shape[1], out)
        return np.sqrt(dev.download(out, 1, len(X))[0]) * np.asarray(outputs, dtype=self.dtype)   # :220-227

    def inv(self, inputs, outputs, noise=False, values=None):
        f = self._factor(inputs, outputs, noise, values)
        return self.device.download(f['a'], 1, f['N'])[0]                                     # :229-234

    def logdet_dinv(self, inputs, outputs, values=None):
        return -self._factor(inputs, outputs, True, values)['st']['logdet']                   # :236-238

    def posterior(self, space, pred, inputs, outputs, noise_pred=False, noise_obs=True, diag=False, inv=False, values=None):
        """rows N.. of chol(joint covariance) [L^-1 outputs; pred] without forming the joint matrix (:239-257)"""
        dev = self.device
        f = self._factor(inputs, outputs, noise_obs, values, space=space)
        M, Mp, Np, d = f['M'], f['Mp'], f['Np'], f['d']
        es = self.dtype.itemsize
        V = dev.wrap(f['K'].ptr + (Np + _lib.G3_RHS_PAD) * f['K'].ld * es, Mp, Np, f['K'].ld, self.dtype, keep=f['K'])
        Kss = dev.alloc(Mp, Mp, self.dtype, zero=True)
        dev.gram(self._prog(noise_pred, values, d), dev.upload(f['S']), None, d, Kss, M, M, 0)
        dev.gemm_nt(Kss, V, V, Mp, Mp, Np, alpha=-1.0, beta=1.0)                              # Schur complement
        Ld = dev.alloc(Mp, Mp, self.dtype, zero=True)
        dev.potrf_robust(Kss, Ld, M)
        return f['mu'] + self._lower_times(Ld, M, Mp, pred)
