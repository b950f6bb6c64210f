"""Kernel algebra and the stationary kernel library of g3py
(g3py/processes/hypers/kernels.py:13-79, 96-110, 192-244, 360-487), MI355X edition.

A kernel object describes a covariance function; `spec(values, d)` turns it into the plain
nested-tuple description that `g3py_amd.device.compile_spec` lowers to a `g3_kernel_prog`,
and `cov(x1, x2=None, params=...)` evaluates it on the GPU through `g3_gram`.
"""
import numpy as np

from . import Hypers, HyperVar, value_of
from .metrics import Delta, Difference, ARD_L1, ARD_L2

pi = np.pi


class Kernel(Hypers):
    def __init__(self, x=None, name=None, metric=Delta, var=None):
        if type(metric) is type:
            self.metric = metric(x)
        else:
            self.metric = metric
        super().__init__(x, name)
        self.var = var

    def check_hypers(self, parent=''):
        if self.var is None:
            self.var = Hypers.FlatExp(parent + self.name + '_var')
        if isinstance(self.var, HyperVar):
            self.hypers += [self.var]
        self.metric.check_hypers(parent + self.name + '_')
        self.hypers += [h for h in self.metric.hypers if isinstance(h, HyperVar)]

    def check_dims(self, x=None):
        super().check_dims(x)
        self.metric.check_dims(x)

    def default_hypers(self, x=None, y=None):
        if isinstance(self.var, HyperVar):
            return {self.var: y.var(), **self.metric.default_hypers(x, y)}
        return self.metric.default_hypers(x, y)

    # ---- algebra (kernels.py:51-75)
    def __mul__(self, other):
        if issubclass(type(other), Kernel):
            return KernelProd(self, other)
        return KernelScale(self, other)
    __imul__ = __mul__

    def __rmul__(self, other):
        if issubclass(type(other), Kernel):
            return KernelProd(other, self)
        return KernelScale(self, other)

    def __add__(self, other):
        if issubclass(type(other), Kernel):
            return KernelSum(self, other)
        return KernelShift(self, other)
    __iadd__ = __add__

    def __radd__(self, other):
        if issubclass(type(other), Kernel):
            return KernelSum(other, self)
        return KernelShift(self, other)

    def __str__(self):
        return str(self.__class__.__name__) + '[m=' + str(self.metric) + ',h=' + str(self.hypers) + ']'
    __repr__ = __str__

    # ---- evaluation
    def spec(self, values, d):
        raise NotImplementedError

    def cov(self, x1, x2=None, params=None, dtype=np.float64, device=None):
        """Kernel.cov(x1, x2=None) -- kernels.py:106-110: Gram matrix on the GPU.
        `params`: natural-space values keyed by hyper name (only needed for free hypers)."""
        from ...device import Device, compile_spec
        from ..._lib import G3_GRAM_SCRUB
        dev = device or Device.default()
        x1 = np.asarray(x1, dtype=dtype)
        x1 = x1[:, None] if x1.ndim == 1 else x1
        d = x1.shape[1]
        prog = compile_spec(self.spec(params or {}, d), d)
        X1 = dev.upload(x1)
        X2 = None
        n2 = x1.shape[0]
        if x2 is not None:
            x2 = np.asarray(x2, dtype=dtype)
            x2 = x2[:, None] if x2.ndim == 1 else x2
            X2 = dev.upload(x2)
            n2 = x2.shape[0]
        out = dev.alloc(x1.shape[0], n2, dtype)
        dev.gram(prog, X1, X2, d, out, x1.shape[0], n2, 0)
        return dev.download(out)
    __call__ = cov


class KernelStationary(Kernel):
    """cov = var * k(metric.gram) -- kernels.py:96-110"""
    kind = None

    def __init__(self, x=None, name=None, metric=ARD_L2, var=None):
        super().__init__(x, name, metric, var)

    def spec(self, values, d):
        return (self.kind, value_of(self.var, values), value_of(self.metric.rate, values),
                self.metric.dims_index(d))


class KernelOperation(Kernel):
    def __init__(self, _k, _element):
        self.k = _k
        self.element = _element
        self.hypers = []
        self.potential = None
        self.op = 'op'

    def check_hypers(self, parent=''):
        self.k.check_hypers(parent=parent)
        self.hypers = self.k.hypers

    def check_dims(self, x=None):
        self.k.check_dims(x)

    def default_hypers_dims(self, x=None, y=None):
        return self.k.default_hypers_dims(x, y)

    def check_potential(self):
        super().check_potential()
        self.k.check_potential()

    @property
    def name(self):
        return str(self.element) + " " + self.op + " " + self.k.name

    def __str__(self):
        return str(self.element) + " " + self.op + " " + str(self.k)
    __repr__ = __str__


class KernelComposition(Kernel):
    def __init__(self, _k1, _k2):
        self.k1 = _k1
        self.k2 = _k2
        self.hypers = []
        self.potential = None
        self.op = 'op'

    def check_hypers(self, parent=''):
        self.k1.check_hypers(parent=parent)
        self.k2.check_hypers(parent=parent)
        self.hypers = self.k1.hypers + self.k2.hypers

    def check_dims(self, x=None):
        self.k1.check_dims(x)
        self.k2.check_dims(x)

    def default_hypers_dims(self, x=None, y=None):
        return {**self.k1.default_hypers_dims(x, y), **self.k2.default_hypers_dims(x, y)}

    def check_potential(self):
        super().check_potential()
        self.k1.check_potential()
        self.k2.check_potential()

    @property
    def name(self):
        return self.k1.name + " " + self.op + " " + self.k2.name

    def __str__(self):
        return str(self.k1) + " " + self.op + " " + str(self.k2)
    __repr__ = __str__


class KernelScale(KernelOperation):
    """element * k.cov -- kernels.py:192-200"""

    def __init__(self, _k, _element):
        super().__init__(_k, _element)
        self.op = '*'

    def spec(self, values, d):
        return ('scale', float(self.element), self.k.spec(values, d))


class KernelShift(KernelOperation):
    """element + k.cov -- kernels.py:203-211"""

    def __init__(self, _k, _element):
        super().__init__(_k, _element)
        self.op = '+'

    def spec(self, values, d):
        return ('shift', float(self.element), self.k.spec(values, d))


class KernelProd(KernelComposition):
    """k1.cov * k2.cov -- kernels.py:214-229"""

    def __init__(self, _k1, _k2):
        super().__init__(_k1, _k2)
        if hasattr(self.k1, 'var') and hasattr(self.k2, 'var'):
            if self.k1.var is None and self.k2.var is None:
                self.k2.var = 1.0
        self.op = '*'

    def spec(self, values, d):
        return ('prod', self.k1.spec(values, d), self.k2.spec(values, d))


class KernelSum(KernelComposition):
    """k1.cov + k2.cov -- kernels.py:232-244"""

    def __init__(self, _k1, _k2):
        super().__init__(_k1, _k2)
        self.op = '+'

    def spec(self, values, d):
        return ('sum', self.k1.spec(values, d), self.k2.spec(values, d))


class KernelNoise(KernelStationary):
    """var * I when square, zeros for a cross block -- kernels.py:360-371"""
    kind = 'NOISE'

    def __init__(self, x=None, name=None, metric=Delta, var=None):
        super().__init__(x, name, metric, var)

    def spec(self, values, d):
        return ('NOISE', value_of(self.var, values))


class WN(KernelStationary):
    """var * I when square, var * Delta.gram for a cross block -- kernels.py:374-385"""
    kind = 'WN'

    def __init__(self, x=None, name=None, metric=Delta, var=None):
        super().__init__(x, name, metric, var)

    def spec(self, values, d):
        return ('WN', value_of(self.var, values), self.metric.dims_index(d))


class RQ(KernelStationary):
    """(1 + d/alpha)^(-alpha) -- kernels.py:388-403"""
    kind = 'RQ'

    def __init__(self, x=None, name=None, metric=ARD_L2, var=None, alpha=None):
        super().__init__(x, name, metric, var)
        self.alpha = alpha

    def check_hypers(self, parent=''):
        super().check_hypers(parent=parent)
        if self.alpha is None:
            self.alpha = Hypers.FlatExp(parent + self.name + '_alpha')
        if isinstance(self.alpha, HyperVar):
            self.hypers += [self.alpha]

    def default_hypers(self, x=None, y=None):
        return {self.alpha: 1.0, **super().default_hypers(x, y)}

    def spec(self, values, d):
        return ('RQ', value_of(self.var, values), value_of(self.metric.rate, values),
                value_of(self.alpha, values), self.metric.dims_index(d))


class MAT32(KernelStationary):
    """(1 + sqrt(3d)) exp(-sqrt(3d)) -- kernels.py:406-412"""
    kind = 'MAT32'


class MAT52(KernelStationary):
    """(1 + sqrt(5d) + 5d/3) exp(-sqrt(5d)) -- kernels.py:415-421"""
    kind = 'MAT52'


class KernelStationaryExponential(KernelStationary):
    """exp(-d) -- kernels.py:424-426"""


class OU(KernelStationaryExponential):
    kind = 'OU'

    def __init__(self, x=None, name=None, metric=ARD_L1, var=None):
        super().__init__(x, name, metric, var)


class SE(KernelStationaryExponential):
    kind = 'SE'

    def __init__(self, x=None, name=None, metric=ARD_L2, var=None):
        super().__init__(x, name, metric, var)


class KernelPeriodic(KernelStationary):
    """periodic family on the Difference metric -- kernels.py:439-459"""
    uses_rate = True

    def __init__(self, x=None, name=None, metric=Difference, var=None, freq=None, rate=None):
        super().__init__(x, name, metric, var)
        self.freq = freq
        self.rate = rate

    def check_hypers(self, parent=''):
        super().check_hypers(parent=parent)
        if self.freq is None:
            self.freq = Hypers.FlatExp(parent + self.name + '_freq', shape=self.shape)
        if self.rate is None:
            self.rate = Hypers.FlatExp(parent + self.name + '_rate', shape=self.shape)
        if isinstance(self.rate, HyperVar):
            self.hypers += [self.rate]
        if isinstance(self.freq, HyperVar):
            self.hypers += [self.freq]

    def default_hypers(self, x=None, y=None):
        r = {}
        if isinstance(self.freq, HyperVar):
            r[self.freq] = 1 / (x.max(axis=0) - x.min(axis=0))
        if isinstance(self.rate, HyperVar):
            r[self.rate] = 1 / np.abs(x[1:] - x[:-1]).mean(axis=0)
        return {**r, **super().default_hypers(x, y)}

    def spec(self, values, d):
        if self.uses_rate:
            return (self.kind, value_of(self.var, values), value_of(self.freq, values),
                    value_of(self.rate, values), self.metric.dims_index(d))
        return (self.kind, value_of(self.var, values), value_of(self.freq, values), self.metric.dims_index(d))


class COS(KernelPeriodic):
    """prod_k cos(2 pi dx_k f_k) -- kernels.py:462-467"""
    kind = 'COS'
    uses_rate = False

    def __init__(self, x=None, name=None, metric=Difference, var=None, freq=None):
        super().__init__(x, name, metric, var, freq, rate=1.0)


class SIN(KernelPeriodic):
    """exp(+2 sum_k rate_k sin^2(pi dx_k f_k)) -- kernels.py:470-472 (sign as in the reference)"""
    kind = 'SIN'


class SINC(KernelPeriodic):
    """prod_k sinc -- kernels.py:475-482"""
    kind = 'SINC'
    uses_rate = False

    def __init__(self, x=None, name=None, metric=Difference, var=None, freq=None):
        super().__init__(x, name, metric, var, freq, rate=1.0)


class SM(KernelPeriodic):
    """exp(-2 pi^2 sum dx^2 rate^2) prod cos(2 pi dx f) -- kernels.py:485-487"""
    kind = 'SM'
