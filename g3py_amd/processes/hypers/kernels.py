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


def _free_slot(owner, attr, label, shape=()):
    """A hyper-parameter slot of a parametric function: `None` means "make it a free positive
    variable called <label>"; whatever ends up in the slot is registered in `owner.hypers` when it
    is a free variable (a number stays a constant).  One helper for var / alpha / freq / rate."""
    value = getattr(owner, attr)
    if value is None:
        value = Hypers.FlatExp(label, shape=shape)
        setattr(owner, attr, value)
    if isinstance(value, HyperVar):
        owner.hypers.append(value)
    return value


def _combine(left, right, node_kernels, node_scalar):
    """`left (op) right` of the kernel algebra: two kernels make a composition (operand order kept),
    a kernel and a number make a scale / shift of the kernel (boundary: kernels.py:51-75)"""
    lk, rk = isinstance(left, Kernel), isinstance(right, Kernel)
    if lk and rk:
        return node_kernels(left, right)
    return node_scalar(left, right) if lk else node_scalar(right, left)


class Kernel(Hypers):
    """covariance function = var * (something of a metric); `metric` may be a class (instantiated on
    the same columns) or a ready instance"""

    METRIC = Delta          # the metric a subclass is built on unless the caller passes one

    def __init__(self, x=None, name=None, metric=None, var=None):
        metric = type(self).METRIC if metric is None else metric
        self.metric = metric(x) if isinstance(metric, type) else metric
        Hypers.__init__(self, x, name)
        self.var = var

    def check_hypers(self, parent=''):
        _free_slot(self, 'var', parent + self.name + '_var')
        self.metric.check_hypers(parent + self.name + '_')
        self.hypers.extend(h for h in self.metric.hypers if isinstance(h, HyperVar))

    def check_dims(self, x=None):
        super().check_dims(x)
        self.metric.check_dims(x)

    def default_hypers(self, x=None, y=None):
        defaults = dict(self.metric.default_hypers(x, y))
        if isinstance(self.var, HyperVar):
            defaults[self.var] = y.var()
        return defaults

    # ---- algebra
    def __mul__(self, other):
        return _combine(self, other, KernelProd, KernelScale)

    def __rmul__(self, other):
        return _combine(other, self, KernelProd, KernelScale)

    def __add__(self, other):
        return _combine(self, other, KernelSum, KernelShift)

    def __radd__(self, other):
        return _combine(other, self, KernelSum, KernelShift)

    __imul__, __iadd__ = __mul__, __add__

    def __repr__(self):
        return '%s[m=%s,h=%s]' % (type(self).__name__, self.metric, self.hypers)
    __str__ = __repr__

    # ---- evaluation
    def spec(self, values, d):
        raise NotImplementedError

    def cov(self, x1, x2=None, params=None, dtype=np.float64, device=None):
        """Kernel.cov(x1, x2=None) -- kernels.py:106-110: Gram matrix on the GPU.
        `params`: natural-space values keyed by hyper name (only needed for free hypers)."""
        from ...device import Device, compile_spec
        dev = device or Device.default()

        def columns(x):
            x = np.asarray(x, dtype=dtype)
            return x[:, None] if x.ndim == 1 else x
        x1 = columns(x1)
        d = x1.shape[1]
        prog = compile_spec(self.spec(params or {}, d), d)
        X1 = dev.upload(x1)
        X2, n2 = None, x1.shape[0]
        if x2 is not None:
            x2 = columns(x2)
            X2, n2 = dev.upload(x2), x2.shape[0]
        out = dev.alloc(x1.shape[0], n2, dtype)
        dev.gram(prog, X1, X2, d, out, x1.shape[0], n2, 0)
        return dev.download(out)
    __call__ = cov


class KernelStationary(Kernel):
    """cov = var * k(metric.gram) -- kernels.py:96-110"""
    kind = None
    METRIC = ARD_L2

    def spec(self, values, d):
        return (self.kind, value_of(self.var, values), value_of(self.metric.rate, values),
                self.metric.dims_index(d))


class _KernelNode(Kernel):
    """Interior node of a kernel expression: it owns no hyper-parameters, it forwards every
    bookkeeping request to its kernel operands (`parts`) and joins the answers.  `op` is the
    symbol shown in names; constants of scale / shift nodes live in `element`."""
    op = 'op'

    def _init_node(self):
        self.hypers = []
        self.potential = None

    @property
    def parts(self):
        raise NotImplementedError

    def _labels(self, how):
        raise NotImplementedError

    def check_hypers(self, parent=''):
        collected = []
        for part in self.parts:
            part.check_hypers(parent=parent)
            collected += part.hypers
        self.hypers = collected if len(self.parts) > 1 else self.parts[0].hypers

    def check_dims(self, x=None):
        for part in self.parts:
            part.check_dims(x)

    def default_hypers_dims(self, x=None, y=None):
        merged = {}
        for part in self.parts:
            merged.update(part.default_hypers_dims(x, y))
        return merged

    def check_potential(self):
        Hypers.check_potential(self)
        for part in self.parts:
            part.check_potential()

    @property
    def name(self):
        return (' ' + self.op + ' ').join(self._labels(lambda k: k.name))

    def __repr__(self):
        return (' ' + self.op + ' ').join(self._labels(str))
    __str__ = __repr__


class KernelOperation(_KernelNode):
    """a kernel combined with a constant (`element`)"""

    def __init__(self, _k, _element):
        self.k, self.element = _k, _element
        self._init_node()

    @property
    def parts(self):
        return (self.k,)

    def _labels(self, how):
        return [str(self.element), how(self.k)]


class KernelComposition(_KernelNode):
    """two kernels combined"""

    def __init__(self, _k1, _k2):
        self.k1, self.k2 = _k1, _k2
        self._init_node()

    @property
    def parts(self):
        return (self.k1, self.k2)

    def _labels(self, how):
        return [how(self.k1), how(self.k2)]


class KernelScale(KernelOperation):
    """element * k.cov -- kernels.py:192-200"""
    op = '*'

    def spec(self, values, d):
        return ('scale', float(self.element), self.k.spec(values, d))


class KernelShift(KernelOperation):
    """element + k.cov -- kernels.py:203-211"""
    op = '+'

    def spec(self, values, d):
        return ('shift', float(self.element), self.k.spec(values, d))


class KernelProd(KernelComposition):
    """k1.cov * k2.cov -- kernels.py:214-229.  Two free variances in a product are not
    identifiable: when neither factor has its variance set, the second one is pinned to 1."""
    op = '*'

    def __init__(self, _k1, _k2):
        super().__init__(_k1, _k2)
        unset = [getattr(k, 'var', 0) is None for k in (_k1, _k2)]
        if all(unset):
            _k2.var = 1.0

    def spec(self, values, d):
        return ('prod', self.k1.spec(values, d), self.k2.spec(values, d))


class KernelSum(KernelComposition):
    """k1.cov + k2.cov -- kernels.py:232-244"""
    op = '+'

    def spec(self, values, d):
        return ('sum', self.k1.spec(values, d), self.k2.spec(values, d))


class KernelNoise(KernelStationary):
    """var * I when square, zeros for a cross block -- kernels.py:360-371"""
    kind = 'NOISE'
    METRIC = Delta

    def spec(self, values, d):
        return ('NOISE', value_of(self.var, values))


class WN(KernelStationary):
    """var * I when square, var * Delta.gram for a cross block -- kernels.py:374-385"""
    kind = 'WN'
    METRIC = Delta

    def spec(self, values, d):
        return ('WN', value_of(self.var, values), self.metric.dims_index(d))


class RQ(KernelStationary):
    """(1 + d/alpha)^(-alpha) -- kernels.py:388-403"""
    kind = 'RQ'

    def __init__(self, x=None, name=None, metric=None, var=None, alpha=None):
        Kernel.__init__(self, x, name, metric, var)
        self.alpha = alpha

    def check_hypers(self, parent=''):
        super().check_hypers(parent=parent)
        _free_slot(self, 'alpha', parent + self.name + '_alpha')

    def default_hypers(self, x=None, y=None):
        defaults = super().default_hypers(x, y)
        defaults[self.alpha] = 1.0
        return defaults

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
    METRIC = ARD_L1


class SE(KernelStationaryExponential):
    kind = 'SE'


class KernelPeriodic(KernelStationary):
    """periodic family on the Difference metric -- kernels.py:439-459"""
    uses_rate = True        # COS and SINC have no decay: their rate slot is the constant 1
    METRIC = Difference

    def __init__(self, x=None, name=None, metric=None, var=None, freq=None, rate=None):
        Kernel.__init__(self, x, name, metric, var)
        self.freq = freq
        self.rate = rate if type(self).uses_rate else 1.0

    def check_hypers(self, parent=''):
        super().check_hypers(parent=parent)
        # creation order freq, rate fixes the variables' positions in the flat parameter vector
        # (boundary: the reference creates them in this order and lists rate first in `hypers`)
        made = {a: getattr(self, a) is None for a in ('freq', 'rate')}
        for attr in ('freq', 'rate'):
            if made[attr]:
                setattr(self, attr, Hypers.FlatExp(parent + self.name + '_' + attr, shape=self.shape))
        self.hypers.extend(h for h in (self.rate, self.freq) if isinstance(h, HyperVar))

    def default_hypers(self, x=None, y=None):
        # one full period over the observed span; decay length of the mean spacing of consecutive rows
        guesses = (('freq', lambda: 1 / (x.max(axis=0) - x.min(axis=0))),
                   ('rate', lambda: 1 / np.abs(np.diff(x, axis=0)).mean(axis=0)))
        defaults = {getattr(self, a): guess() for a, guess in guesses if isinstance(getattr(self, a), HyperVar)}
        for k, v in super().default_hypers(x, y).items():
            defaults[k] = v
        return defaults

    def spec(self, values, d):
        if self.uses_rate:
            return (self.kind, value_of(self.var, values), value_of(self.freq, values),
                    value_of(self.rate, values), self.metric.dims_index(d))
        return (self.kind, value_of(self.var, values), value_of(self.freq, values), self.metric.dims_index(d))


class COS(KernelPeriodic):
    """prod_k cos(2 pi dx_k f_k) -- kernels.py:462-467"""
    kind = 'COS'
    uses_rate = False


class SIN(KernelPeriodic):
    """exp(+2 sum_k rate_k sin^2(pi dx_k f_k)) -- kernels.py:470-472 (sign as in the reference)"""
    kind = 'SIN'


class SINC(KernelPeriodic):
    """prod_k sinc -- kernels.py:475-482"""
    kind = 'SINC'
    uses_rate = False


class SM(KernelPeriodic):
    """exp(-2 pi^2 sum dx^2 rate^2) prod cos(2 pi dx f) -- kernels.py:485-487"""
    kind = 'SM'
