"""Warping maps on the path: Identity, LinearMapping, LogShifted, BoxCoxLinear, ArcsinhLinear
(g3py/processes/hypers/mappings.py:88-215, 309-333): element-wise O(N) host arithmetic
(`__call__`, `inv`, `logdet_dinv`) around the device path."""
import numpy as np

from . import Hypers, Slot, value_of


class Mapping(Hypers):
    def __call__(self, x, values=None):
        raise NotImplementedError

    def inv(self, y, values=None):
        raise NotImplementedError

    def logdet_dinv(self, y, values=None):
        raise NotImplementedError

    def inv_rows(self, y, values_rows, B):
        """inv(y) for B rows of hyper values at once -> (B, N).  Base: row by row."""
        return np.stack([np.asarray(self.inv(y, {k: np.asarray(v)[j] for k, v in values_rows.items()}), dtype=y.dtype)
                         for j in range(B)])

    def logdet_dinv_rows(self, y, values_rows, B):
        with np.errstate(all='ignore'):
            return np.array([self.logdet_dinv(y, {k: np.asarray(v)[j] for k, v in values_rows.items()})
                             for j in range(B)], dtype=y.dtype)

    def grad_rows(self, y, values_rows, B):
        """grad for B rows of hyper values: [(hyper, (B, N), (B,))].  Base: row by row."""
        per = [self.grad(y, {k: np.asarray(v)[j] for k, v in values_rows.items()}) for j in range(B)]
        if not per or not per[0]:
            return []
        return [(per[0][i][0], np.stack([np.asarray(p[i][1], dtype=np.float64) for p in per]),
                 np.array([float(p[i][2]) for p in per])) for i in range(len(per[0]))]

    def grad(self, y, values=None):
        """[(hyper, d inv(y) / d hyper (length N), d logdet_dinv(y) / d hyper)] in natural space:
        the pieces th_dlogp needs from the warping (mappings.py:88-215, 309-333 differentiated)"""
        return []


class Identity(Mapping):
    def __call__(self, x, values=None):
        return x

    def inv(self, y, values=None):
        return y

    def logdet_dinv(self, y, values=None):
        return 0.0

    def inv_rows(self, y, values_rows, B):
        return np.broadcast_to(y, (B, y.shape[0]))

    def logdet_dinv_rows(self, y, values_rows, B):
        return np.zeros(B, dtype=y.dtype)

    def grad_rows(self, y, values_rows, B):
        return []


class LinearMapping(Mapping):
    SLOTS = (Slot('shift', False, '_shift'), Slot('scale', True, '_scale'))

    def default_hypers(self, x=None, y=None):
        return {self.shift: 0.0, self.scale: 1.0}

    def _p(self, values, t):
        return t(value_of(self.shift, values)), t(value_of(self.scale, values))

    def __call__(self, x, values=None):
        shift, scale = self._p(values, x.dtype.type)
        return scale * (x - shift)

    def inv(self, y, values=None):
        shift, scale = self._p(values, y.dtype.type)
        return y / scale + shift

    def logdet_dinv(self, y, values=None):
        shift, scale = self._p(values, y.dtype.type)
        return -y.dtype.type(y.shape[0]) * np.log(scale)

    def _p_rows(self, values_rows, B, t):
        return tuple(np.broadcast_to(np.asarray(value_of(h, values_rows), dtype=np.float64).reshape(-1), (B,)).astype(t)
                     for h in (self.shift, self.scale))

    def inv_rows(self, y, values_rows, B):
        shift, scale = self._p_rows(values_rows, B, y.dtype.type)
        return y[None, :] / scale[:, None] + shift[:, None]

    def logdet_dinv_rows(self, y, values_rows, B):
        shift, scale = self._p_rows(values_rows, B, y.dtype.type)
        return -y.dtype.type(y.shape[0]) * np.log(scale)

    def grad(self, y, values=None):
        t = y.dtype.type
        shift, scale = self._p(values, t)
        return [(self.shift, np.ones_like(y), t(0)), (self.scale, -y / scale ** 2, -t(y.shape[0]) / scale)]

    def grad_rows(self, y, values_rows, B):
        t = y.dtype.type
        shift, scale = self._p_rows(values_rows, B, t)
        return [(self.shift, np.ones((B, y.shape[0]), dtype=y.dtype), np.zeros(B, dtype=y.dtype)),
                (self.scale, -y[None, :] / scale[:, None] ** 2, -t(y.shape[0]) / scale)]


class LogShifted(Mapping):
    SLOTS = (Slot('shift', False, '_shift'),)

    def default_hypers(self, x=None, y=None):
        return {self.shift: y.min() - 1}

    def __call__(self, x, values=None):
        return np.exp(x) + x.dtype.type(value_of(self.shift, values))

    def inv(self, y, values=None):
        t = y.dtype.type
        return np.log(np.maximum(y - t(value_of(self.shift, values)), t(np.float32(1e-32))))

    def logdet_dinv(self, y, values=None):
        with np.errstate(all='ignore'):
            return -np.sum(np.log(y - y.dtype.type(value_of(self.shift, values))))

    def grad(self, y, values=None):
        t = y.dtype.type
        z = y - t(value_of(self.shift, values))
        with np.errstate(all='ignore'):
            return [(self.shift, np.where(z > t(np.float32(1e-32)), -1 / z, t(0)), np.sum(1 / z))]


class BoxCoxLinear(Mapping):
    SLOTS = (Slot('shift', False, '_shift'), Slot('scale', True, '_scale'), Slot('power', True, '_power'))

    def default_hypers(self, x=None, y=None):
        return {self.shift: 1.0, self.scale: 1.0, self.power: 1.0}

    def _p(self, values, t):
        return (t(value_of(self.shift, values)), t(value_of(self.scale, values)),
                t(value_of(self.power, values)))

    def __call__(self, x, values=None):
        t = x.dtype.type
        shift, scale, power = self._p(values, t)
        scaled = power * x + t(1)
        transformed = np.sign(scaled) * np.abs(scaled) ** (t(1) / power)
        return transformed / scale - shift

    def inv(self, y, values=None):
        t = y.dtype.type
        shift, scale, power = self._p(values, t)
        shifted = scale * (y + shift)
        with np.errstate(all='ignore'):
            if power < np.float32(1e-5):
                return np.log(shifted)
            return ((np.sign(shifted) * np.abs(shifted) ** power) - t(1)) / power

    def logdet_dinv(self, y, values=None):
        t = y.dtype.type
        shift, scale, power = self._p(values, t)
        with np.errstate(all='ignore'):
            return (power - t(1)) * np.sum(np.log(np.abs(scale * (y + shift)))) + t(y.shape[0]) * np.log(scale)

    def grad(self, y, values=None):
        t = y.dtype.type
        shift, scale, power = self._p(values, t)
        n = t(y.shape[0])
        sh = scale * (y + shift)
        with np.errstate(all='ignore'):
            a = np.abs(sh)
            if power < np.float32(1e-5):
                d_shift, d_scale, d_power = scale / sh, (y + shift) / sh, np.zeros_like(y)
            else:
                sp = np.sign(sh) * a ** power
                d_shift, d_scale = a ** (power - 1) * scale, a ** (power - 1) * (y + shift)
                d_power = (sp * np.log(a) * power - (sp - 1)) / power ** 2
            return [(self.shift, d_shift, (power - 1) * np.sum(1 / (y + shift))),
                    (self.scale, d_scale, power * n / scale),
                    (self.power, d_power, np.sum(np.log(a)))]


class ArcsinhLinear(Mapping):
    SLOTS = (Slot('shift', False, '_shift'), Slot('scale', True, '_scale'))

    def default_hypers(self, x=None, y=None):
        return {self.shift: np.mean(y), self.scale: np.std(y)}

    def _p(self, values, t):
        return t(value_of(self.shift, values)), t(value_of(self.scale, values))

    def __call__(self, x, values=None):
        shift, scale = self._p(values, x.dtype.type)
        return np.sinh((x - shift) / scale)

    def inv(self, y, values=None):
        shift, scale = self._p(values, y.dtype.type)
        return np.arcsinh(y) * scale + shift

    def logdet_dinv(self, y, values=None):
        t = y.dtype.type
        shift, scale = self._p(values, t)
        return t(y.shape[0]) * np.log(scale) - t(0.5) * np.sum(np.log1p(y ** 2))

    def grad(self, y, values=None):
        t = y.dtype.type
        shift, scale = self._p(values, t)
        return [(self.shift, np.ones_like(y), t(0)), (self.scale, np.arcsinh(y), t(y.shape[0]) / scale)]
