"""Location functions on the path: Zero, Bias, Linear
(g3py/processes/hypers/means.py:6-27, 117-159).  O(N d) host arithmetic feeding the device
path; `eval` takes natural-space hyper values."""
import numpy as np

from . import Hypers, Slot, value_of


class Mean(Hypers):
    def eval(self, x, values):
        raise NotImplementedError

    def __call__(self, x, values=None):
        x = np.asarray(x)
        return self.eval(x[:, self.dims] if self.dims is not None else x, values or {})

    def rows(self, x, values_rows, B):
        """m(x) for B rows of hyper values at once (values_rows[name][j] = row j) -> (B, N).  Base: row by row."""
        return np.stack([self(x, {k: np.asarray(v)[j] for k, v in values_rows.items()}) for j in range(B)])

    # True only for means that are linear in their hypers (d m(x) / d hyper does not depend on the hyper values): the chain
    # gradient then evaluates `jac` once for all rows.  A user-defined mean inherits False and takes the per-row loop.
    JAC_CONSTANT = False

    def jac(self, x, values):
        """[(hyper, d m(x) / d hyper as an N x size array)] for the free hypers -- what Theano's
        reverse mode propagates through means.py:117-159 for th_dlogp"""
        return []

    def grad(self, x, values=None):
        x = np.asarray(x)
        return self.jac(x[:, self.dims] if self.dims is not None else x, values or {})

Location = Mean


class Zero(Mean):
    JAC_CONSTANT = True

    def eval(self, x, values):
        return np.zeros(x.shape[0], dtype=x.dtype)

    def rows(self, x, values_rows, B):
        x = np.asarray(x)
        return np.zeros((B, x.shape[0]), dtype=x.dtype)


class Bias(Mean):
    JAC_CONSTANT = True
    SLOTS = (Slot('bias', False, '_Bias'),)

    def default_hypers(self, x=None, y=None):
        return {self.bias: y.mean()}

    def eval(self, x, values):
        return x.dtype.type(value_of(self.bias, values)) * np.ones(x.shape[0], dtype=x.dtype)

    def rows(self, x, values_rows, B):
        x = np.asarray(x)
        b = np.broadcast_to(np.asarray(value_of(self.bias, values_rows), dtype=np.float64).reshape(-1), (B,))
        return b.astype(x.dtype)[:, None] * np.ones((1, x.shape[0]), dtype=x.dtype)

    def jac(self, x, values):
        return [(self.bias, np.ones((x.shape[0], 1), dtype=x.dtype))]


class Linear(Mean):
    JAC_CONSTANT = True
    SLOTS = (Slot('constant', False, '_Constant'), Slot('coeff', False, '_Coeff', per_column=True))

    def default_hypers(self, x=None, y=None):
        return {self.constant: y.mean(), self.coeff: y.mean() / x.mean(axis=0)}

    def eval(self, x, values):
        coeff = np.broadcast_to(np.asarray(value_of(self.coeff, values), dtype=x.dtype), (x.shape[1],))
        return x.dtype.type(value_of(self.constant, values)) + np.dot(x, coeff)

    def jac(self, x, values):
        return [(self.constant, np.ones((x.shape[0], 1), dtype=x.dtype)), (self.coeff, x)]
