"""Hyper-parametric function base: naming, column selection and the log-transform convention
of the reference (g3py/processes/hypers/__init__.py:35-142, 190-202) without PyMC3.

A hyper-parameter is a `HyperVar` registered in the active `Model` (the stand-in for the
PyMC3 model context).  Positive hypers (`FlatExp`) are stored by users in TRANSFORMED space
under `<name>_log_` exactly as in the reference's parameter dicts
(notebooks/07-Student-t-Process.ipynb:170-173).
"""
import numpy as np

from ...libs import DictObj

_MODEL_STACK = []


class HyperVar:
    """Stand-in for a PyMC3 free RV created by Hypers.Flat / Hypers.FlatExp."""

    def __init__(self, name, shape=(), positive=False):
        self.name = name
        self.shape = () if shape in ((), None) else ((int(shape),) if np.isscalar(shape) else tuple(shape))
        self.positive = positive

    @property
    def key(self):
        """name in the params dict: `<name>_log_` for FlatExp variables"""
        return self.name + '_log_' if self.positive else self.name

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    def __repr__(self):
        return self.key


class Model:
    """Ordered registry of free variables (creation order = pm.ArrayOrdering order,
    g3py/bayesian/models.py:143-155)."""

    def __init__(self, name=''):
        self.name = name
        self.vars = []
        self.potentials = []   # (name, 'L1'|'L2', c, [HyperVar]) -- pm.Potential stand-ins

    def __enter__(self):
        _MODEL_STACK.append(self)
        return self

    def __exit__(self, *exc):
        _MODEL_STACK.pop()

    def add(self, v):
        for u in self.vars:
            if u.name == v.name:
                raise ValueError('Variable name %s already exists.' % v.name)
        self.vars.append(v)
        return v

    @property
    def test_point(self):
        """Flat testval=zeros, FlatExp testval=ones -> log(1)=0 (hypers/__init__.py:116-126)"""
        return DictObj({v.key: np.zeros(v.shape) for v in self.vars})

    @property
    def ndim(self):
        return sum(v.size for v in self.vars)


def modelcontext(model=None):
    if model is not None:
        return model
    if not _MODEL_STACK:
        raise TypeError('No model on context stack.')
    return _MODEL_STACK[-1]


def zeros(shape):
    return np.zeros(shape)


def ones(shape):
    return np.ones(shape)


class Hypers:
    def __init__(self, x=None, name=None):
        self.name = self.__class__.__name__ if name is None else name
        self.hypers = []
        self.shape = None
        self.dims = None
        self.potential = None
        if x is not None:
            self.check_dims(x)

    def __str__(self):
        if len(self.hypers) == 0:
            return str(self.__class__.__name__)
        return str(self.__class__.__name__) + '[h=' + str(self.hypers) + ']'
    __repr__ = __str__

    def check_dims(self, x=None):
        """column selection rules of hypers/__init__.py:55-83"""
        if self.shape is not None:
            return
        if x is not None:
            if type(x) is list:
                self.dims = np.array(x)
                self.shape = self.dims.shape
            elif type(x) is tuple:
                domain, self.dims = x
                self.shape = domain.shape[1] if len(domain.shape) > 1 else 1
            else:
                x = np.asarray(x)
                self.shape = x.shape[1] if len(x.shape) > 1 else 1
                self.dims = slice(0, self.shape)
        else:
            self.shape = None
            self.dims = slice(None)

    def dims_index(self, d):
        """self.dims as an explicit index array for an input with d columns (None = all)"""
        if self.dims is None:
            return None
        idx = np.arange(d)[self.dims]
        if len(idx) == d and np.array_equal(idx, np.arange(d)):
            return None
        return idx

    def check_hypers(self, parent=''):
        pass

    def default_hypers(self, x=None, y=None):
        return {}

    def default_hypers_dims(self, x=None, y=None):
        return dict(self.default_hypers(x[:, self.dims], y))

    def set_potential(self, hypers='', reg='L1', c=1):
        """optional L1 / L2 regulariser on the hypers whose name contains `hypers`
        (hypers/__init__.py:94-95)"""
        self.potential = (hypers, reg, c)

    def check_potential(self):
        """register the potential with the active model (hypers/__init__.py:97-109); the value
        is c * (-sum |h|) or c * (-sum h^2) over the NATURAL-space hypers selected by name"""
        if getattr(self, 'potential', None) is None:
            return None
        hypers, reg, c = self.potential
        sel = [k for k in self.hypers if isinstance(k, HyperVar) and k.name.find(hypers) > 0]
        pot = (self.name + '_' + hypers + '_' + reg, reg, float(c), sel)
        modelcontext().potentials.append(pot)
        return pot

    @staticmethod
    def Flat(name, shape=()):
        return modelcontext().add(HyperVar(name, shape, positive=False))

    @staticmethod
    def FlatExp(name, shape=()):
        return modelcontext().add(HyperVar(name, shape, positive=True))


class Freedom(Hypers):
    """degrees of freedom of the Student-t process: bound + degree with degree > 0
    (hypers/__init__.py:144-160)"""

    def __init__(self, x=None, name=None, degree=None, bound=np.float32(2.0)):
        super().__init__(x, name)
        self.degree = degree
        self.bound = bound

    def check_hypers(self, parent=''):
        super().check_hypers(parent=parent)
        if self.degree is None:
            self.degree = Hypers.FlatExp(parent + self.name + '_degree')
        self.hypers += [self.degree]

    def default_hypers(self, x=None, y=None):
        return {self.degree: np.float64(y.shape[0])}

    def default_hypers_dims(self, x=None, y=None):
        return dict(self.default_hypers(x, y))

    def __call__(self, values=None):
        return float(self.bound) + float(np.asarray(value_of(self.degree, values or {})))


def value_of(h, values):
    """numeric value of a hyper slot: HyperVar -> looked up (natural space), constant -> itself"""
    if isinstance(h, HyperVar):
        return values[h.name]
    return h
