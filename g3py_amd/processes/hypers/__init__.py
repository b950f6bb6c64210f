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
    """the model a new hyper-parameter registers with: the argument, else the innermost `with Model():`"""
    if model is None:
        if not _MODEL_STACK:
            raise TypeError('No model on context stack.')
        model = _MODEL_STACK[-1]
    return model


zeros, ones = np.zeros, np.ones


class Slot:
    """One hyper-parameter a `Hypers` subclass owns: the attribute it lives in, whether it is positive
    (stored as a log, `FlatExp`) or free (`Flat`), the suffix of its registered name, whether it has one
    entry per selected column, and whether the owner's own name is part of the registered name
    (`<parent><Owner><suffix>` vs `<parent><suffix>`).  The registered names are the parameter-dict
    contract of the reference."""
    __slots__ = ('attr', 'positive', 'suffix', 'per_column', 'owner_named')

    def __init__(self, attr, positive, suffix, per_column=False, owner_named=True):
        self.attr, self.positive, self.suffix = attr, positive, suffix
        self.per_column, self.owner_named = per_column, owner_named


def _columns(a):
    return a.shape[1] if len(a.shape) > 1 else 1


class Hypers:
    """Base of every hyper-parametric function.  A subclass lists its parameters in `SLOTS`; construction
    (`Cls(x, name, <slot values...>)`, by position or keyword), registration (`check_hypers`) and the
    `hypers` list all follow from that table."""
    SLOTS = ()

    def __init__(self, x=None, name=None, *given, **named):
        self.name = type(self).__name__ if name is None else name
        self.hypers, self.potential = [], None
        self.shape = self.dims = None
        attrs = [s.attr for s in self.SLOTS]
        if len(given) > len(attrs):
            raise TypeError('%s takes at most %d hyper-parameters' % (type(self).__name__, len(attrs)))
        supplied = dict(zip(attrs, given))
        for k, v in named.items():
            if k not in attrs or k in supplied:
                raise TypeError('%s got an unexpected or repeated argument %r' % (type(self).__name__, k))
            supplied[k] = v
        for a in attrs:
            setattr(self, a, supplied.get(a))
        if x is not None:
            self.check_dims(x)

    def __str__(self):
        return type(self).__name__ + ('[h=%s]' % (self.hypers,) if self.hypers else '')

    __repr__ = __str__

    def check_dims(self, x=None):
        """column selection (hypers/__init__.py:55-83): a list picks columns, a (domain, columns) tuple
        picks columns of a domain of known width, an array means all of its columns; decided once"""
        if self.shape is not None:
            return
        if x is None:
            self.dims = slice(None)
        elif type(x) is list:
            cols = np.array(x)
            self.dims, self.shape = cols, cols.shape
        elif type(x) is tuple:
            domain, cols = x
            self.dims, self.shape = cols, _columns(domain)
        else:
            self.shape = _columns(np.asarray(x))
            self.dims = slice(0, self.shape)

    def dims_index(self, d):
        """self.dims as an explicit index array for an input with d columns (None = all)"""
        if self.dims is None:
            return None
        idx = np.arange(d)[self.dims]
        if len(idx) == d and np.array_equal(idx, np.arange(d)):
            return None
        return idx

    def check_hypers(self, parent=''):
        """register the parameters that were not supplied and list all of them in `self.hypers`"""
        for s in self.SLOTS:
            if getattr(self, s.attr) is None:
                label = parent + (self.name if s.owner_named else '') + s.suffix
                make = Hypers.FlatExp if s.positive else Hypers.Flat
                setattr(self, s.attr, make(label, shape=self.shape) if s.per_column else make(label))
            self.hypers.append(getattr(self, s.attr))

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
        pattern, reg, c = self.potential
        sel = [k for k in self.hypers if isinstance(k, HyperVar) and k.name.find(pattern) > 0]
        pot = ('_'.join((self.name, pattern, reg)), reg, float(c), sel)
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
    SLOTS = (Slot('degree', True, '_degree'),)

    def __init__(self, x=None, name=None, degree=None, bound=np.float32(2.0)):
        Hypers.__init__(self, x, name, degree)
        self.bound = bound

    def default_hypers(self, x=None, y=None):
        return {self.degree: np.float64(y.shape[0])}

    def default_hypers_dims(self, x=None, y=None):
        return dict(self.default_hypers(x, y))

    def __call__(self, values=None):
        return float(self.bound) + float(np.asarray(value_of(self.degree, values or {})))

    def rows(self, values_rows, B):
        """nu for B rows of hyper values -> (B,) float64"""
        return float(self.bound) + np.broadcast_to(np.asarray(value_of(self.degree, values_rows), dtype=np.float64).reshape(-1), (B,))


def value_of(h, values):
    """numeric value of a hyper slot: HyperVar -> looked up (natural space), constant -> itself"""
    if isinstance(h, HyperVar):
        return values[h.name]
    return h
