"""Small host-side helpers carrying the reference's names (boundary: g3py/libs/__init__.py:17-60).

`DictObj` is the return type of `predict`, `scores` and `params`: a dict whose keys also read as
attributes.  Only the behaviour is the boundary -- keys as attributes, AttributeError (not KeyError)
for a missing name, `clone()` / `copy()` giving an independent DictObj, construction from a mapping
passed as `data=`.
"""
import copy as _copy

_MISSING = object()


class DictObj(dict):
    """dict with attribute access."""

    __slots__ = ()

    def __init__(self, data=None, *args, **kwargs):
        dict.__init__(self, *args, **kwargs)
        if data is not None:
            self.update(data)

    # attribute protocol on top of the mapping: one lookup, AttributeError on a miss
    def __getattr__(self, name):
        value = dict.get(self, name, _MISSING)
        if value is _MISSING:
            raise AttributeError("No such attribute: " + name)
        return value

    __setattr__ = dict.__setitem__

    def __delattr__(self, name):
        if dict.pop(self, name, _MISSING) is _MISSING:
            raise AttributeError("No such attribute: " + name)

    def copy(self):
        return type(self)(data=self)

    clone = copy


def clone(c):
    """shallow copy (g3py/libs/__init__.py:55-56)"""
    return _copy.copy(c)
