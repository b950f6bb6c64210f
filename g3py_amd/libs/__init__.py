"""Small host-side helpers with the reference's names (g3py/libs/__init__.py:17-60)."""
from copy import copy


class DictObj(dict):
    """dict with attribute access -- the return type of `predict` and of `params`
    (g3py/libs/__init__.py:17-44)."""

    def __init__(self, data=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if data is not None:
            for k, v in data.items():
                self[k] = v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError("No such attribute: " + name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        if name in self:
            del self[name]
        else:
            raise AttributeError("No such attribute: " + name)

    def clone(self):
        return DictObj(data=self)

    def copy(self):
        return DictObj(data=self)


def clone(c):
    return copy(c)
