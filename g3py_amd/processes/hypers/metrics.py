"""Metrics of the stationary kernels (g3py/processes/hypers/metrics.py:7-13, 30-35, 59-61,
76-108).  They only carry hyper-parameters and defaults; the pairwise arithmetic itself runs
in the HIP Gram kernel (g3py_amd/csrc/g3_gram.hip), which never forms the n1 x n2 x d tensor.
"""
import numpy as np

from . import Hypers


class Metric(Hypers):
    kind = 'abs'


class Delta(Metric):
    """eq(x1 - x2, 0).sum(axis=2) -- metrics.py:30-35"""
    kind = 'delta'


class Difference(Metric):
    """raw difference tensor -- metrics.py:59-61"""
    kind = 'difference'


class ARD(Metric):
    """automatic-relevance-determination metrics: one positive rate per selected column.  The default rate
    is `default_scale` / (mean spacing of consecutive rows) per column (metrics.py:93-95, 104-108)."""
    default_scale = 1.0

    def __init__(self, x, name=None, rate=None):
        super().__init__(x, name)
        self.rate = rate

    def check_hypers(self, parent=''):
        super().check_hypers(parent=parent)
        if self.rate is None:
            self.rate = Hypers.FlatExp(parent + 'rate', shape=self.shape)
        self.hypers.append(self.rate)

    def default_hypers(self, x=None, y=None):
        try:
            spacing = np.abs(np.diff(x, axis=0)).mean(axis=0)
        except Exception:
            return {}
        return {self.rate: self.default_scale / spacing}


class ARD_L1(ARD):
    """dot(|x1 - x2|, rate) -- metrics.py:89-91"""
    kind = 'l1'
    default_scale = 1.0


class ARD_L2(ARD):
    """dot((x1 - x2)**2, 0.5 * rate**2) -- metrics.py:100-102"""
    kind = 'l2'
    default_scale = 0.5
