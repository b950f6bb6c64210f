"""Metrics of the stationary kernels (g3py/processes/hypers/metrics.py:7-13, 30-35, 59-61,
76-108).  They only carry hyper-parameters and defaults; the pairwise arithmetic itself runs
in the HIP Gram kernel (g3py_amd/csrc/g3_gram.hip), which never forms the n1 x n2 x d tensor.
"""
import numpy as np

from . import Hypers, Slot


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
    SLOTS = (Slot('rate', True, 'rate', per_column=True, owner_named=False),)

    def default_hypers(self, x=None, y=None):
        if x is None or np.ndim(x) < 1 or len(x) < 2:
            return {}
        spacing = np.abs(np.diff(np.asarray(x, dtype=float), axis=0)).mean(axis=0)
        return {self.rate: self.default_scale / spacing}


class ARD_L1(ARD):
    """dot(|x1 - x2|, rate) -- metrics.py:89-91"""
    kind = 'l1'
    default_scale = 1.0


class ARD_L2(ARD):
    """dot((x1 - x2)**2, 0.5 * rate**2) -- metrics.py:100-102"""
    kind = 'l2'
    default_scale = 0.5
