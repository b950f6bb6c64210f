"""g3py_amd -- MI355X-native GP inference hot path behind g3py's process / kernel API.

    import g3py_amd as g3
    gp = g3.GaussianProcess(space=x, location=g3.Bias(), kernel=g3.SE(x))
    gp.observed(x_obs, y_obs); gp.logp(params); gp.predict(params)

Python host code -> ctypes -> libg3hip.so (hand-written HIP for gfx950).  No CPU fallback.
"""
from .libs import DictObj, clone
from .libs.tensors import makefn, cholesky_robust, tt_to_num, tt_to_cov, tt_to_bounded, solve_lower_triangular
from .processes import *
from .bayesian import Experiment, random_obs, uniform_obs
from .device import Device, DeviceArray, compile_spec
from ._lib import G3Error

__version__ = '0.1.0'
