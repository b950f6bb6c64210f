from .hypers import Hypers, HyperVar, Model
from .hypers.metrics import *
from .hypers.kernels import *
from .hypers.means import *
from .hypers.mappings import *
from .stochastic import StochasticProcess, GraphicalModel
from .elliptical import EllipticalProcess
from .gaussian import GaussianProcess, WarpedGaussianProcess

# aliases of g3py/processes/__init__.py:9-16
GP = GaussianProcess
WGP = WarpedGaussianProcess
