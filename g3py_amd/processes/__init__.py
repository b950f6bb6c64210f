from .hypers import Hypers, HyperVar, Model, Freedom
from .hypers.metrics import *
from .hypers.kernels import *
from .hypers.means import *
from .hypers.mappings import *
from .hypers.transports import (Transport, TransportComposed, ID, TElemwise, TLinear, TNoLinear, TLocation, TScale,
                                TMapping, TKernel, TTriangular)
from .stochastic import StochasticProcess, GraphicalModel
from .elliptical import EllipticalProcess
from .gaussian import GaussianProcess, WarpedGaussianProcess
from .studentT import StudentTProcess, WarpedStudentTProcess
from .transport import TransportProcess, TransportGaussianProcess, TransportGaussianDistribution

# aliases of g3py/processes/__init__.py:9-16
GP = GaussianProcess
WGP = WarpedGaussianProcess
TP = StudentTProcess
WTP = WarpedStudentTProcess
TGP = TransportGaussianProcess
