"""TransportProcess / TransportGaussianProcess (g3py/processes/transport.py:17-246) on the HIP path.

The process is y = T(x; inputs) with x standard normal and T a composition of transports
(`hypers/transports.py`).  With T = TMapping @ TLocation @ TKernel(noisy) this is the warped GP of
`gaussian.py` written as a push-forward, and its log-density

    logp = -n/2 log 2 pi - 1/2 |T^-1(y)|^2 + logdet dT^-1(y)          (transport.py:222-243)

is the same Gram + Cholesky + forward substitution: `TKernel.inv` / `logdet_dinv` are
`g3_gp_factor`, `TKernel.posterior` the tall factorisation + Schur complement.  Moments are Monte
Carlo over pushed-forward normal draws exactly as in the reference (30 simulations by default).
"""
import numpy as np

from .elliptical import SENTINEL
from .stochastic import StochasticProcess
from .hypers import HyperVar
from .hypers.transports import Transport, ID


class TransportProcess(StochasticProcess):
    def __init__(self, space=None, transport: Transport = None, *args, **kwargs):
        self.f_transport = ID() if transport is None else transport
        kwargs['space'] = space
        super().__init__(*args, **kwargs)

    def _check_hypers(self):
        self.f_transport.check_dims(self._inputs)
        self.f_transport.check_hypers(self.name + '_')
        self.f_transport.check_potential()

    def default_hypers(self):
        return self.f_transport.default_hypers_dims(self.inputs, self.outputs)

    # ---- the transports of transport.py:33-98
    def th_transport(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        v = np.asarray(vector, dtype=self.dtype)
        if prior:
            return self.f_transport(space, v, noise=noise, values=values)
        return self.f_transport.posterior(space, v, inputs, np.asarray(outputs, dtype=self.dtype), noise_pred=noise,
                                          noise_obs=True, values=values)

    def th_transport_diag(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        v = np.asarray(vector, dtype=self.dtype)
        if prior:
            return self.f_transport.diag(space, v, noise=noise, values=values)
        # the reference's diag=True posterior reaches the same code as the plain one (transports.py:27-33, 239)
        return self.th_transport(space, inputs, outputs, vector, params, prior=False, noise=noise)

    def th_transport_inv(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        v = np.asarray(vector, dtype=self.dtype)
        if prior:
            return self.f_transport.inv(space, v, noise=noise, values=values)
        return self.th_transport(space, inputs, outputs, vector, params, prior=False, noise=noise)

    _methods = (('transport', 'th_transport'), ('transport_diag', 'th_transport_diag'),
                ('transport_inv', 'th_transport_inv'))


class TransportGaussianDistribution:
    """Density of y = T(x), x ~ N(0, I) (g3py/processes/transport.py:214-246): the reference's
    pm.Continuous subclass reduced to its arithmetic.  `logp_t` takes numeric arrays and the
    natural-space hyper values; NaN / Inf anywhere gives the reference's -1e30 (transport.py:240-243)."""

    def __init__(self, transport=None, inputs=None, values=None, dtype=np.float64):
        self.transport = ID() if transport is None else transport
        self.th_inputs = inputs
        self.values = values
        self.dtype = np.dtype(dtype)

    @classmethod
    def logp_t(cls, value, transport, inputs, values=None, dtype=np.float64):
        t = np.dtype(dtype).type
        y = np.asarray(value, dtype=dtype)
        with np.errstate(all='ignore'):
            delta = np.asarray(transport.inv(inputs, y, noise=True, values=values))      # transport.py:226
            det_m = transport.logdet_dinv(inputs, y, values=values)                      # transport.py:227
        if not np.all(np.isfinite(delta)) or not np.all(np.isfinite(det_m)):
            return t(SENTINEL)
        npi = t(-0.5) * t(len(y)) * np.log(t(2.0 * np.pi))                               # transport.py:230
        return t(npi + t(-0.5) * t(delta.dot(delta)) + t(det_m))                         # transport.py:231-238

    def logp(self, value):
        return self.logp_t(value, self.transport, self.th_inputs, self.values, self.dtype)


class TransportGaussianProcess(TransportProcess):
    def __init__(self, *args, **kwargs):
        if 'name' not in kwargs:
            kwargs['name'] = 'TGP'
        super().__init__(*args, **kwargs)

    # ---- log-density (TransportGaussianDistribution.logp_t, transport.py:222-243)
    def th_loglike(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        return TransportGaussianDistribution.logp_t(outputs, self.f_transport, inputs, values, self.dtype)

    def th_logp(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        _, logjac = self._values(params)
        t = self.dtype.type
        if prior:
            return t(logjac)
        return t(logjac + self.th_loglike(space, inputs, outputs, vector, params))

    # ---- Monte-Carlo moments (transport.py:169-209)
    def _simulations(self, simulations, params, space, inputs, outputs, prior, noise):
        if simulations is None:
            simulations = 30
        if type(simulations) is int:
            return self.sampler(params=params, space=space, inputs=inputs, outputs=outputs, samples=simulations,
                                prior=prior, noise=noise)
        return simulations

    def mean(self, params=None, space=None, inputs=None, outputs=None, prior=False, noise=False, simulations=None):
        return self._simulations(simulations, params, space, inputs, outputs, prior, noise).mean(axis=1)

    def std(self, params=None, space=None, inputs=None, outputs=None, prior=False, noise=False, simulations=None):
        return self._simulations(simulations, params, space, inputs, outputs, prior, noise).std(axis=1)

    def quantiler(self, params=None, space=None, inputs=None, outputs=None, q=0.975, prior=False, noise=False,
                  simulations=None):
        return np.nanpercentile(self._simulations(simulations, params, space, inputs, outputs, prior, noise), 100 * q, axis=1)

    def sampler(self, params=None, space=None, inputs=None, outputs=None, samples=1, prior=False, noise=False, rand=None):
        """push `samples` standard-normal vectors through the (posterior) transport; `rand`
        (len(space) x samples) may be supplied for reproducibility"""
        if space is None:
            space = self.space
        if rand is None:
            rand = np.random.randn(len(space), samples)
        rand = np.asarray(rand, dtype=self.dtype)
        return np.array([self.transport(params, space, inputs, outputs, vector=rand[:, i], prior=prior, noise=noise).T
                         for i in range(rand.shape[1])]).T

    _methods = TransportProcess._methods + (('logp', 'th_logp'), ('loglike', 'th_loglike'))
