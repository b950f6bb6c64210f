"""GaussianProcess / WarpedGaussianProcess (g3py/processes/gaussian.py:18-260) on the HIP path."""
import numpy as np
from scipy import stats

from .. import _lib
from .elliptical import EllipticalProcess, SENTINEL


def logp_cho_diag(value, mu, sd, mapping, values, dtype):
    """WarpedGaussianDistribution.logp_cho with a DIAGONAL factor (gaussian.py:42-54,192-241):
    independent marginals, O(M) host arithmetic on device-produced vectors."""
    t = np.dtype(dtype).type
    value = np.asarray(value, dtype=dtype)
    with np.errstate(all='ignore'):
        delta = np.asarray(mapping.inv(value, values)) - mu
        if np.any(~np.isfinite(delta)):
            return t(SENTINEL)
        det_m = mapping.logdet_dinv(value, values)
        if np.any(~np.isfinite(det_m)) or np.any(~np.isfinite(sd)):
            return t(SENTINEL)
        lcho = delta / sd
        r = (t(-0.5) * t(len(sd)) * np.log(t(2.0 * np.pi)) + t(-0.5) * lcho.dot(lcho)
             - np.sum(np.log(sd)) + det_m)
    if np.any(~np.isfinite(lcho)):
        return t(SENTINEL)
    return r


class GaussianProcess(EllipticalProcess):
    def __init__(self, *args, **kwargs):
        if 'name' not in kwargs:
            kwargs['name'] = 'GP'
        super().__init__(*args, **kwargs)

    # ---- log marginal likelihood (gaussian.py:192-249; stochastic.py:300-313)
    def th_loglike(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        t = self.dtype.type
        c = self._factor(values, inputs, outputs)
        if not np.all(np.isfinite(c['delta'])):                     # cond1, gaussian.py:234
            return t(SENTINEL)
        if not np.all(np.isfinite(c['det_m'])):                     # cond2, :235
            return t(SENTINEL)
        st = self._solve(c, values, 'logp')
        # cond3 (:236): a factor from a scrubbed covariance (or the 1e-10*I fallback) is finite
        if not np.isfinite(st['logdet']) or st['nonfinite'] > 0:    # cond4, :237
            return t(SENTINEL)
        n = c['N']
        npi = t(-0.5) * t(n) * np.log(t(2.0 * np.pi))               # :218
        return t(npi + t(-0.5) * t(st['quad']) - t(st['logdet']) + c['det_m'])   # :219-232

    def th_logp(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        """sum of the free variables' log-densities (Flat: 0; FlatExp: the log-transform
        Jacobian term) plus the observed term unless prior (stochastic.py:300-306)"""
        _, logjac = self._values(params)
        t = self.dtype.type
        if prior:
            return t(logjac)
        return t(logjac + self.th_loglike(space, inputs, outputs, vector, params))

    def th_logpredictive(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        loc, sd, values = self._loc_sd(space, inputs, outputs, params, prior, noise, sd_noise=True)
        return logp_cho_diag(vector, loc, sd, self.f_mapping, values, self.dtype)

    # ---- quantiles and draws (gaussian.py:56-97)
    def quantiler(self, params=None, space=None, inputs=None, outputs=None, q=0.975, prior=False, noise=False,
                  simulations=None):
        p = stats.norm.ppf(q)
        gp_quantiler = (self.location(params, space, inputs, outputs, prior=prior, noise=noise)
                        + p * self.kernel_sd(params, space, inputs, outputs, prior=prior, noise=noise))
        return self.mapping(params, space, inputs, outputs=gp_quantiler)

    def sampler(self, params=None, space=None, inputs=None, outputs=None, samples=1, prior=False, noise=False,
                rand=None):
        """location + cholesky . randn, mapped (gaussian.py:89-97).  The M x M by M x S product
        runs in the MFMA GEMM.  `rand` (len(space) x samples) may be supplied for reproducibility;
        by default it is drawn with np.random.randn exactly as the reference does."""
        if space is None:
            space = self.space
        M = len(space)
        if rand is None:
            rand = np.random.randn(M, samples)
        rand = np.asarray(rand, dtype=self.dtype)
        S = rand.shape[1]
        loc = self.location(params, space, inputs, outputs, prior=prior, noise=noise)
        L = self.cholesky(params, space, inputs, outputs, prior=prior, noise=noise)
        dev = self.device
        Mp, Sp = _lib.roundup(M, _lib.G3_RHS_PAD), _lib.roundup(S, 64)
        Ld = dev.upload(L, pad_rows=Mp, pad_cols=Mp)
        Zt = dev.upload(np.ascontiguousarray(rand.T), pad_rows=Sp, pad_cols=Mp)
        out = dev.alloc(Sp, Mp, self.dtype)
        dev.gemm_nt(out, Zt, Ld, Sp, Mp, Mp)                      # (L Z)^T
        g = loc[:, None] + dev.download(out, S, M).T
        return np.array([self.mapping(params, space, inputs, outputs=k.T) for k in g.T]).T


class WarpedGaussianProcess(GaussianProcess):
    def __init__(self, *args, **kwargs):
        if 'name' not in kwargs:
            kwargs['name'] = 'WGP'
        super().__init__(*args, **kwargs)

    def gauss_hermite(self, f, mu, sigma, n=10):
        """gaussian.py:162-174"""
        t = self.dtype.type
        _a, _w = np.polynomial.hermite.hermgauss(n)
        a = _a.astype(self.dtype)[:, None]
        w = _w.astype(self.dtype)
        grille = mu + sigma * t(np.sqrt(2)) * a
        return np.dot(w, f(grille.flatten()).reshape(grille.shape)) / t(np.sqrt(np.pi))

    def th_mean(self, space, inputs, outputs, vector, params, prior=False, noise=False, n=10):
        loc, sd, values = self._loc_sd(space, inputs, outputs, params, prior, noise)
        return self.gauss_hermite(lambda v: self.f_mapping(v, values), loc, sd, n)       # gaussian.py:127-141

    def th_variance(self, space, inputs, outputs, vector, params, prior=False, noise=False, n=10):
        loc, sd, values = self._loc_sd(space, inputs, outputs, params, prior, noise)
        m = self.gauss_hermite(lambda v: self.f_mapping(v, values), loc, sd, n)
        return self.gauss_hermite(lambda v: self.f_mapping(v, values) ** 2, loc, sd, n) - m ** 2   # :143-157

    # th_covariance is undefined for the warped process (gaussian.py:159-160): not bound
    _methods = tuple(m for m in GaussianProcess._methods if m[0] != 'covariance')
