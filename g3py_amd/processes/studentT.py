"""StudentTProcess / WarpedStudentTProcess (g3py/processes/studentT.py:16-146) on the HIP path.

Same covariance, Cholesky factor and triangular solves as the Gaussian process -- one
`g3_gp_factor` gives log det and beta = |L^-1 delta|^2 -- plus the scalar Student-t pieces:
the log-density (:114-146), the posterior scaling (nu + beta - 2) / (nu + n - 2) of variances and
covariances (:36-49), t quantiles (:51-55) and inverse-gamma scaled draws (:57-67).
"""
import numpy as np
from scipy import stats
from scipy.special import gammaln, digamma

from .. import _lib
from .elliptical import EllipticalProcess, SENTINEL
from .gaussian import GaussianProcess
from .hypers import Freedom, HyperVar


class StudentTProcess(EllipticalProcess):
    def __init__(self, *args, **kwargs):
        if 'name' not in kwargs:
            kwargs['name'] = 'TP'
        if kwargs.get('degree') is None:
            kwargs['degree'] = Freedom()
        super().__init__(*args, **kwargs)

    # ---- log-density (WarpedStudentTDistribution.logp_cho, studentT.py:114-146)
    def th_loglike(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        t = self.dtype.type
        c = self._factor(values, inputs, outputs)
        if not np.all(np.isfinite(c['delta'])) or not np.all(np.isfinite(c['det_m'])):   # cond1, cond2
            return t(SENTINEL)
        st = self._solve(c, values, 'logp')
        if not np.isfinite(st['logdet']) or st['nonfinite'] > 0:                         # cond3, cond4
            return t(SENTINEL)
        nu = t(self.f_degree(values))                    # th_freedom(prior=True), studentT.py:31
        n, beta = t(c['N']), t(st['quad'])
        with np.errstate(all='ignore'):
            r1 = t(-0.5) * (nu + n) * np.log1p(beta / (nu - t(2)))                       # :124
            if t(np.float32(1e6)) <= nu:                                                 # :125
                r2 = -n * t(0.5) * np.log(t(2.0 * np.float32(np.pi)))
            else:
                r2 = t(gammaln((nu + n) * 0.5) - gammaln(nu * 0.5)) - t(0.5) * n * np.log((nu - t(2)) * t(np.float32(np.pi)))
            return t(r1 + r2 - t(st['logdet']) + c['det_m'])                             # :127-135

    # ---- gradient of logp: the Gaussian chain rule with d logp / d beta = -s/2, s = (nu + n) / (nu - 2 + beta),
    #      plus the degrees-of-freedom term (what Theano's reverse mode gives for studentT.py:114-135)
    th_dlogp = GaussianProcess.th_dlogp
    _dloglike = GaussianProcess._dloglike
    _chain_rule = GaussianProcess._chain_rule
    _flat_gradient = GaussianProcess._flat_gradient
    _potential_gradient = GaussianProcess._potential_gradient

    def _dlogp_scale(self, values, c, st, nat):
        nu, n, beta = float(self.f_degree(values)), float(c['N']), float(st['quad'])
        deg = self.f_degree.degree
        if isinstance(deg, HyperVar) and deg.name in nat:
            g = -0.5 * np.log1p(beta / (nu - 2.0)) + 0.5 * (nu + n) * beta / ((nu - 2.0) * (nu - 2.0 + beta))
            if not float(np.float32(1e6)) <= nu:
                g += 0.5 * digamma((nu + n) * 0.5) - 0.5 * digamma(nu * 0.5) - 0.5 * n / (nu - 2.0)
            nat[deg.name] = nat[deg.name] + g
        return (nu + n) / (nu - 2.0 + beta)

    # ---- chains of hyper-parameter vectors: the Gaussian block path with the Student-t density and scale per row
    logp_chain = GaussianProcess.logp_chain
    dlogp_chain = GaussianProcess.dlogp_chain
    _chain_workspace = GaussianProcess._chain_workspace
    _potential_gradient_rows = GaussianProcess._potential_gradient_rows
    _chain_rule_rows = GaussianProcess._chain_rule_rows
    _flat_gradient_rows = GaussianProcess._flat_gradient_rows

    def _chain_density(self, values_b, st, det_m, N, B):
        """th_loglike above for B rows: st = [logdet, quad, ...] per row"""
        t = self.dtype.type
        nu = self.f_degree.rows(values_b, B).astype(self.dtype)
        n, beta = t(N), st[:, 1].astype(self.dtype)
        r1 = t(-0.5) * (nu + n) * np.log1p(beta / (nu - t(2)))                           # :124
        gauss = -n * t(0.5) * np.log(t(2.0 * np.float32(np.pi)))                         # :125
        r2 = ((gammaln((nu + n) * 0.5) - gammaln(nu * 0.5)).astype(self.dtype)
              - t(0.5) * n * np.log((nu - t(2)) * t(np.float32(np.pi))))
        r2 = np.where(t(np.float32(1e6)) <= nu, gauss, r2)
        return (r1 + r2 - st[:, 0].astype(self.dtype) + det_m).astype(self.dtype)       # :127-135

    def _chain_dlogp_scale(self, values_b, st, N, nat, ok, B):
        """_dlogp_scale above for B rows"""
        nu, n, beta = self.f_degree.rows(values_b, B), float(N), st[:, 1].astype(np.float64)
        deg = self.f_degree.degree
        with np.errstate(all='ignore'):
            if isinstance(deg, HyperVar) and deg.name in nat:
                g = -0.5 * np.log1p(beta / (nu - 2.0)) + 0.5 * (nu + n) * beta / ((nu - 2.0) * (nu - 2.0 + beta))
                g = g + np.where(float(np.float32(1e6)) <= nu, 0.0,
                                 0.5 * digamma((nu + n) * 0.5) - 0.5 * digamma(nu * 0.5) - 0.5 * n / (nu - 2.0))
                nat[deg.name] = nat[deg.name] + np.where(ok, g, 0.0).reshape(nat[deg.name].shape)
            return (nu + n) / (nu - 2.0 + beta)

    # ---- posterior scaling (studentT.py:36-49)
    def th_scaling(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        t = self.dtype.type
        if prior:
            return t(1.0)
        values, _ = self._values(params)
        c = self._factor(values, inputs, outputs)
        st = self._solve(c, values, 'post')              # mapping_outputs = tt_to_num(mapping.inv(outputs))
        nu = self.f_degree(values)
        return t((nu + st['quad'] - 2.0) / (nu + c['N'] - 2.0))

    def th_variance(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        return (super().th_variance(space, inputs, outputs, vector, params, prior=prior, noise=noise)
                * self.th_scaling(space, inputs, outputs, vector, params, prior=prior, noise=noise))

    def th_covariance(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        return (super().th_covariance(space, inputs, outputs, vector, params, prior=prior, noise=noise)
                * self.th_scaling(space, inputs, outputs, vector, params, prior=prior, noise=noise))

    # ---- quantiles and draws (studentT.py:51-67)
    def quantiler(self, params=None, space=None, inputs=None, outputs=None, q=0.975, prior=False, noise=False,
                  simulations=None):
        p = stats.t.ppf(q, df=self.freedom(params, space, inputs, outputs, prior=prior, noise=noise))
        gp_quantiler = (self.location(params, space, inputs, outputs, prior=prior, noise=noise)
                        + p * self.kernel_sd(params, space, inputs, outputs, prior=prior, noise=noise))
        return self.mapping(params, space, inputs, outputs=gp_quantiler)

    def sampler(self, params=None, space=None, inputs=None, outputs=None, samples=1, prior=False, noise=False,
                rand=None):
        """location + cholesky . (randn * invgamma), mapped (studentT.py:57-67); `rand` (len(space) x
        samples, already scaled) may be supplied for reproducibility"""
        if space is None:
            space = self.space
        M = len(space)
        if rand is None:
            free = float(self.freedom(params, space, inputs, outputs, prior=prior, noise=noise))
            rand = np.random.randn(M, samples) * stats.invgamma.rvs(a=free / 2, scale=(free - 2) / 2, size=samples)
        rand = np.asarray(rand, dtype=self.dtype)
        S = rand.shape[1]
        loc = self.location(params, space, inputs, outputs, prior=prior, noise=noise)
        Ld, _, Mp = self._cholesky_dev(params, space, inputs, outputs, prior=prior, noise=noise)   # stays on the device
        dev = self.device
        Sp = _lib.roundup(S, 64)
        Zt = dev.upload(np.ascontiguousarray(rand.T), pad_rows=Sp, pad_cols=Mp)
        out = dev.alloc(Sp, Mp, self.dtype)
        dev.gemm_nt(out, Zt, Ld, Sp, Mp, Mp)                      # (L Z)^T in the MFMA GEMM
        g = loc[:, None] + dev.download(out, S, M).T
        return np.array([self.mapping(params, space, inputs, outputs=k.T) for k in g.T]).T

    _methods = EllipticalProcess._methods + (('dlogp', 'th_dlogp'),)


class WarpedStudentTProcess(StudentTProcess):
    def __init__(self, *args, **kwargs):
        if 'name' not in kwargs:
            kwargs['name'] = 'WTP'
        super().__init__(*args, **kwargs)

    def gauss_hermite(self, f, mu, sigma, n=10):
        """studentT.py:99-102"""
        t = self.dtype.type
        _a, _w = np.polynomial.hermite.hermgauss(n)
        a = _a.astype(self.dtype)[:, None]
        w = _w.astype(self.dtype)
        grille = mu + sigma * t(np.sqrt(2)) * a
        return np.dot(w, f(grille.flatten()).reshape(grille.shape)) / t(np.sqrt(np.pi))

    def th_mean(self, space, inputs, outputs, vector, params, prior=False, noise=False, n=10):
        loc, sd, values = self._loc_sd(space, inputs, outputs, params, prior, noise)
        return self.gauss_hermite(lambda v: self.f_mapping(v, values), loc, sd, n)       # studentT.py:79-85

    def th_variance(self, space, inputs, outputs, vector, params, prior=False, noise=False, n=10):
        loc, sd, values = self._loc_sd(space, inputs, outputs, params, prior, noise)
        m = self.gauss_hermite(lambda v: self.f_mapping(v, values), loc, sd, n)
        return self.gauss_hermite(lambda v: self.f_mapping(v, values) ** 2, loc, sd, n) - m ** 2   # :88-94

    # th_covariance is a stub in the reference (studentT.py:96-97): not bound
    _methods = tuple(m for m in StudentTProcess._methods if m[0] != 'covariance')
