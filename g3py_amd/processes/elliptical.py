"""EllipticalProcess: prior / posterior location, covariance, Cholesky, diagonal and sd of a
GP (g3py/processes/elliptical.py:18-217) computed by libg3hip.

The reference recompiles and recomputes Gram + an O(N^3) LU solve for EVERY requested
statistic.  Here one Cholesky factor of K(X,X)+noise is built per
(params, inputs, outputs) triple by `g3_gp_factor` and kept on the device; every posterior
statistic is a cross-Gram + multi-right-hand-side triangular solve against it
(`g3_gp_cross`).  Equal to the reference's `solve(K, .)` in exact arithmetic.
"""
import numpy as np

from .. import _lib
from ..device import Device, compile_spec
from .hypers import HyperVar
from .hypers.kernels import Kernel, KernelSum, KernelNoise
from .hypers.means import Mean
from .hypers.mappings import Mapping, Identity
from .stochastic import StochasticProcess

SENTINEL = np.float32(-1e30)     # gaussian.py:238-241


class EllipticalProcess(StochasticProcess):
    def __init__(self, space=None, location: Mean = None, kernel: Kernel = None, mapping: Mapping = None,
                 degree=None, noisy=True, var_noise=None, *args, **kwargs):
        self.f_location = location
        self.f_degree = degree
        self.f_mapping = Identity() if mapping is None else mapping
        self.f_kernel = kernel
        if noisy:   # elliptical.py:26-31
            self.f_kernel_noise = KernelSum(self.f_kernel, KernelNoise(name='Noise', var=var_noise))
        else:
            self.f_kernel_noise = self.f_kernel
        self._cache = None
        self._workspace = None
        self._dist = None
        kwargs['space'] = space
        super().__init__(*args, **kwargs)

    @property
    def device(self):
        if self._device is None:
            self._device = Device.default()
        return self._device

    # ---------------------------------------------------------------- several GPUs
    def distribute(self, dist, rank, world, nb=None, transport='rccl'):
        """Evaluate this process on `world` GPUs, one process per GPU (SPMD: every rank runs the same program on the same
        observations and calls the same methods in the same order).  The reference has no counterpart (its only
        parallelism is a process pool over chains, stochastic.py:773-783).  From here on logp / loglike and every
        statistic that comes from the posterior location and variance at `space` -- mean, median, variance, std,
        quantiles, logpredictive -- are computed by the multi-GPU driver inside libg3hip (g3_dist_*: row-block-cyclic
        covariance, library-owned RCCL communicators); each rank holds 1 / world of the covariance and gets the same
        numbers back; `sampler` draws from the f posterior through the driver too (rank 0's normals).  `dlogp` (and so
        `find_MAP`) works too: from the first gradient on, the factorisation carries the identity as right-hand-side rows
        and K^-1 is formed row block by row block where the rows live (g3_dist_gp_dlogp).  The M x M posterior covariance
        (`covariance`, `cholesky`, `predict(cov=True)`) is formed in row chunks and gathered on every rank
        (g3_dist_posterior_cov); `cross_mean` passes its cross kernel to the driver.  `dist`: an initialised torch.distributed (any backend; it
        only carries 256 bytes of communicator ids, or everything with transport='callbacks', the one-GPU rehearsal)."""
        self._dist = dict(dist=dist, rank=int(rank), world=int(world), nb=nb, transport=transport, dgp=None, shape=None,
                          grad=False)
        self._cache = None
        self._workspace = None
        return self

    def undistribute(self):
        """back to one GPU: destroys the multi-GPU driver of this process (collective: call it on every rank)"""
        if self._dist is not None and self._dist['dgp'] is not None:
            self._dist['dgp'].close()
        self._dist = None
        self._cache = None
        self._workspace = None
        return self

    def _dist_step(self, c, values, dl, space, cross_kernel=None):
        """one evaluation on all ranks: (stats, cross-solve results or None); cross_kernel: the kernel of the cross
        covariance K(space, X) when it is not the process's own (th_cross_mean, gaussian.py:99-112)"""
        from ..distributed import NativeDistributedGP
        ds, dev = self._dist, self.device
        N, d = c['N'], c['d']
        S = None if space is None else self._x(space)
        M = 0 if S is None else S.shape[0]
        shape = (N, d, M, self.dtype.str)
        if ds['dgp'] is None or ds['shape'] != shape:
            if ds['dgp'] is not None:
                ds['dgp'].close()
            nb = ds['nb'] or (1024 if ds['world'] <= 4 and N >= 16384 else 512 if N >= 4096 else 128)
            ds['dgp'] = NativeDistributedGP(dev, ds['dist'], ds['rank'], ds['world'], N, d, M, nb=nb, dtype=self.dtype,
                                            transport=ds['transport'])
            ds['shape'] = shape
            if ds.get('grad'):
                ds['dgp'].set_grad(True)
        dgp = ds['dgp']
        dvec = self._workspace['dvec']
        dev.copy_in(dvec, np.where(np.isfinite(dl), dl, 0).astype(self.dtype))
        Sd = dev.upload(S) if M > 0 else c['Xd']
        kc = cross_kernel if cross_kernel is not None else self.f_kernel
        dgp.step(self.f_kernel_noise.spec(values, d), kc.spec(values, d), c['Xd'], Sd, dvec)
        last = dgp.last
        st = dict(logdet=last['logdet'], quad=last['quad'], nonfinite=0 if np.isfinite(last['quad']) else 1,
                  tries=last['tries'], fallback=last['fallback'], info=last['info'])
        cross = None
        if M > 0:
            Mp = _lib.roundup(M, _lib.G3_RHS_PAD)
            cross = dict(kid=(False, id(cross_kernel) if cross_kernel is not None else None), S=S.copy(),
                         out=(None, np.asarray(last['mean'], dtype=self.dtype),
                                                               np.asarray(last['ss'], dtype=self.dtype), M, Mp))
        return st, cross

    def _check_hypers(self):
        """elliptical.py:35-52"""
        x = self._inputs
        self.f_location.check_dims(x)
        self.f_kernel_noise.check_dims(x)
        self.f_mapping.check_dims(x)
        self.f_location.check_hypers(self.name + '_')
        self.f_kernel_noise.check_hypers(self.name + '_')
        self.f_mapping.check_hypers(self.name + '_')
        self.f_location.check_potential()
        self.f_kernel_noise.check_potential()
        self.f_mapping.check_potential()
        if self.f_degree is not None:                     # elliptical.py:49-52
            self.f_degree.check_dims(None)
            self.f_degree.check_hypers(self.name + '_')
            self.f_degree.check_potential()

    def default_hypers(self):
        x, y = self.inputs, self.outputs
        r = {**self.f_location.default_hypers_dims(x, y), **self.f_kernel_noise.default_hypers_dims(x, y),
             **self.f_mapping.default_hypers_dims(x, y)}
        if self.f_degree is not None:
            r.update(self.f_degree.default_hypers_dims(x, y))
        return r

    # ---------------------------------------------------------------- helpers
    def _x(self, a):
        a = np.asarray(a, dtype=self.dtype)
        return a.reshape(len(a), 1) if a.ndim < 2 else np.ascontiguousarray(a)

    def _prog(self, kernel, values, d):
        return compile_spec(kernel.spec(values, d), d)

    def _factor(self, values, inputs, outputs):
        """L = cholesky_robust(tt_to_cov(K_noise(X, X))), a = L^-1 (T^-1(y) - m(X)) on the device
        (elliptical.py:63,68,71; gaussian.py:208-212, 251-260), cached across statistics."""
        X = self._x(inputs)
        y = np.asarray(outputs, dtype=self.dtype).reshape(-1)
        key = (tuple((k, np.asarray(v).tobytes()) for k, v in sorted(values.items())), X.shape)
        c = self._cache
        if c is not None and c['key'] == key and np.array_equal(c['X'], X) and np.array_equal(c['y'], y):
            return c
        dev = self.device
        N, d = X.shape
        with np.errstate(all='ignore'):
            mapped = np.asarray(self.f_mapping.inv(y, values), dtype=self.dtype)      # elliptical.py:63
            mu = self.f_location(X, values)
            delta = mapped - mu                                                       # gaussian.py:208
            det_m = self.f_mapping.logdet_dinv(y, values)                             # gaussian.py:225
        # tt_to_num(mapping.inv(outputs)) is what the posterior uses (elliptical.py:63)
        mapped_num = np.where(np.isnan(mapped), 0, np.where(np.isinf(mapped), self.dtype.type(np.float32(1e10)), mapped))
        Np = _lib.roundup(N)
        # device workspace: re-used while the observations stay the same (optimisers and samplers call
        # logp / dlogp thousands of times on one data set -- hipMalloc and the upload of X would
        # otherwise dominate small problems)
        ws = self._workspace
        if ws is None or ws['shape'] != X.shape or not np.array_equal(ws['X'], X):
            if self._dist is not None:      # several GPUs: the covariance lives in the driver, 1 / world of it per rank
                ws = dict(shape=X.shape, X=X.copy(), Xd=dev.upload(X), Kd=None, ad=None, Wd=None, dvec=dev.alloc(1, N, self.dtype))
            else:
                ws = dict(shape=X.shape, X=X.copy(), Xd=dev.upload(X),
                          Kd=dev.alloc(Np + _lib.G3_RHS_PAD, Np, self.dtype),   # + the right-hand-side block that carries delta
                          ad=dev.alloc(1, Np, self.dtype), Wd=dev.alloc_inverses(Np, self.dtype),
                          dvec=dev.alloc(1, N, self.dtype))
            self._workspace = ws
        Xd, Kd, ad, Wd = ws['Xd'], ws['Kd'], ws['ad'], ws['Wd']
        c = dict(key=key, X=X.copy(), y=y.copy(), N=N, d=d, Np=Np, Xd=Xd, Kd=Kd, Wd=Wd, ad=ad, mu=mu, det_m=det_m,
                 delta=delta, delta_post=mapped_num - mu, stats=None, which=None)
        c['same_delta'] = bool(np.array_equal(c['delta'], c['delta_post']))
        self._cache = c
        return c

    def _solve(self, c, values, which):
        """run g3_gp_factor with delta (logp) or the scrubbed delta (posterior)"""
        if c['which'] == which or (c['which'] is not None and c['same_delta']):
            return c['stats']
        dev = self.device
        dl = c['delta'] if which == 'logp' else c['delta_post']
        finite = np.all(np.isfinite(dl))
        if self._dist is not None:
            # one multi-GPU evaluation gives the factor's scalars AND the cross solve at the process's own space
            sp = getattr(self, 'space', None)         # (a process without a space holds the reference's 2-point dummy)
            if sp is not None and self._x(sp).shape[1] != c['d']:
                sp = None
            st, cross = self._dist_step(c, values, dl, sp)
            st['delta_finite'] = bool(finite)
            c['stats'], c['which'], c['grad'], c['cross'] = st, which, None, cross
            return st
        dvec = self._workspace['dvec']
        dev.copy_in(dvec, np.where(np.isfinite(dl), dl, 0).astype(self.dtype))
        prog = self._prog(self.f_kernel_noise, values, c['d'])
        st = dev.gp_factor(prog, c['Xd'], c['N'], c['d'], dvec, c['Kd'], c['Wd'], c['ad'])
        st['delta_finite'] = bool(finite)
        c['stats'], c['which'] = st, which
        c['grad'] = None            # K^-1 / alpha pieces of th_dlogp belong to this factorisation
        c['cross'] = None           # so does the last cross solve
        return st

    def _cross(self, c, values, space, noise, kernel=None):
        """V = K(space, X) L^-T, mu = V a, ss = |V_i|^2 (elliptical.py:78-91)"""
        dev = self.device
        S = self._x(space)
        M = S.shape[0]
        Mp = _lib.roundup(M, _lib.G3_RHS_PAD)
        # predict() asks for mean, variance, quantiles ... one after the other (the reference compiles and runs one
        # function per statistic, each with its own solves, elliptical.py:81-91): the N^2 M cross solve is shared
        # between them while factor, right-hand side, space and kernel stay the same
        kid = (bool(noise), id(kernel) if kernel is not None else None)
        cc = c.get('cross')
        if self._dist is not None:
            # the Noise term contributes nothing to a cross block (kernels.py:367-371): one cross solve serves both flags
            kd = id(kernel) if kernel is not None else None
            if cc is None or cc['kid'][1] != kd or cc['S'].shape != S.shape or not np.array_equal(cc['S'], S):
                dl = c['delta'] if c['which'] == 'logp' else c['delta_post']
                st, cc = self._dist_step(c, values, dl, S, cross_kernel=kernel)   # another space / cross kernel: one more evaluation
                cc['kernel_ref'] = kernel          # keeps the object alive: its id() cannot be handed to another kernel
                c['stats'], c['cross'] = dict(st, delta_finite=c['stats']['delta_finite']), cc
            return cc['out']
        if cc is not None and cc['kid'] == kid and cc['S'].shape == S.shape and np.array_equal(cc['S'], S):
            return cc['out']
        Sd = dev.upload(S)
        V = dev.alloc(Mp, c['Np'], self.dtype)
        mu = dev.alloc(1, Mp, self.dtype)
        ss = dev.alloc(1, Mp, self.dtype)
        kern = kernel if kernel is not None else (self.f_kernel_noise if noise else self.f_kernel)
        dev.gp_cross(self._prog(kern, values, c['d']), Sd, M, c['Xd'], c['N'], c['d'], c['Kd'], c['Wd'], c['ad'], V, mu, ss)
        out = (V, dev.download(mu, 1, M)[0], dev.download(ss, 1, M)[0], M, Mp)
        c['cross'] = dict(kid=kid, S=S.copy(), out=out, kernel_ref=kernel)   # (the reference pins id(kernel))
        return out

    def _prior_gram(self, values, space, noise, pad=False):
        """prior_kernel_space = tt_to_cov(K_noise(space)) / prior_kernel_f_space = K_f(space)
        (elliptical.py:70,74) as a device matrix"""
        dev = self.device
        S = self._x(space)
        M, d = S.shape
        Mp = _lib.roundup(M, _lib.G3_RHS_PAD) if pad else M
        Sd = dev.upload(S)
        K = dev.alloc(Mp, Mp, self.dtype)
        kern = self.f_kernel_noise if noise else self.f_kernel
        dev.gram(self._prog(kern, values, d), Sd, None, d, K, Mp, Mp, _lib.G3_GRAM_SCRUB if noise else 0)
        if noise:
            dev.cov_lift(K, M)
        return K, M, Mp

    def _prior_diag(self, values, space, noise):
        dev = self.device
        S = self._x(space)
        M, d = S.shape
        Sd = dev.upload(S)
        out = dev.alloc(1, M, self.dtype)
        kern = self.f_kernel_noise if noise else self.f_kernel
        dev.gram_diag(self._prog(kern, values, d), Sd, d, out)
        dg = dev.download(out, 1, M)[0]
        if noise:   # tt_to_cov acts on the whole matrix; its effect on the diagonal:
            dg = np.where(np.isnan(dg), 0, np.where(np.isinf(dg), self.dtype.type(np.float32(1e10)), dg))
            m = dg.min()
            if not m > 0:
                dg = dg + (self.dtype.type(np.float32(1e-6)) - m)
        return dg

    # ---------------------------------------------------------------- statistics (th_* of the reference)
    def th_define_process(self):
        pass   # the symbolic graph of elliptical.py:60-107 is replaced by the methods below

    def th_freedom(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        """prior degrees of freedom, plus the number of observations a posteriori (elliptical.py:109-113)"""
        if self.f_degree is None:
            return None
        values, _ = self._values(params)
        nu = self.f_degree(values)
        return self.dtype.type(nu if prior else nu + len(np.asarray(inputs)))

    def th_logp(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        """sum of the free variables' log-densities (Flat: 0; FlatExp: the log-transform Jacobian
        term; potentials) plus the observed term unless prior (stochastic.py:300-306)"""
        _, logjac = self._values(params)
        t = self.dtype.type
        if prior:
            return t(logjac)
        return t(logjac + self.th_loglike(space, inputs, outputs, vector, params))

    def th_mapping_inv(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        with np.errstate(all='ignore'):
            r = np.asarray(self.f_mapping.inv(np.asarray(outputs, dtype=self.dtype), values))
        return np.where(np.isnan(r), 0, np.where(np.isinf(r), self.dtype.type(np.float32(1e10)), r))

    def th_mapping(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        with np.errstate(all='ignore'):
            r = np.asarray(self.f_mapping(np.asarray(outputs, dtype=self.dtype), values))
        return np.where(np.isnan(r), 0, np.where(np.isinf(r), self.dtype.type(np.float32(1e10)), r))

    def th_location(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        S = self._x(space)
        loc = self.f_location(S, values)
        if prior:
            return loc                                                    # elliptical.py:122-123
        c = self._factor(values, inputs, outputs)
        self._solve(c, values, 'post')
        _, mu, _, _, _ = self._cross(c, values, space, noise)
        return loc + mu                                                   # elliptical.py:81-84

    def th_cross_mean(self, space, inputs, outputs, vector, params, prior=False, noise=False, cross_kernel=None):
        """location of one process given another through a cross kernel (gaussian.py:99-112)"""
        values, _ = self._values(params)
        loc = self.f_location(self._x(space), values)
        if prior:
            return loc
        c = self._factor(values, inputs, outputs)
        self._solve(c, values, 'post')
        _, mu, _, _, _ = self._cross(c, values, space, False, kernel=cross_kernel)
        return loc + mu

    def _kernel_dev(self, space, inputs, outputs, params, prior, noise):
        """prior / posterior covariance of `space` as a DEVICE matrix: (K, M, rows of K)"""
        values, _ = self._values(params)
        dev = self.device
        if prior:
            K, M, Mp = self._prior_gram(values, space, noise)
            return K, M, Mp
        c = self._factor(values, inputs, outputs)
        self._solve(c, values, c['which'] or 'post')
        if self._dist is not None:
            # several GPUs: V = K(space, X) L^-T lives in the driver's right-hand-side rows; every rank forms the rows of
            # its chunks, the M x M matrix is gathered on all of them (g3_dist_posterior_cov)
            _, _, _, M, Mp = self._cross(c, values, space, noise)        # the driver's last evaluation is at this space
            K = dev.alloc(Mp, Mp, self.dtype)
            kern = self.f_kernel_noise if noise else self.f_kernel
            self._dist['dgp'].posterior_cov(kern.spec(values, c['d']), dev.upload(self._x(space)), K)
            return K, M, Mp
        V, _, _, M, Mp = self._cross(c, values, space, noise)
        K, _, _ = self._prior_gram(values, space, noise, pad=True)
        dev.gemm_nt(K, V, V, Mp, Mp, c['Np'], alpha=-1.0, beta=1.0)        # elliptical.py:86-91
        return K, M, Mp

    def th_kernel(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        K, M, _ = self._kernel_dev(space, inputs, outputs, params, prior, noise)
        return self.device.download(K, M, M)

    def th_cholesky_dev(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        """cholesky_robust of the prior / posterior covariance, kept on the device: (L zero-padded to
        a multiple of 128, M, padded size) -- what the samplers multiply the normal draws with"""
        dev = self.device
        K, M, _ = self._kernel_dev(space, inputs, outputs, params, prior, noise)
        Mp = _lib.roundup(M, _lib.G3_RHS_PAD)
        Ld = dev.alloc(Mp, Mp, self.dtype, zero=True)
        dev.potrf_robust(K, Ld, M)                                        # elliptical.py:72,76,88,92
        return Ld, M, Mp

    def th_cholesky(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        Ld, M, _ = self.th_cholesky_dev(space, inputs, outputs, vector, params, prior=prior, noise=noise)
        return self.device.download(Ld, M, M)

    def th_kernel_diag(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        dg = self._prior_diag(values, space, noise)
        if not prior:
            c = self._factor(values, inputs, outputs)
            self._solve(c, values, c['which'] or 'post')
            _, _, ss, _, _ = self._cross(c, values, space, noise)
            dg = dg - ss
        return np.where(dg < 0, self.dtype.type(0), dg)                   # tt_to_bounded(., 0) elliptical.py:94-97

    def th_kernel_sd(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        return np.sqrt(self.th_kernel_diag(space, inputs, outputs, vector, params, prior=prior, noise=noise))

    def th_cholesky_diag(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        return np.diag(self.th_kernel_sd(space, inputs, outputs, vector, params, prior=prior, noise=noise))

    def _loc_sd(self, space, inputs, outputs, params, prior, noise, sd_noise=None):
        """location and sd from ONE cross solve (the reference runs two compiled functions)"""
        values, _ = self._values(params)
        S = self._x(space)
        loc = self.f_location(S, values)
        sd_noise = noise if sd_noise is None else sd_noise
        dg = self._prior_diag(values, space, sd_noise)
        if not prior:
            c = self._factor(values, inputs, outputs)
            self._solve(c, values, 'post')
            _, mu, ss, _, _ = self._cross(c, values, space, noise)
            loc = loc + mu
            dg = dg - ss
        dg = np.where(dg < 0, self.dtype.type(0), dg)
        return loc, np.sqrt(dg), values

    def th_median(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        return self.f_mapping(self.th_location(space, inputs, outputs, vector, params, prior, noise), values)

    def th_mean(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        values, _ = self._values(params)
        return self.f_mapping(self.th_location(space, inputs, outputs, vector, params, prior, noise), values)

    def th_variance(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        return self.th_kernel_diag(space, inputs, outputs, vector, params, prior=prior, noise=noise)

    def th_std(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        return np.sqrt(self.th_variance(space, inputs, outputs, vector, params, prior=prior, noise=noise))

    def th_covariance(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        return self.th_kernel(space, inputs, outputs, vector, params, prior=prior, noise=noise)

    # ---- errors of the predictive mean against a supplied vector (stochastic.py:315-326)
    def th_error_l1(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        mean = self.th_mean(space, inputs, outputs, vector, params, prior=prior, noise=noise)
        return np.mean(np.abs(np.asarray(vector, dtype=self.dtype) - mean))

    def th_error_l2(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        mean = self.th_mean(space, inputs, outputs, vector, params, prior=prior, noise=noise)
        return np.mean((np.asarray(vector, dtype=self.dtype) - mean) ** 2)

    def th_error_mse(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        d = np.abs(np.asarray(vector, dtype=self.dtype) - np.asarray(outputs, dtype=self.dtype))
        return np.mean(d) ** 2 + np.var(d)

    _methods = (('mean', 'th_mean'), ('median', 'th_median'), ('variance', 'th_variance'), ('std', 'th_std'),
                ('covariance', 'th_covariance'), ('logpredictive', 'th_logpredictive'), ('logp', 'th_logp'),
                ('loglike', 'th_loglike'), ('mapping', 'th_mapping'), ('mapping_inv', 'th_mapping_inv'),
                ('location', 'th_location'), ('kernel', 'th_kernel'), ('cholesky', 'th_cholesky'),
                ('kernel_diag', 'th_kernel_diag'), ('kernel_sd', 'th_kernel_sd'),
                ('cholesky_diag', 'th_cholesky_diag'), ('cross_mean', 'th_cross_mean'), ('freedom', 'th_freedom'),
                ('error_l1', 'th_error_l1'), ('error_l2', 'th_error_l2'), ('error_mse', 'th_error_mse'),
                ('_cholesky_dev', 'th_cholesky_dev'))
