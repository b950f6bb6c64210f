"""GaussianProcess / WarpedGaussianProcess (g3py/processes/gaussian.py:18-260) on the HIP path."""
import numpy as np
from scipy import stats

from .. import _lib
from ..device import compile_spec_rows
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


class _Ref:
    """placeholder a kernel's spec() carries where a free hyper-parameter's value would go"""

    def __init__(self, name):
        self.name = name


class _Refs(dict):
    def __getitem__(self, name):
        return _Ref(name)


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

    # ---- gradient of logp (stochastic.py:308-309; tensors.py:11-22, 224-260)
    def th_dlogp(self, space, inputs, outputs, vector, params, prior=False, noise=False):
        """tt_to_num(gradient(th_logp, free variables)): flat vector over the model's variables in
        creation order, in TRANSFORMED space (d/d log h = h * d/dh for the FlatExp hypers).
        The reference differentiates the Theano graph; the same chain rule in closed form is
            d loglike = 1/2 sum_ij (alpha alpha^T - K^-1)_ij dK_ij  +  alpha . dm  -  alpha . dT^-1(y)  +  d logdet
        with K the (possibly jittered) matrix that was factored -- CholeskyRobust.grad re-uses the
        jittered factor the same way -- and zero when logp takes its constant -1e30 branch.
        K^-1 and the kernel-parameter sums are computed on the device (g3_gp_dlogp)."""
        values, _ = self._values(params)
        nat = self._potential_gradient(values)
        if not prior:
            ll = self.th_loglike(space, inputs, outputs, vector, params)
            if np.isfinite(ll) and ll != self.dtype.type(SENTINEL):
                self._dloglike(values, inputs, outputs, nat)
        return self._flat_gradient(values, nat)

    def _dlogp_scale(self, values, c, st, nat):
        return 1.0

    def _dloglike(self, values, inputs, outputs, nat):
        """adds d loglike / d (natural-space hyper) into `nat`"""
        dev = self.device
        c = self._factor(values, inputs, outputs)
        ds = self._dist
        if ds is not None and not ds.get('grad'):
            # several GPUs: from the first gradient on, every evaluation of this process carries the identity as
            # right-hand-side rows (the rank's rows of L^-T, g3_dist_set_grad); an evaluation cached without them is redone
            ds['grad'] = True
            if ds['dgp'] is not None:
                ds['dgp'].set_grad(True)
            c['which'] = None
        st = self._solve(c, values, 'logp')
        N, d, Np = c['N'], c['d'], c['Np']
        # d logp / d beta = -s / 2 with beta = |L^-1 delta|^2: s = 1 for the Gaussian density; the
        # Student-t density supplies its own s (and its degrees-of-freedom term) through the hook
        s = float(self._dlogp_scale(values, c, st, nat))
        if c.get('grad') is None and ds is not None:
            # K^-1 never exists in one place: gathered panels of L^-T, the rank's rows of K^-1, one all-reduce of the sums
            prog, gmap, slots, alpha = ds['dgp'].dlogp(self.f_kernel_noise.spec(values, d), c['Xd'], np.sqrt(s))
            c['grad'] = dict(prog=prog, gmap=gmap, slots=slots, alpha=np.sqrt(s) * alpha)
        if c.get('grad') is None:
            prog = self._prog(self.f_kernel_noise, values, d)
            gmap = dev.grad_layout(prog)
            ws = self._workspace                 # K^-1 workspaces live with the factor workspace (same data set)
            if 'Y' not in ws:
                ws['Y'], ws['Kinv'] = dev.alloc(Np, Np, self.dtype), dev.alloc(Np, Np, self.dtype)
                ws['alpha'] = dev.alloc(1, Np, self.dtype)
            Y, Kinv, alpha = ws['Y'], ws['Kinv'], ws['alpha']
            ad = c['ad']
            if s != 1.0:      # G = s alpha alpha^T - K^-1: hand the device sqrt(s) a, it returns sqrt(s) alpha
                ad = dev.upload((dev.download(c['ad']) * np.sqrt(s)).astype(self.dtype))
            slots = dev.gp_dlogp(prog, gmap, c['Xd'], N, d, c['Kd'], c['Wd'], ad, Y, Kinv, alpha)
            c['grad'] = dict(prog=prog, gmap=gmap, slots=slots,
                             alpha=np.sqrt(s) * dev.download(alpha, 1, N)[0].astype(np.float64))
        g = c['grad']
        self._chain_rule(values, inputs, outputs, nat, g['prog'], g['gmap'], g['slots'], g['alpha'], d)

    def _chain_rule(self, values, inputs, outputs, nat, prog, gmap, slots, alpha, d):
        """host part of d loglike: route the device's per-leaf parameter sums (`slots`) to the model's
        variables and add the O(N) location / warping terms on alpha = s K^-1 delta"""
        from ..device import spec_leaves
        # kernel hypers: leaf parameter slots -> the HyperVars that fed them
        by_name = {v.name: v for v in self.model.vars}
        refs = spec_leaves(self.f_kernel_noise.spec(_Refs(), d))
        fields = {'SE': dict(rate=2), 'OU': dict(rate=2), 'MAT32': dict(rate=2), 'MAT52': dict(rate=2),
                  'RQ': dict(rate=2, alpha=3), 'COS': dict(freq=2), 'SINC': dict(freq=2),
                  'SIN': dict(freq=2, rate=3), 'SM': dict(freq=2, rate=3), 'NOISE': {}, 'WN': {}}
        for l, lf in enumerate(refs):
            nd = prog.leaf[l].ndims
            for pname, idx in dict(var=1, **fields[lf[0]]).items():
                ref = lf[idx]
                if not isinstance(ref, _Ref) or ref.name not in by_name:
                    continue
                slot = getattr(gmap, pname)[l]
                if pname in ('var', 'alpha'):
                    nat[ref.name] = nat[ref.name] + slots[slot]
                else:
                    gk = slots[slot:slot + nd]
                    nat[ref.name] = nat[ref.name] + (gk if by_name[ref.name].shape else gk.sum())
        # location and warping hypers: O(N) host arithmetic on alpha = K^-1 delta
        X = self._x(inputs)
        y = np.asarray(outputs, dtype=self.dtype).reshape(-1)
        for h, J in self.f_location.grad(X, values):
            if getattr(h, 'name', None) in by_name:
                gj = alpha.dot(np.asarray(J, dtype=np.float64))
                nat[h.name] = nat[h.name] + (gj.reshape(by_name[h.name].shape) if by_name[h.name].shape else gj.sum())
        with np.errstate(all='ignore'):
            for h, dinv, dlogdet in self.f_mapping.grad(y, values):
                if getattr(h, 'name', None) in by_name:
                    nat[h.name] = nat[h.name] + (-alpha.dot(np.asarray(dinv, dtype=np.float64)) + float(dlogdet))

    def _flat_gradient(self, values, nat):
        """natural-space gradients -> flat vector in creation order, transformed space, tt_to_num'ed"""
        flat = []
        for v in self.model.vars:
            g = np.asarray(nat[v.name], dtype=np.float64).reshape(v.shape)
            if v.positive:
                g = g * np.asarray(values[v.name], dtype=np.float64)
            flat.append(g.reshape(-1))
        flat = np.concatenate(flat) if flat else np.zeros(0)
        flat = np.where(np.isnan(flat), 0.0, np.where(np.isinf(flat), float(np.float32(1e10)), flat))   # tt_to_num
        return flat.astype(self.dtype)

    def _potential_gradient(self, values):
        nat = {v.name: np.zeros(v.shape, dtype=np.float64) for v in self.model.vars}
        for _, reg, c, sel in self.model.potentials:                # hypers/__init__.py:97-109
            for h in sel:
                hv = np.asarray(values[h.name], dtype=np.float64)
                nat[h.name] = nat[h.name] + (-c * np.sign(hv) if reg == 'L1' else -2.0 * c * hv)
        return nat

    def _chain_density(self, values_b, st, det_m, N, B):
        """the rows' observed log-density from the device statistics st = [logdet, quad, ...] (B, 6): the Gaussian
        -n/2 log 2 pi - beta/2 - sum log L_ii + logdet_dinv (gaussian.py:208-224), evaluated in the process dtype"""
        t = self.dtype.type
        npi = t(-0.5) * t(N) * np.log(t(2.0 * np.pi))
        return npi + t(-0.5) * st[:, 1].astype(self.dtype) - st[:, 0].astype(self.dtype) + det_m

    def _chain_dlogp_scale(self, values_b, st, N, nat, ok, B):
        """rows' s in d logp / d beta = -s / 2 (None: the Gaussian's s = 1); a density with its own hypers adds their
        gradient terms to `nat` here (rows with ok False: none)"""
        return None

    def _chain_workspace(self, batch, Np, grad):
        """device buffers of logp_chain / dlogp_chain, kept between calls (samplers and optimisers evaluate chain after
        chain on one data set; a GB-sized hipMalloc + hipFree per call costs more than the evaluation: 250 - 390 ms spikes
        against 7 ms).  One workspace serves both: it only grows (more members, or the two extra matrices of the gradient)
        and is replaced when the padded size or the dtype changes."""
        ws = getattr(self, '_chain_ws', None)
        same = ws is not None and ws['Np'] == int(Np) and ws['dtype'] == self.dtype.str
        if same and ws['cap'] >= batch and (ws['grad'] or not grad):
            return ws
        cap, grad = int(batch), bool(grad)
        if ws is not None:
            if same:                 # grow, never shrink: logp_chain and dlogp_chain alternate inside one optimiser
                cap, grad = max(cap, ws['cap']), grad or ws['grad']
            for b in ws['bufs']:
                b.free()
        dev = self.device
        n3 = 3 if grad else 1
        mats = [dev.alloc(cap * (Np + _lib.G3_RHS_PAD), Np, self.dtype) for _ in range(n3)]
        W = dev.alloc(cap * Np, _lib.G3_PAD, self.dtype)
        vecs = [dev.alloc(cap, Np, self.dtype) for _ in range(2 if grad else 1)]
        ws = dict(Np=int(Np), dtype=self.dtype.str, cap=cap, grad=grad, K=mats[0], Y=mats[1] if grad else None,
                  Ki=mats[2] if grad else None, W=W, a=vecs[0], al=vecs[1] if grad else None, bufs=mats + [W] + vecs)
        self._chain_ws = ws
        return ws

    def dlogp_chain(self, chain, batch=None):
        """one dlogp per row of a flat-parameter chain, shape (rows, ndim) -- what fixed_dlogp averages
        (stochastic.py:554-564: a loop of single gradients in the reference).  `batch` rows at a time
        share ONE Gram launch, ONE factorisation sweep (g3_gp_factor_batched_fields) and ONE K^-1 sweep with the
        members' alpha and kernel-parameter sums in launches that carry the member in grid.y
        (g3_gp_dlogp_batched_fields); the O(N) chain-rule pieces are array arithmetic over the rows on the host."""
        chain = np.atleast_2d(np.asarray(chain, dtype=np.float64))
        n_rows = len(chain)
        out = np.zeros((n_rows, self.active.ndim), dtype=self.dtype)
        custom_scale = (type(self)._dlogp_scale is not GaussianProcess._dlogp_scale
                        and type(self)._chain_dlogp_scale is GaussianProcess._chain_dlogp_scale)
        if not self.is_observed or n_rows == 0 or custom_scale:
            for i in range(n_rows):          # a density with a per-member scale and no row form of it: one at a time
                out[i] = self.dlogp(chain[i], array=True)
            return out
        dev = self.device
        X = self._x(self.inputs)
        y = np.asarray(self.outputs, dtype=self.dtype).reshape(-1)
        N, d = X.shape
        Np = _lib.roundup(N)
        kstride = (Np + _lib.G3_RHS_PAD) * Np
        if batch is None:
            batch = int(4e9 // (3 * kstride * self.dtype.itemsize))
        batch = max(1, min(int(batch), n_rows, _lib.G3_MAX_BATCH))
        Xd = dev.upload(X)
        ws = self._chain_workspace(batch, Np, True)
        K, Y, Ki, W, a, al = ws['K'], ws['Y'], ws['Ki'], ws['W'], ws['a'], ws['al']
        t = self.dtype.type
        for lo in range(0, n_rows, batch):
            hi = min(lo + batch, n_rows)
            B = hi - lo
            # the block's host side in NumPy passes over all rows (as logp_chain): values, programs as template + fields,
            # warped observations, mean; the device then does factor, K^-1, alpha and the kernel-parameter sums of every
            # member in batched launches, and the chain rule below is array arithmetic over the rows
            values_b, _ = self._values_rows(chain[lo:hi])
            tmpl, offs, fields = compile_spec_rows(self.f_kernel_noise.spec(values_b, d),
                                                   self.f_kernel_noise.spec(self._values_row(values_b, 0), d), d, B)
            with np.errstate(all='ignore'):
                delta = (np.asarray(self.f_mapping.inv_rows(y, values_b, B), dtype=self.dtype)
                         - np.asarray(self.f_location.rows(X, values_b, B), dtype=self.dtype))
                det_m = np.asarray(self.f_mapping.logdet_dinv_rows(y, values_b, B), dtype=self.dtype)
            bad = ~(np.isfinite(delta).all(axis=1) & np.isfinite(det_m))                  # constant -1e30 branch
            if bad.any():
                delta = np.where(bad[:, None], t(0), delta)
            st = dev.gp_factor_batched_fields(tmpl, offs, fields, Xd, N, d, dev.upload(np.ascontiguousarray(delta, dtype=self.dtype)),
                                              K, kstride, W, a)
            gmap = dev.grad_layout(tmpl)
            ok = ~bad & np.isfinite(st[:, 0]) & (st[:, 2] == 0)
            nat = self._potential_gradient_rows(values_b, B)
            s = self._chain_dlogp_scale(values_b, st, N, nat, ok, B)
            if s is not None:     # G = s alpha alpha^T - K^-1: the device gets sqrt(s) a and returns sqrt(s) alpha
                with np.errstate(all='ignore'):
                    rs = np.where(ok, np.sqrt(s), 1.0)
                dev.copy_in(a, (dev.download(a, B, Np) * rs[:, None].astype(self.dtype)).astype(self.dtype))
            slots = dev.gp_dlogp_batched_fields(tmpl, offs, fields, gmap, Xd, N, d, K, kstride, W, a, Y, Ki, al)
            alphas = dev.download(al, B, N).astype(np.float64)
            if s is not None:
                alphas = alphas * rs[:, None]
            self._chain_rule_rows(values_b, X, y, nat, tmpl, gmap, slots, alphas, d, ok, B)
            out[lo:hi] = self._flat_gradient_rows(values_b, nat, B)
        return out

    def _potential_gradient_rows(self, values_b, B):
        nat = {v.name: np.zeros((B,) + tuple(v.shape)) for v in self.model.vars}
        for _, reg, c, sel in self.model.potentials:                # hypers/__init__.py:97-109
            for h in sel:
                hv = np.asarray(values_b[h.name], dtype=np.float64)
                nat[h.name] = nat[h.name] + (-c * np.sign(hv) if reg == 'L1' else -2.0 * c * hv)
        return nat

    def _chain_rule_rows(self, values_b, X, y, nat, prog, gmap, slots, alphas, d, ok, B):
        """_chain_rule for B rows at once: slots (B, nslots), alphas (B, N); rows with ok False get no likelihood term"""
        from ..device import spec_leaves
        by_name = {v.name: v for v in self.model.vars}
        with np.errstate(all='ignore'):
            slots = np.where(ok[:, None], slots, 0.0)
            alphas = np.where(ok[:, None], alphas, 0.0)

        def add(name, g):            # g: (B,) for a scalar hyper, (B, size) otherwise
            shp = tuple(by_name[name].shape)
            nat[name] = nat[name] + np.asarray(g, dtype=np.float64).reshape((B,) + shp)
        refs = spec_leaves(self.f_kernel_noise.spec(_Refs(), d))
        fld = {'SE': dict(rate=2), 'OU': dict(rate=2), 'MAT32': dict(rate=2), 'MAT52': dict(rate=2),
               'RQ': dict(rate=2, alpha=3), 'COS': dict(freq=2), 'SINC': dict(freq=2),
               'SIN': dict(freq=2, rate=3), 'SM': dict(freq=2, rate=3), 'NOISE': {}, 'WN': {}}
        for l, lf in enumerate(refs):
            nd = prog.leaf[l].ndims
            for pname, idx in dict(var=1, **fld[lf[0]]).items():
                ref = lf[idx]
                if not isinstance(ref, _Ref) or ref.name not in by_name:
                    continue
                slot = getattr(gmap, pname)[l]
                if pname in ('var', 'alpha'):
                    add(ref.name, slots[:, slot])
                else:
                    gk = slots[:, slot:slot + nd]
                    add(ref.name, gk if by_name[ref.name].shape else gk.sum(axis=1))
        # location: d m / d hyper is the same for every row for the means on the path (linear in their hypers)
        if getattr(self.f_location, 'JAC_CONSTANT', False):
            for h, J in self.f_location.grad(X, self._values_row(values_b, 0)):
                if getattr(h, 'name', None) in by_name:
                    gj = alphas.dot(np.asarray(J, dtype=np.float64))           # (B, size)
                    add(h.name, gj if by_name[h.name].shape else gj.sum(axis=1))
        else:
            for j in range(B):
                for h, J in self.f_location.grad(X, self._values_row(values_b, j)):
                    if getattr(h, 'name', None) in by_name:
                        gj = alphas[j].dot(np.asarray(J, dtype=np.float64))
                        nat[h.name][j] = nat[h.name][j] + (gj.reshape(by_name[h.name].shape) if by_name[h.name].shape else gj.sum())
        with np.errstate(all='ignore'):
            for h, dinv, dlogdet in self.f_mapping.grad_rows(y, values_b, B):
                if getattr(h, 'name', None) in by_name:
                    g = -(alphas * np.asarray(dinv, dtype=np.float64)).sum(axis=1) + np.asarray(dlogdet, dtype=np.float64)
                    add(h.name, np.where(ok, g, 0.0))

    def _flat_gradient_rows(self, values_b, nat, B):
        flat = []
        for v in self.model.vars:
            g = np.asarray(nat[v.name], dtype=np.float64).reshape(B, -1)
            if v.positive:
                g = g * np.asarray(values_b[v.name], dtype=np.float64).reshape(B, -1)
            flat.append(g)
        flat = np.concatenate(flat, axis=1) if flat else np.zeros((B, 0))
        flat = np.where(np.isnan(flat), 0.0, np.where(np.isinf(flat), float(np.float32(1e10)), flat))   # tt_to_num
        return flat.astype(self.dtype)

    # ---- many hyper-parameter vectors on the same observations (stochastic.py:515-520)
    def logp_chain(self, chain, prior=False, batch=None):
        """one logp per row of a flat-parameter chain.  The reference loops over the rows
        (stochastic.py:515-520; fixed_logp :526-532, "TODO: Vectorized"); here `batch` rows at a
        time go through ONE Gram launch and ONE factorisation sweep (g3_gp_factor_batched), which
        is what fills the GPU when a single N x N problem is too small to.  `batch` defaults to
        as many members as fit in 4 GB of covariance workspace."""
        chain = np.atleast_2d(np.asarray(chain, dtype=np.float64))
        n_rows = len(chain)
        t = self.dtype.type
        out = np.empty(n_rows, dtype=self.dtype)
        if prior or not self.is_observed or n_rows == 0:
            if n_rows:                      # the free variables' terms only (th_logp with prior=True): no device work
                out[:] = self._values_rows(chain)[1].astype(self.dtype)
            return out
        dev = self.device
        X = self._x(self.inputs)
        y = np.asarray(self.outputs, dtype=self.dtype).reshape(-1)
        N, d = X.shape
        Np = _lib.roundup(N)
        kstride = (Np + _lib.G3_RHS_PAD) * Np
        if batch is None:
            batch = int(4e9 // (kstride * self.dtype.itemsize))
        batch = max(1, min(int(batch), n_rows, _lib.G3_MAX_BATCH))
        Xd = dev.upload(X)
        ws = self._chain_workspace(batch, Np, False)
        K, W, a = ws['K'], ws['W'], ws['a']
        for lo in range(0, n_rows, batch):
            hi = min(lo + batch, n_rows)
            B = hi - lo
            # the whole block at once on the host: values, programs (one template + the hyper values that differ per
            # row), warped observations and the mean -- O(B) NumPy passes, no per-row Python
            values_b, logjac = self._values_rows(chain[lo:hi])
            values0 = self._values_row(values_b, 0)
            tmpl, offs, fields = compile_spec_rows(self.f_kernel_noise.spec(values_b, d),
                                                   self.f_kernel_noise.spec(values0, d), d, B)
            with np.errstate(all='ignore'):
                delta = (np.asarray(self.f_mapping.inv_rows(y, values_b, B), dtype=self.dtype)
                         - np.asarray(self.f_location.rows(X, values_b, B), dtype=self.dtype))
                det_m = np.asarray(self.f_mapping.logdet_dinv_rows(y, values_b, B), dtype=self.dtype)
            bad = ~(np.isfinite(delta).all(axis=1) & np.isfinite(det_m))                  # gaussian.py:234-235
            if bad.any():
                delta = np.where(bad[:, None], t(0), delta)        # evaluated like the others, result replaced below
            dd = dev.upload(np.ascontiguousarray(delta, dtype=self.dtype))
            st = dev.gp_factor_batched_fields(tmpl, offs, fields, Xd, N, d, dd, K, kstride, W, a)
            with np.errstate(all='ignore'):
                lp = logjac.astype(self.dtype) + self._chain_density(values_b, st, det_m, N, B).astype(self.dtype)
            bad |= ~np.isfinite(st[:, 0]) | (st[:, 2] > 0)                                # gaussian.py:237
            out[lo:hi] = np.where(bad, logjac.astype(self.dtype) + t(SENTINEL), lp)
        return out

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
        loc = self.location(params, space, inputs, outputs, prior=prior, noise=noise)
        if self._dist is not None and not prior:
            # several GPUs: every rank must use the SAME normals (rank 0's are broadcast), the posterior covariance of
            # the f process is formed and factored by the driver (g3_dist_posterior_draws) from the cross solve that
            # `location` has just left in it
            if noise:
                raise _lib.G3Error('draws with the noise term are not available on a distributed process')
            ds = self._dist
            if ds['world'] > 1:
                box = [rand if ds['rank'] == 0 else None]
                ds['dist'].broadcast_object_list(box, src=0)
                rand = np.asarray(box[0], dtype=self.dtype)
            values, _ = self._values(params)
            S_ = self._x(space)
            g = ds['dgp'].draws(self.f_kernel.spec(values, S_.shape[1]), self.device.upload(S_), np.asarray(loc, dtype=np.float64), rand)
            return self.mapping(params, space, inputs, outputs=g.astype(self.dtype))
        Ld, _, _ = self._cholesky_dev(params, space, inputs, outputs, prior=prior, noise=noise)   # stays on the device
        g = self.device.gp_sample(Ld, M, loc, rand)                # loc + L Z, one call (g3_gp_sample)
        # the mapping is element-wise: one vectorised pass over the M x S draws instead of the
        # reference's per-sample loop (gaussian.py:97)
        return self.mapping(params, space, inputs, outputs=g)

    _methods = EllipticalProcess._methods + (('dlogp', 'th_dlogp'),)


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
