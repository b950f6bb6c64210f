"""StochasticProcess: the user-facing API, parameter conventions and compiled-method dispatch
of g3py/processes/stochastic.py:20-105, 150-201, 300-313, 385-430, 444-513, rebuilt on the
HIP path.  Optimisers, MCMC drivers, plotting and pickling are out of scope.
"""
import types

import numpy as np

from ..libs import DictObj, clone
from ..libs.tensors import makefn
from .hypers import Model


class GraphicalModel:
    """The slice of g3py/bayesian/models.py:56-182 the path depends on: the variable registry,
    default / current parameters and the dict <-> flat-array bijection (models.py:143-155)."""
    active = None

    def __init__(self, name='GM'):
        self.name = name
        self.model = Model(name)
        self.components = DictObj()
        self.current_params = None
        self.fix_vars()

    def add_component(self, component):
        self.components[component.name] = component

    def compile_components(self, precompile=False):
        pass

    # ---- bijection (pm.DictToArrayBijection over vars in creation order)
    @property
    def ndim(self):
        return self.model.ndim

    def dict_to_array(self, params):
        return np.concatenate([np.asarray(params[v.key], dtype=np.float64).reshape(-1)
                               for v in self.model.vars]) if self.model.vars else np.zeros(0)

    def array_to_dict(self, array):
        array = np.asarray(array, dtype=np.float64).reshape(-1)
        r, o = DictObj(), 0
        for v in self.model.vars:
            r[v.key] = array[o:o + v.size].reshape(v.shape)
            o += v.size
        return r

    class _Bijection:
        def __init__(self, gm):
            self.map = gm.dict_to_array
            self.rmap = gm.array_to_dict

    @property
    def bijection(self):
        return GraphicalModel._Bijection(self)

    # ---- fixed chains (models.py:270-296): some variables are pinned to the rows of a chain / datatrace,
    #      the others are the ones an optimiser or sampler moves
    def fix_vars(self, datatrace=None, keys=None):
        """`datatrace`: 2-D array or DataFrame whose first `ndim` columns are flat parameter vectors;
        `keys`: the names (transformed keys or natural names) of the variables that stay fixed"""
        if datatrace is None or keys is None:
            self.fixed_keys, self.fixed_datatrace, self.fixed_chain, self.fixed_dims = [], None, None, []
            return
        self.fixed_keys = list(keys)
        self.fixed_datatrace = datatrace.copy()
        values = getattr(self.fixed_datatrace, 'values', self.fixed_datatrace)
        self.fixed_chain = np.array(np.asarray(values)[:, :self.ndim], dtype=np.float64)
        dims, o = [], 0
        for v in self.model.vars:
            if v.key in self.fixed_keys or v.name in self.fixed_keys:
                dims += list(range(o, o + v.size))
            o += v.size
        self.fixed_dims = sorted(dims)

    @property
    def sampling_dims(self):
        return sorted(set(range(self.ndim)) - set(self.fixed_dims))

    def sampling_params(self, params):
        """the entries an optimiser / sampler moves, from a dict or a flat vector"""
        flat = self.dict_to_array(params) if isinstance(params, dict) else np.asarray(params)
        return flat[self.sampling_dims]

    def dict_from_sampling_array(self, params):
        """inverse of sampling_params: the moved entries spliced into the current parameter vector"""
        flat = np.asarray(params, dtype=np.float64)
        if self.fixed_datatrace is not None:
            full = self.dict_to_array(self.params)
            full[self.sampling_dims] = flat
            flat = full
        return self.array_to_dict(flat)

    # ---- parameters
    def set_params(self, params=None):
        self.current_params = None if params is None else DictObj(params)

    @property
    def params(self):
        """a copy of the parameters set with set_params, else the defaults"""
        return self.params_default if self.current_params is None else clone(self.current_params)

    @property
    def params_test(self):
        return DictObj(self.model.test_point)

    @property
    def params_default(self):
        """models.py:174-182 + transformed_hypers (:46-53): defaults in transformed space"""
        out = self.params_test
        mine = set(self.model.vars)
        for component in self.components.values():
            for var, natural in component.default_hypers().items():
                if var not in mine:
                    continue
                value = np.asarray(natural, dtype=np.float64)
                if var.positive:
                    with np.errstate(all='ignore'):
                        value = np.log(value)
                out[var.key] = value.reshape(var.shape)
        return out

    def transform_params(self, params, to_dict=True, to_transformed=True, complete=False):
        """natural <-> transformed names and values (models.py:232-260)"""
        if not isinstance(params, dict):
            params = self.array_to_dict(params)
        r = DictObj(self.params) if (complete or not to_dict) else DictObj()
        by_name = {v.name: v for v in self.model.vars}
        by_key = {v.key: v for v in self.model.vars}
        for k, val in params.items():
            if to_transformed and k in by_name and by_name[k].positive:
                r[by_name[k].key] = np.log(val)
            elif (not to_transformed) and k in by_key and by_key[k].positive:
                r[by_key[k].name] = np.exp(val)
            else:
                r[k] = val
        return r if to_dict else self.dict_to_array(r)


class StochasticProcess:
    """stochastic.py:20-105 (construction), :150-201 (space/observations), :385-430
    (method dispatch), :444-513 (predict)."""

    def __init__(self, space=None, order=None, inputs=None, outputs=None, hidden=None, index=None,
                 name='SP', distribution=None, active=False, precompile=False, file=None, load=True,
                 compile_logp=True, dtype=np.float64, device=None, *args, **kwargs):
        self.makefn = makefn
        self.nspace = self._columns_of(space)
        self.name = name
        self.dtype = np.dtype(dtype)
        self._device = device
        # until set_space / observed supply data every array is the reference's 2-point dummy (:46-56)
        pair = np.array([0.0, 1.0], dtype=self.dtype)
        grid = np.tile(pair[:, None], (1, self.nspace))
        self._order, self._index, self._outputs = pair.copy(), pair.copy(), pair.copy()
        self._space, self._inputs = grid.copy(), grid.copy()
        self.is_observed, self.np_hidden, self.distribution = False, None, distribution
        self.active = self._graph_for(active)
        self.active.add_component(self)
        self.compiles = DictObj()
        self.precompile = precompile
        self.file = file   # accepted for signature compatibility; pickling is out of scope
        with self.model:
            self._check_hypers()
            self.th_define_process()
            self.active.compile_components()
        self.set_space(space=space, hidden=hidden, order=order, inputs=inputs, outputs=outputs, index=index)
        self._compile_methods(compile_logp)
        if hidden is None:
            self.hidden = hidden

    @staticmethod
    def _columns_of(space):
        """number of input columns: from an array's second axis, or given directly as an int"""
        if space is None:
            return 1
        shape = getattr(space, 'shape', None)
        if shape is None:
            return int(space)
        return shape[1] if len(shape) > 1 else 1

    def _graph_for(self, active):
        """True: the shared model of all active=True processes; False: a private one; else the given one"""
        if active is True:
            if GraphicalModel.active is None:
                GraphicalModel.active = GraphicalModel('GM_' + self.name)
            return GraphicalModel.active
        return GraphicalModel('GM_' + self.name) if active is False else active

    # ---- parameters
    def set_params(self, *args, **kwargs):
        return self.active.set_params(*args, **kwargs)

    def transform_params(self, *args, **kwargs):
        return self.active.transform_params(*args, **kwargs)

    @property
    def model(self):
        return self.active.model

    @property
    def params(self):
        return self.active.params

    @property
    def params_default(self):
        return self.active.params_default

    @property
    def params_test(self):
        return self.active.params_test

    def filter_params(self, params):
        return {k: params[k] for k in self.model.test_point}   # models.py:471-473

    # ---- space and observations (boundary: stochastic.py:150-201, 219-259 -- copy on set)
    # how each array handed to set_space is normalised: True = matrix (a vector becomes one column),
    # False = vector (a matrix is flattened to its first axis)
    _ARRAY_RANK2 = (('space', True), ('hidden', False), ('order', False), ('inputs', True),
                    ('outputs', False), ('index', False))
    # a vector that, when not supplied for a one-dimensional space, mirrors a matrix; and the
    # matrix whose length it must match otherwise (it falls back to 0..n-1)
    _MIRRORS = (('order', 'space'), ('index', 'inputs'))

    def set_space(self, space=None, hidden=None, order=None, inputs=None, outputs=None, index=None):
        given = dict(space=space, hidden=hidden, order=order, inputs=inputs, outputs=outputs, index=index)
        for name, rank2 in self._ARRAY_RANK2:
            a = given[name]
            if a is None:
                continue
            a = np.asarray(a)
            n = len(a)
            if rank2 and a.ndim < 2:
                a = a.reshape(n, 1)
            elif not rank2 and a.ndim > 1:
                a = a.reshape(n)
            setattr(self, name, a)
        for vec, mat in self._MIRRORS:
            m = getattr(self, mat)
            if given[vec] is None and self.nspace == 1:
                setattr(self, vec, m.reshape(len(m)))
            if len(getattr(self, vec)) != len(m):
                setattr(self, vec, np.arange(len(m)))

    def observed(self, inputs=None, outputs=None, order=None, index=None, hidden=None):
        self.set_space(inputs=inputs, outputs=outputs, order=order, index=index, hidden=hidden)
        self.is_observed = not (inputs is None and outputs is None)

    def _get(self, name):
        return getattr(self, '_' + name).copy()

    def _set(self, name, value):
        setattr(self, '_' + name, np.array(value, dtype=self.dtype, copy=True))

    space = property(lambda s: s._get('space'), lambda s, v: s._set('space', v))
    inputs = property(lambda s: s._get('inputs'), lambda s, v: s._set('inputs', v))
    outputs = property(lambda s: s._get('outputs'), lambda s, v: s._set('outputs', v))
    order = property(lambda s: s._get('order'), lambda s, v: s._set('order', v))
    index = property(lambda s: s._get('index'), lambda s, v: s._set('index', v))

    @property
    def hidden(self):
        return self.np_hidden

    @hidden.setter
    def hidden(self, value):
        self.np_hidden = value

    def _values(self, params):
        """transformed-space params dict -> natural-space values by hyper name, and the terms every
        process adds to logp besides the observed density: the log-Jacobian of the FlatExp variables
        (hypers/__init__.py:199-200) and the L1 / L2 potentials registered by check_potential
        (hypers/__init__.py:94-109; stochastic.py:305).  Shared by the elliptical and the transport
        processes."""
        memo = getattr(self, '_values_memo', None)
        if memo is not None and memo[0] is params:          # th_logp -> th_loglike -> ... share one params object
            return memo[1], memo[2]
        values, logjac = {}, 0.0
        for v in self.model.vars:
            p = np.asarray(params[v.key], dtype=np.float64)
            if v.positive:
                with np.errstate(over='ignore'):
                    e = np.exp(p)
                logjac += float(np.sum(np.where(e > 1e-6, 0.0, -np.inf)))
                values[v.name] = e
            else:
                values[v.name] = p
        # optional L1 / L2 potentials enter th_logp like pm.Potential terms (stochastic.py:305)
        for _, reg, c, sel in self.model.potentials:
            if reg == 'L1':
                logjac += c * -float(sum(np.sum(np.abs(values[h.name])) for h in sel))
            elif reg == 'L2':
                logjac += c * -float(sum(np.sum(np.asarray(values[h.name]) ** 2) for h in sel))
        self._values_memo = (params, values, logjac)
        return values, logjac

    def _values_rows(self, chain):
        """_values for every row of a flat-parameter chain at once: {hyper name: Rows (B, *shape)} and the (B,) extra
        logp terms.  Same arithmetic as _values, reductions taken per row."""
        from ..device import Rows
        chain = np.asarray(chain, dtype=np.float64)
        B = len(chain)
        values, logjac, o = {}, np.zeros(B), 0
        for v in self.model.vars:
            p = chain[:, o:o + v.size].reshape((B,) + tuple(v.shape))
            o += v.size
            if v.positive:
                with np.errstate(over='ignore'):
                    p = np.exp(p)
                logjac = logjac + np.where(p > 1e-6, 0.0, -np.inf).reshape(B, -1).sum(axis=1)
            values[v.name] = np.ascontiguousarray(p).view(Rows)
        for _, reg, c, sel in self.model.potentials:
            tot = np.zeros(B)
            for h in sel:
                hv = np.asarray(values[h.name]).reshape(B, -1)
                tot = tot + (np.abs(hv).sum(axis=1) if reg == 'L1' else (hv ** 2).sum(axis=1))
            if reg in ('L1', 'L2'):
                logjac = logjac + c * -tot
        return values, logjac

    @staticmethod
    def _values_row(values_rows, j):
        """row j of _values_rows as the plain dict _values returns"""
        return {k: np.asarray(v)[j] for k, v in values_rows.items()}

    # ---- to be provided by subclasses
    def default_hypers(self):
        return {}

    def _check_hypers(self):
        pass

    def th_define_process(self):
        pass

    def sampler(self, samples=1, prior=False, noise=False):
        pass

    def quantiler(self, q=0.975, prior=False, noise=False, simulations=None):
        pass

    # names of the statistics a subclass implements as  fn(ctx, prior, noise, **kw)
    _methods = ()

    def _compile_methods(self, compile_logp=True):
        """bind every statistic as a lazily 'compiled' method (stochastic.py:328-380)"""
        if self.compiles is None:
            self.compiles = DictObj()
        for public, th_name in self._methods:
            setattr(self, public, types.MethodType(self._method_name(th_name), self))
        if compile_logp:
            # the reference force-compiles logp on the 2-point dummy data here; our kernels are
            # already compiled, so only the registry entries are created
            for prior in (False, True):
                self._compiled('th_logp', prior, False, True, (), {})

    def _compiled(self, method, prior, noise, array, args, kwargs):
        # registry key, e.g. posterior_logp, prior_kernel_sd_noise (boundary: stochastic.py:404-415)
        name = ''.join(['prior' if prior else 'posterior', method.replace('th', ''), '_noise' if noise else '',
                        str(args) if args else '', str(kwargs) if kwargs else ''])
        if name not in self.compiles:
            impl = getattr(self, method)

            def fn(space, inputs, outputs, vector, params, _impl=impl):
                return _impl(space, inputs, outputs, vector, params, prior=prior, noise=noise, *args, **kwargs)
            th_vars = [self.name + '_space_th', self.name + '_inputs_th', self.name + '_outputs_th',
                       self.name + '_vector_th'] + [v.key for v in self.model.vars]
            self.compiles[name] = self.makefn(th_vars, fn, givens=[('space', 'space_th'), ('inputs', 'inputs_th'),
                                                                    ('outputs', 'outputs_th')],
                                              bijection=None, precompile=self.precompile)
        if not array:
            return self.compiles[name]
        flat = 'array_' + name               # same function behind the flat-vector -> dict bijection
        if flat not in self.compiles:
            self.compiles[flat] = self.compiles[name].clone(self.active.bijection.rmap)
        return self.compiles[flat]

    def _call_defaults(self, params, space, inputs, outputs, prior, array):
        """fill in what a caller left out (boundary: stochastic.py:387-402): current parameters (as a
        flat vector when array=True; a supplied dict is cut down to the model's variables), the
        stored space / observations, and the prior when nothing has been observed"""
        if params is None:
            params = self.active.dict_to_array(self.params) if array else self.params
        elif not array:
            params = self.filter_params(params)
        unobserved_call = inputs is None and not self.is_observed
        stored = (self.space, self.inputs, self.outputs)
        space, inputs, outputs = (s if a is None else a for a, s in zip((space, inputs, outputs), stored))
        return params, space, inputs, outputs, (True if unobserved_call else prior)

    @staticmethod
    def _method_name(method=None):
        """public statistic bound to an instance: signature (params, space, inputs, outputs, vector,
        prior, noise, array) as in the reference; resolves defaults, then runs the registry entry"""
        def statistic(self, params=None, space=None, inputs=None, outputs=None, vector=[], prior=False,
                      noise=False, array=False, *args, **kwargs):
            params, space, inputs, outputs, prior = self._call_defaults(params, space, inputs, outputs, prior, array)
            kwargs.pop('simulations', None)   # accepted and unused, as in the reference's th_* methods
            fn = self._compiled(method, prior, noise, array, args, kwargs)
            return fn(params, space, inputs, outputs, vector)
        statistic.__name__ = str(method)
        return statistic

    @property
    def executed(self):
        return {k: v.executed for k, v in self.compiles.items()}

    # ---- predict (boundary: stochastic.py:444-513 -- switches and returned keys)
    # (switch, returned key, method, extra keyword arguments, noise override)
    _PREDICT_TABLE = (
        ('mean', 'mean', 'mean', {}, None),
        ('var', 'variance', 'variance', {}, None),
        ('std', 'std', 'std', {}, None),
        ('cov', 'covariance', 'covariance', {}, None),
        ('median', 'median', 'median', {}, None),
        ('quantiles', 'quantile_up', 'quantiler', {'q': 0.975}, None),
        ('quantiles', 'quantile_down', 'quantiler', {'q': 0.025}, None),
        ('quantiles_noise', 'noise_std', 'std', {}, True),
        ('quantiles_noise', 'noise_up', 'quantiler', {'q': 0.975}, True),
        ('quantiles_noise', 'noise_down', 'quantiler', {'q': 0.025}, True),
    )

    def predict(self, params=None, space=None, inputs=None, outputs=None, mean=True, std=True, var=False,
                cov=False, median=False, quantiles=False, quantiles_noise=False, samples=0, distribution=False,
                prior=False, noise=False, simulations=None):
        params = self.params if params is None else params
        prior = prior or not self.is_observed
        space = self.space if space is None else space
        inputs = self.inputs if inputs is None else inputs
        outputs = self.outputs if outputs is None else outputs
        where = (params, space, inputs, outputs)
        if type(simulations) is int:    # the reference draws them here and then ignores them downstream
            self.sampler(*where, prior=prior, noise=noise, samples=simulations)
        switches = dict(mean=mean, var=var, std=std, cov=cov, median=median, quantiles=quantiles,
                        quantiles_noise=quantiles_noise)
        values = DictObj()
        for switch, key, method, extra, noise_override in self._PREDICT_TABLE:
            if switches[switch]:
                values[key] = getattr(self, method)(*where, prior=prior,
                                                    noise=noise if noise_override is None else noise_override, **extra)
        if samples > 0:
            values['samples'] = self.sampler(*where, samples=samples, prior=prior, noise=noise)
        if distribution:
            def logpredictive(x, _where=where, _prior=prior):
                return self.logpredictive(*_where, vector=x, prior=_prior, noise=True)
            values['logpredictive'] = logpredictive
        return values

    def sample(self, params=None, space=None, inputs=None, outputs=None, samples=1, prior=False, noise=False):
        """draws only (boundary: models.py:443-446)"""
        return self.predict(params=params, space=space, inputs=inputs, outputs=outputs, mean=False, std=False,
                            samples=samples, prior=prior, noise=noise)['samples']

    def scores(self, params=None, space=None, hidden=None, inputs=None, outputs=None, logp=False, logpred=False,
               bias=True, variance=False, median=False, *args, **kwargs):
        """scoring harness over mean / variance / median / logpredictive against the hidden truth
        (boundary: models.py:449-469 -- the keys `_l1`, `_l2`, `_mse`, `_rmse`, `_median_l1`,
        `_median_l2`, `_logp`, `_loglike`, `_logprior`, `_nlpd`)"""
        truth = self.hidden if hidden is None else hidden
        pred = self.predict(params=params, space=space, inputs=inputs, outputs=outputs, mean=True, var=variance,
                            median=median, distribution=logpred)

        def err(which, power):
            return np.mean(np.abs(pred[which] - truth) ** power)
        out = DictObj()
        if bias:
            out.update(_l1=err('mean', 1), _l2=err('mean', 2))
        if variance:
            mse = np.mean((pred.mean - truth) ** 2 + pred.variance)
            out.update(_mse=mse, _rmse=np.sqrt(mse))
        if median:
            out.update(_median_l1=err('median', 1), _median_l2=err('median', 2))
        if logp:
            out.update(_logp=self.logp(params), _loglike=self.loglike(params), _logprior=self.logp(params, prior=True))
        if logpred:
            out['_nlpd'] = -pred.logpredictive(truth) / len(truth)
        return out

    def logp_chain(self, chain, prior=False):
        """stochastic.py:515-520: one logp per row of a flat-parameter chain"""
        return np.array([self.logp(row, array=True, prior=prior) for row in chain], dtype=np.float64)

    # ---- averages over a fixed chain (stochastic.py:522-564): the rows of active.fixed_chain with
    #      the sampling dimensions overwritten by `sampling_params`
    def _fixed_rows(self, sampling_params):
        if self.active.fixed_chain is None:
            raise ValueError('no fixed chain: call active.fix_vars(datatrace, keys) first')
        self.active.fixed_chain[:, self.active.sampling_dims] = sampling_params
        return self.active.fixed_chain

    def fixed_logp(self, sampling_params, return_array=False):
        r = self.logp_chain(self._fixed_rows(sampling_params))          # one batched sweep where available
        return r if return_array else np.mean(r)

    def fixed_logprior(self, sampling_params, return_array=False):
        r = self.logp_chain(self._fixed_rows(sampling_params), prior=True)
        return r if return_array else np.mean(r)

    def fixed_loglike(self, sampling_params, return_array=False):
        rows = self._fixed_rows(sampling_params)
        r = self.logp_chain(rows) - self.logp_chain(rows, prior=True)
        return r if return_array else np.mean(r)

    def fixed_dlogp(self, sampling_params, return_array=False):
        rows = self._fixed_rows(sampling_params)
        if hasattr(self, 'dlogp_chain'):      # one batched factor + K^-1 sweep where the process provides it
            r = np.asarray(self.dlogp_chain(rows))[:, self.active.sampling_dims]
        else:
            r = np.array([self.dlogp(p, array=True)[self.active.sampling_dims] for p in rows])
        return r if return_array else np.mean(r, axis=0)
