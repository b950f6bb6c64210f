"""The run / score harness around the hot path (boundary: g3py/bayesian/selection.py:43-338 and the
subsampling helpers of g3py/libs/data.py:129-236).

`Experiment` repeatedly splits a data set into observations / (hold-out) / test points, conditions
every model on the observations, obtains parameters, and records `scores()` on the three sets together
with how long each stage took (`time_params`, `time_obs`, `time_valid`, `time_test`) -- the consumer
of mean / variance / median / logpredictive that SURVEY.md section 8f ranks fourth.  What is kept is
the protocol (method names, the order of observed / set_space / scores calls, the result columns);
what is out of scope stays out: the optimisers (`find_MAP`) and MCMC starts are not rebuilt, so
parameter selection is either the model's defaults or a callable the user supplies; plotting and
the pickle / HDF5 persistence (`file=`) are accepted and ignored.
"""
import time
from datetime import datetime

import numpy as np

def _pandas():
    """the result tables are pandas frames, as in the reference; nothing else in g3py_amd needs pandas, so it is
    imported only when an Experiment is created"""
    try:
        import pandas
    except ImportError as e:
        raise ImportError('g3py_amd.Experiment keeps its simulation / result tables in pandas DataFrames '
                          '(boundary: g3py/bayesian/selection.py:43-60): install pandas to use it') from e
    return pandas


SIM_COLUMNS = ('obs', 'valid', 'test', 'datetime')
RESULT_COLUMNS = ('n_sim', 'model', 'selected', 'start', 'params', 'scores_obs', 'scores_valid', 'scores_test',
                  'time_params', 'time_obs', 'time_valid', 'time_test', 'datetime')


def _split(x, y, obs_j):
    test_j = np.setdiff1d(np.arange(len(x)), obs_j)
    return obs_j, x[obs_j], y[obs_j], test_j, x[test_j], y[test_j]


def random_obs(x, y, p=0.2, s=1.0, include_min=False, plot=False, plot_independent=False):
    """random subsample of int(len(x) * s * p) of the first int(len(x) * s) points; the rest is the
    test set (data.py:129-160).  Draws from np.random like the reference (seed it for repeatability)."""
    x, y = np.asarray(x), np.asarray(y)
    n = int(len(x) * s)
    k = int(n * p)
    obs_j = np.sort(np.random.choice(n, k, replace=False))
    if include_min and k > 0:
        lowest = int(np.argmin(y))
        if lowest not in obs_j:
            obs_j[np.random.randint(k)] = lowest
            obs_j = np.sort(obs_j)
    return _split(x, y, obs_j)


def uniform_obs(x, y, p=0.2, s=1.0, include_min=False, plot=False, plot_independent=False):
    """evenly spaced subsample of the first int(len(x) * s) points (data.py:190-236)"""
    x, y = np.asarray(x), np.asarray(y)
    n = int(len(x) * s)
    k = max(int(n * p), 1)
    obs_j = np.unique(np.linspace(0, n - 1, k).astype(int))
    if include_min:
        obs_j = np.unique(np.append(obs_j, int(np.argmin(y))))
    return _split(x, y, obs_j)


class Experiment:
    def __init__(self, models=None, file=None, load=True):
        self.models = models
        self.file = file                      # persistence is out of scope: kept for signature compatibility
        self.data_x = self.data_y = self.data_p = None
        self.data_limit, self.data_min, self.data_method = 1, True, random_obs
        self.scores_mean = self.scores_median = self.scores_variance = self.scores_logpred = True
        self.find_MAP, self.selector, self.holdout, self.holdout_p = False, None, None, 0
        pd = _pandas()
        self.simulations_raw = pd.DataFrame(columns=list(SIM_COLUMNS))
        self.results_raw = pd.DataFrame(columns=list(RESULT_COLUMNS))

    # ---- configuration (selection.py:145-190)
    def data(self, x, y, p, limit=1.0, method='random', include_min=False):
        self.data_x, self.data_y, self.data_p = np.asarray(x), np.asarray(y), p
        self.data_limit, self.data_min = limit, include_min
        self.data_method = {'random': random_obs, 'uniform': uniform_obs}[method]

    def scores(self, logpred=True, mean=True, median=False, variance=False):
        self.scores_mean, self.scores_median = mean, median
        self.scores_variance, self.scores_logpred = variance, logpred

    def model_selection(self, find_MAP=False, selector=None, holdout=None, holdout_p=0, **ignored):
        """`selector(sp) -> params` stands in for the reference's find_MAP / MCMC starts (optimisers are
        out of scope); with neither, every model runs at its default parameters."""
        if find_MAP and selector is None:
            raise NotImplementedError('find_MAP is outside the hot-path scope: pass selector=callable(sp) -> params')
        self.find_MAP, self.selector = bool(find_MAP), selector
        self.holdout, self.holdout_p = holdout, holdout_p

    # ---- one simulation
    def new_data(self, plot=False):
        """observations / optional hold-out / test indices and arrays (selection.py:158-169)"""
        obs_j, x_obs, y_obs, test_j, x_test, y_test = self.data_method(
            x=self.data_x, y=self.data_y, p=self.data_p, s=self.data_limit, include_min=self.data_min, plot=False)
        valid_j = x_valid = y_valid = None
        if self.holdout_p > 0:
            held, _, _, kept, _, _ = self.data_method(x=obs_j, y=obs_j, p=self.holdout_p, include_min=False, plot=False)
            valid_j, obs_j = obs_j[held], obs_j[kept]
            x_obs, y_obs = self.data_x[obs_j], self.data_y[obs_j]
            x_valid, y_valid = self.data_x[valid_j], self.data_y[valid_j]
        return obs_j, x_obs, y_obs, valid_j, x_valid, y_valid, test_j, x_test, y_test

    def calc_scores(self, sp, params):
        return sp.scores(params, logpred=self.scores_logpred, bias=self.scores_mean, median=self.scores_median,
                         variance=self.scores_variance)

    def select_model(self, sp, x_valid=None, y_valid=None):
        """(selected, start, params)"""
        if self.selector is not None:
            start = sp.params_default
            return getattr(self.selector, '__name__', 'selector'), start, self.selector(sp)
        return None, None, sp.params_default

    def add_simulation(self, index, obs, valid, test):
        self.simulations_raw.loc[index] = {'obs': obs, 'valid': valid, 'test': test, 'datetime': str(datetime.now())}

    def add_result(self, **row):
        row['datetime'] = str(datetime.now())
        self.results_raw.loc[len(self.results_raw)] = row

    def run(self, n_simulations=1, repeat=(), plot=False):
        """n_simulations new splits plus the stored splits listed in `repeat`; every model is scored on the
        observations, the hold-out (if any) and -- re-conditioned on observations + hold-out -- the test
        points, each stage timed (selection.py:237-292)"""
        first = len(self.simulations_raw)
        todo = list(range(first, first + n_simulations)) + (list(range(repeat)) if isinstance(repeat, int) else list(repeat))
        for n_sim in todo:
            if n_sim in self.simulations_raw.index:
                sim = self.simulations_raw.loc[n_sim]
                obs_j, valid_j, test_j = sim['obs'], sim['valid'], sim['test']
            else:
                obs_j, _, _, valid_j, _, _, test_j, _, _ = self.new_data()
                self.add_simulation(n_sim, obs_j, valid_j, test_j)
            part = lambda j: (None, None) if j is None else (self.data_x[j], self.data_y[j])
            (x_obs, y_obs), (x_valid, y_valid), (x_test, y_test) = part(obs_j), part(valid_j), part(test_j)
            for sp in self.models:
                clock = [time.time()]

                def lap():
                    clock.append(time.time())
                    return clock[-1] - clock[-2]
                sp.observed(x_obs, y_obs)
                selected, start, params = self.select_model(sp, x_valid, y_valid)
                time_params = lap()
                sp.set_params(params)
                sp.set_space(x_obs, y_obs, obs_j)              # space, hidden truth, order
                scores_obs = self.calc_scores(sp, params)
                time_obs = lap()
                scores_valid = {}
                if valid_j is not None:
                    sp.set_space(x_valid, y_valid, valid_j)
                    scores_valid = self.calc_scores(sp, params)
                time_valid = lap()
                if valid_j is not None:
                    sp.observed(np.concatenate([x_obs, x_valid]), np.concatenate([y_obs, y_valid]))
                sp.set_space(x_test, y_test, test_j)
                scores_test = self.calc_scores(sp, params)
                time_test = lap()
                self.add_result(n_sim=n_sim, model=sp.name, selected=selected, start=start, params=params,
                                scores_obs=scores_obs, scores_valid=scores_valid, scores_test=scores_test,
                                time_params=time_params, time_obs=time_obs, time_valid=time_valid, time_test=time_test)

    # ---- reporting (selection.py:297-329)
    def describe(self):
        return {k: v for k, v in self.__dict__.items() if k not in ('results_raw', 'simulations_raw')}

    def simulations(self):
        return self.simulations_raw

    def results(self, model=None, scores_columns=True, params_columns=False, raw=False, like=None):
        """one row per (simulation, model): timings plus `obs_*`, `valid_*`, `test_*` score columns"""
        if raw or self.results_raw.empty:
            return self.results_raw
        wanted = None if model is None else (model if isinstance(model, list) else [model])
        rows = []
        for _, r in self.results_raw.iterrows():
            if wanted is not None and r.model not in wanted:
                continue
            row = {k: r[k] for k in ('n_sim', 'model', 'selected', 'time_params', 'time_obs', 'time_valid', 'time_test',
                                     'start', 'datetime')}
            for prefix, sc in (('obs', r.scores_obs), ('valid', r.scores_valid), ('test', r.scores_test)):
                if scores_columns:
                    row.update({prefix + k: v for k, v in sc.items()})
                else:
                    row[prefix] = sc
            if params_columns:
                row.update({'p_' + k: v for k, v in r.params.items()})
            else:
                row['params'] = r.params
            rows.append(row)
        df = _pandas().DataFrame(rows)
        if like is not None:
            df = df[[c for c in df.columns if like in c or c in ('n_sim', 'model')]]
        return df
