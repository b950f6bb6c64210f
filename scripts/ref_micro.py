"""The reference's only recorded timings (notebooks/04-Random-Fields-Fixed.ipynb
:3505-3643): WGP, SE + noise, BoxCoxLinear, N = 125 observations, d = 8, on an unknown CPU in
float32 -- logp 878 us / call, fixed_logp over 10 chain rows 10.1 ms, fixed_dlogp 18.6 ms,
dict_to_array 67.1 us.  Same model shape here on synthetic data (the abalone subsample is not
available), through the same public methods."""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import g3py_amd as g3


def timeit(f, n):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n


def main():
    rng = np.random.default_rng(0)
    N, d = 125, 8
    X = rng.uniform(0, 1, (N, d))
    y = np.exp(0.3 * np.sin(X.sum(1)) + 0.05 * rng.standard_normal(N)) + 0.5
    for dtype in (np.float32, np.float64):
        gp = g3.WGP(space=X, location=g3.Bias(), kernel=g3.SE(X), mapping=g3.BoxCoxLinear(), dtype=dtype)
        gp.observed(X, y)
        p = gp.params
        a = gp.active.dict_to_array(p)
        chain = a + 0.01 * rng.standard_normal((10, len(a)))
        gp.active.fix_vars(chain, [gp.name + '_Noise_var_log_'])
        sp = gp.active.sampling_params(a)
        print(np.dtype(dtype).name, 'logp %.1f' % gp.logp(p))
        # a different parameter vector on every call: the process caches the factor of the last one
        arrs = [a + 1e-3 * rng.standard_normal(len(a)) for _ in range(64)]
        dicts = [gp.active.array_to_dict(v) for v in arrs]
        it = {'i': 0}

        def nxt(seq):
            it['i'] += 1
            return seq[it['i'] % len(seq)]
        print('  logp(params)            %8.1f us   (reference: 878 us)' % (timeit(lambda: gp.logp(nxt(dicts)), 200) * 1e6))
        print('  logp(array)             %8.1f us' % (timeit(lambda: gp.logp(nxt(arrs), array=True), 200) * 1e6))
        print('  dlogp(params)           %8.1f us' % (timeit(lambda: gp.dlogp(nxt(dicts)), 100) * 1e6))
        sps = [gp.active.sampling_params(v) for v in arrs]
        print('  fixed_logp (10 rows)    %8.1f us   (reference: 10100 us)' % (timeit(lambda: gp.fixed_logp(nxt(sps)), 100) * 1e6))
        print('  fixed_dlogp (10 rows)   %8.1f us   (reference: 18600 us)' % (timeit(lambda: gp.fixed_dlogp(nxt(sps)), 20) * 1e6))
        print('  dict_to_array           %8.1f us   (reference: 67.1 us)' % (timeit(lambda: gp.active.dict_to_array(p), 2000) * 1e6))


if __name__ == '__main__':
    main()
