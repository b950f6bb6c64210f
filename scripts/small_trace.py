"""Kernel timeline of one small evaluation (development aid): N = 125, d = 8 WGP logp, 20 calls."""
import sys
import numpy as np
sys.path.insert(0, '.')
import g3py_amd as g3
rng = np.random.default_rng(0)
N, d = 125, 8
X = rng.uniform(0, 1, (N, d))
y = np.exp(0.3 * np.sin(X.sum(1)) + 0.05 * rng.standard_normal(N)) + 0.5
gp = g3.WGP(space=X, location=g3.Bias(), kernel=g3.SE(X), mapping=g3.BoxCoxLinear(), dtype=np.float64)
gp.observed(X, y)
a = gp.active.dict_to_array(gp.params)
for i in range(20):
    gp.logp(a + 1e-3 * rng.standard_normal(len(a)), array=True)
g3.Device.close_all()
