import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import g3py_amd as g3
rng = np.random.default_rng(0)
N, d, B = 128, 3, 4096
X = rng.uniform(0, N ** (1 / d), (N, d))
y = np.sin(X.sum(1) / 2) + 0.1 * rng.standard_normal(N)
gp = g3.GaussianProcess(space=X, location=g3.Bias(), kernel=g3.MAT52(X))
gp.observed(X, y)
a0 = gp.active.dict_to_array(gp.params_default)
chain = a0 + 0.15 * rng.standard_normal((B, len(a0)))
for _ in range(3):
    gp.dlogp_chain(chain)
