"""GPU test of the multi-rank driver with the real HIP tile operations: world_size 1 and 2
(two ranks sharing cuda:0, gloo transport -- RCCL refuses two ranks on one device; the driver
only uses broadcast and all_reduce, which both backends provide)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dist_helpers import worker, synth  # noqa: E402
from test_distributed_cpu import _free_port  # noqa: E402


@pytest.mark.parametrize('world,N,nb', [(1, 1500, 512), (2, 1500, 512), (2, 2048, 256)])
def test_hip_block_cyclic_matches_oracle(tmp_path, world, N, nb):
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    d, M = 4, 50
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(worker, args=(world, _free_port(), N, d, M, nb, 'gloo', True, spec_f, 0.1, out), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-8 * abs(ref)
    np.testing.assert_allclose(r['mean'], gp.mean(Xs, X, y), atol=1e-8)
    np.testing.assert_allclose(r['var'], gp.variance(Xs, X, y), atol=1e-8)
