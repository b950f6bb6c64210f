"""GPU test of the multi-rank driver with the real HIP tile operations: world_size 1 and 2
(two ranks sharing cuda:0, gloo transport -- RCCL refuses two ranks on one device; the driver
only uses broadcast and all_reduce, which both backends provide)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dist_helpers import worker, synth, chain_worker, native_worker  # noqa: E402
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


def test_hip_one_rank_through_rccl(tmp_path, monkeypatch):
    """G3_DIST_COLLECTIVES=1, backend nccl, world 1: every diagonal-factor broadcast (on its own communicator),
    every panel all-gather (all_gather_into_tensor, asynchronous work handles waited on from two streams) and the
    closing all-reduces really pass through ProcessGroupNCCL = RCCL instead of being short-circuited -- the
    stream semantics of the 8-GPU run, exercised on the one GPU a development box has"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    monkeypatch.setenv('G3_DIST_COLLECTIVES', '1')
    N, d, M, nb = 3000, 4, 200, 256
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(worker, args=(1, _free_port(), N, d, M, nb, 'nccl', True, spec_f, 0.1, out), nprocs=1, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-10 * abs(ref)
    np.testing.assert_allclose(r['mean'], gp.mean(Xs, X, y), atol=1e-8)
    np.testing.assert_allclose(r['var'], gp.variance(Xs, X, y), atol=1e-8)
    assert int(r['comm_calls']) >= 2 * (N // nb)         # one broadcast + one all-gather per row block + the all-reduces


@pytest.mark.parametrize('world', [1, 2])
def test_hip_staircase_longer_than_one_launch(tmp_path, monkeypatch, world):
    """more row blocks than one staircase launch may describe (N / nb > 160; here the limit is
    lowered to 3): the update is cut into row and column chunks, same result"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    monkeypatch.setenv('G3_STAIR_MAX', '3')
    N, d, M, nb = 2300, 4, 300, 128
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(worker, args=(world, _free_port(), N, d, M, nb, 'gloo', True, spec_f, 0.1, out, False, 5), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-9 * abs(ref)
    np.testing.assert_allclose(r['var'], gp.variance(Xs, X, y), atol=1e-8)
    Z = np.random.default_rng(5).standard_normal((M, 5))
    np.testing.assert_allclose(r['draws'], gp.sampler(Xs, X, y, rand=Z), atol=1e-7)


@pytest.mark.parametrize('world', [1, 2])
def test_hip_exhausted_jitter_falls_back(tmp_path, world):
    """indefinite matrix: the distributed driver installs the reference's 1e-10 * I fallback
    (tensors.py:215-222) on every rank instead of raising; same logp as the oracle"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    N, d, M, nb = 500, 1, 20, 256
    spec_f = ('SIN', 1.0, np.full(d, 0.37), np.full(d, 40.0), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(worker, args=(world, _free_port(), N, d, M, nb, 'gloo', True, spec_f, None, out), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    assert bool(r['fallback']) and int(r['tries']) == 20
    ref = orc.GP(spec_f, None).logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-8 * abs(ref)


@pytest.mark.parametrize('world,N,nb,M,dtype,tol', [(1, 1500, 512, 300, 'f64', 1e-7), (2, 1400, 256, 300, 'f64', 1e-7),
                                                    (2, 1400, 256, 300, 'f32', 2e-3)])
def test_hip_posterior_draws(tmp_path, world, N, nb, M, dtype, tol):
    """BASELINE config 5's extras through the HIP tile operations: posterior covariance, its robust
    Cholesky and loc + L Z draws (gaussian.py:75-97) against the oracle's sampler with the same normals"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    d, S = 4, 8
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(worker, args=(world, _free_port(), N, d, M, nb, 'gloo', True, spec_f, 0.1, out, False, S, dtype),
             nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    Z = np.random.default_rng(5).standard_normal((M, S))
    ref = orc.GP(spec_f, 0.1).sampler(Xs, X, y, rand=Z)
    np.testing.assert_allclose(r['draws'], ref, atol=tol)


@pytest.mark.parametrize('world', [1, 2, 3])
def test_logp_chain_sharded_over_replicas(tmp_path, world):
    """chain rows dealt to the ranks as replicas: equal to one-at-a-time logp"""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'res.npz')
    mp.spawn(chain_worker, args=(world, _free_port(), 300, 3, 11, out), nprocs=world, join=True)
    r = np.load(out)
    np.testing.assert_allclose(r['got'], r['ref'], rtol=1e-11)


# ---- the driver inside libg3hip (g3_dist_*, g3py_amd/csrc/g3_dist.hip)
@pytest.mark.parametrize('world,N,nb,M', [(1, 1500, 512, 50), (2, 1500, 512, 50), (2, 2048, 256, 300), (3, 2300, 128, 130),
                                          (3, 256, 128, 10), (4, 1100, 128, 130), (5, 640, 128, 129)])   # up to five ranks on the one GPU
def test_native_driver_matches_oracle(tmp_path, world, N, nb, M):
    """the C++ per-panel loop (three streams, diagonal-factor broadcast + panel all-gather + staircase updates) with
    `world` ranks sharing cuda:0: collectives served through the callback transport over gloo"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    d = 4
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, 0.1, out), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-10 * abs(ref)
    np.testing.assert_allclose(r['mean'], gp.mean(Xs, X, y), atol=1e-8)
    np.testing.assert_allclose(r['var'], gp.variance(Xs, X, y), atol=1e-8)


def test_native_driver_one_rank_through_rccl(tmp_path):
    """the product transport: the library dlopens librccl, creates its two communicators from ids made by rank 0 and
    issues ncclBroadcast / ncclAllGather / ncclAllReduce on its own streams (world 1: all a one-GPU box allows)"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    N, d, M, nb = 3000, 4, 200, 256
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(1, _free_port(), N, d, M, nb, 'rccl', spec_f, 0.1, out, False, 6), nprocs=1, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-10 * abs(ref)
    np.testing.assert_allclose(r['mean'], gp.mean(Xs, X, y), atol=1e-8)
    np.testing.assert_allclose(r['var'], gp.variance(Xs, X, y), atol=1e-8)
    Z = np.random.default_rng(5).standard_normal((M, 6))
    np.testing.assert_allclose(r['draws'], gp.sampler(Xs, X, y, rand=Z), atol=1e-7)
    assert int(r['comm_calls']) >= 2 * (N // nb)


@pytest.mark.parametrize('world,dtype,tol', [(1, 'f64', 1e-7), (2, 'f64', 1e-7), (3, 'f32', 2e-3)])
def test_native_driver_posterior_draws(tmp_path, world, dtype, tol):
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    N, d, M, nb, S = 1400, 4, 300, 256, 8
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, 0.1, out, False, S, dtype),
             nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    Z = np.random.default_rng(5).standard_normal((M, S))
    ref = orc.GP(spec_f, 0.1).sampler(Xs, X, y, rand=Z)
    np.testing.assert_allclose(r['draws'], ref, atol=tol)


@pytest.mark.parametrize('world', [1, 2])
def test_native_driver_jitter_and_fallback(tmp_path, world):
    """CholeskyRobust's schedule replicated on every rank (tensors.py:203-213) and, when its 20 steps are exhausted,
    the 1e-10 * I fallback (tensors.py:215-222) -- through the C++ driver"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    # singular covariance (duplicated inputs, no noise): the jitter path
    N, d, M, nb = 600, 2, 20, 128
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res1.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, None, out, True), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    X[1::2] = X[0::2][:len(X[1::2])]
    y = np.sin(X.sum(1) / np.sqrt(d))
    ref = orc.GP(spec_f, None).logp(X, y)
    assert int(r['tries']) >= 1 and not bool(r['fallback'])
    assert abs(float(r['logp']) - ref) <= 1e-8 * abs(ref)
    # indefinite matrix: fallback
    N, d, M, nb = 500, 1, 20, 256
    spec_f = ('SIN', 1.0, np.full(d, 0.37), np.full(d, 40.0), None)
    out = str(tmp_path / 'res2.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, None, out), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    assert bool(r['fallback']) and int(r['tries']) == 20
    ref = orc.GP(spec_f, None).logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-8 * abs(ref)


def test_native_driver_staircase_longer_than_one_launch(tmp_path, monkeypatch):
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    monkeypatch.setenv('G3_STAIR_MAX', '3')
    world, N, d, M, nb = 2, 2300, 4, 300, 128
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, 0.1, out, False, 5), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-9 * abs(ref)
    Z = np.random.default_rng(5).standard_normal((M, 5))
    np.testing.assert_allclose(r['draws'], gp.sampler(Xs, X, y, rand=Z), atol=1e-7)


@pytest.mark.parametrize('world,warped', [(1, False), (2, False), (3, True)])
def test_public_api_on_several_ranks(tmp_path, world, warped):
    """GaussianProcess.distribute(): the user API itself on `world` ranks (SPMD, callback transport on the one GPU):
    logp, mean, variance, std, median, quantiles and logpredictive equal the one-GPU process's values"""
    import torch.multiprocessing as mp
    import g3py_amd as g3
    from dist_helpers import api_worker
    N, d, M = 900, 3, 150
    out = str(tmp_path / 'res.npz')
    mp.spawn(api_worker, args=(world, _free_port(), N, d, M, 'callbacks', out, warped), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    if warped:
        y = y - y.min() + 1.0
        gp = g3.WarpedGaussianProcess(space=Xs, location=g3.Bias(), kernel=g3.SE(X), mapping=g3.BoxCoxLinear())
    else:
        gp = g3.GaussianProcess(space=Xs, location=g3.Bias(), kernel=g3.MAT52(X) + g3.COS(X))
    gp.observed(X, y)
    params = dict(gp.params)
    for k in params:
        if k.endswith('_var_log_') and 'Noise' not in k:
            params[k] = np.log(1.1)
        elif k.endswith('_rate_log_'):
            params[k] = np.log(np.full(d, 0.9))
        elif 'Noise' in k:
            params[k] = np.log(0.1)
        elif k.endswith('_freq_log_'):
            params[k] = np.log(np.full(d, 0.2))
    lp = gp.logp(params)
    pr = gp.predict(params, mean=True, var=True, std=True, median=True, quantiles=True)
    assert abs(float(r['logp']) - lp) <= 1e-10 * abs(lp) and float(r['logp2']) == float(r['logp'])
    for key, want in (('mean', pr.mean), ('var', pr.variance), ('std', pr.std), ('median', pr.median),
                      ('qu', pr.quantile_up), ('qd', pr.quantile_down)):
        np.testing.assert_allclose(r[key], want, atol=1e-8, err_msg=key)
    lpred = gp.logpredictive(params, vector=np.asarray(pr.median) + 0.01)
    assert abs(float(r['lpred']) - lpred) <= 1e-8 * abs(lpred)
    np.testing.assert_allclose(r['other'], gp.mean(params, space=Xs[: M // 2]), atol=1e-8)
    Z = np.random.default_rng(100).standard_normal((M, 4))
    np.testing.assert_allclose(r['smp'], gp.sampler(params, samples=4, rand=Z), atol=1e-7)     # rank 0's normals, same draws
