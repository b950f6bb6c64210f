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
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, 0.1, out, False, 0, 'f64', True),
             nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-10 * abs(ref)
    np.testing.assert_allclose(r['mean'], gp.mean(Xs, X, y), atol=1e-8)
    np.testing.assert_allclose(r['var'], gp.variance(Xs, X, y), atol=1e-8)
    # the same ranks then switched to gradient mode (g3_dist_set_grad + g3_dist_gp_dlogp): parameter sums and alpha
    _check_gradient(r, orc.with_noise(spec_f, 0.1), X, y, d)


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
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, None, out, False, 0, 'f64', True),
             nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    assert bool(r['fallback']) and int(r['tries']) == 20
    ref = orc.GP(spec_f, None).logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-8 * abs(ref)
    # gradient mode on the fallback factor 1e-10 * I: L^-T = 1e10 * I rides in the identity rows; the sums equal what the
    # one-GPU g3_gp_dlogp makes of the same fallback factor
    import g3py_amd as g3
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec
    dev = g3.Device(0)
    prog = compile_spec(spec_f, d)
    Np = _lib.roundup(N)
    Kd, W, a = dev.alloc(Np + 128, Np, np.float64), dev.alloc_inverses(Np, np.float64), dev.alloc(1, Np, np.float64)
    Xd = dev.upload(X)
    st = dev.gp_factor(prog, Xd, N, d, dev.upload(y), Kd, W, a)
    assert st['fallback']
    Y, Ki, al = dev.alloc(Np, Np, np.float64), dev.alloc(Np, Np, np.float64), dev.alloc(1, Np, np.float64)
    one = dev.gp_dlogp(prog, dev.grad_layout(prog), Xd, N, d, Kd, W, a, Y, Ki, al)
    np.testing.assert_allclose(r['slots'], one, rtol=1e-9)
    np.testing.assert_allclose(r['alpha'], dev.download(al, 1, N)[0], rtol=1e-9)


def test_native_driver_staircase_longer_than_one_launch(tmp_path, monkeypatch):
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    monkeypatch.setenv('G3_STAIR_MAX', '3')
    world, N, d, M, nb = 2, 2300, 4, 300, 128
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, 0.1, out, False, 5, 'f64', True),
             nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-9 * abs(ref)
    Z = np.random.default_rng(5).standard_normal((M, 5))
    np.testing.assert_allclose(r['draws'], gp.sampler(Xs, X, y, rand=Z), atol=1e-7)
    # the gradient stage's staircase launches (K^-1 row blocks, diagonal blocks cut at the diagonal) are chunked too
    alpha, ref_g = _grad_reference(orc.with_noise(spec_f, 0.1), X, y)
    np.testing.assert_allclose(r['alpha'], alpha, rtol=1e-7, atol=1e-8 * np.abs(alpha).max())
    from g3py_amd.device import compile_spec
    import g3py_amd as g3
    gmap = g3.Device(0).grad_layout(compile_spec(orc.with_noise(spec_f, 0.1), d))
    for (leaf, pname, k, want, scale) in ref_g:
        got = r['slots'][getattr(gmap, pname)[leaf] + (0 if k is None else k)]
        assert abs(got - want) < 1e-8 * scale, (leaf, pname, k, got, want)


def _grad_reference(spec_n, X, y):
    """slots of g3_gp_dlogp from the oracle's dK/dparam tensors: 1/2 sum_ij (alpha alpha^T - K^-1)_ij dK_ij"""
    from oracle import g3_oracle as orc
    K, grads = orc.kernel_cov_grads(spec_n, X)
    Kinv = np.linalg.inv(K)
    alpha = Kinv @ y
    G = np.outer(alpha, alpha) - Kinv
    return alpha, [(leaf, pname, k, 0.5 * np.sum(G * dK), 0.5 * np.sum(np.abs(G * dK)) + 1e-30) for (leaf, pname, k, dK) in grads]


def _check_gradient(r, spec_n, X, y, d, rtol=1e-8):
    from g3py_amd.device import compile_spec
    import g3py_amd as g3
    alpha, ref = _grad_reference(spec_n, X, y)
    assert abs(float(r['logp_grad']) - float(r['logp'])) <= 1e-11 * abs(float(r['logp']))
    np.testing.assert_allclose(r['alpha'], alpha, rtol=1e-7, atol=1e-8 * np.abs(alpha).max())
    gmap = g3.Device(0).grad_layout(compile_spec(spec_n, d))
    assert len(r['slots']) == gmap.nslots == len(ref)
    for (leaf, pname, k, want, scale) in ref:
        got = r['slots'][getattr(gmap, pname)[leaf] + (0 if k is None else k)]
        assert abs(got - want) < rtol * scale, (leaf, pname, k, got, want)


@pytest.mark.parametrize('world,N,nb,M,kern', [(2, 1100, 128, 130, 'm52cos'), (3, 700, 128, 0, 'rq')])   # SE, 1 - 5 ranks: in test_native_driver_matches_oracle
def test_native_driver_gradient(tmp_path, world, N, nb, M, kern):
    """g3_dist_set_grad + g3_dist_gp_dlogp: the identity rides through the factorisation as right-hand-side rows (the
    rank's rows of L^-T), K^-1 rows by gathered panels + staircase products, the gradient kernel over the rank's row
    blocks, one all-reduce -- parameter sums and alpha = K^-1 delta against the oracle (tensors.py:224-260 is what the
    reference differentiates through), ragged N, up to five ranks on the one GPU"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    from g3py_amd.device import compile_spec
    import g3py_amd as g3
    d = 3
    r_ = np.array([0.9, 1.1, 0.7])
    spec_f = {'se': ('SE', 1.2, r_, None), 'rq': ('RQ', 0.8, r_, 1.7, None),
              'm52cos': ('sum', ('MAT52', 1.1, r_, None), ('COS', 0.4, np.array([0.2, 0.15, 0.1]), None))}[kern]
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, max(M, 1), nb, 'callbacks', spec_f, 0.1, out, False, 0, 'f64', True),
             nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, max(M, 1), 77)
    _check_gradient(r, orc.with_noise(spec_f, 0.1), X, y, d)


def test_native_driver_gradient_one_rank_through_rccl(tmp_path):
    """gradient mode with the product transport: the panels of L^-T and the alpha pieces go through ncclAllGather, the
    parameter sums through ncclAllReduce (world 1: all a one-GPU box allows)"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    from g3py_amd.device import compile_spec
    import g3py_amd as g3
    N, d, M, nb = 1300, 3, 60, 256
    spec_f = ('MAT32', 1.2, np.array([0.9, 1.1, 0.7]), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(1, _free_port(), N, d, M, nb, 'rccl', spec_f, 0.1, out, False, 0, 'f64', True), nprocs=1, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    spec_n = orc.with_noise(spec_f, 0.1)
    alpha, ref = _grad_reference(spec_n, X, y)
    np.testing.assert_allclose(r['alpha'], alpha, rtol=1e-7, atol=1e-8 * np.abs(alpha).max())
    gmap = g3.Device(0).grad_layout(compile_spec(spec_n, d))
    for (leaf, pname, k, want, scale) in ref:
        got = r['slots'][getattr(gmap, pname)[leaf] + (0 if k is None else k)]
        assert abs(got - want) < 1e-8 * scale, (leaf, pname, k, got, want)


def test_native_driver_gradient_fp32(tmp_path):
    """the gradient stage in fp32 (config 5's arithmetic) on three ranks: alpha and the parameter sums to fp32 accuracy"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    from g3py_amd.device import compile_spec
    import g3py_amd as g3
    world, N, d, M, nb = 3, 600, 3, 30, 128
    spec_f = ('SE', 1.2, np.array([0.9, 1.1, 0.7]), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, 0.3, out, False, 0, 'f32', True),
             nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    spec_n = orc.with_noise(spec_f, 0.3)
    X32, y32 = X.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64)
    alpha, ref = _grad_reference(spec_n, X32, y32)
    np.testing.assert_allclose(r['alpha'], alpha, rtol=2e-3, atol=2e-3 * np.abs(alpha).max())
    gmap = g3.Device(0).grad_layout(compile_spec(spec_n, d))
    for (leaf, pname, k, want, scale) in ref:
        got = r['slots'][getattr(gmap, pname)[leaf] + (0 if k is None else k)]
        assert abs(got - want) < 5e-3 * scale, (leaf, pname, k, got, want)


def test_gram_grad_rows_add_up():
    """g3_gram_grad_rows over disjoint row ranges (ragged last range) sums to g3_gram_grad, fast path and interpreter"""
    import g3py_amd as g3
    from g3py_amd.device import compile_spec
    from oracle import g3_oracle as orc
    dev = g3.Device(0)
    N, d = 700, 4
    rng = np.random.default_rng(3)
    X = rng.uniform(0, 3, (N, d))
    A = rng.standard_normal((N, N))
    G = np.tril((A + A.T) / 2)
    alpha = rng.standard_normal(N)
    Xd, ad = dev.upload(X), dev.upload(alpha)
    for spec in (orc.with_noise(('MAT52', 1.3, np.full(d, 0.8), None), 0.2),
                 ('sum', ('prod', ('SE', 1.0, np.full(d, 0.7), None), ('SINC', 0.7, np.full(d, 0.2), None)), ('NOISE', 0.2))):
        prog = compile_spec(spec, d)
        gmap = dev.grad_layout(prog)
        full = dev.gram_grad(prog, gmap, Xd, N, d, dev.upload(G), ad)
        tot = np.zeros_like(full)
        for r0 in range(0, N, 256):
            nr = min(256, N - r0)
            tot += dev.gram_grad_rows(prog, gmap, Xd, N, d, r0, nr, dev.upload(np.ascontiguousarray(G[r0:r0 + nr])), ad)
        np.testing.assert_allclose(tot, full, rtol=1e-12, atol=1e-12 * np.abs(full).max())


@pytest.mark.parametrize('world,warped', [(1, False), (2, False), (3, True)])
def test_public_api_on_several_ranks(tmp_path, world, warped):
    """GaussianProcess.distribute(): the user API itself on `world` ranks (SPMD, callback transport on the one GPU):
    logp, mean, variance, std, median, quantiles, logpredictive and dlogp equal the one-GPU process's values"""
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
    rc = np.load(out.replace('.npz', '_cov.npz'))        # the full posterior covariance, f and noisy, and its Cholesky factor
    np.testing.assert_allclose(rc['cov'], np.asarray(gp.kernel(params)), atol=1e-8)
    np.testing.assert_allclose(rc['cov_n'], np.asarray(gp.kernel(params, noise=True)), atol=1e-8)
    np.testing.assert_allclose(rc['cov'], rc['cov'].T, atol=0)
    np.testing.assert_allclose(rc['chol'], np.asarray(gp.cholesky(params)), atol=1e-6)
    g1 = np.asarray(gp.dlogp(params))          # dlogp on the distributed process = the one-GPU gradient (kernel, location, warping hypers)
    np.testing.assert_allclose(r['grad'], g1, rtol=1e-7, atol=1e-8 * np.abs(g1).max())


def test_student_t_process_on_two_ranks(tmp_path):
    """logp, the scaled posterior variance and dlogp of a StudentTProcess on two ranks (callback transport on the one GPU)
    equal the same object evaluated on one GPU: the Student-t density's s = (nu + n) / (nu - 2 + beta) reaches the
    distributed gradient through alpha_scale"""
    import torch.multiprocessing as mp
    from dist_helpers import tp_worker
    out = str(tmp_path / 'res.npz')
    mp.spawn(tp_worker, args=(2, _free_port(), 700, 3, 40, out), nprocs=2, join=True)
    r = np.load(out)
    assert int(r['ok']) == 1 and np.all(np.isfinite(r['grad']))


@pytest.mark.parametrize('world', [1, 2, 3])
def test_native_driver_random_shapes(tmp_path, world):
    """seeded random shapes (ragged N around the block height, M around the 128-row chunks incl. M < world chunks, several
    block heights) through one process group per world size: logp, posterior mean / variance and the gradient against the
    oracle -- what the fixed cases cannot see: an off-by-one in the dealing of blocks, chunks or identity rows"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    from dist_helpers import native_multi_worker
    rng = np.random.default_rng(400 + world)
    cases = []
    for _ in range(4):
        nb = int(rng.choice([128, 256]))
        N = int(rng.choice([nb - 1, nb + 1, 2 * nb, 3 * nb + 17, 5 * nb - 3, 7 * nb + 1]))
        cases.append((N, int(rng.integers(1, 4)), int(rng.choice([1, 127, 128, 129, 260])), nb, int(rng.integers(1, 1000))))
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_multi_worker, args=(world, _free_port(), cases, out), nprocs=world, join=True)
    r = np.load(out)
    for ci, (N, d, M, nb, seed) in enumerate(cases):
        X, y, Xs = synth(N, d, M, seed)
        spec_f = ('MAT32', 1.1, np.linspace(0.7, 1.2, d), None)
        gp = orc.GP(spec_f, 0.2)
        ref = gp.logp(X, y)
        assert abs(float(r['logp%d' % ci]) - ref) <= 1e-10 * abs(ref), (cases[ci], float(r['logp%d' % ci]), ref)
        np.testing.assert_allclose(r['mean%d' % ci], gp.mean(Xs, X, y), atol=1e-8, err_msg=str(cases[ci]))
        prior = np.diag(orc.kernel_cov(spec_f, Xs))
        np.testing.assert_allclose(np.maximum(prior - r['ss%d' % ci], 0), gp.variance(Xs, X, y), atol=1e-8, err_msg=str(cases[ci]))
        rr = {'logp': r['logp%d' % ci], 'logp_grad': r['logp%d' % ci], 'alpha': r['alpha%d' % ci], 'slots': r['slots%d' % ci]}
        _check_gradient(rr, orc.with_noise(spec_f, 0.2), X, y, d)


@pytest.mark.parametrize('world,N,nb,M', [(3, 1500, 128, 200), (4, 2300, 256, 130)])
def test_replayed_rank_equals_the_rank_of_a_real_run(tmp_path, world, N, nb, M):
    """VERDICT r3 item 2.  The replay transport plays ONE rank of a P-rank evaluation alone on the GPU, every collective
    a device copy of the bytes the rank would receive out of a world-1 reference run.  What each replayed rank
    contributes to the closing all-reduce -- its share of the log-determinant, of a^T a, of the posterior means and sums
    of squares -- must be what the same rank contributed in a REAL P-rank run (callback transport over gloo, the ranks
    sharing this GPU), and the contributions must add up to the one-GPU logp."""
    import torch.multiprocessing as mp
    from dist_helpers import contrib_worker
    d = 3
    mp.spawn(contrib_worker, args=(world, _free_port(), N, d, M, nb, str(tmp_path)), nprocs=world, join=True)
    real = [np.load(str(tmp_path / ('rank%d.npz' % r))) for r in range(world)]
    import g3py_amd as g3
    from g3py_amd.distributed import NativeDistributedGP
    from oracle import g3_oracle as orc
    X, y, Xs = synth(N, d, M, 77)
    spec_f = ('SE', 1.0, np.ones(d), None)
    spec_n = orc.with_noise(spec_f, 0.1)
    dev = g3.Device.default()
    Xd, Xsd, yd = dev.upload(X), dev.upload(Xs), dev.upload(y)
    ref = NativeDistributedGP(dev, None, 0, 1, N, d, M, nb=nb, transport='callbacks', keep=True)
    lp1 = ref.step(spec_n, spec_f, Xd, Xsd, yd)
    tot = np.zeros(2 + 2 * M)
    for r in range(world):
        rp = NativeDistributedGP(dev, None, r, world, N, d, M, nb=nb, transport='replay', reference=ref)
        rp.step(spec_n, spec_f, Xd, Xsd, yd)
        mine = np.concatenate([[rp.last['logdet'], rp.last['quad']], rp.last['mean'], rp.last['ss']])
        want = real[r]['contrib']
        np.testing.assert_allclose(mine, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
        cs = rp.comm_stats()
        assert cs['allgather']['calls'] == -(-N // nb) - 1           # one panel all-gather per block column but the last
        tot += mine
        rp.close()
    lp = -0.5 * N * np.log(2 * np.pi) - 0.5 * tot[1] - tot[0]
    assert abs(lp - lp1) <= 1e-12 * abs(lp1) and abs(lp - float(real[0]['logp'])) <= 1e-12 * abs(lp1)
    np.testing.assert_allclose(tot[2:2 + M], real[0]['mean'], rtol=1e-11, atol=1e-12)
    ref_lp = orc.GP(spec_f, 0.1).logp(X, y)
    assert abs(lp - ref_lp) <= 1e-9 * abs(ref_lp)
    # a replay must refuse what it cannot replay
    with pytest.raises(Exception):
        NativeDistributedGP(dev, None, 0, 2, N + 128, d, M, nb=nb, transport='replay', reference=ref)
    ref.close()


@pytest.mark.parametrize('world,N,nb,M', [(2, 900, 128, 40), (3, 1300, 256, 140)])
def test_python_twin_issues_the_collectives_of_the_native_driver(tmp_path, world, N, nb, M):
    """the Python DistributedGP is the readable twin of the C++ schedule in g3_dist.hip -- the two must
    not drift.  On the same plan both drivers are asked for one evaluation and every collective they issue is logged
    (kind, payload bytes, root / reduction): per kind the two sequences are identical, rank by rank, and so is logp."""
    import json
    import torch.multiprocessing as mp
    from dist_helpers import twin_worker
    mp.spawn(twin_worker, args=(world, _free_port(), N, 3, M, nb, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        j = json.load(open(str(tmp_path / ('rank%d.json' % r))))
        assert abs(j['logp'][0] - j['logp'][1]) <= 1e-12 * abs(j['logp'][0])
        for kind in ('bcast', 'allgather', 'allreduce'):
            a = [tuple(e[1:]) for e in j['native'] if e[0] == kind]
            b = [tuple(e[1:]) for e in j['python'] if e[0] == kind]
            assert a == b, (r, kind, a[:6], b[:6], len(a), len(b))
        assert len([e for e in j['native'] if e[0] == 'allgather']) == -(-N // nb) - 1


@pytest.mark.parametrize('transport', ['callbacks', 'callbacks_async'])
@pytest.mark.parametrize('world,N,nb,M', [(2, 1500, 256, 50), (3, 2300, 128, 130), (4, 1100, 128, 130)])
def test_native_driver_on_both_callback_transports(tmp_path, monkeypatch, transport, world, N, nb, M):
    """the schedule of g3_dist.hip with collectives genuinely in flight.  'callbacks_async': two worker
    threads of the library serve the all-gathers and the broadcasts from two gloo groups while the host runs ahead and the
    three streams keep working (stream-ordered like RCCL calls); 'callbacks': every collective a blocking host call behind a
    stream synchronisation (the debugging aid).  Same numbers from both, and from the oracle (the rest of this file runs on
    the asynchronous one by default: dist_helpers.cb_transport)"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    monkeypatch.setenv('G3_TEST_TRANSPORT', transport)
    d = 4
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, 'callbacks', spec_f, 0.1, out, False, 4), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-10 * abs(ref)
    np.testing.assert_allclose(r['mean'], gp.mean(Xs, X, y), atol=1e-8)
    np.testing.assert_allclose(r['var'], gp.variance(Xs, X, y), atol=1e-8)
    Z = np.random.default_rng(5).standard_normal((M, 4))
    np.testing.assert_allclose(r['draws'], gp.sampler(Xs, X, y, rand=Z), atol=1e-7)


@pytest.mark.parametrize('world,transport', [(3, 'callbacks'), (3, 'callbacks_async'), (1, 'rccl')])
def test_native_driver_serial_collectives_knob(tmp_path, monkeypatch, world, transport):
    """G3_DIST_SERIAL_COLL=1 makes the diagonal-factor broadcast wait for the previous panel's all-gather, so
    the two communicators are never in flight together (the conservative schedule for the first multi-GPU runs): same
    numbers as the overlapped default"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    monkeypatch.setenv('G3_DIST_SERIAL_COLL', '1')
    if transport.startswith('callbacks'):
        monkeypatch.setenv('G3_TEST_TRANSPORT', transport)
        transport = 'callbacks'
    N, d, M, nb = 1100, 3, 70, 128
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(native_worker, args=(world, _free_port(), N, d, M, nb, transport, spec_f, 0.1, out), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-10 * abs(ref)
    np.testing.assert_allclose(r['mean'], gp.mean(Xs, X, y), atol=1e-9)


def test_eight_rank_schedule_at_the_full_headline_size_reproduces_the_oracle_pin():
    """Multi-rank parity at FULL size inside the driver-run suite ("what's weak"): BASELINE config 4
    (N=32768, d=4, M=1024, fp64) as EIGHT ranks with 1024-row blocks, every rank replayed alone on this GPU (replay
    transport: the schedule, the block dealing, the staircase launches and the gather tables of the 8-rank run; the bytes
    a rank would receive come from a world-1 reference pass).  The ranks' contributions must add up to the CPU oracle's
    full-size pin (tests/golden/fullsize.json) at 1e-8, posterior means / variances included."""
    import json
    import os
    import g3py_amd as g3
    from g3py_amd.distributed import NativeDistributedGP
    from oracle import g3_oracle as orc
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'fullsize.json')))['c4']
    N, d, M, P, nb = gold['N'], gold['d'], gold['M'], 8, 1024
    rng = np.random.Generator(np.random.PCG64(gold['seed']))
    Lbox = N ** (1.0 / d)
    X = rng.uniform(0, Lbox, (N, d)); Xs = rng.uniform(0, Lbox, (M, d))
    y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(N)
    spec_f = ('SE', 1.0, np.ones(d), None)
    spec_n = orc.with_noise(spec_f, gold['noise'])
    dev = g3.Device(0)
    Xd, Xsd, yd = dev.upload(X), dev.upload(Xs), dev.upload(y)
    ref = NativeDistributedGP(dev, None, 0, 1, N, d, M, nb=nb, transport='callbacks', keep=True)
    ref.step(spec_n, spec_f, Xd, Xsd, yd)
    tot = np.zeros(2 + 2 * M)
    recv = []
    for r in range(P):
        rp = NativeDistributedGP(dev, None, r, P, N, d, M, nb=nb, transport='replay', reference=ref)
        rp.step(spec_n, spec_f, Xd, Xsd, yd)
        tot += np.concatenate([[rp.last['logdet'], rp.last['quad']], rp.last['mean'], rp.last['ss']])
        cs = rp.comm_stats()
        recv.append(cs['bcast']['bytes'] + cs['allgather']['bytes'] / 2)
        rp.close()
    ref.close()
    lp = -0.5 * N * np.log(2 * np.pi) - 0.5 * tot[1] - tot[0]
    assert abs(lp - gold['logp']) <= 1e-8 * abs(gold['logp']), (lp, gold['logp'])
    assert abs(tot[0] - gold['logdet']) <= 1e-8 * abs(gold['logdet'])
    nq = len(gold['mean'])
    np.testing.assert_allclose(tot[2:2 + nq], gold['mean'], rtol=0, atol=1e-8)
    np.testing.assert_allclose(np.maximum(1.0 - tot[2 + M:2 + M + nq], 0), gold['variance'], rtol=0, atol=1e-8)
    # what a rank would receive: (P - 1) / P of the lower triangle in panels + the diagonal factors it does not own
    assert 4.0e9 < min(recv) and max(recv) < 5.5e9, recv
    dev.close()


def test_posterior_covariance_refuses_other_prediction_points():
    """the cross-solve rows V = K(Xs, X) L^-T live in the driver from the last evaluation: g3_dist_posterior_cov / _draws
    at any OTHER Xs would silently combine K(Xs', Xs') with the V of Xs -- the driver remembers the points it evaluated
    and refuses; at the same points the covariance equals the oracle's"""
    import g3py_amd as g3
    from g3py_amd.distributed import NativeDistributedGP
    from g3py_amd import _lib
    from oracle import g3_oracle as orc
    N, d, M = 500, 2, 70
    X, y, Xs = synth(N, d, M, 91)
    spec_f = ('SE', 1.0, np.ones(d), None)
    spec_n = orc.with_noise(spec_f, 0.1)
    dev = g3.Device.default()
    Xd, Xsd, yd = dev.upload(X), dev.upload(Xs), dev.upload(y)
    gp = NativeDistributedGP(dev, None, 0, 1, N, d, M, nb=256, transport='callbacks')
    gp.step(spec_n, spec_f, Xd, Xsd, yd)
    Mp = _lib.roundup(M)
    out = dev.alloc(Mp, Mp, np.float64)
    gp.posterior_cov(spec_f, Xsd, out)
    cov = dev.download(out, M, M)
    Kxx = orc.kernel_cov(spec_f, X, None, np.float64) + 0.1 * np.eye(N)
    Ksx = orc.kernel_cov(spec_f, Xs, X, np.float64)
    want = orc.kernel_cov(spec_f, Xs, None, np.float64) - Ksx @ np.linalg.solve(Kxx, Ksx.T)
    np.testing.assert_allclose(cov, want, atol=1e-9)
    Xs2 = Xs.copy()
    Xs2[3, 0] += 1e-3
    with pytest.raises(g3.G3Error, match='other prediction points'):
        gp.posterior_cov(spec_f, dev.upload(Xs2), out)
    with pytest.raises(g3.G3Error, match='other prediction points'):
        gp.draws(spec_f, dev.upload(Xs2), np.zeros(M), np.zeros((M, 2)))
    gp.step(spec_n, spec_f, Xd, dev.upload(Xs2), yd)            # evaluated there: accepted now
    gp.posterior_cov(spec_f, dev.upload(Xs2), out)
    gp.close()
