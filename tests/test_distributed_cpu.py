"""world_size-2 (and 3, 4) gloo runs of the multi-rank driver on CPU: row-block ownership, the
diagonal-factor broadcast with look-ahead, the panel all-gather and its re-ordering into global
block order, the right-hand-side chunks, the closing all-reduce.  The tile arithmetic is a NumPy test double
(tests/dist_helpers.py); the expected values come from the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dist_helpers import worker, synth  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize('world,N,nb,M', [(2, 700, 256, 20), (3, 520, 128, 300), (2, 300, 512, 20), (4, 1100, 128, 130),
                                          (1, 400, 128, 20), (8, 2100, 128, 20),
                                          (3, 256, 128, 10), (5, 640, 128, 129), (2, 1280, 128, 257)])   # fewer blocks than ranks; one block per rank; a long pipeline
def test_block_cyclic_driver_matches_oracle(tmp_path, world, N, nb, M):
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    d = 3
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(worker, args=(world, _free_port(), N, d, M, nb, 'gloo', False, spec_f, 0.1, out), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    gp = orc.GP(spec_f, 0.1)
    ref = gp.logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-9 * abs(ref)
    np.testing.assert_allclose(r['mean'], gp.mean(Xs, X, y), atol=1e-9)
    np.testing.assert_allclose(r['var'], gp.variance(Xs, X, y), atol=1e-9)


def test_distributed_jitter_schedule(tmp_path):
    """a singular covariance (duplicated inputs, no noise): every rank follows the same jitter
    schedule (tensors.py:203-213) and the result equals the oracle's robust path"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    world, N, d, M, nb = 2, 300, 2, 10, 128
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(worker, args=(world, _free_port(), N, d, M, nb, 'gloo', False, spec_f, None, out, True), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    X[1::2] = X[0::2][:len(X[1::2])]
    y = np.sin(X.sum(1) / np.sqrt(d))
    ref = orc.GP(spec_f, None).logp(X, y)
    assert int(r['tries']) >= 1
    assert abs(float(r['logp']) - ref) <= 1e-8 * abs(ref)


def test_distributed_exhausted_jitter_falls_back(tmp_path):
    """an indefinite 'covariance' (SIN kernel with a large positive exponent, kernels.py:472, no noise):
    the 20-step schedule is exhausted and every rank installs the reference's 1e-10 * I fallback
    (tensors.py:215-222) instead of raising; logp equals the oracle's"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    world, N, d, M, nb = 2, 200, 1, 10, 128
    spec_f = ('SIN', 1.0, np.full(d, 0.37), np.full(d, 40.0), None)
    X, y, Xs = synth(N, d, M, 77)
    K = orc.tt_to_cov(orc.tt_to_num(orc.kernel_cov(spec_f, X)))
    _, tries, fb = orc.cholesky_robust(K, return_info=True)
    assert fb, 'test premise: the oracle itself must reach the fallback'
    out = str(tmp_path / 'res.npz')
    mp.spawn(worker, args=(world, _free_port(), N, d, M, nb, 'gloo', False, spec_f, None, out), nprocs=world, join=True)
    r = np.load(out)
    assert bool(r['fallback']) and int(r['tries']) == 20
    ref = orc.GP(spec_f, None).logp(X, y)
    assert abs(float(r['logp']) - ref) <= 1e-9 * abs(ref)


@pytest.mark.parametrize('world,N,nb,M', [(2, 600, 128, 200), (3, 500, 128, 300), (1, 300, 128, 130)])
def test_distributed_posterior_draws(tmp_path, world, N, nb, M):
    """config-5 extras on the multi-rank driver: posterior covariance, its Cholesky and L Z draws
    (gaussian.py:75-97) equal the oracle's sampler with the same normals"""
    import torch.multiprocessing as mp
    from oracle import g3_oracle as orc
    d, S = 3, 6
    spec_f = ('SE', 1.0, np.ones(d), None)
    out = str(tmp_path / 'res.npz')
    mp.spawn(worker, args=(world, _free_port(), N, d, M, nb, 'gloo', False, spec_f, 0.1, out, False, S), nprocs=world, join=True)
    r = np.load(out)
    X, y, Xs = synth(N, d, M, 77)
    Z = np.random.default_rng(5).standard_normal((M, S))
    gp = orc.GP(spec_f, 0.1)
    ref = gp.sampler(Xs, X, y, rand=Z)
    np.testing.assert_allclose(r['draws'], ref, atol=1e-7)


def test_block_ranges_and_ownership():
    from g3py_amd.distributed import block_ranges
    assert block_ranges(640, 256) == [(0, 256), (256, 256), (512, 128)]
    assert block_ranges(128, 512) == [(0, 128)]


@pytest.mark.parametrize('limit', [1, 2, 3, 160])
def test_stair_chunks_cover_the_staircase_once(limit):
    """a staircase with more row blocks than one launch can describe (N / nb > 160: ADVICE r2) is cut into
    row / column chunks; applying the chunks equals applying the whole staircase, with and without a block table"""
    from g3py_amd.distributed import stair_chunks
    rng = np.random.default_rng(11)
    nbr, k = 4, 3
    seg_rows = [4, 4, 8, 4, 0, 4, 12]
    seg_cols = [4, 8, 12, 16, 16, 24, 28]
    nblk = max(seg_cols) // nbr
    perm = list(rng.permutation(nblk + 2)[:nblk])
    A = rng.standard_normal((sum(seg_rows), k))
    Bl = rng.standard_normal((nblk * nbr, k))                      # logical order
    Bp = np.zeros(((nblk + 2) * nbr, k))
    for s_, p_ in enumerate(perm):
        Bp[p_ * nbr:(p_ + 1) * nbr] = Bl[s_ * nbr:(s_ + 1) * nbr]
    ref = np.zeros((sum(seg_rows), max(seg_cols)))
    r = 0
    for rows, cols in zip(seg_rows, seg_cols):
        ref[r:r + rows, :cols] -= A[r:r + rows] @ Bl[:cols].T
        r += rows
    for table in (perm, None):
        got = np.zeros_like(ref)
        for (r0, rows, c0, cols, pm) in stair_chunks(seg_rows, seg_cols, nbr if table is not None else 0, table, limit):
            assert len(rows) <= limit and (pm is None or len(pm) <= limit)
            if pm is not None:
                b = np.concatenate([Bp[q * nbr:(q + 1) * nbr] for q in pm])
            else:
                b = Bl[c0:]
            rr = r0
            for rws, cls in zip(rows, cols):
                got[rr:rr + rws, c0:c0 + cls] -= A[rr:rr + rws] @ b[:cls].T
                rr += rws
        np.testing.assert_allclose(got, ref, atol=1e-13)


@pytest.mark.parametrize('bad', [(0, 1), (1,), (0,)])
def test_native_driver_handshake_is_collective_when_rccl_is_missing(tmp_path, bad):
    """a rank that cannot load librccl (G3_RCCL_PATH=/nonexistent) used to raise BEFORE the id broadcast and
    leave its peers blocked in it.  Now every rank probes, the ranks exchange the outcome, and all raise the same error --
    whichever rank is the broken one -- and stay in step (one more all-reduce goes through)."""
    import torch.multiprocessing as mp
    from dist_helpers import handshake_worker
    world = 2
    mp.spawn(handshake_worker, args=(world, _free_port(), tuple(bad), str(tmp_path)), nprocs=world, join=True)
    msgs = []
    for r in range(world):
        lines = open(str(tmp_path / ('rank%d.txt' % r))).read().split('\n')
        msgs.append(lines[0])
        assert float(lines[1]) == 3.0                       # 1 + 2: the closing all-reduce saw both ranks
    assert msgs[0] and msgs[0] == msgs[1]                   # the same error everywhere
    assert 'native multi-GPU driver unavailable' in msgs[0] and 'of 2 ranks' in msgs[0]
