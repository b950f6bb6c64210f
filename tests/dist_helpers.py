"""Test doubles and workers for the multi-rank driver (g3py_amd.distributed).

NumpyPanelOps implements the PanelOps interface with NumPy/SciPy on CPU tensors so that the
distribution logic (ownership, broadcast order, reductions) can run under gloo without a GPU.
It lives under tests/ (it uses the oracle) and is never imported by the product."""
import os
import sys

import numpy as np
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def cb_transport(name='callbacks'):
    """the native driver's test transport for world > 1 on one GPU: the ASYNCHRONOUS one by default (worker threads,
    stream-ordered collectives in flight beside the three streams: g3_dist_create_callbacks_async); G3_TEST_TRANSPORT=
    callbacks selects the blocking one, kept as a debugging aid.  Other names pass through."""
    if name != 'callbacks':
        return name
    return os.environ.get('G3_TEST_TRANSPORT', 'callbacks_async')


class NumpyPanelOps:
    def __init__(self, torch):
        self.torch = torch

    def alloc(self, rows, cols):
        return self.torch.zeros((rows, cols), dtype=self.torch.float64)

    zeros = alloc

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))

    def gram_rows(self, out, X, N, r0, nb, spec):
        from oracle import g3_oracle as orc
        o = out.numpy()
        o[:, :r0 + nb] = 0
        rows = max(min(nb, N - r0), 0)
        cols = min(N, r0 + nb)
        if rows > 0:
            K = orc.tt_to_num(orc.kernel_cov(spec, X.numpy()[:cols], None))      # square semantics (noise on the diagonal)
            o[:rows, :cols] = K[r0:r0 + rows]
        for i in range(rows, nb):          # identity padding
            o[i, r0 + i] = 1.0

    def rhs_rows(self, out, chunk, Xs, M, X, N, spec, delta):
        from oracle import g3_oracle as orc
        o = out.numpy()
        o[:] = 0
        if chunk == 0:
            o[0, :N] = delta.numpy()[:N]
            return
        s0 = (chunk - 1) * 128
        m = max(min(128, M - s0), 0)
        if m > 0:
            o[:m, :N] = orc.tt_to_num(orc.kernel_cov(spec, Xs.numpy()[s0:s0 + m], X.numpy()[:N]))

    def diag_min(self, blk, n):
        dg = np.diag(blk.numpy()[:n, :n])
        return float(dg.min()), float(dg.mean())

    def diag_add(self, blk, n, value):
        p = blk.numpy()
        p[np.arange(n), np.arange(n)] += value

    def potrf_block(self, L, nb, W):
        if getattr(self, '_info', 0):
            return
        p = L.numpy()
        A = np.tril(p[:nb, :nb])
        A = A + np.tril(A, -1).T
        try:
            F = scipy.linalg.cholesky(A, lower=True)
        except Exception:
            self._info = max(getattr(self, '_info', 0), 1)
            return
        p[:nb, :nb] = F

    def reset_info(self):
        self._info = 0

    def read_info(self):
        return getattr(self, '_info', 0)

    def lookahead(self):
        import contextlib
        return contextlib.nullcontext()

    def join_lookahead(self):
        pass

    def bulk(self):
        import contextlib
        return contextlib.nullcontext()

    def mark(self):
        return None

    def wait_event(self, ev):
        pass

    def join_bulk(self):
        pass

    def trsm(self, L, nb, W, B, m):
        # like the device kernels, everything after a failed pivot is skipped (the info flag)
        if m > 0 and not getattr(self, '_info', 0):
            b = B.numpy()
            b[:m, :nb] = scipy.linalg.solve_triangular(np.tril(L.numpy()[:nb, :nb]), b[:m, :nb].T, lower=True).T

    def invert_block(self, L, nb, W, V):
        # the full inverse the owner broadcasts instead of (L, block inverses): distributed.py::_factor_block
        if not getattr(self, '_info', 0):
            V.numpy()[:nb, :nb] = scipy.linalg.solve_triangular(np.tril(L.numpy()[:nb, :nb]), np.eye(nb), lower=True)

    def solve_full(self, V, nb, B, m):
        if m > 0 and not getattr(self, '_info', 0):
            b = B.numpy()
            b[:m, :nb] = b[:m, :nb] @ np.tril(V.numpy()[:nb, :nb]).T

    def gemm_sub(self, C, A, B, m, n, k, lower_only=False):
        if m > 0 and n > 0 and not getattr(self, '_info', 0):
            c = C.numpy()
            upd = A.numpy()[:m, :k] @ B.numpy()[:n, :k].T
            if lower_only:
                upd = np.where(np.tril(np.ones((m, n), bool)), upd, 0.0)
            c[:m, :n] -= upd

    def gemm_sub_stair(self, C, A, B, k, seg_rows, seg_cols, b_block_rows=0, b_perm=None, seg_diag=None):
        if getattr(self, '_info', 0):
            return
        c, a, b = C.numpy(), A.numpy(), B.numpy()
        if b_perm is not None:      # logical block s of B lives at physical block b_perm[s]
            nbr = b_block_rows
            b = np.concatenate([b[p * nbr:(p + 1) * nbr] for p in b_perm])
        r = 0
        for rows, cols in zip(seg_rows, seg_cols):
            if rows > 0 and cols > 0:
                c[r:r + rows, :cols] -= a[r:r + rows, :k] @ b[:cols, :k].T
            r += rows

    def fill_zero(self, t):
        t.zero_()

    def scale_cols(self, t, n, factor):
        t[:, :n] *= factor

    def gram_block(self, out, Xa, ma, Xb, mb, spec, scrub=True):
        from oracle import g3_oracle as orc
        K = orc.kernel_cov(spec, Xa.numpy()[:ma], Xb.numpy()[:mb])
        out.numpy()[:ma, :mb] = orc.tt_to_num(K) if scrub else K

    def chol_draws(self, cov, M, loc, Z):
        from oracle import g3_oracle as orc
        L, tries, fb = orc.cholesky_robust(cov.numpy()[:M, :M], return_info=True)
        return np.asarray(loc)[:, None] + np.tril(L).dot(Z), tries, fb

    def logdet_block(self, D, nv):
        return float(np.sum(np.log(np.diag(D.numpy()[:nv, :nv])))) if nv > 0 else 0.0

    def rows_dot(self, V, a, n):
        v = V.numpy()[:, :n]
        return v @ a.numpy()[0, :n], (v ** 2).sum(1)

    def sync(self):
        pass


def synth(N, d, M, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    Lbox = N ** (1.0 / d)
    X = rng.uniform(0, Lbox, (N, d))
    Xs = rng.uniform(0, Lbox, (M, d))
    y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(N)
    return X, y, Xs


def worker(rank, world, port, N, d, M, nb, backend, use_gpu, spec_f, noise, out_path, dup=False, draws=0, dtype='f64'):
    """one rank of a distributed run; rank 0 writes (logp, mean, var[, draws]) to out_path"""
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from g3py_amd.distributed import DistributedGP
        from oracle import g3_oracle as orc
        X, y, Xs = synth(N, d, M, 77)
        if dup:                      # duplicated inputs: singular without noise -> the jitter schedule
            X[1::2] = X[0::2][:len(X[1::2])]
            y = np.sin(X.sum(1) / np.sqrt(d))
        spec_n = orc.with_noise(spec_f, noise) if noise is not None else spec_f
        if use_gpu:
            import g3py_amd as g3
            tdev = torch.device('cuda', 0)
            torch.cuda.set_device(0)
            dev = g3.Device(0)
            st = torch.cuda.Stream()
            torch.cuda.set_stream(st)
            dev.set_stream(st.cuda_stream)
            dgp = DistributedGP(dev, dist, rank, world, N, d, M, nb=nb, torch_device=tdev,
                                dtype=np.float32 if dtype == 'f32' else np.float64)
        else:
            dgp = DistributedGP(None, dist, rank, world, N, d, M, nb=nb, ops=NumpyPanelOps(torch))
        o = dgp.ops
        Z = np.random.default_rng(5).standard_normal((M, draws)) if draws else None
        lp = dgp.step(spec_n, spec_f, o.from_host(X), o.from_host(Xs), o.from_host(y), Z=Z)
        if rank == 0:
            prior = np.diag(orc.kernel_cov(spec_f, Xs))
            np.savez(out_path, logp=lp, mean=dgp.last['mean'], var=np.maximum(prior - dgp.last['ss'], 0),
                     tries=dgp.last['tries'], fallback=dgp.last['fallback'],
                     draws=dgp.last['draws'] if draws else np.zeros(0),
                     comm_calls=sum(v['calls'] for v in dgp.comm_stats().values()))
    finally:
        dist.destroy_process_group()


def chain_worker(rank, world, port, N, d, rows, out_path):
    """one rank of a replica-sharded logp_chain run (two ranks may share cuda:0 over gloo)"""
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import g3py_amd as g3
        from g3py_amd.distributed import logp_chain_sharded
        X, y, _ = synth(N, d, 4, 91)
        gp = g3.GaussianProcess(space=X[:4], location=g3.Zero(), kernel=g3.SE(X))
        gp.observed(X, y)
        rng = np.random.default_rng(3)
        chain = gp.active.dict_to_array(gp.params)[None, :] + 0.2 * rng.standard_normal((rows, gp.active.ndim))
        got = logp_chain_sharded(gp, chain, dist, rank, world)
        if rank == 0:
            ref = np.array([gp.logp(c, array=True) for c in chain])
            np.savez(out_path, got=got, ref=ref)
    finally:
        dist.destroy_process_group()


def native_worker(rank, world, port, N, d, M, nb, transport, spec_f, noise, out_path, dup=False, draws=0, dtype='f64', grad=False):
    """one rank of the driver INSIDE libg3hip (g3_dist_*): transport 'callbacks' lets `world` ranks share cuda:0 over
    gloo (host-staged collectives), 'rccl' is the product transport (one rank per GPU: world 1 on a one-GPU box)"""
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import g3py_amd as g3
        from g3py_amd.distributed import NativeDistributedGP
        from oracle import g3_oracle as orc
        X, y, Xs = synth(N, d, M, 77)
        if dup:
            X[1::2] = X[0::2][:len(X[1::2])]
            y = np.sin(X.sum(1) / np.sqrt(d))
        spec_n = orc.with_noise(spec_f, noise) if noise is not None else spec_f
        npdt = np.float32 if dtype == 'f32' else np.float64
        dev = g3.Device(0)
        dgp = NativeDistributedGP(dev, dist, rank, world, N, d, M, nb=nb, dtype=npdt, transport=cb_transport(transport))
        Xd, Xsd, yd = dev.upload(X.astype(npdt)), dev.upload(Xs.astype(npdt)), dev.upload(y.astype(npdt))
        Z = np.random.default_rng(5).standard_normal((M, draws)) if draws else None
        lp = dgp.step(spec_n, spec_f, Xd, Xsd, yd, Z=Z)
        lp2 = dgp.step(spec_n, spec_f, Xd, Xsd, yd)          # a second evaluation on the same plan: same numbers
        assert lp2 == lp or (np.isnan(lp) and np.isnan(lp2)), (lp, lp2)
        cs = dgp.comm_stats()
        slots, alpha, lp3, names = np.zeros(0), np.zeros(0), lp, []
        if grad:
            # gradient mode: the same step with the identity riding along, then the parameter sums and alpha
            dgp.set_grad(True)
            lp3 = dgp.step(spec_n, spec_f, Xd, Xsd, yd)
            mean3 = dgp.last['mean'].copy()
            prog, gmap, slots, alpha = dgp.dlogp(spec_n, Xd)
            _, _, slots2, _ = dgp.dlogp(spec_n, Xd)              # repeatable from the same factorisation
            assert np.array_equal(slots, slots2)
            dgp.set_grad(False)
            lp4 = dgp.step(spec_n, spec_f, Xd, Xsd, yd)          # and back: the plain plan again
            assert lp4 == lp, (lp4, lp)
            assert np.allclose(mean3, dgp.last['mean'], rtol=1e-9, atol=1e-11)
        if rank == 0:
            prior = np.diag(orc.kernel_cov(spec_f, Xs))
            np.savez(out_path, logp=lp, mean=dgp.last['mean'], var=np.maximum(prior - dgp.last['ss'], 0),
                     tries=dgp.last['tries'], fallback=dgp.last['fallback'],
                     draws=dgp.last['draws'] if draws else np.zeros(0), slots=slots, alpha=alpha, logp_grad=lp3,
                     comm_calls=sum(v['calls'] for v in cs.values()), comm_bytes=sum(v['bytes'] for v in cs.values()))
        dgp.close()
    finally:
        dist.destroy_process_group()


def api_worker(rank, world, port, N, d, M, transport, out_path, warped=False):
    """the PUBLIC API on several ranks (SPMD): GaussianProcess.distribute(...) then logp / predict / logpredictive exactly
    as on one GPU; rank 0 writes what it got"""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import g3py_amd as g3
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
        gp.distribute(dist, rank, world, nb=128, transport=cb_transport(transport))
        lp = gp.logp(params)
        pr = gp.predict(params, mean=True, var=True, std=True, median=True, quantiles=True)
        lpred = gp.logpredictive(params, vector=np.asarray(pr.median) + 0.01)
        lp2 = gp.logp(params)                               # cached factor: no second evaluation, same number
        other = gp.mean(params, space=Xs[: max(M // 2, 1)])  # another space than the process's own
        gp.mean(params)                                     # back to the process's own space: the driver holds its cross solve
        Z = np.random.default_rng(100 + rank).standard_normal((M, 4))      # ranks draw DIFFERENT normals: rank 0's must win
        smp = gp.sampler(params, samples=4, rand=Z)
        k2 = g3.OU(X, var=0.7, metric=g3.ARD_L1(X, rate=np.full(d, 0.8)))
        cm = np.asarray(gp.cross_mean(params, cross_kernel=k2))   # another kernel for K(space, X) (gaussian.py:99-112)
        m_again = np.asarray(gp.mean(params))                      # ... must not poison the cached cross solve
        assert np.allclose(m_again, np.asarray(pr.mean), rtol=0, atol=1e-12)
        grad = np.asarray(gp.dlogp(params))                 # switches the driver to gradient mode (identity rows ride along)
        lp3 = gp.logp(params)
        assert abs(lp3 - lp) <= 1e-11 * abs(lp), (lp3, lp)
        p2 = {k: (v + 0.05) for k, v in params.items()}     # another parameter vector in gradient mode: logp and dlogp again
        lp_b, grad_b = gp.logp(p2), np.asarray(gp.dlogp(p2))
        if rank == 0:
            np.savez(out_path, logp=lp, logp2=lp2, mean=pr.mean, var=pr.variance, std=pr.std, median=pr.median,
                     qu=pr.quantile_up, qd=pr.quantile_down, lpred=lpred, other=other, smp=smp, grad=grad)
        cov = np.asarray(gp.kernel(params))              # formed in row chunks, gathered on every rank
        cov_n = np.asarray(gp.kernel(params, noise=True))
        chol = np.asarray(gp.cholesky(params))
        if rank == 0:
            np.savez(out_path.replace('.npz', '_cov.npz'), cov=cov, cov_n=cov_n, chol=chol)
        gp.undistribute()
        assert abs(gp.logp(params) - lp) <= 1e-10 * abs(lp)       # and the same process object works on one GPU again
        g1, g1b = np.asarray(gp.dlogp(params)), np.asarray(gp.dlogp(p2))      # ... where K^-1 is one matrix
        np.testing.assert_allclose(cm, np.asarray(gp.cross_mean(params, cross_kernel=k2)), atol=1e-8)
        assert abs(gp.logp(p2) - lp_b) <= 1e-10 * abs(lp_b)
        np.testing.assert_allclose(grad, g1, rtol=1e-7, atol=1e-8 * np.abs(g1).max())
        np.testing.assert_allclose(grad_b, g1b, rtol=1e-7, atol=1e-8 * np.abs(g1b).max())
    finally:
        dist.destroy_process_group()


def tp_worker(rank, world, port, N, d, M, out_path):
    """Student-t process on several ranks: logp, variance (posterior scaling) and dlogp -- the density hands the driver
    sqrt(s) through alpha_scale (d logp / d beta = -s / 2, studentT.py:114-135) -- equal the same object on one GPU"""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import g3py_amd as g3
        X, y, Xs = synth(N, d, M, 77)
        tp = g3.StudentTProcess(space=Xs, location=g3.Bias(), kernel=g3.SE(X))
        tp.observed(X, y)
        params = dict(tp.params)
        for k in params:
            if k.endswith('_var_log_') and 'Noise' not in k:
                params[k] = np.log(1.1)
            elif k.endswith('_rate_log_'):
                params[k] = np.log(np.full(d, 0.9))
            elif 'Noise' in k:
                params[k] = np.log(0.1)
        tp.distribute(dist, rank, world, nb=128, transport=cb_transport())
        lp, var, g = tp.logp(params), np.asarray(tp.variance(params)), np.asarray(tp.dlogp(params))
        tp.undistribute()
        lp1, var1, g1 = tp.logp(params), np.asarray(tp.variance(params)), np.asarray(tp.dlogp(params))
        assert abs(lp - lp1) <= 1e-10 * abs(lp1), (lp, lp1)
        np.testing.assert_allclose(var, var1, atol=1e-8)
        np.testing.assert_allclose(g, g1, rtol=1e-7, atol=1e-8 * np.abs(g1).max())
        if rank == 0:
            np.savez(out_path, logp=lp, grad=g, ok=1)
    finally:
        dist.destroy_process_group()


def native_multi_worker(rank, world, port, cases, out_path):
    """several seeded random shapes through ONE process group: cases = [(N, d, M, nb, seed), ...]; rank 0 writes logp, mean,
    ss, gradient sums and alpha of every case (driver re-created per case: plan, buffers, streams come and go)"""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import g3py_amd as g3
        from g3py_amd.distributed import NativeDistributedGP
        from oracle import g3_oracle as orc
        dev = g3.Device(0)
        res = {}
        for ci, (N, d, M, nb, seed) in enumerate(cases):
            X, y, Xs = synth(N, d, M, seed)
            spec_f = ('MAT32', 1.1, np.linspace(0.7, 1.2, d), None)
            spec_n = orc.with_noise(spec_f, 0.2)
            dgp = NativeDistributedGP(dev, dist, rank, world, N, d, M, nb=nb, dtype=np.float64, transport=cb_transport())
            Xd, Xsd, yd = dev.upload(X), dev.upload(Xs), dev.upload(y)
            dgp.set_grad(True)
            lp = dgp.step(spec_n, spec_f, Xd, Xsd, yd)
            _, _, slots, alpha = dgp.dlogp(spec_n, Xd)
            res['logp%d' % ci], res['mean%d' % ci], res['ss%d' % ci] = lp, dgp.last['mean'].copy(), dgp.last['ss'].copy()
            res['slots%d' % ci], res['alpha%d' % ci] = slots, alpha
            dgp.close()
        if rank == 0:
            np.savez(out_path, **res)
    finally:
        dist.destroy_process_group()


def handshake_worker(rank, world, port, bad_ranks, out_dir):
    """start-up handshake of the native driver with librccl unavailable on `bad_ranks`: every rank must raise the SAME
    collective error and stay in step with its peers (no GPU needed: the probe fails before any HIP call)"""
    import torch.distributed as dist
    import types
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    if rank in bad_ranks:
        os.environ['G3_RCCL_PATH'] = '/nonexistent/librccl.so'
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from g3py_amd import _lib
    from g3py_amd.distributed import NativeDistributedGP
    dev = types.SimpleNamespace(lib=_lib.load(), ctx=None)
    msg = ''
    try:
        NativeDistributedGP(dev, dist, rank, world, 256, 2, 8, nb=128)
    except _lib.G3Error as e:
        msg = str(e)
    # the ranks are still in step: one more collective goes through
    import torch
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    with open(os.path.join(out_dir, 'rank%d.txt' % rank), 'w') as f:
        f.write('%s\n%g\n' % (msg, float(t.item())))
    dist.destroy_process_group()


def contrib_worker(rank, world, port, N, d, M, nb, out_dir):
    """one rank of a world-rank evaluation over the callback transport (gloo, one GPU): writes the rank's OWN
    contribution to the closing all-reduce -- [log-det part, a^T a part, mean parts (M), sum-of-squares parts (M)]"""
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import g3py_amd as g3
        from g3py_amd.distributed import NativeDistributedGP
        from oracle import g3_oracle as orc
        X, y, Xs = synth(N, d, M, 77)
        spec_f = ('SE', 1.0, np.ones(d), None)
        dev = g3.Device(0)
        dgp = NativeDistributedGP(dev, dist, rank, world, N, d, M, nb=nb, transport=cb_transport())
        lp = dgp.step(orc.with_noise(spec_f, 0.1), spec_f, dev.upload(X), dev.upload(Xs), dev.upload(y))
        assert dgp.last_allreduce_in.shape == (2 + 2 * M,)
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), contrib=dgp.last_allreduce_in, logp=lp, mean=dgp.last['mean'], ss=dgp.last['ss'])
        dgp.close()
    finally:
        dist.destroy_process_group()


def twin_worker(rank, world, port, N, d, M, nb, out_dir):
    """both drivers on the same plan: the collectives each issues, in order and per kind"""
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import g3py_amd as g3
        from g3py_amd.distributed import DistributedGP, NativeDistributedGP
        from oracle import g3_oracle as orc
        X, y, Xs = synth(N, d, M, 77)
        spec_f = ('SE', 1.0, np.ones(d), None)
        spec_n = orc.with_noise(spec_f, 0.1)
        dev = g3.Device(0)
        nat = NativeDistributedGP(dev, dist, rank, world, N, d, M, nb=nb, transport=cb_transport())
        nat.coll_trace = []
        lp_n = nat.step(spec_n, spec_f, dev.upload(X), dev.upload(Xs), dev.upload(y))
        nat_trace = list(nat.coll_trace)
        nat.close()
        tdev = torch.device('cuda', 0)
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()                 # the Python driver mixes torch ops and library calls: one stream for both
        torch.cuda.set_stream(st)
        dev.set_stream(st.cuda_stream)
        py = DistributedGP(dev, dist, rank, world, N, d, M, nb=nb, torch_device=tdev)
        py.coll_trace = []
        o = py.ops
        lp_p = py.step(spec_n, spec_f, o.from_host(X), o.from_host(Xs), o.from_host(y))
        import json
        with open(os.path.join(out_dir, 'rank%d.json' % rank), 'w') as f:
            json.dump({'native': nat_trace, 'python': py.coll_trace, 'logp': [lp_n, lp_p]}, f)
    finally:
        dist.destroy_process_group()
