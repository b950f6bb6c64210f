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


class NumpyPanelOps:
    def __init__(self, torch):
        self.torch = torch

    def alloc(self, rows, cols):
        return self.torch.zeros((rows, cols), dtype=self.torch.float64)

    zeros = alloc

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))

    def gram_panel(self, out, X, N, Np, r0, nb, spec):
        from oracle import g3_oracle as orc
        Xn = X.numpy()
        o = out.numpy()
        o[:] = 0
        rows = max(N - r0, 0)
        cols = max(min(nb, N - r0), 0)
        if cols > 0:
            K = orc.tt_to_num(orc.kernel_cov(spec, Xn[r0:], None))[:, :cols]
            o[:rows, :cols] = K
        for i in range(Np - r0):       # identity padding
            if i >= rows and i < nb:
                o[i, i] = 1.0

    def diag_min(self, panel, n):
        dg = np.diag(panel.numpy()[:n, :n])
        return float(dg.min()), float(dg.mean())

    def diag_add(self, panel, n, value):
        p = panel.numpy()
        p[np.arange(n), np.arange(n)] += value

    def potrf_panel(self, panel, rows, nb, W):
        p = panel.numpy()
        A = np.tril(p[:nb, :nb])
        A = A + np.tril(A, -1).T
        try:
            L = scipy.linalg.cholesky(A, lower=True)
        except Exception:
            return 1
        p[:nb, :nb] = np.where(np.tril(np.ones((nb, nb), bool)), L, p[:nb, :nb])
        if rows > nb:
            p[nb:rows] = scipy.linalg.solve_triangular(L, p[nb:rows].T, lower=True).T
        return 0

    def syrk_update(self, C, A, B, m, n, k):
        c = C.numpy()
        upd = A.numpy()[:m, :k] @ B.numpy()[:n, :k].T
        mask = np.tril(np.ones((m, n), bool))
        c[:m, :n] -= np.where(mask, upd, 0.0)

    def rhs_block(self, out, Xs, M, X, N, r0, nb, spec, delta):
        from oracle import g3_oracle as orc
        o = out.numpy()
        o[:] = 0
        cols = max(min(nb, N - r0), 0)
        if cols > 0:
            o[128:128 + M, :cols] = orc.tt_to_num(orc.kernel_cov(spec, Xs.numpy(), X.numpy()[r0:r0 + cols]))
            o[0, :cols] = delta.numpy()[r0:r0 + cols]

    def block_stats(self, rhs, Ljj, M, nb, nvalid):
        x = rhs.numpy()
        a = x[0, :nb]
        V = x[128:128 + M, :nb]
        ld = float(np.sum(np.log(np.diag(Ljj.numpy()[:nvalid, :nvalid])))) if nvalid > 0 else 0.0
        return ld, float(a @ a), V @ a, (V ** 2).sum(1)

    def sync(self):
        pass


def synth(N, d, M, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    Lbox = N ** (1.0 / d)
    X = rng.uniform(0, Lbox, (N, d))
    Xs = rng.uniform(0, Lbox, (M, d))
    y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(N)
    return X, y, Xs


def worker(rank, world, port, N, d, M, nb, backend, use_gpu, spec_f, noise, out_path):
    """one rank of a distributed run; rank 0 writes (logp, mean, var) to out_path"""
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
        spec_n = orc.with_noise(spec_f, noise)
        if use_gpu:
            import g3py_amd as g3
            tdev = torch.device('cuda', 0)
            torch.cuda.set_device(0)
            dev = g3.Device(0)
            st = torch.cuda.Stream()
            torch.cuda.set_stream(st)
            dev.set_stream(st.cuda_stream)
            dgp = DistributedGP(dev, dist, rank, world, N, d, M, nb=nb, torch_device=tdev)
        else:
            dgp = DistributedGP(None, dist, rank, world, N, d, M, nb=nb, ops=NumpyPanelOps(torch))
        o = dgp.ops
        lp = dgp.step(spec_n, spec_f, o.from_host(X), o.from_host(Xs), o.from_host(y))
        if rank == 0:
            prior = np.diag(orc.kernel_cov(spec_f, Xs))
            np.savez(out_path, logp=lp, mean=dgp.last['mean'], var=np.maximum(prior - dgp.last['ss'], 0))
    finally:
        dist.destroy_process_group()
