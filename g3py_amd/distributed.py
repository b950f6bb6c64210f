"""Multi-GPU GP hot path: the N x N covariance block-partitioned over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI).  The lower triangle
of K is cut into NB-wide block columns, dealt round-robin to the ranks (1-D block-cyclic); a
rank stores each of its block columns as one tall row-major panel (rows r_j.. of columns
r_j..r_j+nb).  The reference has no distributed code at all; the
algebra is the same Gram + Cholesky + triangular solves as the single-GPU path
(g3py/libs/tensors.py:197-222, g3py/processes/gaussian.py:208-224, elliptical.py:81-97).

  Gram        every rank builds exactly its own panels from the replicated N x d input: no
              communication.
  Cholesky    right-looking over panels with one-panel look-ahead: the owner factors the
              diagonal block and solves the panel (g3_potrf + g3_trsm_rlt), BROADCASTS the
              panel; the owner of the next panel updates and factors it first and its
              broadcast is issued asynchronously while all ranks apply the current panel to
              the rest of their block columns (MFMA SYRK/GEMM).
  Solves      the right-hand sides B = [delta^T; K(Xs, X)] are appended as extra ROWS to every
              panel (block column j of B lives with block column j of K), so the panel solves
              and trailing updates of the factorisation also compute B L^-T: no separate
              triangular-solve phase and no extra communication (the broadcast panels simply
              carry 128 + roundup(M, 128) more rows).
  Scalars     log det, a^T a, posterior mean / variance pieces: one all-reduce of a short vector.

Only broadcast and all_reduce are used, so the same driver runs on RCCL and, for tests, on gloo.
All tile arithmetic goes through a `PanelOps` object: `HipPanelOps` (libg3hip, the product)
or a test double supplied by tests/ (world_size-2 gloo runs on CPU).
"""
import numpy as np


def block_ranges(Np, nb):
    r = list(range(0, Np, nb))
    return [(a, min(nb, Np - a)) for a in r]


class HipPanelOps:
    """Tile operations on torch CUDA tensors through the C ABI (no CPU fallback)."""

    def __init__(self, dev, torch, device, dtype=np.float64):
        self.dev, self.torch, self.device = dev, torch, device
        self.dtype = np.dtype(dtype)
        self.tdtype = torch.float64 if self.dtype == np.float64 else torch.float32
        from . import _lib
        from .device import compile_spec
        self._lib, self._compile = _lib, compile_spec

    def _w(self, t, rows=None, cols=None):
        rows = t.shape[0] if rows is None else rows
        cols = t.shape[1] if cols is None else cols
        return self.dev.wrap(t.data_ptr(), rows, cols, t.stride(0), self.dtype, keep=t)

    def alloc(self, rows, cols):
        return self.torch.empty((rows, cols), dtype=self.tdtype, device=self.device)

    def zeros(self, rows, cols):
        return self.torch.zeros((rows, cols), dtype=self.tdtype, device=self.device)

    def from_host(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=self.dtype)).to(self.device)

    def gram_panel(self, out, X, N, Np, r0, nb, spec):
        """out[(Np-r0) x nb] = lower part of tt_to_num(K(X, X))[r0:, r0:r0+nb] (+ identity padding)"""
        d = X.shape[1]
        Xv = X[r0:] if r0 < N else X[:1]
        n_rows = max(N - r0, 0)
        prog = self._compile(spec, d)
        flags = self._lib.G3_GRAM_LOWER | self._lib.G3_GRAM_SCRUB | self._lib.G3_GRAM_PAD_EYE
        xw = self.dev.wrap(Xv.data_ptr(), n_rows, d, X.stride(0), self.dtype, keep=X)
        self.dev.gram(prog, xw, None, d, self._w(out), Np - r0, nb, flags)

    def diag_min(self, panel, n):
        import ctypes as C
        out = (C.c_double * 3)()
        rc = self.dev.lib.g3_diag_stats(self.dev.ctx, panel.data_ptr(), n, panel.stride(0), self._lib.dtype_code(self.dtype), out)
        if rc:
            raise self._lib.G3Error('g3_diag_stats failed %d' % rc)
        return out[0], out[1]

    def diag_add(self, panel, n, value):
        import ctypes as C
        rc = self.dev.lib.g3_diag_add(self.dev.ctx, panel.data_ptr(), n, panel.stride(0), self._lib.dtype_code(self.dtype), float(value))
        if rc:
            raise self._lib.G3Error('g3_diag_add failed %d' % rc)

    def potrf_panel(self, panel, rows, nb, W):
        """factor the top nb x nb block in place, then solve the rows below against it"""
        import ctypes as C
        info = C.c_int(0)
        dt = self._lib.dtype_code(self.dtype)
        rc = self.dev.lib.g3_potrf(self.dev.ctx, panel.data_ptr(), nb, panel.stride(0), dt, W.data_ptr(), C.byref(info))
        if rc:
            raise self._lib.G3Error('g3_potrf failed %d' % rc)
        if rows > nb and info.value == 0:
            below = panel[nb:]
            rc = self.dev.lib.g3_trsm_rlt(self.dev.ctx, panel.data_ptr(), nb, panel.stride(0), below.data_ptr(), rows - nb,
                                          below.stride(0), dt, W.data_ptr())
            if rc:
                raise self._lib.G3Error('g3_trsm_rlt failed %d' % rc)
        return info.value

    def syrk_update(self, C_, A, B, m, n, k):
        """C[m x n] -= A[m x k] B[n x k]^T on and below C's diagonal"""
        self.dev.gemm_nt(self._w(C_, m, n), self._w(A, m, k), self._w(B, n, k), m, n, k, alpha=-1.0, beta=1.0, lower_only=True)

    def rhs_block(self, out, Xs, M, X, N, r0, nb, spec, delta):
        """out[E x nb], E = 128 + Mp: row 0 = delta[r0:r0+nb]; rows 128..128+M = tt_to_num(K(Xs, X[r0:r0+nb]))"""
        d = X.shape[1]
        out.zero_()
        ncols = max(min(nb, N - r0), 0)
        if ncols > 0:
            prog = self._compile(spec, d)
            xs = self.dev.wrap(Xs.data_ptr(), M, d, Xs.stride(0), self.dtype, keep=Xs)
            xv = X[r0:]
            xw = self.dev.wrap(xv.data_ptr(), ncols, d, X.stride(0), self.dtype, keep=X)
            v = out[128:]
            self.dev.gram(prog, xs, xw, d, self._w(v), M, ncols, self._lib.G3_GRAM_SCRUB)
            out[0, :ncols] = delta[r0:r0 + ncols]

    def block_stats(self, rhs, Ljj, M, nb, nvalid):
        """(sum log diag L_jj over valid rows, a_j^T a_j, dot[M], ss[M]) with a_j = row 0 of the
        solved right-hand-side block and V_j = its rows 128..128+M"""
        a = rhs[0:1]
        V = rhs[128:]
        st = self.dev.logp_terms(self._w(Ljj, nb, nb), max(nvalid, 1), self._w(a, 1, nb)) if nvalid > 0 else [0.0, 0.0, 0, 0]
        dot, ss = self.alloc(1, M), self.alloc(1, M)
        self.dev.rows_dot_ss(self._w(V, M, nb), M, nb, self._w(a, 1, nb), self._w(dot, 1, M), self._w(ss, 1, M))
        # a beyond nvalid is exactly zero (zero right-hand side in the identity padding)
        quad = float((a[0, :] * a[0, :]).sum().item())
        return st[0], quad, dot[0].double().cpu().numpy(), ss[0].double().cpu().numpy()

    def sync(self):
        self.dev.sync()


class DistributedGP:
    """One evaluation of logp + posterior mean / variance over `world` ranks."""

    def __init__(self, dev, dist, rank, world, N, d, M, nb=1024, torch_device=None, ops=None, dtype=np.float64):
        import torch
        self.dist, self.rank, self.world = dist, rank, world
        self.N, self.d, self.M = N, d, M
        pad = 128
        self.nb = max(pad, (nb // pad) * pad)
        self.Np = (N + pad - 1) // pad * pad
        self.Mp = (M + pad - 1) // pad * pad
        self.E = 128 + self.Mp                          # right-hand-side rows: delta block + K(Xs, .)
        self.R = self.Np + self.E
        self.blocks = block_ranges(self.Np, self.nb)
        self.ops = ops if ops is not None else HipPanelOps(dev, torch, torch_device, dtype)
        self.torch = torch
        o = self.ops
        # owned panels: rows r_j..Np of block column j, then the E right-hand-side rows
        self.panels, self.W = {}, {}
        for j, (r0, nbj) in enumerate(self.blocks):
            if j % world == rank:
                self.panels[j] = o.alloc(self.R - r0, nbj)
                self.W[j] = o.alloc(nbj, 128)
        self.recv = [o.alloc(self.R, self.nb), o.alloc(self.R, self.nb)]   # double-buffered panel landing zone
        self.last = {}
        self._alias = {}

    def owner(self, j):
        return j % self.world

    # ---------------------------------------------------------------- build
    def _build(self, spec, spec_cross, X, Xs, delta, jitter):
        o = self.ops
        lmin = np.inf
        for j, P in self.panels.items():
            r0, nbj = self.blocks[j]
            o.gram_panel(P[:self.Np - r0], X, self.N, self.Np, r0, nbj, spec)
            o.rhs_block(P[self.Np - r0:], Xs, self.M, X, self.N, r0, nbj, spec_cross, delta)
            nv = max(min(nbj, self.N - r0), 0)
            if nv > 0:
                lmin = min(lmin, o.diag_min(P, nv)[0])
        # tt_to_cov (tensors.py:95-98): min over the WHOLE diagonal
        t = self.torch.tensor([lmin if np.isfinite(lmin) else 1e300], dtype=self.torch.float64)
        gmin = float(self._allreduce(t, 'min')[0])
        add = jitter
        if not gmin > 0:
            add += float(np.float32(1e-6)) - gmin
        if add != 0.0:
            for j, P in self.panels.items():
                r0, nbj = self.blocks[j]
                nv = max(min(nbj, self.N - r0), 0)
                if nv > 0:
                    o.diag_add(P, nv, add)

    def _allreduce(self, t, op='sum'):
        dist = self.dist
        if self.world == 1:
            return t
        opmap = {'sum': dist.ReduceOp.SUM, 'min': dist.ReduceOp.MIN, 'max': dist.ReduceOp.MAX}
        dev_t = t.to(self.recv[0].device) if self.recv[0].is_cuda else t
        dist.all_reduce(dev_t, op=opmap[op])
        return dev_t.cpu()

    def _panel(self, k):
        """the (rows x nb_k) tensor holding panel k on this rank (own storage or landing zone)"""
        r0, nbk = self.blocks[k]
        if self.owner(k) == self.rank:
            return self.panels[k]
        if k not in self._alias:   # contiguous (rows x nb_k) alias of the landing zone
            rows = self.R - r0
            self._alias[k] = self.recv[k % 2].view(-1)[:rows * nbk].view(rows, nbk)
        return self._alias[k]

    def _bcast(self, k, async_op):
        if self.world == 1:
            return None
        return self.dist.broadcast(self._panel(k), src=self.owner(k), async_op=async_op)

    def _apply(self, k, j):
        """block column j (and its right-hand-side rows) -= panel k contribution (lower part only)"""
        rk, nbk = self.blocks[k]
        rj, nbj = self.blocks[j]
        rows = self._panel(k)[rj - rk:]
        self.ops.syrk_update(self.panels[j], rows, rows, self.R - rj, nbj, nbk)

    # ---------------------------------------------------------------- factorisation + solves in one sweep
    def factor(self, spec, spec_cross, X, Xs, delta, jitter=0.0):
        """returns the global potrf info (0 = success)"""
        o = self.ops
        self._alias = {}
        self._build(spec, spec_cross, X, Xs, delta, jitter)
        nblk = len(self.blocks)
        info = 0
        if self.owner(0) == self.rank:
            info = max(info, o.potrf_panel(self.panels[0], self.R, self.blocks[0][1], self.W[0]))
        work = self._bcast(0, async_op=False)
        for k in range(nblk):
            if work is not None and hasattr(work, 'wait'):
                work.wait()
            work = None
            if k + 1 < nblk:
                if self.owner(k + 1) == self.rank:      # look-ahead: next panel first
                    self._apply(k, k + 1)
                    r1, nb1 = self.blocks[k + 1]
                    info = max(info, o.potrf_panel(self.panels[k + 1], self.R - r1, nb1, self.W[k + 1]))
                work = self._bcast(k + 1, async_op=True)
            for j in self.panels:
                if j > k + 1:
                    self._apply(k, j)
        t = self._allreduce(self.torch.tensor([float(info)], dtype=self.torch.float64), 'max')
        return int(t[0])

    def factor_robust(self, spec, spec_cross, X, Xs, delta):
        """CholeskyRobust's schedule (tensors.py:197-222) around the distributed factorisation"""
        info = self.factor(spec, spec_cross, X, Xs, delta)
        tries, fallback = 0, False
        if info != 0:
            # jitter from the diagonal of the (lifted) covariance: rebuild and gather mean / min
            self._build(spec, spec_cross, X, Xs, delta, 0.0)
            s, cnt, mn = 0.0, 0, np.inf
            for j, P in self.panels.items():
                r0, nbj = self.blocks[j]
                nv = max(min(nbj, self.N - r0), 0)
                if nv > 0:
                    a, b = self.ops.diag_min(P, nv)
                    mn, s, cnt = min(mn, a), s + b * nv, cnt + nv
            t = self._allreduce(self.torch.tensor([s, float(cnt)], dtype=self.torch.float64), 'sum')
            mean = float(t[0]) / max(float(t[1]), 1.0)
            gmin = float(self._allreduce(self.torch.tensor([mn if np.isfinite(mn) else 1e300], dtype=self.torch.float64), 'min')[0])
            c6, c10 = float(np.float32(1e-6)), 10.0
            dK, lift = mean * c6, 0.0
            if gmin <= 0.0:
                lift = mean * c6 - gmin
            ok = False
            for _ in range(20):
                tries += 1
                if self.factor(spec, spec_cross, X, Xs, delta, jitter=lift + dK) == 0:
                    ok = True
                    break
                dK *= c10
            if not ok:
                raise RuntimeError('distributed Cholesky: jitter schedule exhausted (the 1e-10*I fallback of the '
                                   'reference is a single-GPU path)')
        self.last.update(info=info, tries=tries, fallback=fallback)
        return info

    def stats(self):
        """(logdet, quad, mean_pieces[M], ss[M]) summed over all ranks"""
        o, M = self.ops, self.M
        acc = np.zeros(2 + 2 * M)
        for j, P in self.panels.items():
            rj, nbj = self.blocks[j]
            nv = max(min(nbj, self.N - rj), 0)
            ld, q, dot, ss = o.block_stats(P[self.Np - rj:], P, M, nbj, nv)
            acc[0] += ld
            acc[1] += q
            acc[2:2 + M] += dot
            acc[2 + M:] += ss
        t = self._allreduce(self.torch.from_numpy(acc), 'sum').numpy()
        return float(t[0]), float(t[1]), t[2:2 + M], t[2 + M:]

    def step(self, spec_noise, spec_f, X, Xs, delta):
        """one pass of the hot path; returns logp (mean / variance pieces in self.last)"""
        Xt = X._keep if hasattr(X, '_keep') and X._keep is not None else X
        Xst = Xs._keep if hasattr(Xs, '_keep') and Xs._keep is not None else Xs
        dt = delta._keep if hasattr(delta, '_keep') and delta._keep is not None else delta
        self.factor_robust(spec_noise, spec_f, Xt, Xst, dt.reshape(-1))
        logdet, quad, mean, ss = self.stats()
        self.ops.sync()
        logp = -0.5 * self.N * np.log(2 * np.pi) - 0.5 * quad - logdet
        self.last.update(logdet=logdet, quad=quad, mean=mean, ss=ss, logp=logp)
        return logp
