"""Multi-GPU GP hot path: the N x N covariance block-partitioned over the GPUs of one node.

One process per GPU.  The reference has no distributed code at all; the algebra is the same
Gram + Cholesky + triangular solves as the single-GPU path (g3py/libs/tensors.py:197-222,
g3py/processes/gaussian.py:208-224, elliptical.py:81-97).

Two drivers with ONE schedule:
  * NativeDistributedGP -- the product: a thin wrapper of the driver inside libg3hip (g3_dist_*,
    g3py_amd/csrc/g3_dist.hip): the library owns the streams, the buffers, the per-panel loop and the RCCL
    communicators; nothing here loops over panels.
  * DistributedGP -- the same schedule written out on torch.distributed (backend "nccl" = RCCL over xGMI, or gloo),
    with all tile arithmetic behind a `PanelOps` object: `HipPanelOps` (libg3hip) or a test double supplied by tests/
    (world_size 1-8 gloo runs on CPU).  It is the readable statement of the algorithm and bench.py's fallback.

Layout: ROW-block-cyclic.  The (identity-padded) covariance is cut into nb-row blocks dealt in boustrophedon order to
the ranks; a rank keeps its row blocks stacked in one local matrix (full width, so the rows below any block are one
contiguous slab), followed by its share of the right-hand-side rows [delta^T; K(Xs, X)] (128-row chunks, dealt
round-robin), which ride through the factorisation exactly as on one GPU.

Why rows and not block columns: with block columns the owner of column k+1 needs ALL of panel k before it can factor,
so the chain  receive panel -> update -> factor -> send panel  serialises a whole-panel transfer per step.  With row
blocks only the nb x nb diagonal factor travels on the critical path; the panel itself is computed in parallel (every
rank solves its own rows) and exchanged by an all-gather in which every GPU sends and receives over all of its xGMI
links at once.

Schedule (three streams per rank; G_k = the gathered panel k), step k:
  bulk stream    d1. block column k+2 of everything the rank owns from block k+2 down -= P_k G_k^T      -> event B_k
                 d2. block columns >= k+3 of its blocks >= k+3 and of the right-hand-side rows: ONE staircase MFMA-GEMM
                     launch (g3_gemm_nt_stair: each row block has its own width and ends in its diagonal block, of
                     which only the tiles on or below the diagonal are launched; the rank-major order of the gathered
                     panel is a block table of the B operand -- no re-ordering copy)
  caller's       a.  (after B_{k-1}) block column k+1 of its blocks >= k+2 -= P_k G_k^T
  stream         b.  (after the broadcast of L_{k+1,k+1}) solve its rows of panel k+1, all-gather them (asynchronous)
  look-ahead     c.  (after B_k) the owner of block k+2 applies panel k+1 to its diagonal block from its own rows,
  stream             factors it and broadcasts the nb x nb factor (+ the inverses of its 128 x 128 diagonal blocks) on
                     the broadcast's OWN communicator, so it never queues behind a panel all-gather

Gram: every rank builds exactly its own rows from the replicated N x d input: no communication.
Scalars (log det, a^T a, posterior mean / variance pieces): one broadcast of a = L^-1 delta and one all-reduce of a
short vector.  Only broadcast, all_gather and all_reduce are used.
"""
import sys as _sys

import numpy as np


def deal_blocks(P, nblk):
    """owner[I] of row block I -- the line-by-line mirror of g3_host.h::g3h_deal: blocks are dealt from the top in rounds of
    P, every rank one block per round, the least loaded rank so far the heaviest block of the round (block I weighs
    I^2 + 6 I + 1: its trailing updates, column updates and panel solves over the sweep); ties go to the lowest rank"""
    import os
    if os.environ.get('G3_DIST_DEAL', '') == 'snake':       # the boustrophedon of rounds 1-4 (A/B measurements, as g3_dist.hip)
        return [(I % (2 * P)) if (I % (2 * P)) < P else 2 * P - 1 - (I % (2 * P)) for I in range(max(nblk, 0))]
    owner = [0] * max(nblk, 0)
    load = [0] * P
    top = nblk - 1
    while top >= 0:
        order = sorted(range(P), key=lambda q: (load[q], q))
        for i in range(P):
            I = top - i
            if I < 0:
                break
            owner[I] = order[i]
            load[order[i]] += I * I + 6 * I + 1
        top -= P
    return owner


def block_ranges(Np, nb):
    r = list(range(0, Np, nb))
    return [(a, min(nb, Np - a)) for a in r]


def stair_chunks(seg_rows, seg_cols, b_block_rows, b_perm, limit, seg_diag=None):
    """cut a staircase update into launches of at most `limit` row segments and `limit` blocks of B.
    Yields (first row, segment rows, first column, segment columns, block table[, diagonal flags]) per launch;
    columns are cut at multiples of b_block_rows (the whole width when B has no block table and fits).  With
    seg_diag the sixth item flags the segments whose diagonal block (their last seg_rows columns) lies in the chunk"""
    nseg = len(seg_rows)
    width = max(seg_cols)
    cstep = limit * b_block_rows if b_block_rows > 0 else width
    r0 = 0
    for s0 in range(0, nseg, limit):
        rows = list(seg_rows[s0:s0 + limit])
        for c0 in range(0, width, max(cstep, 1)):
            cols = [max(min(c - c0, cstep), 0) for c in seg_cols[s0:s0 + limit]]
            if sum(rows) > 0 and max(cols) > 0:
                perm = None
                if b_perm is not None:
                    b0 = c0 // b_block_rows
                    perm = list(b_perm[b0:b0 + limit])
                if seg_diag is None:
                    yield r0, rows, c0, cols, perm
                else:
                    dg = [1 if (g and c0 < c <= c0 + cstep) else 0
                          for g, c in zip(seg_diag[s0:s0 + limit], seg_cols[s0:s0 + limit])]
                    yield r0, rows, c0, cols, perm, dg
        r0 += sum(rows)


class HipPanelOps:
    """Tile operations on torch CUDA tensors through the C ABI (no CPU fallback)."""

    def __init__(self, dev, torch, device, dtype=np.float64):
        self.dev, self.torch, self.device = dev, torch, device
        self.dtype = np.dtype(dtype)
        self.tdtype = torch.float64 if self.dtype == np.float64 else torch.float32
        from . import _lib
        from .device import compile_spec
        self._lib, self._compile = _lib, compile_spec
        self._dt = _lib.dtype_code(self.dtype)
        # look-ahead: the next diagonal block is factored on a second stream through a second
        # library context (its own info flag and scratch, so the two streams never share state)
        # while the main stream applies the trailing update; potrf info is accumulated on the
        # device and read once per sweep
        self.dev_main = dev
        self.dev_side = type(dev)(dev.index)
        self.side = torch.cuda.Stream(device=device, priority=-1)
        self.dev_side.set_stream(self.side.cuda_stream)
        # the bulk of every trailing update runs on a third, low-priority stream (as in the one-GPU sweep,
        # g3_potrf.hip::potrf_lookahead): the panel solve, the next block column and the collectives of step
        # k + 1 are issued on the caller's stream and overlap it
        self.dev_bulk = type(dev)(dev.index)
        self.bulk_s = torch.cuda.Stream(device=device, priority=0)
        self.dev_bulk.set_stream(self.bulk_s.cuda_stream)
        self.info_dev = torch.zeros(1, dtype=torch.int32, device=device)

    def _chk(self, rc, what):
        if rc:
            raise self._lib.G3Error('%s failed (%d): %s' % (what, rc, self.dev.lib.g3_last_error(self.dev.ctx).decode()))

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

    def gram_rows(self, out, X, N, r0, nb, spec):
        """out[nb x >= r0+nb] <- rows [r0, r0+nb), columns [0, r0+nb) of tt_to_num(K(X, X)), identity padded"""
        import ctypes as C
        d = X.shape[1]
        prog = self._compile(spec, d)
        flags = self._lib.G3_GRAM_SCRUB | self._lib.G3_GRAM_PAD_EYE
        rc = self.dev.lib.g3_gram_rows(self.dev.ctx, C.byref(prog), X.data_ptr(), N, X.stride(0), d, r0, nb, self._dt,
                                       out.data_ptr(), out.stride(0), flags)
        self._chk(rc, 'g3_gram_rows')

    def rhs_rows(self, out, chunk, Xs, M, X, N, spec, delta):
        """out[128 x Np]: chunk 0 = [delta; 0 ...]; chunk c >= 1 = tt_to_num(K(Xs[(c-1)*128 : c*128], X)), zero padded"""
        out.zero_()
        if chunk == 0:
            out[0, :N] = delta[:N]
            return
        s0 = (chunk - 1) * 128
        m = max(min(128, M - s0), 0)
        if m > 0:
            d = X.shape[1]
            prog = self._compile(spec, d)
            xs = self.dev.wrap(Xs[s0:].data_ptr(), m, d, Xs.stride(0), self.dtype, keep=Xs)
            xw = self.dev.wrap(X.data_ptr(), N, d, X.stride(0), self.dtype, keep=X)
            self.dev.gram(prog, xs, xw, d, self._w(out), m, N, self._lib.G3_GRAM_SCRUB)

    def diag_min(self, blk, n):
        import ctypes as C
        out = (C.c_double * 3)()
        self._chk(self.dev.lib.g3_diag_stats(self.dev.ctx, blk.data_ptr(), n, blk.stride(0), self._dt, out), 'g3_diag_stats')
        return out[0], out[1]

    def diag_add(self, blk, n, value):
        self._chk(self.dev.lib.g3_diag_add(self.dev.ctx, blk.data_ptr(), n, blk.stride(0), self._dt, float(value)), 'g3_diag_add')

    def potrf_block(self, L, nb, W):
        """factor the nb x nb block in place (lower), W <- inverses of its 128 x 128 diagonal blocks;
        no host synchronisation: the first failing pivot is kept in info_dev (see read_info)"""
        self._chk(self.dev.lib.g3_potrf_nowait(self.dev.ctx, L.data_ptr(), nb, L.stride(0), self._dt, W.data_ptr(),
                                               self.info_dev.data_ptr()), 'g3_potrf_nowait')

    def reset_info(self):
        self.info_dev.zero_()

    def read_info(self):
        return int(self.info_dev.item())

    def lookahead(self):
        """context: work issued inside runs on the look-ahead stream, after everything queued so far"""
        import contextlib
        torch, ops = self.torch, self

        @contextlib.contextmanager
        def cm():
            main = torch.cuda.current_stream(ops.device)
            ops.side.wait_stream(main)
            with torch.cuda.stream(ops.side):
                ops.dev = ops.dev_side
                try:
                    yield
                finally:
                    ops.dev = ops.dev_main
        return cm()

    def join_lookahead(self):
        self.torch.cuda.current_stream(self.device).wait_stream(self.side)

    def bulk(self):
        """context: work issued inside runs on the bulk stream, after everything queued so far on the caller's"""
        import contextlib
        torch, ops = self.torch, self

        @contextlib.contextmanager
        def cm():
            ops.bulk_s.wait_stream(torch.cuda.current_stream(ops.device))
            with torch.cuda.stream(ops.bulk_s):
                prev, ops.dev = ops.dev, ops.dev_bulk
                try:
                    yield
                finally:
                    ops.dev = prev
        return cm()

    def mark(self):
        """an event behind everything queued so far on the current stream"""
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream(self.device))
        return ev

    def wait_event(self, ev):
        if ev is not None:
            self.torch.cuda.current_stream(self.device).wait_event(ev)

    def join_bulk(self):
        self.torch.cuda.current_stream(self.device).wait_stream(self.bulk_s)

    def trsm(self, L, nb, W, B, m):
        """B[m x nb] <- B L^-T"""
        if m > 0:
            self._chk(self.dev.lib.g3_trsm_rlt(self.dev.ctx, L.data_ptr(), nb, L.stride(0), B.data_ptr(), m, B.stride(0),
                                               self._dt, W.data_ptr()), 'g3_trsm_rlt')

    def invert_block(self, L, nb, W, V):
        """V (nb x nb, compact) <- L^-1 by recursive doubling from the 128-block inverses W (g3_trtri_full); nb = 128 * 2^q"""
        key = (nb, self.dev is self.dev_side)
        if getattr(self, '_inv_scratch_key', None) != key:
            self._inv_scratch = (self.zeros(nb, nb), self.zeros(nb, nb))
            self._inv_scratch_key = key
        Vt, U = self._inv_scratch
        self._chk(self.dev.lib.g3_trtri_full(self.dev.ctx, L.data_ptr(), nb, W.data_ptr(), V.data_ptr(), Vt.data_ptr(), U.data_ptr(),
                                             self._dt), 'g3_trtri_full')

    def solve_full(self, V, nb, B, m):
        """B[m x nb] <- B V^T = B L^-T as ONE K-triangular product (g3_trsm_full), through a compact temporary"""
        if m > 0:
            tmp = self.alloc(m, nb)
            self._chk(self.dev.lib.g3_trsm_full(self.dev.ctx, V.data_ptr(), nb, V.stride(0), B.data_ptr(), m, B.stride(0),
                                                tmp.data_ptr(), nb, self._dt), 'g3_trsm_full')
            B.copy_(tmp)

    def gemm_sub(self, C_, A, B, m, n, k, lower_only=False):
        """C[m x n] -= A[m x k] B[n x k]^T"""
        if m > 0 and n > 0:
            self.dev.gemm_nt(self._w(C_, m, n), self._w(A, m, k), self._w(B, n, k), m, n, k, alpha=-1.0, beta=1.0,
                             lower_only=lower_only)

    def gemm_sub_stair(self, C_, A, B, k, seg_rows, seg_cols, b_block_rows=0, b_perm=None, seg_diag=None):
        """stacked row segments of C: C[rows_s, :seg_cols[s]] -= A[rows_s, :k] B[:seg_cols[s], :k]^T in one launch;
        logical row block s of B (b_block_rows rows) is stored at block b_perm[s]; seg_diag[s] != 0: the last
        seg_rows[s] columns of segment s are its diagonal block, only the lower triangle of which is needed"""
        if sum(seg_rows) <= 0 or max(seg_cols) <= 0:
            return
        # one launch describes at most STAIR_MAX row segments and STAIR_MAX blocks of B (the raster table travels
        # in the kernel arguments, g3_gemm.hip::RasterTab): longer staircases -- N / nb > 160 row blocks -- are
        # cut into row chunks and column chunks, each its own launch on the same stream
        es = C_.element_size()
        diag = seg_diag if seg_diag is not None else [0] * len(seg_rows)
        for (r0, rows, c0, cols, perm, dg) in stair_chunks(seg_rows, seg_cols, b_block_rows, b_perm, self.STAIR_MAX, diag):
            self.dev.gemm_nt_stair(C_.data_ptr() + (r0 * C_.stride(0) + c0) * es, C_.stride(0),
                                   A.data_ptr() + r0 * A.stride(0) * es, A.stride(0),
                                   B.data_ptr() + (0 if perm is not None else c0 * B.stride(0) * es), B.stride(0), k,
                                   rows, cols, self.dtype, alpha=-1.0, beta=1.0, b_block_rows=b_block_rows, b_perm=perm,
                                   seg_diag=dg if seg_diag is not None else None)

    import os as _os
    STAIR_MAX = int(_os.environ.get('G3_STAIR_MAX', '160'))     # tests lower it to exercise the chunking

    def fill_zero(self, t):
        t.zero_()

    def scale_cols(self, t, n, factor):
        """t[:, :n] *= factor (the 1e-10 * I fallback solves its right-hand sides by a division)"""
        t[:, :n].mul_(factor)

    def gram_block(self, out, Xa, ma, Xb, mb, spec, scrub=True):
        """out[ma x >= mb] <- tt_to_num(K(Xa[:ma], Xb[:mb])) (cross semantics: no noise term)"""
        d = Xa.shape[1]
        prog = self._compile(spec, d)
        xa = self.dev.wrap(Xa.data_ptr(), ma, d, Xa.stride(0), self.dtype, keep=Xa)
        xb = self.dev.wrap(Xb.data_ptr(), mb, d, Xb.stride(0), self.dtype, keep=Xb)
        self.dev.gram(prog, xa, xb, d, self._w(out), ma, mb, self._lib.G3_GRAM_SCRUB if scrub else 0)

    def chol_draws(self, cov, M, loc, Z):
        """cholesky_robust of the M x M covariance (tensors.py:197-222) and loc + L Z  (gaussian.py:92-95);
        cov is a padded device matrix, Z an M x S host array; returns (draws M x S, tries, fallback)"""
        torch = self.torch
        Mp = cov.shape[0]
        S = Z.shape[1]
        Sp = (S + 63) // 64 * 64
        L = self.zeros(Mp, Mp)
        tries, fb, _ = self.dev.potrf_robust(self._w(cov, M, M), self._w(L, M, M), M)
        Zt = self.zeros(Sp, Mp)
        Zt[:S, :M] = torch.from_numpy(np.ascontiguousarray(np.asarray(Z, dtype=self.dtype).T)).to(self.device)
        out = self.zeros(Mp, Sp)
        self.dev.gemm_nt(self._w(out), self._w(L), self._w(Zt), Mp, Sp, Mp)
        d = out[:M, :S].double().cpu().numpy()
        return np.asarray(loc, dtype=np.float64)[:, None] + d, tries, fb

    def logdet_block(self, D, nv):
        """sum of log of the first nv diagonal entries"""
        if nv <= 0:
            return 0.0
        return self.dev.logp_terms(self._w(D, nv, nv), nv, None)[0]

    def rows_dot(self, V, a, n):
        """(V a, row sums of V^2) over the first n columns, as float64 NumPy vectors"""
        m = V.shape[0]
        dot, ss = self.alloc(1, m), self.alloc(1, m)
        self.dev.rows_dot_ss(self._w(V, m, n), m, n, self._w(a, 1, n), self._w(dot, 1, m), self._w(ss, 1, m))
        return dot[0].double().cpu().numpy(), ss[0].double().cpu().numpy()

    def sync(self):
        self.dev.sync()


class DistributedGP:
    """One evaluation of logp + posterior mean / variance over `world` ranks."""

    def __init__(self, dev, dist, rank, world, N, d, M, nb=512, torch_device=None, ops=None, dtype=np.float64,
                 group_backend=None):
        import torch
        self.dist, self.rank, self.world = dist, rank, world
        self.N, self.d, self.M = N, d, M
        pad = 128
        self.nb = nb = max(pad, (nb // pad) * pad)
        self.Np = (N + nb - 1) // nb * nb                # identity-padded to whole row blocks
        self.nblk = self.Np // nb
        self.Mp = (M + pad - 1) // pad * pad
        self.nchunk = 1 + self.Mp // pad                 # right-hand-side chunks: delta block + K(Xs, .) rows
        self.ops = ops if ops is not None else HipPanelOps(dev, torch, torch_device, dtype)
        self.torch = torch
        o = self.ops
        self._owner = deal_blocks(world, self.nblk)
        self.my_blocks = [I for I in range(self.nblk) if self.owner(I) == rank]
        self.my_chunks = [c for c in range(self.nchunk) if c % world == rank]
        self.rows_mat = len(self.my_blocks) * nb
        self.rows_rhs = len(self.my_chunks) * pad
        self.loff = {I: t * nb for t, I in enumerate(self.my_blocks)}
        self.A = o.zeros(self.rows_mat + self.rows_rhs, self.Np)          # local rows, full width
        # Panel solve as ONE product (as g3_dist.hip, round 5): the owner of a diagonal block inverts the whole nb x nb factor
        # and broadcasts V = L^-1 (nb x nb) instead of (L, its 128-block inverses); a rank's rows of the panel are X V^T.
        # Needs nb = 128 * 2^q <= 2048; G3_DIST_FULLINV=0 keeps the blocked solve (the same switch as the native driver).
        import os as _os
        q = nb // pad
        self.fullinv = (_os.environ.get('G3_DIST_FULLINV', '1') != '0' and nb <= 2048 and (q & (q - 1)) == 0
                        and hasattr(o, 'invert_block'))
        # nb x nb diagonal factor, the inverses of its 128 x 128 diagonal blocks, [its full inverse] (double buffer)
        dlen = nb * nb + nb * pad + (nb * nb if self.fullinv else 0)
        self.dbuf = [o.zeros(1, dlen), o.zeros(1, dlen)]
        self._perms = {}
        cmax = max(self._perm(0)[0], 1) if self.nblk > 1 else 1
        # panel rows on their way out / gathered from every rank (double buffers: the gather of panel k+1 runs
        # under the update with panel k)
        self.send = [o.zeros(max(cmax, 1) * nb, nb), o.zeros(max(cmax, 1) * nb, nb)]
        # three gather buffers: the bulk update with panel k (bulk stream) may still be reading its buffer
        # while panel k + 2 is already being gathered
        self.gath = [o.zeros(world * max(cmax, 1) * nb, nb) for _ in range(3)]
        self.last = {}
        # G3_DIST_COLLECTIVES=1: do not short-circuit the collectives of a one-rank run -- every broadcast,
        # all-gather and all-reduce then really passes through the process group (RCCL stream semantics and
        # work handles exercised on the one GPU a development box has)
        import os
        self.use_coll = world > 1 or os.environ.get('G3_DIST_COLLECTIVES', '0') == '1'
        # The diagonal-factor broadcast (critical path, <= 9 MiB) gets its OWN process group = its own RCCL
        # communicator and stream: on one communicator it would queue behind the panel all-gather (up to
        # 134 MB) issued just before it.  Both are issued in the same order on every rank.
        # group_backend (e.g. 'nccl' when the default group is gloo): both groups are created with that backend
        self.g_bcast = self.g_main = None
        if self.use_coll and dist is not None:
            kw = {'backend': group_backend} if group_backend else {}
            self.g_bcast = dist.new_group(ranks=list(range(world)), **kw)
            if group_backend:
                self.g_main = dist.new_group(ranks=list(range(world)), **kw)
        self.comm = {k: {'calls': 0, 'bytes': 0, 'wait_s': 0.0} for k in ('bcast', 'allgather', 'allreduce')}

    def owner(self, I):
        """the rank that holds row block I (deal_blocks: from the top, in load-balanced rounds -- the ranks' total work
        is within 2 % of the mean at P = 8 with 32 blocks; round-robin leaves the last rank at 1.4x)"""
        return self._owner[I]

    def _perm(self, k):
        """(blocks per rank in the padded gather of step k, block table): entry s of the table (global
        block k + 1 + s) = position of that block in the rank-major gather buffer"""
        if k not in self._perms:
            P = self.world
            nbelow = self.nblk - 1 - k
            cnt = (max(len([I for I in range(k + 1, self.nblk) if self.owner(I) == q]) for q in range(P))
                   if nbelow > 0 else 0)
            seen = [0] * P
            idx = []
            for I in range(k + 1, self.nblk):
                q = self.owner(I)
                idx.append(q * cnt + seen[q])
                seen[q] += 1
            self._perms[k] = (cnt, idx)
        return self._perms[k]

    def _LW(self, k):
        nb = self.nb
        f = self.dbuf[k % 2]
        return f[0, :nb * nb].view(nb, nb), f[0, nb * nb:nb * nb + nb * 128].view(nb, 128)

    def _V(self, k):
        nb = self.nb
        return self.dbuf[k % 2][0, nb * nb + nb * 128:].view(nb, nb)

    def _bc(self, k):
        """what travels for diagonal block k: V alone with the full inverse, else (L, block inverses)"""
        nb = self.nb
        f = self.dbuf[k % 2]
        return f[:, nb * nb + nb * 128:] if self.fullinv else f

    def _diag(self, I):
        """the nb x nb diagonal block of an owned row block (a view of the local matrix)"""
        nb = self.nb
        return self.A[self.loff[I]:self.loff[I] + nb, I * nb:(I + 1) * nb]

    # ---------------------------------------------------------------- build
    def _build(self, spec, spec_cross, X, Xs, delta, jitter):
        o, nb = self.ops, self.nb
        lmin = np.inf
        for I in self.my_blocks:
            o.gram_rows(self.A[self.loff[I]:self.loff[I] + nb], X, self.N, I * nb, nb, spec)
            nv = max(min(nb, self.N - I * nb), 0)
            if nv > 0:
                lmin = min(lmin, o.diag_min(self._diag(I), nv)[0])
        for t, c in enumerate(self.my_chunks):
            o.rhs_rows(self.A[self.rows_mat + t * 128:self.rows_mat + (t + 1) * 128], c, Xs, self.M, X, self.N, spec_cross, delta)
        # tt_to_cov (tensors.py:95-98): min over the WHOLE diagonal
        t = self.torch.tensor([lmin if np.isfinite(lmin) else 1e300], dtype=self.torch.float64)
        gmin = float(self._allreduce(t, 'min')[0])
        add = jitter
        if not gmin > 0:
            add += float(np.float32(1e-6)) - gmin
        if add != 0.0:
            for I in self.my_blocks:
                nv = max(min(nb, self.N - I * nb), 0)
                if nv > 0:
                    o.diag_add(self._diag(I), nv, add)

    def _count(self, kind, t, peers, tag=None):
        """bytes this rank moves over the fabric for one collective on tensor t (sent + received); with
        self.coll_trace a list, the collective is also logged as (kind, payload bytes, root / op) -- the test that keeps
        this driver and the native one (g3_dist.hip) from drifting compares the two logs"""
        c = self.comm[kind]
        c['calls'] += 1
        c['bytes'] += int(t.numel() * t.element_size() * peers)
        tr = getattr(self, 'coll_trace', None)
        if tr is not None:
            tr.append((kind, int(t.numel() * t.element_size()), tag))

    def _allreduce(self, t, op='sum'):
        dist = self.dist
        if not self.use_coll:
            return t
        import time
        opmap = {'sum': dist.ReduceOp.SUM, 'min': dist.ReduceOp.MIN, 'max': dist.ReduceOp.MAX}
        dev_t = t.to(self.A.device) if self.A.is_cuda else t
        t0 = time.perf_counter()
        dist.all_reduce(dev_t, op=opmap[op], group=self.g_main)
        out = dev_t.cpu()
        self.comm['allreduce']['wait_s'] += time.perf_counter() - t0
        self._count('allreduce', dev_t, 2 * (self.world - 1) / max(self.world, 1), op)
        return out

    def _bcast(self, t, src, async_op):
        if not self.use_coll:
            return None
        self._count('bcast', t, 1 if self.world > 1 else 0, int(src))
        return self.dist.broadcast(t, src=src, group=self.g_bcast, async_op=async_op)

    def _allgather(self, out, inp):
        """out (world * rows x nb) <- every rank's inp (rows x nb); asynchronous"""
        if not self.use_coll:
            out.copy_(inp)
            return None
        self._count('allgather', inp, 2 * (self.world - 1))
        if hasattr(self.dist, 'all_gather_into_tensor'):
            try:
                return self.dist.all_gather_into_tensor(out, inp, group=self.g_main, async_op=True)
            except (RuntimeError, NotImplementedError):
                pass
        chunks = list(out.view(self.world, inp.shape[0], inp.shape[1]).unbind(0))
        return self.dist.all_gather(chunks, inp, group=self.g_main, async_op=True)

    def _wait(self, work, kind=None):
        """make the current stream wait for the collective (RCCL: a stream dependency, the host does not block;
        gloo: the host blocks); the host time spent here is kept per kind"""
        if work is not None and hasattr(work, 'wait'):
            import time
            t0 = time.perf_counter()
            work.wait()
            if kind:
                self.comm[kind]['wait_s'] += time.perf_counter() - t0

    def comm_stats(self):
        """per-rank collective counts, bytes moved (sent + received over the fabric) and host seconds spent
        waiting in them since the last call"""
        out = {k: dict(v) for k, v in self.comm.items()}
        for v in self.comm.values():
            v.update(calls=0, bytes=0, wait_s=0.0)
        return out

    # ---------------------------------------------------------------- factorisation + solves in one sweep
    def _factor_block(self, k):
        """owner only: factor diagonal block k into the broadcast buffer"""
        o = self.ops
        L, W = self._LW(k)
        D = self._diag(k)
        L.copy_(D)
        o.potrf_block(L, self.nb, W)
        D.copy_(L)                                       # kept for the log-determinant
        if self.fullinv:
            o.invert_block(L, self.nb, W, self._V(k))

    def _solve_and_gather(self, k):
        """panel k: solve my rows below block k (right-hand-side rows included) against L_kk, then start the
        all-gather of those rows into gath[k % 2]; returns the collective's work handle (None on the last block)"""
        o, nb, P, A = self.ops, self.nb, self.world, self.A
        L, W = self._LW(k)
        c0, c1 = k * nb, (k + 1) * nb
        r_lo = sum(1 for I in self.my_blocks if I <= k) * nb
        if self.fullinv:
            o.solve_full(self._V(k), nb, A[r_lo:, c0:c1], A.shape[0] - r_lo)
        else:
            o.trsm(L, nb, W, A[r_lo:, c0:c1], A.shape[0] - r_lo)
        if self.nblk - 1 - k <= 0:
            return None
        cnt, _ = self._perm(k)
        send, gath = self.send[k % 2], self.gath[k % 3]
        mine = self.rows_mat - r_lo
        if mine > 0:
            send[:mine].copy_(A[r_lo:self.rows_mat, c0:c1])
        return self._allgather(gath[:P * cnt * nb], send[:cnt * nb])

    def _lookahead(self, j, ev):
        """diagonal block j: its owner applies the update with panel j - 1 from its own panel rows (the
        updates with the panels before it arrived with the bulk stream's column launches, event `ev`), factors
        the block and broadcasts factor + block inverses -- all on the look-ahead stream; the others post the
        receive"""
        o, nb = self.ops, self.nb
        if self.owner(j) != self.rank:
            return self._bcast(self._bc(j), self.owner(j), async_op=True)
        with o.lookahead():
            o.wait_event(ev)
            lo = self.loff[j]
            Pn = self.A[lo:lo + nb, (j - 1) * nb:j * nb]
            o.gemm_sub(self.A[lo:lo + nb, j * nb:(j + 1) * nb], Pn, Pn, nb, nb, nb, lower_only=True)
            self._factor_block(j)
            return self._bcast(self._bc(j), self.owner(j), async_op=True)

    def factor(self, spec, spec_cross, X, Xs, delta, jitter=0.0):
        """returns the global potrf info (0 = success).

        Software-pipelined over the row-block steps on three streams, the schedule of the one-GPU sweep
        (g3_potrf.hip::potrf_lookahead) with the collectives in it.  With G_k the gathered panel k, step k is

          bulk stream   d1. block column k+2 of everything I own from block k+2 down (right-hand-side rows
                            included) -= P_k G_k^T; its completion is event B_k
                        d2. block columns >= k+3 of my blocks >= k+3 and of the right-hand-side rows: the
                            bulk of the work, ONE staircase launch
          caller's      a.  (after B_{k-1}) block column k+1 of my blocks >= k+2 -= P_k G_k^T
          stream        b.  (after the broadcast of L_{k+1,k+1}) solve my rows of panel k+1, start their
                            all-gather
          look-ahead    c.  (after B_k) the owner of block k+2 applies panel k+1 to its diagonal block from
          stream            its own rows, factors it and broadcasts it (on the broadcast's own communicator)

        so while d2 of step k streams through the chip, a - c of step k+1 and the collectives run beside it.
        """
        o, nb, P, A = self.ops, self.nb, self.world, self.A
        self._build(spec, spec_cross, X, Xs, delta, jitter)
        o.reset_info()
        if self.owner(0) == self.rank:
            self._factor_block(0)
        self._wait(self._bcast(self._bc(0), self.owner(0), async_op=False), 'bcast')
        work_g = self._solve_and_gather(0)
        work_b = self._lookahead(1, None) if self.nblk > 1 else None
        ev_prev = None                                   # B_{k-1}
        rhs = [self.rows_rhs] if self.rows_rhs > 0 else []
        for k in range(self.nblk - 1):
            c0, c1, c2, c3 = k * nb, (k + 1) * nb, (k + 2) * nb, (k + 3) * nb
            cnt, perm = self._perm(k)
            self._wait(work_g, 'allgather')
            G = self.gath[k % 3][:P * cnt * nb]
            ev_k = None
            with o.bulk():
                if k + 2 < self.nblk:
                    # d1. block column k+2 (block k+2's diagonal block included: the look-ahead adds panel k+1 only)
                    mine = [I for I in self.my_blocks if I >= k + 2]
                    seg_rows = [nb] * len(mine) + rhs
                    if seg_rows:
                        lo = self.loff[mine[0]] if mine else self.rows_mat
                        o.gemm_sub_stair(A[lo:, c2:c3], A[lo:, c0:c1], G, nb, seg_rows, [nb] * len(seg_rows), nb, perm[1:2],
                                         seg_diag=[1 if I == k + 2 else 0 for I in mine] + [0] * len(rhs))
                    ev_k = o.mark()
                    # d2. the rest
                    mine = [I for I in self.my_blocks if I >= k + 3]
                    seg_rows = [nb] * len(mine) + rhs
                    seg_cols = [(I - k - 2) * nb for I in mine] + ([(self.nblk - k - 3) * nb] if rhs else [])
                    if seg_rows and max(seg_cols) > 0:
                        lo = self.loff[mine[0]] if mine else self.rows_mat
                        o.gemm_sub_stair(A[lo:, c3:], A[lo:, c0:c1], G, nb, seg_rows, seg_cols, nb, perm[2:],
                                         seg_diag=[1] * len(mine) + [0] * len(rhs))
            # a. block column k+1 (it carries the updates up to panel k-1 once B_{k-1} has fired)
            o.wait_event(ev_prev)
            mine = [I for I in self.my_blocks if I >= k + 2]
            seg_rows = [nb] * len(mine) + rhs
            if seg_rows:
                lo = self.loff[mine[0]] if mine else self.rows_mat
                o.gemm_sub_stair(A[lo:, c1:c2], A[lo:, c0:c1], G, nb, seg_rows, [nb] * len(seg_rows), nb, perm[:1])
            # b. panel k+1
            self._wait(work_b, 'bcast')
            if self.owner(k + 1) == self.rank:
                o.join_lookahead()
            work_g = self._solve_and_gather(k + 1)
            # c. diagonal block k+2
            work_b = self._lookahead(k + 2, ev_k) if k + 2 < self.nblk else None
            ev_prev = ev_k
        o.join_bulk()
        o.join_lookahead()
        t = self._allreduce(self.torch.tensor([float(o.read_info())], dtype=self.torch.float64), 'max')
        return int(t[0])

    def factor_robust(self, spec, spec_cross, X, Xs, delta):
        """CholeskyRobust's schedule (tensors.py:197-222) around the distributed factorisation"""
        info = self.factor(spec, spec_cross, X, Xs, delta)
        tries, fallback = 0, False
        if info != 0:
            # jitter from the diagonal of the (lifted) covariance: rebuild and gather mean / min
            self._build(spec, spec_cross, X, Xs, delta, 0.0)
            s, cnt, mn = 0.0, 0, np.inf
            for I in self.my_blocks:
                nv = max(min(self.nb, self.N - I * self.nb), 0)
                if nv > 0:
                    a, b = self.ops.diag_min(self._diag(I), nv)
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
                # CholeskyRobust.perform never raises: the factor becomes 1e-10 * I (tensors.py:215-222),
                # identity on the padding; the right-hand-side rows are rebuilt and solved against it
                fallback = True
                c = float(np.float32(1e-10))
                o, nb = self.ops, self.nb
                o.fill_zero(self.A[:self.rows_mat])
                for I in self.my_blocks:
                    nv = max(min(nb, self.N - I * nb), 0)
                    D = self._diag(I)
                    if nv > 0:
                        o.diag_add(D, nv, c)
                    if nv < nb:
                        o.diag_add(D[nv:, nv:], nb - nv, 1.0)
                for t, ch in enumerate(self.my_chunks):
                    o.rhs_rows(self.A[self.rows_mat + t * 128:self.rows_mat + (t + 1) * 128], ch, Xs, self.M, X, self.N,
                               spec_cross, delta)
                if self.rows_rhs > 0:
                    o.scale_cols(self.A[self.rows_mat:], self.N, 1.0 / c)
        self.last.update(info=info, tries=tries, fallback=fallback)
        return info

    def stats(self):
        """(logdet, quad, mean_pieces[M], ss[M]) summed over all ranks"""
        o, M, nb, Np = self.ops, self.M, self.nb, self.Np
        acc = np.zeros(2 + 2 * M)
        for I in self.my_blocks:
            acc[0] += o.logdet_block(self._diag(I), max(min(nb, self.N - I * nb), 0))
        # a = L^-1 delta is row 0 of right-hand-side chunk 0 (rank 0); everyone needs it for V a
        a = o.zeros(1, Np)
        if 0 in self.my_chunks:
            a.copy_(self.A[self.rows_mat:self.rows_mat + 1])
        self._wait(self._bcast(a, 0, async_op=False))            # chunk 0 lives on rank 0 (chunks are dealt c % world)
        if self.rank == 0:
            acc[1] = float((a[0].double() * a[0].double()).sum().item())
        for t, c in enumerate(self.my_chunks):
            if c == 0:
                continue
            s0 = (c - 1) * 128
            m = max(min(128, M - s0), 0)
            if m > 0:
                V = self.A[self.rows_mat + t * 128:self.rows_mat + t * 128 + m]
                dot, ss = o.rows_dot(V, a, Np)
                acc[2 + s0:2 + s0 + m] += dot
                acc[2 + M + s0:2 + M + s0 + m] += ss
        t = self._allreduce(self.torch.from_numpy(acc), 'sum').numpy()
        return float(t[0]), float(t[1]), t[2:2 + M], t[2 + M:]

    # ---------------------------------------------------------------- posterior covariance and draws
    def posterior_draws(self, spec_pred, Xs, loc, Z):
        """Posterior covariance K(Xs, Xs) - V V^T (elliptical.py:86-91), its robust Cholesky
        (elliptical.py:88,92; tensors.py:197-222) and the draws loc + L_post Z (gaussian.py:75-97,
        before the mapping) -- BASELINE config 5's extra work.

        V = K(Xs, X) L^-T sits in the right-hand-side chunks of the factored local matrices, full rows
        per chunk.  All-gather V (M x N: 1 GiB at config 5), every rank forms the covariance rows of
        ITS chunks with one MFMA GEMM against the gathered V (rank-major: block table again), the
        M x M covariance is all-gathered (64 MiB) and factored redundantly on every rank -- M^3/3
        flops, nothing to exchange -- so all ranks hold the same draws.  spec_pred is the kernel of
        the latent process f (posterior_kernel_f_space, elliptical.py:90-92: the prior part is the
        plain f_kernel.cov, not scrubbed), loc the posterior location."""
        o, P, M, Np, pad = self.ops, self.world, self.M, self.Np, 128
        nch = self.Mp // pad                                       # chunks of Xs rows (right-hand-side chunks 1..nch)
        owner = [(c + 1) % P for c in range(nch)]
        cmax = max(max(sum(1 for c in range(nch) if owner[c] == q) for q in range(P)), 1)
        mine = [c for c in range(nch) if owner[c] == self.rank]
        # my V chunks, in chunk order (chunk c of Xs = right-hand-side chunk c + 1)
        send = o.zeros(cmax * pad, Np)
        for i, c in enumerate(mine):
            t = self.my_chunks.index(c + 1)
            send[i * pad:(i + 1) * pad].copy_(self.A[self.rows_mat + t * pad:self.rows_mat + (t + 1) * pad])
        Vall = o.zeros(P * cmax * pad, Np)
        self._wait(self._allgather(Vall, send))
        seen, perm = [0] * P, []
        for c in range(nch):
            perm.append(owner[c] * cmax + seen[owner[c]])
            seen[owner[c]] += 1
        # covariance rows of my chunks: K(Xs_c, Xs) - V_c Vall^T
        rows = o.zeros(cmax * pad, self.Mp)
        for i, c in enumerate(mine):
            m = max(min(pad, M - c * pad), 0)
            if m > 0:
                o.gram_block(rows[i * pad:(i + 1) * pad], Xs[c * pad:], m, Xs, M, spec_pred, scrub=False)
        if mine:
            o.gemm_sub_stair(rows, send, Vall, Np, [len(mine) * pad], [self.Mp], pad, perm)
        call = o.zeros(P * cmax * pad, self.Mp)
        self._wait(self._allgather(call, rows))
        cov = o.zeros(self.Mp, self.Mp)
        for c in range(nch):
            cov[c * pad:(c + 1) * pad].copy_(call[perm[c] * pad:(perm[c] + 1) * pad])
        draws, tries, fb = o.chol_draws(cov, M, loc, Z)
        self.last.update(cov_tries=tries, cov_fallback=fb)
        return draws

    def step(self, spec_noise, spec_f, X, Xs, delta, Z=None, loc_prior=None):
        """one pass of the hot path; returns logp (mean / variance pieces in self.last).  With Z (M x S
        standard normals, gaussian.py:91) also the posterior covariance, its Cholesky and the latent
        draws loc + L_post Z (self.last['draws']; the caller applies the mapping)"""
        Xt = X._keep if hasattr(X, '_keep') and X._keep is not None else X
        Xst = Xs._keep if hasattr(Xs, '_keep') and Xs._keep is not None else Xs
        dt = delta._keep if hasattr(delta, '_keep') and delta._keep is not None else delta
        self.factor_robust(spec_noise, spec_f, Xt, Xst, dt.reshape(-1))
        logdet, quad, mean, ss = self.stats()
        if Z is not None:
            loc = mean if loc_prior is None else np.asarray(loc_prior, dtype=np.float64) + mean
            self.last['draws'] = self.posterior_draws(spec_f, Xst, loc, Z)
        self.ops.sync()
        logp = -0.5 * self.N * np.log(2 * np.pi) - 0.5 * quad - logdet
        self.last.update(logdet=logdet, quad=quad, mean=mean, ss=ss, logp=logp)
        return logp


# --------------------------------------------------------------------------- the driver inside the library
class NativeDistributedGP:
    """The multi-GPU evaluation behind the C ABI (g3_dist_*, g3py_amd/csrc/g3_dist.hip): the library owns the
    three HIP streams, the device buffers, the per-panel loop and the RCCL communicators (two: panel all-gathers and
    diagonal-factor broadcasts never share a queue).  This class only boots it -- rank 0 makes the communicator
    ids, `dist` (any torch.distributed backend, gloo is enough) carries their 2 x 128 bytes to the other ranks --
    and wraps one call per evaluation; there is no Python loop over panels.  Same `step()` / `last` contract as
    DistributedGP, which stays as the readable reference of the schedule (and runs on CPU with a test double).

    transport='callbacks' (tests only): the collectives are served by `dist` through host staging, so several ranks
    can share ONE GPU -- RCCL refuses that -- and the library's schedule is checked for world > 1 on a one-GPU box.
    Every collective there is a blocking host call behind a stream synchronisation (a debugging aid);
    transport='callbacks_async' is the faithful one: the library's two worker threads serve the panel all-gathers and the
    diagonal-factor broadcasts from two gloo groups while the host thread runs ahead and the three streams keep working
    -- the stream semantics of the RCCL path (g3_dist_create_callbacks_async, include/g3hip.h)."""

    def __init__(self, dev, dist, rank, world, N, d, M, nb=512, dtype=np.float64, transport='rccl', reference=None, keep=False):
        import ctypes as C
        from . import _lib
        from .device import compile_spec
        self._C, self._lib, self._compile = C, _lib, compile_spec
        self.dev, self.dist, self.rank, self.world = dev, dist, rank, world
        self.N, self.d, self.M = N, d, M
        self.nb = max(128, (nb // 128) * 128)
        self.dtype = np.dtype(dtype)
        self._dt = _lib.dtype_code(self.dtype)
        self.last = {}
        self.grad = False
        self.h = C.c_void_p()
        lib = dev.lib
        if transport == 'rccl':
            # Collective-safe start-up: EVERY rank probes the library (g3_dist_unique_id loads librccl and makes an id; only
            # rank 0's ids are used), the ranks exchange what they found, and either all go on or all raise the same error --
            # a rank that cannot load RCCL must never leave its peers inside a broadcast or inside ncclCommInitRank.
            bufs, why = [], ''
            for _ in range(2):
                b = C.create_string_buffer(_lib.G3_DIST_ID_BYTES)
                rc = lib.g3_dist_unique_id(b)
                if rc:
                    why = 'g3_dist_unique_id failed (%d) on rank %d: is librccl available? (G3_RCCL_PATH)' % (rc, rank)
                    break
                bufs.append(bytes(b.raw))
            votes = [why]
            if world > 1:
                votes = [None] * world
                dist.all_gather_object(votes, why)
            bad = [w for w in votes if w]
            if bad:
                raise _lib.G3Error('native multi-GPU driver unavailable on %d of %d ranks: %s' % (len(bad), world, bad[0]))
            ids = [bufs if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            a, b = ids[0]
            rc = lib.g3_dist_create(dev.ctx, a, b, rank, world, C.byref(self.h))
            msg = ('g3_dist_create failed (%d) on rank %d: %s' % (rc, rank, (lib.g3_last_error(dev.ctx) or b'').decode())) if rc else ''
            outcome = [msg]
            if world > 1:
                outcome = [None] * world
                dist.all_gather_object(outcome, msg)
            bad = [w for w in outcome if w]
            if bad:                                    # every rank leaves together, with its communicators released
                if not rc and self.h:
                    lib.g3_dist_destroy(self.h)
                    self.h = C.c_void_p()
                raise _lib.G3Error(bad[0])
        elif transport == 'callbacks':
            self._cb = self._make_callbacks()
            rc = lib.g3_dist_create_callbacks(dev.ctx, C.byref(self._cb), rank, world, C.byref(self.h))
            if rc:
                raise _lib.G3Error('g3_dist_create_callbacks failed (%d)' % rc)
        elif transport == 'callbacks_async':
            self._cb = self._make_host_callbacks()
            rc = lib.g3_dist_create_callbacks_async(dev.ctx, C.byref(self._cb), rank, world, C.byref(self.h))
            if rc:
                raise _lib.G3Error('g3_dist_create_callbacks_async failed (%d)' % rc)
        elif transport == 'replay':
            # measurement: this object plays rank `rank` of a `world`-rank evaluation alone on the GPU; `reference` is a
            # world-1 NativeDistributedGP created with keep=True that has evaluated the same problem (g3hip.h)
            if reference is None or not reference.h:
                raise ValueError('the replay transport needs the reference driver')
            self._reference = reference                    # (must outlive this object)
            rc = lib.g3_dist_create_replay(dev.ctx, reference.h, rank, world, C.byref(self.h))
            if rc:
                raise _lib.G3Error('g3_dist_create_replay failed (%d): the reference must be a planned world-1 driver with '
                                   'keep=True' % rc)
        else:
            raise ValueError(transport)
        self.transport = transport
        if keep:
            self._chk(lib.g3_dist_set_keep(self.h, 1), 'g3_dist_set_keep')
        self._chk(lib.g3_dist_plan(self.h, N, d, M, self.nb, self._dt), 'g3_dist_plan')

    def _chk(self, rc, what):
        if rc:
            raise self._lib.G3Error('%s failed (%d): %s' % (what, rc, (self.dev.lib.g3_dist_last_error(self.h) or b'').decode()))

    def _make_callbacks(self):
        """collectives for the test transport: device buffer -> host (C ABI memcpy) -> torch.distributed on CPU
        tensors -> device"""
        import torch
        C, lib, dev, dist, world = self._C, self.dev.lib, self.dev, self.dist, self.world

        def d2h(ptr, n):
            a = np.empty(n, dtype=np.uint8)
            assert lib.g3_memcpy_d2h(dev.ctx, a.ctypes.data, ptr, n) == 0
            return a

        def h2d(ptr, a):
            assert lib.g3_memcpy_h2d(dev.ctx, ptr, a.ctypes.data, a.nbytes) == 0

        self.coll_trace = None      # a list: every collective the LIBRARY asks for is logged as (kind, bytes, root / op)

        def bcast(user, buf, nbytes, root):
            try:
                if self.coll_trace is not None:
                    self.coll_trace.append(('bcast', int(nbytes), int(root)))
                t = torch.from_numpy(d2h(buf, nbytes))
                if world > 1:
                    dist.broadcast(t, src=root)
                h2d(buf, t.numpy())
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def allgather(user, sendp, recvp, nbytes):
            try:
                if self.coll_trace is not None:
                    self.coll_trace.append(('allgather', int(nbytes), None))
                t = torch.from_numpy(d2h(sendp, nbytes))
                out = [torch.empty_like(t) for _ in range(world)]
                if world > 1:
                    dist.all_gather(out, t)
                else:
                    out = [t]
                h2d(recvp, np.ascontiguousarray(np.concatenate([o.numpy() for o in out])))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def allreduce(user, vals, n, op):
            try:
                a = np.ctypeslib.as_array(vals, shape=(n,))
                if self.coll_trace is not None:
                    self.coll_trace.append(('allreduce', int(n) * 8, ('sum', 'min', 'max')[op]))
                self.last_allreduce_in = a.copy()      # this rank's own contribution (tests compare it with a replay)
                t = torch.from_numpy(a.copy())
                if world > 1:
                    dist.all_reduce(t, op=[dist.ReduceOp.SUM, dist.ReduceOp.MIN, dist.ReduceOp.MAX][op])
                a[:] = t.numpy()
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._cb_keep = (self._lib.DIST_BCAST_CB(bcast), self._lib.DIST_ALLGATHER_CB(allgather),
                         self._lib.DIST_ALLREDUCE_CB(allreduce))
        return self._lib.DistCallbacks(None, *self._cb_keep)

    def _make_host_callbacks(self):
        """collectives for the asynchronous test transport: the library hands HOST staging buffers to these callbacks from
        its two worker threads -- `bcast` from the broadcast worker, `allgather` / `allreduce` from the gather worker -- so
        each kind gets a gloo group of its own (two communicators, as in the RCCL path) and works in place on the buffer"""
        import torch
        C, dist, world = self._C, self.dist, self.world
        self.coll_trace = None
        self._g_gather = self._g_bcast = None
        if world > 1:
            ranks = list(range(world))
            self._g_gather = dist.new_group(ranks=ranks, backend='gloo')
            self._g_bcast = dist.new_group(ranks=ranks, backend='gloo')

        def view(ptr, n):
            return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * n).from_address(ptr)))

        def bcast(user, buf, nbytes, root):
            try:
                if self.coll_trace is not None:
                    self.coll_trace.append(('bcast', int(nbytes), int(root)))
                if world > 1:
                    dist.broadcast(view(buf, nbytes), src=root, group=self._g_bcast)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def allgather(user, sendp, recvp, nbytes):
            try:
                if self.coll_trace is not None:
                    self.coll_trace.append(('allgather', int(nbytes), None))
                if world > 1:
                    out = view(recvp, nbytes * world)
                    mine = view(sendp, nbytes).clone()          # (the send part lies inside the receive buffer)
                    dist.all_gather(list(out.view(world, nbytes).unbind(0)), mine, group=self._g_gather)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def allreduce(user, vals, n, op):
            try:
                a = np.ctypeslib.as_array(vals, shape=(n,))
                if self.coll_trace is not None:
                    self.coll_trace.append(('allreduce', int(n) * 8, ('sum', 'min', 'max')[op]))
                self.last_allreduce_in = a.copy()
                if world > 1:
                    t = torch.from_numpy(a.copy())
                    dist.all_reduce(t, op=[dist.ReduceOp.SUM, dist.ReduceOp.MIN, dist.ReduceOp.MAX][op], group=self._g_gather)
                    a[:] = t.numpy()
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        self._cb_keep = (self._lib.DIST_BCAST_CB(bcast), self._lib.DIST_ALLGATHER_CB(allgather),
                         self._lib.DIST_ALLREDUCE_CB(allreduce))
        return self._lib.DistHostCallbacks(None, *self._cb_keep)

    @staticmethod
    def _ptr_ld(a):
        """(device pointer, row stride in elements) of a DeviceArray or a 2-D / 1-D torch tensor"""
        if hasattr(a, 'ptr'):
            return a.ptr, a.ld
        return a.data_ptr(), (a.stride(0) if a.dim() == 2 else a.shape[-1])

    def step(self, spec_noise, spec_f, X, Xs, delta, Z=None, loc_prior=None):
        """one pass of the hot path on all ranks; returns logp (mean / variance pieces in self.last).  With Z (M x S
        standard normals, gaussian.py:91) also the posterior covariance, its Cholesky and the latent draws
        loc + L_post Z (self.last['draws']; the caller applies the mapping)"""
        C, lib = self._C, self.dev.lib
        pn, pf = self._compile(spec_noise, self.d), self._compile(spec_f, self.d)
        xp, ldx = self._ptr_ld(X)
        xsp, ldxs = self._ptr_ld(Xs)
        dp, _ = self._ptr_ld(delta)
        out = (C.c_double * 5)()
        M = self.M
        mean = np.zeros(max(M, 1))
        ss = np.zeros(max(M, 1))
        self._chk(lib.g3_dist_gp_factor_predict(self.h, C.byref(pn), C.byref(pf), xp, ldx, dp, xsp, ldxs, out,
                                                mean.ctypes.data_as(C.POINTER(C.c_double)),
                                                ss.ctypes.data_as(C.POINTER(C.c_double))), 'g3_dist_gp_factor_predict')
        logdet, quad = out[0], out[1]
        mean, ss = mean[:M], ss[:M]
        self.last.update(logdet=logdet, quad=quad, info=int(out[2]), tries=int(out[3]), fallback=bool(out[4]), mean=mean, ss=ss)
        if Z is not None:
            loc = mean if loc_prior is None else np.asarray(loc_prior, dtype=np.float64) + mean
            self.draws(spec_f, Xs, loc, Z)
        logp = -0.5 * self.N * np.log(2 * np.pi) - 0.5 * quad - logdet
        self.last['logp'] = logp
        return logp

    def draws(self, spec_f, Xs, loc, Z):
        """after step() with the same Xs: posterior covariance K_f(Xs, Xs) - V V^T, its robust Cholesky and the latent
        draws loc + L_post Z (gaussian.py:75-97, before the mapping); every rank returns the same M x S matrix"""
        C, lib = self._C, self.dev.lib
        pf = self._compile(spec_f, self.d)
        xsp, ldxs = self._ptr_ld(Xs)
        M = self.M
        Zc = np.ascontiguousarray(Z, dtype=self.dtype)
        S = Zc.shape[1]
        locc = np.ascontiguousarray(loc, dtype=self.dtype)
        out = np.empty((M, S), dtype=self.dtype)
        tries, fb = C.c_int(0), C.c_int(0)
        self._chk(lib.g3_dist_posterior_draws(self.h, C.byref(pf), xsp, ldxs, locc.ctypes.data, Zc.ctypes.data, S,
                                              out.ctypes.data, C.byref(tries), C.byref(fb)), 'g3_dist_posterior_draws')
        self.last.update(draws=out.astype(np.float64), cov_tries=tries.value, cov_fallback=bool(fb.value))
        return self.last['draws']

    def posterior_cov(self, spec, Xs, out):
        """after step() with the same Xs: the posterior covariance K(Xs, Xs) - V V^T (elliptical.py:86-91) into the device
        matrix `out` (roundup(M, 128) square) on every rank; spec with or without the Noise term"""
        C, lib = self._C, self.dev.lib
        pk = self._compile(spec, self.d)
        xsp, ldxs = self._ptr_ld(Xs)
        op, ldo = self._ptr_ld(out)
        self._chk(lib.g3_dist_posterior_cov(self.h, C.byref(pk), xsp, ldxs, op, ldo), 'g3_dist_posterior_cov')
        return out

    def set_grad(self, on=True):
        """gradient mode: every factorisation also carries the identity as right-hand-side rows (the rank's rows of
        L^-T), which `dlogp` needs; costs N^3 / 3 more flops per step over all ranks and doubles the local matrix"""
        self._chk(self.dev.lib.g3_dist_set_grad(self.h, 1 if on else 0), 'g3_dist_set_grad')
        self.grad = bool(on)

    def dlogp(self, spec_noise, X, alpha_scale=1.0):
        """after step() in gradient mode: (prog, gmap, slots, alpha) -- the per-leaf parameter sums
        1/2 sum_ij (alpha_i alpha_j - K^-1_ij) dK_ij/dparam of g3_gp_dlogp and alpha = alpha_scale * K^-1 delta,
        identical on every rank (stochastic.py:308-309 through tensors.py:224-260 in the reference)"""
        C, lib = self._C, self.dev.lib
        pn = self._compile(spec_noise, self.d)
        gmap = self.dev.grad_layout(pn)
        xp, ldx = self._ptr_ld(X)
        slots = (C.c_double * max(gmap.nslots, 1))()
        alpha = np.zeros(self.N)
        self._chk(lib.g3_dist_gp_dlogp(self.h, C.byref(pn), C.byref(gmap), xp, ldx, float(alpha_scale), slots,
                                       alpha.ctypes.data_as(C.POINTER(C.c_double))), 'g3_dist_gp_dlogp')
        return pn, gmap, np.array(slots[:gmap.nslots]), alpha

    def comm_stats(self):
        """per collective kind since the last call: calls, bytes sent + received by this rank, device milliseconds
        inside the collectives (HIP events on the stream each ran on)"""
        out = (self._C.c_double * 9)()
        self._chk(self.dev.lib.g3_dist_comm_stats(self.h, out), 'g3_dist_comm_stats')
        return {k: {'calls': out[3 * i], 'bytes': out[3 * i + 1], 'device_ms': out[3 * i + 2]}
                for i, k in enumerate(('bcast', 'allgather', 'allreduce'))}

    def phase_stats(self):
        """as of the last comm_stats(): diagonal blocks this rank updated + factored and the device milliseconds of
        that, panel solves of its rows and their device milliseconds"""
        out = (self._C.c_double * 4)()
        self._chk(self.dev.lib.g3_dist_phase_stats(self.h, out), 'g3_dist_phase_stats')
        return {'diag': {'calls': out[0], 'device_ms': out[1]}, 'solve': {'calls': out[2], 'device_ms': out[3]}}

    def prof_enable(self, on=2):
        """HIP events around the MFMA GEMM launches of the driver's bulk stream (its staircase updates)"""
        self._chk(self.dev.lib.g3_dist_prof_enable(self.h, int(on)), 'g3_dist_prof_enable')

    def prof_collect(self):
        n = len(self._lib.PROF_TAGS)
        out = (self._C.c_double * (3 * n))()
        self._chk(self.dev.lib.g3_dist_prof_collect(self.h, out), 'g3_dist_prof_collect')
        return {t: {'count': int(out[3 * i]), 'ms': out[3 * i + 1], 'work': out[3 * i + 2]} for i, t in enumerate(self._lib.PROF_TAGS)}

    def close(self):
        """destroy the driver (communicators, streams, buffers).  Idempotent; a no-op once the device context it lives
        on has been closed (Device.close() releases the context first only at interpreter exit)"""
        if self.h and self.dev.ctx:
            self.dev.lib.g3_dist_destroy(self.h)
        self.h = self._C.c_void_p()

    def __del__(self):
        try:                              # (no import here: at interpreter shutdown the import machinery is already gone)
            if _sys is None or _sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------- replicas
def logp_chain_sharded(process, chain, dist, rank, world, prior=False, torch_device=None):
    """logp of every row of a flat-parameter chain with the rows dealt round-robin to the ranks.

    The second way the path scales: `find_MAP` restarts, `fixed_logp`,
    `sample_hypers` evaluate MANY independent hyper-parameter vectors on the same observations
    (g3py/processes/stochastic.py:515-564, 740-783 -- a process pool in the reference).  These are
    replicas: every rank holds the same `process` (same observations) on its own GPU, evaluates rows
    rank, rank + world, ... through its batched sweep (`logp_chain` -> g3_gp_factor_batched) and the
    scalars are combined by one all-reduce of a vector that is zero outside a rank's own rows.
    No covariance data ever crosses xGMI."""
    import torch
    chain = np.atleast_2d(np.asarray(chain, dtype=np.float64))
    out = np.zeros(len(chain), dtype=np.float64)
    mine = np.arange(rank, len(chain), world)
    if len(mine):
        out[mine] = np.asarray(process.logp_chain(chain[mine], prior=prior), dtype=np.float64)
    if world > 1:
        t = torch.from_numpy(out)
        if torch_device is not None:
            t = t.to(torch_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out = t.cpu().numpy()
    return out
