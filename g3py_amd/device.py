"""Device plumbing over the C ABI: contexts, device arrays, kernel-program compilation.

Nothing here computes on the CPU: every numerical routine is a call into libg3hip.so.
"""
import atexit
import ctypes as C
import sys
import weakref

import numpy as np

from . import _lib
from ._lib import G3Error, KernelProg, roundup


def _check(dev, rc, what):
    if rc != 0:
        msg = ''
        if rc == -1000 and dev is not None and dev.ctx:
            msg = ': ' + (dev.lib.g3_last_error(dev.ctx) or b'').decode()
        raise G3Error('%s failed with status %d%s' % (what, rc, msg))


class DeviceArray:
    """A row-major device matrix (rows x cols, leading dimension ld, in elements)."""

    def __init__(self, dev, ptr, rows, cols, ld, dtype, owned=True, keep=None):
        self.dev, self.ptr, self.rows, self.cols, self.ld = dev, ptr, int(rows), int(cols), int(ld)
        self.dtype = np.dtype(dtype)
        self.owned = owned
        self._keep = keep
        if owned and self.ptr:
            dev._owned.add(self)

    @property
    def nbytes(self):
        return self.rows * self.ld * self.dtype.itemsize

    def offset(self, row, col=0):
        return self.ptr + (row * self.ld + col) * self.dtype.itemsize

    def free(self):
        """hipFree an owned buffer; a no-op once the context is closed (Device.close() has already
        released everything the context owned, and nothing may call into HIP after that)"""
        if self.owned and self.ptr and self.dev.ctx:
            self.dev.lib.g3_free(self.dev.ctx, self.ptr)
        self.ptr = 0

    def __del__(self):
        # never call into the HIP runtime from interpreter finalisation: by then the runtime (or a
        # profiler tool library layered over it) may already be shutting down
        if sys is None or sys.is_finalizing():
            return
        try:
            self.free()
        except Exception:
            pass


class Device:
    """One g3_ctx on one GPU.  Raises if the library or the GPU is missing."""
    _default = {}
    _live = weakref.WeakSet()      # every open context, closed by the atexit hook below

    def __init__(self, index=0):
        self.ctx = None
        self._owned = weakref.WeakSet()
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.g3_ctx_create(int(index), C.byref(h))
        if rc != 0:
            raise G3Error('g3_ctx_create(device=%d) failed with status %d: no usable MI355X; '
                          'g3py_amd has no CPU fallback' % (index, rc))
        self.ctx = h
        self.index = index
        Device._live.add(self)

    @classmethod
    def default(cls, index=0):
        if index not in cls._default:
            cls._default[index] = cls(index)
        return cls._default[index]

    def close(self):
        """Release everything the context owns -- device buffers handed out by alloc()/upload(), the
        library's streams, events, pinned mirrors and workspaces -- while the HIP runtime is alive.
        Idempotent; afterwards DeviceArray.free() is a no-op and any other call raises."""
        if not self.ctx:
            return
        try:
            self.lib.g3_ctx_sync(self.ctx)
        except Exception:
            pass
        for a in list(self._owned):
            try:
                a.free()
            except Exception:
                pass
        self.lib.g3_ctx_destroy(self.ctx)
        self.ctx = None
        Device._live.discard(self)
        for k in [k for k, v in Device._default.items() if v is self]:
            del Device._default[k]

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def close_all(cls):
        """explicit teardown of every live context (registered with atexit: Python's exit hooks run
        before the C runtime's static destructors, i.e. while HIP and any profiler tool are intact)"""
        for d in list(cls._live):
            try:
                d.close()
            except Exception:
                pass
        cls._default.clear()

    def set_stream(self, stream_handle):
        _check(self, self.lib.g3_ctx_set_stream(self.ctx, stream_handle), 'g3_ctx_set_stream')

    def sync(self):
        _check(self, self.lib.g3_ctx_sync(self.ctx), 'g3_ctx_sync')

    # ---- profiling (HIP events on the context's stream)
    def prof_enable(self, on=True):
        _check(self, self.lib.g3_prof_enable(self.ctx, int(on)), 'g3_prof_enable')

    def prof_reset(self):
        _check(self, self.lib.g3_prof_reset(self.ctx), 'g3_prof_reset')

    def prof_collect(self):
        out = (C.c_double * (3 * len(_lib.PROF_TAGS)))()
        _check(self, self.lib.g3_prof_collect(self.ctx, out), 'g3_prof_collect')
        return {t: dict(count=int(out[3 * i]), ms=out[3 * i + 1], work=out[3 * i + 2])
                for i, t in enumerate(_lib.PROF_TAGS)}

    # ---- memory
    def alloc(self, rows, cols, dtype, ld=None, zero=False):
        ld = int(cols if ld is None else ld)
        p = C.c_void_p()
        nbytes = int(rows) * ld * np.dtype(dtype).itemsize
        _check(self, self.lib.g3_malloc(self.ctx, nbytes, C.byref(p)), 'g3_malloc(%d)' % nbytes)
        a = DeviceArray(self, p.value or 0, rows, cols, ld, dtype)
        if zero and nbytes:
            _check(self, self.lib.g3_memset(self.ctx, a.ptr, 0, nbytes), 'g3_memset')
        return a

    def wrap(self, ptr, rows, cols, ld, dtype, keep=None):
        """Borrow foreign device memory (e.g. torch.Tensor.data_ptr())."""
        return DeviceArray(self, int(ptr), rows, cols, ld, dtype, owned=False, keep=keep)

    def upload(self, a, dtype=None, pad_rows=None, pad_cols=None):
        a = np.asarray(a)
        if dtype is not None:
            a = a.astype(dtype, copy=False)
        if a.ndim == 1:
            a = a[None, :]
        rows, cols = a.shape
        pr = rows if pad_rows is None else int(pad_rows)
        pc = cols if pad_cols is None else int(pad_cols)
        if (pr, pc) != (rows, cols):
            b = np.zeros((pr, pc), dtype=a.dtype)
            b[:rows, :cols] = a
            a = b
        a = np.ascontiguousarray(a)
        d = self.alloc(pr, pc, a.dtype)
        if a.nbytes:
            _check(self, self.lib.g3_memcpy_h2d(self.ctx, d.ptr, a.ctypes.data, a.nbytes), 'g3_memcpy_h2d')
        d.rows, d.cols = pr, pc
        return d

    def copy_in(self, d, a):
        """host array -> an existing device buffer of the same shape"""
        a = np.ascontiguousarray(a, dtype=d.dtype)
        if a.nbytes > d.nbytes:
            raise G3Error('copy_in: %d bytes do not fit in a buffer of %d' % (a.nbytes, d.nbytes))
        if a.nbytes:
            _check(self, self.lib.g3_memcpy_h2d(self.ctx, d.ptr, a.ctypes.data, a.nbytes), 'g3_memcpy_h2d')

    def download(self, d, rows=None, cols=None):
        rows = d.rows if rows is None else rows
        cols = d.cols if cols is None else cols
        full = np.empty((d.rows, d.ld), dtype=d.dtype)
        if full.nbytes:
            _check(self, self.lib.g3_memcpy_d2h(self.ctx, full.ctypes.data, d.ptr, full.nbytes), 'g3_memcpy_d2h')
        return np.ascontiguousarray(full[:rows, :cols])

    def gram_path_stats(self):
        """Gram launches of this context so far by kernel: compile-time table / generated at first use / interpreted"""
        out = (C.c_double * 3)()
        _check(self, self.lib.g3_gram_path_stats(self.ctx, out), 'g3_gram_path_stats')
        return {'table': int(out[0]), 'generated': int(out[1]), 'interpreted': int(out[2])}

    def grad_path_stats(self):
        """the same for the gradient's kernel-parameter sums"""
        out = (C.c_double * 3)()
        _check(self, self.lib.g3_grad_path_stats(self.ctx, out), 'g3_grad_path_stats')
        return {'table': int(out[0]), 'generated': int(out[1]), 'interpreted': int(out[2])}

    # ---- kernels (thin, argument-checked wrappers)
    def gram(self, prog, X1, X2, d, out, n1pad, n2pad, flags):
        dt = _lib.dtype_code(out.dtype)
        rc = self.lib.g3_gram(self.ctx, C.byref(prog), X1.ptr, X1.rows, X1.ld,
                              X2.ptr if X2 is not None else None, X2.rows if X2 is not None else 0,
                              X2.ld if X2 is not None else 0, d, dt, out.ptr, out.ld, n1pad, n2pad, flags)
        _check(self, rc, 'g3_gram')

    def gram_diag(self, prog, X, d, out):
        rc = self.lib.g3_gram_diag(self.ctx, C.byref(prog), X.ptr, X.rows, X.ld, d,
                                   _lib.dtype_code(out.dtype), out.ptr)
        _check(self, rc, 'g3_gram_diag')

    def cov_lift(self, K, n):
        _check(self, self.lib.g3_cov_lift(self.ctx, K.ptr, n, K.ld, _lib.dtype_code(K.dtype)), 'g3_cov_lift')

    def scrub(self, A, n1, n2):
        _check(self, self.lib.g3_scrub(self.ctx, A.ptr, n1, n2, A.ld, _lib.dtype_code(A.dtype)), 'g3_scrub')

    def gemm_nt(self, Cm, A, B, m, n, k, alpha=1.0, beta=0.0, lower_only=False, c_off=0, a_off=0, b_off=0):
        rc = self.lib.g3_gemm_nt(self.ctx, Cm.ptr + c_off, Cm.ld, A.ptr + a_off, A.ld, B.ptr + b_off, B.ld,
                                 m, n, k, alpha, beta, _lib.dtype_code(Cm.dtype), int(lower_only))
        _check(self, rc, 'g3_gemm_nt')

    def gemm_nt_stair(self, c_ptr, ldc, a_ptr, lda, b_ptr, ldb, k, seg_rows, seg_cols, dtype, alpha=1.0, beta=0.0,
                      b_block_rows=0, b_perm=None, seg_diag=None):
        """one launch over stacked row segments: segment s gets its first seg_cols[s] columns;
        b_perm[s] = physical position of logical row block s of B (blocks of b_block_rows rows);
        seg_diag[s] != 0: the last seg_rows[s] columns of segment s are its diagonal block (lower triangle wanted)"""
        n = len(seg_rows)
        rows = (C.c_int64 * n)(*[int(v) for v in seg_rows])
        cols = (C.c_int64 * n)(*[int(v) for v in seg_cols])
        perm, nperm = None, 0
        if b_perm is not None:
            nperm = len(b_perm)
            perm = (C.c_int32 * nperm)(*[int(v) for v in b_perm])
        diag = (C.c_int64 * n)(*[int(v) for v in seg_diag]) if seg_diag is not None else None
        rc = self.lib.g3_gemm_nt_stair(self.ctx, c_ptr, ldc, a_ptr, lda, b_ptr, ldb, k, rows, cols, n, alpha, beta,
                                       _lib.dtype_code(dtype), int(b_block_rows), perm, nperm, diag)
        _check(self, rc, 'g3_gemm_nt_stair')

    def potrf(self, A, n):
        info = C.c_int(0)
        rc = self.lib.g3_potrf(self.ctx, A.ptr, n, A.ld, _lib.dtype_code(A.dtype), None, C.byref(info))
        _check(self, rc, 'g3_potrf')
        return info.value

    def potrf_robust(self, K, L, n, maxtries=20):
        tries, fb, jit = C.c_int(0), C.c_int(0), C.c_double(0)
        rc = self.lib.g3_potrf_robust(self.ctx, K.ptr, K.ld, L.ptr, L.ld, n, _lib.dtype_code(K.dtype),
                                      maxtries, C.byref(tries), C.byref(fb), C.byref(jit))
        _check(self, rc, 'g3_potrf_robust')
        return tries.value, bool(fb.value), jit.value

    def trsm_rlt(self, L, n, B, m, have_inverses=False):
        rc = self.lib.g3_trsm_rlt(self.ctx, L.ptr, n, L.ld, B.ptr, m, B.ld, _lib.dtype_code(L.dtype), None)
        _check(self, rc, 'g3_trsm_rlt')

    def logp_terms(self, L, n, a):
        out = (C.c_double * 4)()
        rc = self.lib.g3_logp_terms(self.ctx, L.ptr, n, L.ld, a.ptr if a is not None else None,
                                    _lib.dtype_code(L.dtype), out)
        _check(self, rc, 'g3_logp_terms')
        return list(out)

    def rows_dot_ss(self, V, m, n, a, dot, ss):
        rc = self.lib.g3_rows_dot_ss(self.ctx, V.ptr, m, n, V.ld, a.ptr if a is not None else None,
                                     _lib.dtype_code(V.dtype), dot.ptr if dot is not None else None,
                                     ss.ptr if ss is not None else None)
        _check(self, rc, 'g3_rows_dot_ss')

    def alloc_inverses(self, Np, dtype):
        """buffer for the 128 x 128 diagonal-block inverses that belong to one factor"""
        return self.alloc(Np // _lib.G3_PAD * _lib.G3_PAD, _lib.G3_PAD, dtype)

    def gp_factor(self, prog, X, N, d, delta, K, W, a):
        """K must have roundup(N) + 128 rows (the trailing block carries delta through the factorisation)"""
        if K.rows < roundup(N) + _lib.G3_RHS_PAD:
            raise G3Error('g3_gp_factor needs a K buffer with roundup(N) + 128 rows')
        out = (C.c_double * 6)()
        rc = self.lib.g3_gp_factor(self.ctx, C.byref(prog), X.ptr, N, X.ld, d, delta.ptr,
                                   _lib.dtype_code(K.dtype), K.ptr, K.ld, W.ptr, a.ptr, out)
        _check(self, rc, 'g3_gp_factor')
        return dict(logdet=out[0], quad=out[1], nonfinite=out[2], tries=int(out[3]),
                    fallback=bool(out[4]), info=int(out[5]))

    def gp_factor_predict(self, prog, prog_cross, X, N, d, delta, Xs, M, K, W, a, mu, ss):
        """K must have roundup(N) + 128 + roundup(M, 128) rows"""
        out = (C.c_double * 6)()
        rc = self.lib.g3_gp_factor_predict(self.ctx, C.byref(prog), C.byref(prog_cross), X.ptr, N, X.ld, d, delta.ptr,
                                           Xs.ptr, M, Xs.ld, _lib.dtype_code(K.dtype), K.ptr, K.ld, W.ptr, a.ptr,
                                           mu.ptr, ss.ptr, out)
        _check(self, rc, 'g3_gp_factor_predict')
        return dict(logdet=out[0], quad=out[1], nonfinite=out[2], tries=int(out[3]),
                    fallback=bool(out[4]), info=int(out[5]))

    def gp_factor_batched(self, progs, X, N, d, delta, K, kstride, W, a, raw=False):
        """len(progs) evaluations in one sweep; K holds the members `kstride` elements apart.  `progs`: a list of
        KernelProg or an existing ctypes array of them (long chains: packing 4096 programs costs more than evaluating
        them); raw=True returns the (B, 6) array [logdet, quad, nonfinite, tries, fallback, info] instead of dicts"""
        B = len(progs)
        arr = progs if isinstance(progs, C.Array) else (_lib.KernelProg * B)(*progs)
        out = np.empty((B, 6))
        rc = self.lib.g3_gp_factor_batched(self.ctx, arr, B, X.ptr, N, X.ld, d, delta.ptr, delta.ld,
                                           _lib.dtype_code(K.dtype), K.ptr, K.ld, kstride, W.ptr, a.ptr,
                                           out.ctypes.data_as(C.POINTER(C.c_double)))
        _check(self, rc, 'g3_gp_factor_batched')
        if raw:
            return out
        return [dict(logdet=r[0], quad=r[1], nonfinite=r[2], tries=int(r[3]), fallback=bool(r[4]), info=int(r[5]))
                for r in out]

    def gp_factor_batched_fields(self, tmpl, offsets, fields, X, N, d, delta, K, kstride, W, a):
        """gp_factor_batched with the members given as one template program plus a (B, nfield) float64 matrix of the
        values that differ and the byte offsets in g3_kernel_prog they belong at (see compile_spec_rows); the library
        expands the programs on the device.  Returns the raw (B, 6) statistics."""
        fields = np.ascontiguousarray(fields, dtype=np.float64)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        B, nf = fields.shape
        if len(offsets) != nf:
            raise G3Error('gp_factor_batched_fields: %d offsets for %d fields' % (len(offsets), nf))
        out = np.empty((B, 6))
        rc = self.lib.g3_gp_factor_batched_fields(self.ctx, C.byref(tmpl), B, fields.ctypes.data, offsets.ctypes.data, nf,
                                                  X.ptr, N, X.ld, d, delta.ptr, delta.ld, _lib.dtype_code(K.dtype),
                                                  K.ptr, K.ld, kstride, W.ptr, a.ptr,
                                                  out.ctypes.data_as(C.POINTER(C.c_double)))
        _check(self, rc, 'g3_gp_factor_batched_fields')
        return out

    def gp_cross(self, prog, Xs, M, X, N, d, L, W, a, V, mu, ss):
        rc = self.lib.g3_gp_cross(self.ctx, C.byref(prog), Xs.ptr, M, Xs.ld, X.ptr, N, X.ld, d, L.ptr, L.ld,
                                  W.ptr, a.ptr if a is not None else None, _lib.dtype_code(L.dtype), V.ptr, V.ld,
                                  mu.ptr if mu is not None else None, ss.ptr if ss is not None else None)
        _check(self, rc, 'g3_gp_cross')

    def gp_dlogp_batched(self, progs, gmap, X, N, d, K, kstride, W, a, Y, Kinv, alpha):
        """after gp_factor_batched: per member K^-1, alpha and the kernel-parameter sums; returns (B, nslots)"""
        B = len(progs)
        arr = (_lib.KernelProg * B)(*progs)
        out = (C.c_double * (max(gmap.nslots, 1) * B))()
        rc = self.lib.g3_gp_dlogp_batched(self.ctx, arr, B, C.byref(gmap), X.ptr, N, X.ld, d, K.ptr, K.ld, kstride,
                                          W.ptr, a.ptr, _lib.dtype_code(K.dtype), Y.ptr, Kinv.ptr, alpha.ptr, out)
        _check(self, rc, 'g3_gp_dlogp_batched')
        return np.array(out[:]).reshape(B, max(gmap.nslots, 1))[:, :gmap.nslots]

    def gp_dlogp_batched_fields(self, tmpl, offsets, fields, gmap, X, N, d, K, kstride, W, a, Y, Kinv, alpha):
        """gp_dlogp_batched for members given as template + fields (after gp_factor_batched_fields on the same buffers);
        returns (B, nslots)"""
        fields = np.ascontiguousarray(fields, dtype=np.float64)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        B, nf = fields.shape
        ns = max(gmap.nslots, 1)
        out = np.empty((B, ns))
        rc = self.lib.g3_gp_dlogp_batched_fields(self.ctx, C.byref(tmpl), B, fields.ctypes.data, offsets.ctypes.data, nf,
                                                 C.byref(gmap), X.ptr, N, X.ld, d, K.ptr, K.ld, kstride, W.ptr, a.ptr,
                                                 _lib.dtype_code(K.dtype), Y.ptr, Kinv.ptr, alpha.ptr,
                                                 out.ctypes.data_as(C.POINTER(C.c_double)))
        _check(self, rc, 'g3_gp_dlogp_batched_fields')
        return out[:, :gmap.nslots]

    def gp_sample(self, L, M, loc, Z):
        """loc[:, None] + L Z for host normals Z (M x S); L is the padded device factor"""
        Z = np.ascontiguousarray(Z, dtype=L.dtype)
        loc = np.ascontiguousarray(loc, dtype=L.dtype)
        S = Z.shape[1]
        out = np.empty((M, S), dtype=L.dtype)
        rc = self.lib.g3_gp_sample(self.ctx, L.ptr, M, L.ld, loc.ctypes.data, Z.ctypes.data, S,
                                   _lib.dtype_code(L.dtype), out.ctypes.data)
        _check(self, rc, 'g3_gp_sample')
        return out

    # ---- gradient of logp
    def grad_layout(self, prog):
        m = _lib.GradMap()
        _check(self, self.lib.g3_grad_layout(C.byref(prog), C.byref(m)), 'g3_grad_layout')
        return m

    def potri(self, L, n, W, Y, Kinv):
        rc = self.lib.g3_potri(self.ctx, L.ptr, n, L.ld, W.ptr if W is not None else None,
                               _lib.dtype_code(L.dtype), Y.ptr, Y.ld, Kinv.ptr, Kinv.ld)
        _check(self, rc, 'g3_potri')

    def gram_grad(self, prog, gmap, X, N, d, Kinv, alpha):
        out = (C.c_double * max(gmap.nslots, 1))()
        rc = self.lib.g3_gram_grad(self.ctx, C.byref(prog), C.byref(gmap), X.ptr, N, X.ld, d,
                                   _lib.dtype_code(Kinv.dtype), Kinv.ptr, Kinv.ld, alpha.ptr, out)
        _check(self, rc, 'g3_gram_grad')
        return np.array(out[:gmap.nslots])

    def gram_grad_rows(self, prog, gmap, X, N, d, row0, nrows, Kinv_rows, alpha):
        """rows [row0, row0 + nrows) of gram_grad's sum; Kinv_rows holds those rows of K^-1"""
        out = (C.c_double * max(gmap.nslots, 1))()
        rc = self.lib.g3_gram_grad_rows(self.ctx, C.byref(prog), C.byref(gmap), X.ptr, N, X.ld, d,
                                        _lib.dtype_code(Kinv_rows.dtype), row0, nrows, Kinv_rows.ptr, Kinv_rows.ld,
                                        alpha.ptr, out)
        _check(self, rc, 'g3_gram_grad_rows')
        return np.array(out[:gmap.nslots])

    def gp_dlogp(self, prog, gmap, X, N, d, L, W, a, Y, Kinv, alpha):
        """after gp_factor: K^-1, alpha = K^-1 delta and 1/2 sum G_ij dK_ij/dparam per slot"""
        out = (C.c_double * max(gmap.nslots, 1))()
        rc = self.lib.g3_gp_dlogp(self.ctx, C.byref(prog), C.byref(gmap), X.ptr, N, X.ld, d, L.ptr, L.ld, W.ptr,
                                  a.ptr, _lib.dtype_code(L.dtype), Y.ptr, Y.ld, Kinv.ptr, Kinv.ld, alpha.ptr, out)
        _check(self, rc, 'g3_gp_dlogp')
        return np.array(out[:gmap.nslots])


atexit.register(Device.close_all)


# --------------------------------------------------------------------------- kernel programs
def _expand(spec, leaves):
    """Expand a kernel spec tree (see g3py_amd.processes.hypers.kernels.Kernel.spec) into
    shift + sum_p coef_p * prod leaf  (KernelSum/Prod/Scale/Shift, kernels.py:192-244)."""
    op = spec[0]
    if op == 'sum':
        c1, t1 = _expand(spec[1], leaves)
        c2, t2 = _expand(spec[2], leaves)
        return c1 + c2, t1 + t2
    if op == 'prod':
        c1, t1 = _expand(spec[1], leaves)
        c2, t2 = _expand(spec[2], leaves)
        terms = [(a * b, fa + fb) for a, fa in t1 for b, fb in t2]
        if c1 != 0.0:
            terms += [(c1 * b, fb) for b, fb in t2]
        if c2 != 0.0:
            terms += [(c2 * a, fa) for a, fa in t1]
        return c1 * c2, terms
    if op == 'scale':
        c, t = _expand(spec[2], leaves)
        e = float(spec[1])
        return e * c, [(e * a, f) for a, f in t]
    if op == 'shift':
        c, t = _expand(spec[2], leaves)
        return float(spec[1]) + c, t
    leaves.append(spec)
    return 0.0, [(1.0, (len(leaves) - 1,))]


def spec_leaves(spec):
    """the leaves of a spec tree in the order compile_spec numbers them"""
    leaves = []
    _expand(spec, leaves)
    return leaves


def compile_spec(spec, d):
    """spec tree -> g3_kernel_prog for an input with d columns."""
    leaves = []
    shift, terms = _expand(spec, leaves)
    if len(leaves) > _lib.G3_MAXLEAF or len(terms) > _lib.G3_MAXPROD:
        raise G3Error('kernel expression too large for one g3_kernel_prog (%d leaves, %d products)'
                      % (len(leaves), len(terms)))
    if d > _lib.G3_MAXCOLS:
        raise G3Error('inputs with more than %d columns are not supported' % _lib.G3_MAXCOLS)
    prog = KernelProg()
    prog.nleaf, prog.nprod, prog.shift = len(leaves), len(terms), float(shift)
    for i, lf in enumerate(leaves):
        L = prog.leaf[i]
        kind = lf[0]
        L.kind = _lib.KINDS[kind]
        L.var = float(lf[1])
        dims = lf[-1] if kind != 'NOISE' else None
        dims = np.arange(d) if dims is None else np.atleast_1d(np.asarray(dims)).astype(int)
        dims = np.where(dims < 0, dims + d, dims)
        if len(dims) > _lib.G3_MAXD or (len(dims) and (dims.min() < 0 or dims.max() >= d)):
            raise G3Error('bad dims for kernel leaf %s' % kind)
        L.ndims = len(dims)
        for k, c in enumerate(dims):
            L.dims[k] = int(c)

        def put(dst, v):
            v = np.broadcast_to(np.asarray(v, dtype=np.float64), (len(dims),))
            for k in range(len(dims)):
                dst[k] = float(v[k])
        if kind in ('SE', 'OU', 'MAT32', 'MAT52'):
            put(L.rate, lf[2])
        elif kind == 'RQ':
            put(L.rate, lf[2])
            L.alpha = float(lf[3])
        elif kind in ('COS', 'SINC'):
            put(L.freq, lf[2])
        elif kind in ('SIN', 'SM'):
            put(L.freq, lf[2])
            put(L.rate, lf[3])
    for p, (coef, fac) in enumerate(terms):
        if len(fac) > _lib.G3_MAXFAC:
            raise G3Error('product of more than %d kernels is not supported' % _lib.G3_MAXFAC)
        prog.prod[p].coef = float(coef)
        prog.prod[p].nfac = len(fac)
        for k, f in enumerate(fac):
            prog.prod[p].fac[k] = int(f)
    return prog


class Rows(np.ndarray):
    """a hyper value with a leading axis over chain rows (marks batched values apart from constants in a spec)"""


def compile_spec_rows(spec_rows, spec0, d, B):
    """(template, offsets, fields) for gp_factor_batched_fields: `spec_rows` is a spec tree whose free hyper values are
    Rows arrays (B, ...), `spec0` the same tree for row 0.  Scale / shift constants are not hypers, so products and
    the shift are the template's; every Rows leaf value becomes field columns."""
    tmpl = compile_spec(spec0, d)
    leaves = spec_leaves(spec_rows)
    leaf0 = KernelProg.leaf.offset
    lsz = C.sizeof(_lib.Leaf)
    offs, cols = [], []

    def put(i, member, v, n):
        if not isinstance(v, Rows):
            return
        v = np.broadcast_to(np.asarray(v, dtype=np.float64).reshape(B, -1), (B, n))
        base = leaf0 + i * lsz + getattr(_lib.Leaf, member).offset
        for k in range(n):
            offs.append(base + 8 * k)
            cols.append(v[:, k])
    for i, lf in enumerate(leaves):
        kind, nd = lf[0], int(tmpl.leaf[i].ndims)
        put(i, 'var', lf[1], 1)
        if kind in ('SE', 'OU', 'MAT32', 'MAT52'):
            put(i, 'rate', lf[2], nd)
        elif kind == 'RQ':
            put(i, 'rate', lf[2], nd)
            put(i, 'alpha', lf[3], 1)
        elif kind in ('COS', 'SINC'):
            put(i, 'freq', lf[2], nd)
        elif kind in ('SIN', 'SM'):
            put(i, 'freq', lf[2], nd)
            put(i, 'rate', lf[3], nd)
    fields = np.stack(cols, axis=1) if cols else np.zeros((B, 0))
    return tmpl, np.asarray(offs, dtype=np.int32), np.ascontiguousarray(fields)
