"""ctypes binding of libg3hip.so (the C ABI declared in include/g3hip.h).

There is no CPU fallback: importing this module without the built library, or creating a
context without a GPU, raises.  Build with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C g3py_amd/csrc`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('G3_LIB_PATH') or os.path.join(_HERE, 'lib', 'libg3hip.so')   # G3_LIB_PATH: another build (A/B runs)

G3_MAXD, G3_MAXCOLS, G3_MAXLEAF, G3_MAXPROD, G3_MAXFAC = 32, 40, 8, 16, 4
G3_F64, G3_F32 = 0, 1
G3_GRAM_LOWER, G3_GRAM_SCRUB, G3_GRAM_PAD_EYE = 1, 2, 4
G3_PAD = 128       # matrices are padded to a multiple of the panel block (G3_LB in the library)
G3_RHS_PAD = 128   # right-hand-side blocks are padded to a multiple of the 128-row tile
G3_MAX_BATCH = 4096
KINDS = dict(SE=0, OU=1, MAT32=2, MAT52=3, RQ=4, COS=5, SIN=6, SINC=7, SM=8, NOISE=9, WN=10)


class Leaf(C.Structure):
    _fields_ = [('kind', C.c_int32), ('ndims', C.c_int32), ('dims', C.c_int32 * G3_MAXD),
                ('var', C.c_double), ('alpha', C.c_double),
                ('rate', C.c_double * G3_MAXD), ('freq', C.c_double * G3_MAXD)]


class Prod(C.Structure):
    _fields_ = [('coef', C.c_double), ('nfac', C.c_int32), ('fac', C.c_int32 * G3_MAXFAC),
                ('_pad', C.c_int32 * 3)]


class KernelProg(C.Structure):
    _fields_ = [('nleaf', C.c_int32), ('nprod', C.c_int32), ('shift', C.c_double),
                ('leaf', Leaf * G3_MAXLEAF), ('prod', Prod * G3_MAXPROD)]


class GradMap(C.Structure):
    """g3_grad_map: output slot of each leaf parameter (-1 = not wanted)"""
    _fields_ = [('nslots', C.c_int32), ('var', C.c_int32 * G3_MAXLEAF), ('alpha', C.c_int32 * G3_MAXLEAF),
                ('rate', C.c_int32 * G3_MAXLEAF), ('freq', C.c_int32 * G3_MAXLEAF)]


_P = C.c_void_p
_I64 = C.c_int64

G3_DIST_ID_BYTES = 128
DIST_BCAST_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
DIST_ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
DIST_ALLREDUCE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int)


class DistCallbacks(C.Structure):
    """g3_dist_callbacks: the three collectives as blocking host callbacks on device buffers (test transport)"""
    _fields_ = [('user', C.c_void_p), ('bcast', DIST_BCAST_CB), ('allgather', DIST_ALLGATHER_CB),
                ('allreduce', DIST_ALLREDUCE_CB)]

class DistHostCallbacks(C.Structure):
    """g3_dist_host_callbacks: the collectives as host callbacks on HOST staging buffers, served by the library's two
    worker threads (asynchronous test transport)"""
    _fields_ = [('user', C.c_void_p), ('bcast', DIST_BCAST_CB), ('allgather', DIST_ALLGATHER_CB),
                ('allreduce', DIST_ALLREDUCE_CB)]

_SIGS = {
    'g3_version': ([], C.c_int),
    'g3_ctx_create': ([C.c_int, C.POINTER(_P)], C.c_int),
    'g3_ctx_destroy': ([_P], C.c_int),
    'g3_ctx_set_stream': ([_P, _P], C.c_int),
    'g3_ctx_sync': ([_P], C.c_int),
    'g3_last_error': ([_P], C.c_char_p),
    'g3_malloc': ([_P, C.c_size_t, C.POINTER(_P)], C.c_int),
    'g3_free': ([_P, _P], C.c_int),
    'g3_memcpy_h2d': ([_P, _P, _P, C.c_size_t], C.c_int),
    'g3_memcpy_d2h': ([_P, _P, _P, C.c_size_t], C.c_int),
    'g3_memcpy_d2d': ([_P, _P, _P, C.c_size_t], C.c_int),
    'g3_memset': ([_P, _P, C.c_int, C.c_size_t], C.c_int),
    'g3_copy2d': ([_P, _P, _I64, _P, _I64, _I64, _I64, C.c_int], C.c_int),
    'g3_gram': ([_P, C.POINTER(KernelProg), _P, _I64, _I64, _P, _I64, _I64, C.c_int, C.c_int, _P, _I64,
                 _I64, _I64, C.c_uint], C.c_int),
    'g3_gram_rows': ([_P, C.POINTER(KernelProg), _P, _I64, _I64, C.c_int, _I64, _I64, C.c_int, _P, _I64, C.c_uint], C.c_int),
    'g3_gram_diag': ([_P, C.POINTER(KernelProg), _P, _I64, _I64, C.c_int, C.c_int, _P], C.c_int),
    'g3_cov_lift': ([_P, _P, _I64, _I64, C.c_int], C.c_int),
    'g3_gram_path_stats': ([_P, C.POINTER(C.c_double)], C.c_int),
    'g3_grad_path_stats': ([_P, C.POINTER(C.c_double)], C.c_int),
    'g3_gram_jit_check': ([C.POINTER(KernelProg), C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_char_p, C.c_int64], C.c_int),
    'g3_grad_jit_check': ([C.POINTER(KernelProg), C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_char_p, C.c_int64], C.c_int),
    'g3_scrub': ([_P, _P, _I64, _I64, _I64, C.c_int], C.c_int),
    'g3_gemm_nt': ([_P, _P, _I64, _P, _I64, _P, _I64, _I64, _I64, _I64, C.c_double, C.c_double, C.c_int,
                    C.c_int], C.c_int),
    'g3_gemm_nt_stair': ([_P, _P, _I64, _P, _I64, _P, _I64, _I64, C.POINTER(_I64), C.POINTER(_I64), C.c_int,
                          C.c_double, C.c_double, C.c_int, _I64, C.POINTER(C.c_int32), C.c_int, C.POINTER(_I64)], C.c_int),
    'g3_potrf': ([_P, _P, _I64, _I64, C.c_int, _P, C.POINTER(C.c_int)], C.c_int),
    'g3_potrf_nowait': ([_P, _P, _I64, _I64, C.c_int, _P, _P], C.c_int),
    'g3_potrf_robust': ([_P, _P, _I64, _P, _I64, _I64, C.c_int, C.c_int, C.POINTER(C.c_int),
                         C.POINTER(C.c_int), C.POINTER(C.c_double)], C.c_int),
    'g3_trsm_rlt': ([_P, _P, _I64, _I64, _P, _I64, _I64, C.c_int, _P], C.c_int),
    'g3_trtri_full': ([_P, _P, _I64, _P, _P, _P, _P, C.c_int], C.c_int),
    'g3_trsm_full': ([_P, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, C.c_int], C.c_int),
    'g3_logp_terms': ([_P, _P, _I64, _I64, _P, C.c_int, C.POINTER(C.c_double)], C.c_int),
    'g3_diag_stats': ([_P, _P, _I64, _I64, C.c_int, C.POINTER(C.c_double)], C.c_int),
    'g3_diag_add': ([_P, _P, _I64, _I64, C.c_int, C.c_double], C.c_int),
    'g3_rows_dot_ss': ([_P, _P, _I64, _I64, _I64, _P, C.c_int, _P, _P], C.c_int),
    'g3_gp_factor': ([_P, C.POINTER(KernelProg), _P, _I64, _I64, C.c_int, _P, C.c_int, _P, _I64, _P, _P,
                      C.POINTER(C.c_double)], C.c_int),
    'g3_gp_factor_predict': ([_P, C.POINTER(KernelProg), C.POINTER(KernelProg), _P, _I64, _I64, C.c_int, _P, _P, _I64,
                              _I64, C.c_int, _P, _I64, _P, _P, _P, _P, C.POINTER(C.c_double)], C.c_int),
    'g3_gp_factor_batched': ([_P, C.POINTER(KernelProg), C.c_int, _P, _I64, _I64, C.c_int, _P, _I64, C.c_int, _P, _I64,
                              _I64, _P, _P, C.POINTER(C.c_double)], C.c_int),
    'g3_gp_factor_batched_fields': ([_P, C.POINTER(KernelProg), C.c_int, _P, _P, C.c_int, _P, _I64, _I64, C.c_int, _P,
                                     _I64, C.c_int, _P, _I64, _I64, _P, _P, C.POINTER(C.c_double)], C.c_int),
    'g3_gp_cross': ([_P, C.POINTER(KernelProg), _P, _I64, _I64, _P, _I64, _I64, C.c_int, _P, _I64, _P, _P,
                     C.c_int, _P, _I64, _P, _P], C.c_int),
    'g3_gp_sample': ([_P, _P, _I64, _I64, _P, _P, _I64, C.c_int, _P], C.c_int),
    'g3_grad_layout': ([C.POINTER(KernelProg), C.POINTER(GradMap)], C.c_int),
    'g3_potri': ([_P, _P, _I64, _I64, _P, C.c_int, _P, _I64, _P, _I64], C.c_int),
    'g3_gram_grad': ([_P, C.POINTER(KernelProg), C.POINTER(GradMap), _P, _I64, _I64, C.c_int, C.c_int, _P, _I64,
                      _P, C.POINTER(C.c_double)], C.c_int),
    'g3_gram_grad_rows': ([_P, C.POINTER(KernelProg), C.POINTER(GradMap), _P, _I64, _I64, C.c_int, C.c_int, _I64, _I64,
                           _P, _I64, _P, C.POINTER(C.c_double)], C.c_int),
    'g3_gp_dlogp': ([_P, C.POINTER(KernelProg), C.POINTER(GradMap), _P, _I64, _I64, C.c_int, _P, _I64, _P, _P,
                     C.c_int, _P, _I64, _P, _I64, _P, C.POINTER(C.c_double)], C.c_int),
    'g3_gp_dlogp_batched': ([_P, C.POINTER(KernelProg), C.c_int, C.POINTER(GradMap), _P, _I64, _I64, C.c_int, _P, _I64,
                             _I64, _P, _P, C.c_int, _P, _P, _P, C.POINTER(C.c_double)], C.c_int),
    'g3_gp_dlogp_batched_fields': ([_P, C.POINTER(KernelProg), C.c_int, _P, _P, C.c_int, C.POINTER(GradMap), _P, _I64, _I64,
                                    C.c_int, _P, _I64, _I64, _P, _P, C.c_int, _P, _P, _P, C.POINTER(C.c_double)], C.c_int),
    'g3_dist_unique_id': ([_P], C.c_int),
    'g3_dist_create': ([_P, _P, _P, C.c_int, C.c_int, C.POINTER(_P)], C.c_int),
    'g3_dist_create_callbacks': ([_P, C.POINTER(DistCallbacks), C.c_int, C.c_int, C.POINTER(_P)], C.c_int),
    'g3_dist_create_callbacks_async': ([_P, C.POINTER(DistHostCallbacks), C.c_int, C.c_int, C.POINTER(_P)], C.c_int),
    'g3_dist_destroy': ([_P], C.c_int),
    'g3_dist_last_error': ([_P], C.c_char_p),
    'g3_dist_plan': ([_P, _I64, C.c_int, _I64, _I64, C.c_int], C.c_int),
    'g3_dist_gp_factor_predict': ([_P, C.POINTER(KernelProg), C.POINTER(KernelProg), _P, _I64, _P, _P, _I64,
                                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)], C.c_int),
    'g3_dist_posterior_draws': ([_P, C.POINTER(KernelProg), _P, _I64, _P, _P, _I64, _P, C.POINTER(C.c_int),
                                 C.POINTER(C.c_int)], C.c_int),
    'g3_dist_posterior_cov': ([_P, C.POINTER(KernelProg), _P, _I64, _P, _I64], C.c_int),
    'g3_dist_set_grad': ([_P, C.c_int], C.c_int),
    'g3_dist_gp_dlogp': ([_P, C.POINTER(KernelProg), C.POINTER(GradMap), _P, _I64, C.c_double, C.POINTER(C.c_double),
                          C.POINTER(C.c_double)], C.c_int),
    'g3_dist_comm_stats': ([_P, C.POINTER(C.c_double)], C.c_int),
    'g3_dist_phase_stats': ([_P, C.POINTER(C.c_double)], C.c_int),
    'g3_dist_set_keep': ([_P, C.c_int], C.c_int),
    'g3_dist_create_replay': ([_P, _P, C.c_int, C.c_int, C.POINTER(_P)], C.c_int),
    'g3_dist_prof_enable': ([_P, C.c_int], C.c_int),
    'g3_dist_prof_collect': ([_P, C.POINTER(C.c_double)], C.c_int),
    'g3_dist_local_rows': ([_P, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)], C.c_int),
    'g3_prof_enable': ([_P, C.c_int], C.c_int),
    'g3_prof_reset': ([_P], C.c_int),
    'g3_prof_collect': ([_P, C.POINTER(C.c_double)], C.c_int),
}
EXPORTS = tuple(_SIGS)
PROF_TAGS = ('gemm_bulk', 'gram', 'potrf', 'trsv', 'cross_gram', 'trsm', 'reduce', 'gemm_mid',
             'gemm_small', 'diag128')

_lib = None


class G3Error(RuntimeError):
    pass


def load():
    """Load libg3hip.so; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise G3Error('libg3hip.so not found at %s: build it first (make -C g3py_amd/csrc); '
                      'g3py_amd has no CPU fallback' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in _SIGS.items():
        f = getattr(lib, name)      # AttributeError if the symbol is missing
        f.argtypes = args
        f.restype = res
    _lib = lib
    return lib


def dtype_code(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return G3_F64
    if dtype == np.float32:
        return G3_F32
    raise G3Error('unsupported dtype %s' % dtype)


def roundup(n, m=G3_PAD):
    return (int(n) + m - 1) // m * m
