"""CPU tests of the drop-in boundary: libg3hip.so loads, exports every symbol include/g3hip.h
declares, the ctypes mirror of the structs matches the C layout, and the product fails loudly
without a GPU (no CPU fallback).  No compute call is made here."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'g3hip.h')


@pytest.fixture(scope='module')
def lib():
    from g3py_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return _lib.load()


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(g3_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported(lib):
    from g3py_amd import _lib
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), 'libg3hip.so does not export ' + n
    assert sorted(_lib.EXPORTS) == names, 'ctypes signature table and header disagree'
    assert lib.g3_version() >= 100


def test_struct_layout_matches_c(tmp_path):
    from g3py_amd import _lib
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "g3hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %d\\n",'
                   'sizeof(g3_leaf),sizeof(g3_prod),sizeof(g3_kernel_prog),offsetof(g3_leaf,rate),'
                   'offsetof(g3_kernel_prog,leaf),offsetof(g3_kernel_prog,prod),sizeof(g3_grad_map),sizeof(g3_dist_callbacks),'
                   'offsetof(g3_dist_callbacks,allreduce),G3_DIST_ID_BYTES);return 0;}\n')
    exe = tmp_path / 'sz'
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)])
    got = list(map(int, subprocess.check_output([str(exe)]).split()))
    assert got == [ctypes.sizeof(_lib.Leaf), ctypes.sizeof(_lib.Prod), ctypes.sizeof(_lib.KernelProg),
                   _lib.Leaf.rate.offset, _lib.KernelProg.leaf.offset, _lib.KernelProg.prod.offset,
                   ctypes.sizeof(_lib.GradMap), ctypes.sizeof(_lib.DistCallbacks), _lib.DistCallbacks.allreduce.offset,
                   _lib.G3_DIST_ID_BYTES]


def test_header_compiles_as_plain_c(tmp_path):
    src = tmp_path / 'h.c'
    src.write_text('#include "g3hip.h"\nint main(void){return G3_OK;}\n')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), '-c', str(src),
                           '-o', str(tmp_path / 'h.o')])


def test_no_cpu_fallback(lib):
    """without a GPU the product raises; it never routes to the oracle or any CPU path"""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import g3py_amd as g3
    with pytest.raises(g3.G3Error):
        g3.Device(0)
    import numpy as np
    x = np.linspace(0, 1, 8)[:, None]
    gp = g3.GaussianProcess(space=x, location=g3.Zero(), kernel=g3.SE(x))   # construction is host-only
    gp.observed(x, np.sin(x[:, 0]))
    with pytest.raises(g3.G3Error):
        gp.logp()
    # nothing under g3py_amd imports the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'g3py_amd')):
        for f in files:
            if f.endswith('.py') or f.endswith('.hip') or f.endswith('.h'):
                assert 'oracle' not in open(os.path.join(dirpath, f)).read().replace('the oracle', ''), f


def test_null_context_is_an_error_not_a_crash(lib):
    assert lib.g3_ctx_sync(None) == -1
    assert lib.g3_gemm_nt(None, None, 0, None, 0, None, 0, 0, 0, 0, 1.0, 0.0, 0, 0) == -1
    assert lib.g3_ctx_destroy(None) == -1
