"""CPU sanitizer leg: the host-side schedule / table builders of libg3hip
(g3py_amd/csrc/g3_host.h -- tile rasters incl. the multi-GPU staircase and its chunking, the stripe-solve op list,
panel boundaries, the Gram fast-path matcher, the kernel-program ring, the jitter schedule) compiled with
g++ -fsanitize=address,undefined and driven by tests/host_asan/harness.cpp.  No GPU involved; never a GPU sanitizer run."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_host_logic_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'host_asan')
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all',
           '-I', os.path.join(ROOT, 'include'), '-I', os.path.join(ROOT, 'g3py_amd', 'csrc'),
           os.path.join(ROOT, 'tests', 'host_asan', 'harness.cpp'), '-o', exe]
    subprocess.check_call(cmd)
    env = dict(os.environ)
    env.pop('LD_PRELOAD', None)            # ASan must come first in the link order of the test binary
    for k in ('G3_NB_TAIL', 'G3_NB_MIN', 'G3_STAIR_MAX'):
        env.pop(k, None)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'host_asan ok' in r.stdout
    # the chunking again with a tiny launch limit (what the GPU tests use to exercise it)
    env['G3_STAIR_MAX'] = '3'
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    # the block dealing of the multi-GPU drivers: the Python twin (distributed.py::deal_blocks) deals exactly as g3_host.h::g3h_deal
    import sys
    sys.path.insert(0, ROOT)
    from g3py_amd.distributed import deal_blocks
    for P, nblk in [(1, 5), (2, 7), (3, 13), (4, 32), (5, 32), (8, 32), (8, 33), (8, 64), (7, 50), (8, 1), (3, 2)]:
        r = subprocess.run([exe, 'deal', str(P), str(nblk)], capture_output=True, text=True, timeout=60, env=env)
        assert r.returncode == 0
        assert [int(v) for v in r.stdout.split()] == deal_blocks(P, nblk), (P, nblk)
