import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _usable_cores():
    """cores this job may use: the cgroup quota when there is one (a GPU box hands a one-GPU job 16 cores of a host with
    hundreds), else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return n


# The oracle side of the tests is NumPy / SciPy: a BLAS pool sized to the HOST's core count inside a 16-core cgroup
# quota is throttled to a crawl (an N = 2300 inverse took a minute on the GPU box).  Size the pools to the quota --
# here for this process, through the environment for the ranks the multi-process tests spawn.
_CORES = _usable_cores()
for _v in ('OPENBLAS_NUM_THREADS', 'OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
    os.environ.setdefault(_v, str(max(1, min(_CORES, 16))))
_TP_LIMIT = None


def pytest_configure(config):
    global _TP_LIMIT
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')
    try:                                  # NumPy may have been imported (and its pool sized) before this file ran
        from threadpoolctl import threadpool_limits
        _TP_LIMIT = threadpool_limits(limits=max(1, min(_CORES, 16)))
    except Exception:
        pass


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _built_library():
    """build libg3hip.so in-tree when the snapshot does not carry it (hipcc cross-compiles without a GPU)"""
    from g3py_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
