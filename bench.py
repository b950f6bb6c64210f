#!/usr/bin/env python3
"""bench.py -- headline benchmark of the GP inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json: "GP logp+predict end-to-end sec and Cholesky TFLOP/s at N=32768
fp64"): SE-kernel GaussianProcess, fp64, N=32768 observations in d=4, M=1024 prediction
points, synthetic inputs (seed 1004) already resident in HBM.  One
"step" = one full pass: Gram assembly -> blocked Cholesky -> L^-1 delta (log marginal
likelihood) -> cross-Gram -> M-right-hand-side triangular solve -> posterior mean and variance.

Prints ONE JSON line (rank 0).  `value` = algorithmic flops of the step / wall time of the
step (whole job), `e2e_sec` and `cholesky_tflops` are the two halves of the baseline metric.
`roofline` is measured live with HIP events around every launch of the dominant kernel (the
MFMA GEMM that carries the panel updates of the blocked Cholesky); `cpu_baseline` times the
CPU oracle (NumPy/SciPy restatement, same LAPACK entry points as the reference) on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix (= vector) peak, AMD public spec; see DESIGN.md
FP32_MATRIX_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = FP32 vector peak
HBM_PEAK_GBS = 8000.0


def _traffic():
    """HBM bytes per bulk-GEMM launch from the committed rocprofv3 PMC passes (profiles/rNN_traffic.json:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE x2 for gfx950), or None"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')))
    if not fs:
        return None
    try:
        return float(json.load(open(fs[-1]))['hbm_bytes_per_launch'])
    except Exception:
        return None


def shutil_which(name):
    import shutil
    return shutil.which(name) is not None


def _measure_traffic(argv_workload):
    """`--measure-traffic`: HBM bytes per bulk-GEMM launch measured NOW, by two child runs of this script under
    `rocprofv3 --kernel-trace --pmc <counter>` (one counter per pass, no other trace domain -- the guide's rule),
    corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950, both in KB).  Returns (bytes, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    per = {}
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        tmp = tempfile.mkdtemp(prefix='g3pmc_', dir=os.environ.get('TMPDIR', '/tmp'))
        try:
            cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', tmp, '--',
                   sys.executable, os.path.abspath(__file__), '--steps', '1', '--warmup', '0', '--cpu-n', '0',
                   '--skip-events'] + argv_workload
            # the profiler and the profiled python run in their own session: on a timeout the WHOLE group is
            # killed and reaped before tmp goes away, so no grandchild is left holding the GPU
            import signal
            child = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp, start_new_session=True)
            try:
                rc = child.wait(timeout=150)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(child.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                child.wait()
                return None, 'rocprofv3 --pmc %s child timed out (process group killed)' % counter
            fs = glob.glob(os.path.join(tmp, '**', '*_counter_collection.csv'), recursive=True)
            if rc != 0 or not fs:
                return None, 'rocprofv3 --pmc %s child failed (rc %s)' % (counter, rc)
            n, kb = 0, 0.0
            for row in csv.DictReader(open(fs[0])):
                name = row['Kernel_Name']
                if 'gemm_nt_kernel' in name and ', 128, 128, 64, ' in name and \
                        int(row['Grid_Size']) // int(row['Workgroup_Size']) >= 4096:
                    n += 1
                    kb += float(row['Counter_Value'])
            if n == 0:
                return None, 'no bulk GEMM launch in the %s pass' % counter
            per[counter] = kb * 1024.0 / n
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return 2.0 * per['FETCH_SIZE'] + per['WRITE_SIZE'], \
        'measured in this run: two child passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- bench.py --steps 1 ' \
        '--warmup 0 --skip-events` (FETCH_SIZE x2, gfx950 correction; fetch %.3f GB + write %.3f GB per launch).  These are the ' \
        'L2\'s fabric-side requests: reads that hit the 256 MiB Infinity Cache are counted (MI355X_MICROARCH.md, HBM) -- the <= 268 MB ' \
        'panel every tile re-reads lives there, so this is an upper bound on HBM bytes; algorithmic: C read + written + one panel' \
        % (2.0 * per['FETCH_SIZE'] / 1e9, per['WRITE_SIZE'] / 1e9)


def _golden_logp(N, d, M, seed, kernel):
    """logp of the CPU oracle at a full benchmark configuration, or None when it was never generated"""
    try:
        gold = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'fullsize.json')))
    except Exception:
        return None
    for g in gold.values():
        if (g['N'], g['d'], g['M'], g['seed'], g['kernel']) == (N, d, M, seed, kernel):
            return float(g['logp'])
    return None


def _golden_c5(N, d, M, seed, S):
    """the fp64 oracle's pin of BASELINE config 5 (tests/golden/fullsize.json: c5 / c5mini), or None"""
    try:
        gold = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'fullsize.json')))
    except Exception:
        return None
    for g in gold.values():
        if 'draw_values' in g and (g['N'], g['d'], g['M'], g['seed'], g['draws']) == (N, d, M, seed, S):
            return g
    return None


def _api_timing(g3, X, y, Xs, d, kernel, npdt):
    """What the user calls: GaussianProcess.logp and .predict through the public API from HOST
    NumPy inputs -- parameter dict -> kernel program, H2D of X / y / Xs, Gram, Cholesky, solves, D2H of the
    statistics.  A fresh parameter vector per call, so the factor cache never hits; reported beside the
    device-resident step, never part of `value`."""
    import time as _t
    kern = g3.SE(X) if kernel == 'se' else g3.MAT52(X) + g3.COS(X)
    gp = g3.GaussianProcess(space=Xs, location=g3.Zero(), kernel=kern)
    t0 = _t.perf_counter()
    gp.observed(X.astype(npdt), y.astype(npdt))
    t_obs = _t.perf_counter() - t0
    res = {'observed_ms': t_obs * 1e3}
    for rep in range(2):                      # rep 0 allocates the device workspace; rep 1 is the steady state
        params = dict(gp.params)
        for k_ in params:
            if k_.endswith('_var_log_') and 'Noise' not in k_:
                params[k_] = np.log(1.0 + 1e-3 * (rep + 1))
            elif k_.endswith('_rate_log_'):
                params[k_] = np.log(np.ones(d))
            elif 'Noise' in k_:
                params[k_] = np.log(0.1)
            elif k_.endswith('_freq_log_') or k_.endswith('_freq'):
                params[k_] = np.log(np.full(d, 0.125)) if k_.endswith('_log_') else np.full(d, 0.125)
        t0 = _t.perf_counter()
        lp = gp.logp(params)
        t1 = _t.perf_counter()
        pr = gp.predict(params, mean=True, var=True, std=False)
        t2 = _t.perf_counter()
        res.update(logp_ms=(t1 - t0) * 1e3, predict_ms=(t2 - t1) * 1e3, logp=float(lp),
                   mean0=float(np.asarray(pr.mean)[0]), var0=float(np.asarray(pr.variance)[0]))
    res['note'] = ('public API, host NumPy in / out, second of two evaluations with distinct hyper-parameters; predict = '
                   'posterior mean + variance at the M test points re-using the factor of the logp call (one cross solve)')
    return res


def synth(N, d, M, seed):
    """SURVEY.md section 8(d)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    Lbox = N ** (1.0 / d)
    X = rng.uniform(0, Lbox, (N, d))
    Xs = rng.uniform(0, Lbox, (M, d))
    y = np.sin(X.sum(1) / np.sqrt(d)) + 0.1 * rng.standard_normal(N)
    return X, y, Xs


def step_flops(N, M, S=0):
    """algorithmic flops of one step: potrf N^3/3, trsv N^2,
    trsm N^2 M, mean+variance 2 N M; with S > 0 draws also the posterior covariance N M^2, its
    Cholesky M^3/3 and L Z: M^2 S"""
    f = N ** 3 / 3.0 + float(N) ** 2 + float(N) ** 2 * M + 2.0 * N * M
    if S > 0:
        f += float(N) * M * M + M ** 3 / 3.0 + float(M) * M * S
    return f


_REAL_STDOUT = [None]     # a private duplicate of the process's stdout, taken before anything redirects fd 1


def _emit_line(obj):
    """the ONE line, on the real stdout even while fd 1 points at stderr (a watchdog may fire inside such a block)"""
    data = (json.dumps(obj) + '\n').encode()
    try:
        sys.stdout.flush()
    except Exception:
        pass
    fd = _REAL_STDOUT[0] if _REAL_STDOUT[0] is not None else 1
    while data:
        n = os.write(fd, data)
        data = data[n:]


class _StdoutToStderr:
    """fd 1 points at stderr inside the block: gloo and RCCL announce themselves on the C stdout (connection counts, the
    version banner at the first communicator), and the bench's stdout carries exactly ONE line"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def usable_cores():
    """CPU cores this process may actually use: the cgroup quota when there is one (a GPU box hands a
    one-GPU job a share of the host), else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return n


def cpu_default_n(N):
    """size of the default CPU-baseline pass: the whole workload when it fits the host's free memory (N x N fp64 plus
    a few row tiles and the LAPACK copies: ~3.5 matrices) and the default run's time budget (N <= 32768), else 16384"""
    if N > 32768:
        return 16384
    try:
        import psutil
        free = psutil.virtual_memory().available
    except Exception:
        free = 0
    need = 3.5 * 8.0 * N * N
    return N if (N <= 16384 or free >= max(need, 48e9 if N > 16384 else 0)) else 16384


def cpu_baseline(N, d, M, seed, n_cpu):
    """the CPU oracle (NumPy/SciPy restatement: same LAPACK entry points as the reference) on a bounded
    sample of the workload, with the BLAS pool sized to the cores this job may use"""
    from oracle import g3_oracle as orc
    X, y, Xs = synth(N, d, M, seed)
    X, y = X[:n_cpu], y[:n_cpu]
    cores = usable_cores()
    blas = None
    try:
        import threadpoolctl
        with threadpoolctl.threadpool_limits(limits=cores):
            blas = sorted({p.get('num_threads', 1) for p in threadpoolctl.threadpool_info()})
            t0 = time.perf_counter()
            lp, mean, var, tm = orc.cpu_hot_path(X, y, Xs)
            dt = time.perf_counter() - t0
    except ImportError:
        t0 = time.perf_counter()
        lp, mean, var, tm = orc.cpu_hot_path(X, y, Xs)
        dt = time.perf_counter() - t0
    threads = max(blas) if blas else cores
    scale = (float(N) / n_cpu) ** 3
    what = 'one pass of the WHOLE workload' if n_cpu >= N else 'one pass of the same workload cut to N=%d' % n_cpu
    return dict(value=step_flops(n_cpu, M) / dt / 1e12, unit='TFLOP/s', cores=int(min(threads, cores)), kind='port',
                sample='%s (N=%d, d=%d, M=%d, same generator and seed), '
                       'oracle.cpu_hot_path: NumPy Gram + scipy dpotrf (tensors.py:198; above N=16384 as a sweep over 8192-wide '
                       'panels of dpotrf / dtrsm / dgemm: one call over the 8.6 GB matrix segfaults in this OpenBLAS) + '
                       'solve_triangular; BLAS threads %s, host reports %d CPUs, cgroup/affinity allows %d'
                       % (what, n_cpu, d, M, blas, os.cpu_count() or 0, cores),
                seconds=dt, potrf_gflops=(n_cpu ** 3 / 3.0) / tm['potrf'] / 1e9, phases_sec=tm,
                full_size_seconds_extrapolated=dt * scale if n_cpu < N else dt,
                extrapolation='N^3 from the sample (flagged: not measured)' if n_cpu < N else None), lp


def _pass_summary(args, N, M, S, pd, pin):
    """one timed pass of a multi-rank run (a collective schedule) in the `schedules` block of the line"""
    sec = pd['elapsed'] / args.steps
    med = float(np.median(pd['step_s'])) if pd['step_s'] else sec
    fl = step_flops(N, M, S)
    o = {'ms_per_step': sec * 1e3, 'ms_per_step_median': med * 1e3, 'value': fl / med / 1e12, 'value_mean': fl / sec / 1e12,
         'logp': pd['logp']}
    if pin is not None:
        o['logp_rel_err'] = abs(pd['logp'] - pin) / abs(pin)
        o['pin_ok'] = bool(o['logp_rel_err'] <= 1e-8)
    if pd['comm_all'] is not None:
        o['comm'] = {'per_rank': [{k: _per_step(v, args.steps) for k, v in r.items()} for r in pd['comm_all']]}
    return o


def _per_step(v, steps):
    o = {'calls_per_step': v['calls'] / steps, 'bytes_per_step': v['bytes'] / steps}
    if 'device_ms' in v:
        o['device_ms_per_step'] = v['device_ms'] / steps
    else:
        o['host_wait_s_per_step'] = v['wait_s'] / steps
    return o


def _pass_line(args, world, N, d, M, S, pd, sched):
    """a complete line from ONE pass: what is printed if a later pass never comes back"""
    sec = pd['elapsed'] / args.steps
    med = float(np.median(pd['step_s'])) if pd['step_s'] else sec
    fl = step_flops(N, M, S)
    seed = 1005 if args.f32 else (1003 if args.kernel == 'mat52cos' else (1002 if N == 8192 else 1004))
    pin = _golden_logp(N, d, M, seed, args.kernel) if not (args.f32 or S > 0) else None
    ln = {'metric': 'GP logp+predict end-to-end (Gram+Cholesky+solves), N=%d %s: algorithmic TFLOP/s' % (N, 'fp32' if args.f32 else 'fp64'),
          'value': fl / med / 1e12, 'unit': 'TFLOP/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
          'ms_per_step': sec * 1e3, 'ms_per_step_median': med * 1e3, 'value_mean': fl / sec / 1e12, 'higher_is_better': True,
          'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32' if args.f32 else 'f64', 'data': 'synthetic',
          'config': {'workload': args.kernel.upper() + '-kernel GaussianProcess, N=%d d=%d, M=%d test points: Gram + blocked Cholesky + '
                                 'L^-1 y (logp) + cross-Gram + %d-rhs trsm + posterior mean/variance' % (N, d, M, M),
                     'N': N, 'd': d, 'M': M, 'parallelism': pd['parallelism']},
          'e2e_sec': med, 'logp': pd['logp'], 'schedule': sched,
          'schedules': {sched: _pass_summary(args, N, M, S, pd, pin)}}
    g = {k: sum(pd['prof'][t][k] for t in ('gemm_bulk', 'gemm_mid', 'gemm_small')) for k in ('count', 'ms', 'work')}
    if g['count'] and g['ms'] > 0:
        ach = g['work'] / (g['ms'] * 1e-3) / 1e12
        peak = FP32_MATRIX_PEAK_TFLOPS if args.f32 else FP64_MATRIX_PEAK_TFLOPS
        ln['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
                          'kernel': 'gemm_nt_kernel<*>: rank 0, staircase launches of the bulk stream + a 1-in-16 sample of the chain stream\'s',
                          'launches_per_step': g['count'] / args.steps, 'avg_launch_ms': g['ms'] / g['count'],
                          'avg_launch_flops': g['work'] / g['count']}
    if pin is not None:
        ln['logp_ref'] = pin
        ln['logp_rel_err'] = abs(pd['logp'] - pin) / abs(pin)
    return ln


def main():
    if _REAL_STDOUT[0] is None:
        _REAL_STDOUT[0] = os.dup(1)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--points', dest='n', type=int, default=32768, help='N observations')
    ap.add_argument('--dims', dest='d', type=int, default=4, help='input columns d')
    ap.add_argument('--queries', dest='m', type=int, default=1024, help='M prediction points')
    ap.add_argument('--kernel', default='se', choices=['se', 'mat52cos'],
                    help='se: BASELINE configs 2/4; mat52cos: MAT52 + periodic COS sum kernel (config 3)')
    ap.add_argument('--f32', action='store_true', help='float32 arithmetic (config 5 runs in fp32)')
    ap.add_argument('--cpu-n', type=int, default=-1, help='N of the CPU-baseline pass (0 = skip; default: the whole workload up to N=32768 -- about 75 s on 16 cores -- or a 16384 sample when the host has less than 48 GB free)')
    ap.add_argument('--skip-events', dest='no_prof', action='store_true', help='do not record HIP events in the timed region')
    ap.add_argument('--measure-traffic', dest='measure_traffic', action='store_true', default=None,
                    help='measure roofline.traffic now: two rocprofv3 --pmc child passes of this workload (~10 s).  Default: on for '
                         'the one-GPU fp64 headline workload, off otherwise')
    ap.add_argument('--no-measure-traffic', dest='measure_traffic', action='store_false',
                    help='quote profiles/rNN_traffic.json (committed PMC passes of the same command) instead')
    ap.add_argument('--grad', action='store_true', help='also time dlogp (K^-1 + kernel-parameter sums, SURVEY.md 8f rank 1) '
                                                        'after the timed region; reported under "dlogp", never part of value')
    ap.add_argument('--draws', type=int, default=-1, help='posterior draws S through the warped-GP path (BASELINE config 5: posterior '
                                                          'covariance + second Cholesky + L Z + mapping); default 16 with --f32, else 0')
    ap.add_argument('--api', action='store_true', default=None, help='also time GaussianProcess.logp + predict through the public API '
                                                                     'from host NumPy inputs ("api_ms"; default: on for the one-GPU run)')
    ap.add_argument('--no-api', dest='api', action='store_false')
    ap.add_argument('--panel', dest='nb', type=int, default=0, help='row-block height of the multi-GPU distribution (0 = 1024)')
    args = ap.parse_args()

    if args.api is None:
        args.api = args.gpus == 1 and not args.no_prof
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` typed directly: start one FRESH child process per GPU through
        # torch.distributed.run and exit with its status.  Nothing in this process has touched the GPU
        # yet (torch is not even imported), and nothing is exec'ed: the ranks are ordinary children.
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:      # checked before anything initialises the GPU
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torchrun --nproc-per-node %d, or '
                         'run `python bench.py --gpus %d` and let it start the ranks)' % (args.gpus, world, args.gpus, args.gpus))
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # all ranks are on ONE node (the bench contract): RCCL's bootstrap sockets go over loopback, whatever the
        # container's hostname / interfaces resolve to (checked with one rank: profiles/r03_bench_dist1_nccl.json)
        os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
        os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')    # the gloo group (ids, barriers, timing) too
    # One rank per GPU.  G3_DIST_DRIVER=native (default): the multi-GPU driver INSIDE libg3hip (g3_dist_*: the library
    # owns the RCCL communicators, streams and the per-panel loop); torch.distributed over gloo only carries the
    # 2 x 128 communicator-id bytes, the barriers and the max-over-ranks of the timing.  G3_DIST_DRIVER=python: the
    # torch.distributed driver (g3py_amd/distributed.py::DistributedGP) over ProcessGroupNCCL.  If the native driver
    # cannot be created on some rank (no librccl ...) every rank falls back to the python driver together.
    # G3_DIST_BACKEND=gloo (python driver only) rehearses N > 1 on a one-GPU box: ranks may then share a device.
    driver = os.environ.get('G3_DIST_DRIVER', 'native')
    backend = os.environ.get('G3_DIST_BACKEND', 'nccl')
    # gloo + native: host-staged collectives served by the library's worker threads, stream-ordered like RCCL calls (rehearsal)
    native_transport = os.environ.get('G3_DIST_CALLBACKS', 'callbacks_async') if backend == 'gloo' else 'rccl'
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU fallback')
    if world > 1 and backend == 'nccl' and ndev < world:
        raise SystemExit('bench.py: %d ranks over RCCL need %d GPUs, %d visible (G3_DIST_BACKEND=gloo rehearses '
                         'several ranks on one GPU)' % (world, world, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    # G3_FORCE_DIST=1 on one GPU: the multi-rank driver with ONE rank.  The native driver then still creates its RCCL
    # communicators and issues every collective; the python driver does so with G3_DIST_COLLECTIVES=1
    force_dist = os.environ.get('G3_FORCE_DIST', '0') == '1'
    solo_pg = world == 1 and force_dist and driver == 'python' and os.environ.get('G3_DIST_COLLECTIVES', '0') == '1'
    if solo_pg:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as s_:
                s_.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(s_.getsockname()[1])
    if world > 1 or solo_pg:
        # gloo announces its connections on the C stdout: the bench's stdout carries ONE line, so fd 1 points at stderr
        # while the group comes up
        with _StdoutToStderr():
            if driver == 'native':
                dist.init_process_group('gloo', rank=rank, world_size=world)
            elif backend == 'nccl':
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            if dist.get_backend() == 'gloo':
                dist.barrier()

    tdev0 = torch.device('cuda', local_rank)
    import g3py_amd as g3
    from g3py_amd import _lib
    from g3py_amd.device import compile_spec

    # seed = 1000 + BASELINE config index (config 2: N=8192 SE, 3: MAT52+COS, 4: N=32768 SE, 5: fp32)
    seed = 1005 if args.f32 else (1003 if args.kernel == 'mat52cos' else (1002 if args.n == 8192 else 1004))
    N, d, M = args.n, args.d, args.m
    X, y, Xs = synth(N, d, M, seed)
    S = args.draws if args.draws >= 0 else (16 if args.f32 else 0)
    if S > 0:
        # BASELINE config 5: warped GP, BoxCoxLinear(shift=1, scale=1, power=1.2) on y - min(y) + 1
        from g3py_amd.processes.hypers.mappings import BoxCoxLinear
        yw = (y - y.min() + 1.0).astype(np.float32 if args.f32 else np.float64)
        warp = BoxCoxLinear(shift=1.0, scale=1.0, power=1.2)
        Z = np.random.Generator(np.random.PCG64(seed + 100)).standard_normal((M, S))
        delta = np.asarray(warp.inv(yw), dtype=np.float64)        # refreshed inside every timed step
    else:
        delta = y.copy()                  # Zero mean, identity mapping: delta = y
    if args.kernel == 'se':
        spec_f = ('SE', 1.0, np.ones(d), None)
    else:
        spec_f = ('sum', ('MAT52', 1.0, np.ones(d), None), ('COS', 0.5, np.full(d, 0.125), None))
    spec_n = ('sum', spec_f, ('NOISE', 0.1))
    npdt = np.float32 if args.f32 else np.float64
    tdt = torch.float32 if args.f32 else torch.float64
    # G3_BENCH_IDLE_STREAMS=n: n pairs of idle streams created first -- a different stream history of the process, hence a
    # different placement of the library's streams on hardware queues (A/B measurements of the placement probe)
    _idle = [torch.cuda.Stream(device=torch.device('cuda', local_rank), priority=pr)
             for _ in range(int(os.environ.get('G3_BENCH_IDLE_STREAMS', '0'))) for pr in (-1, 0)]
    dev = g3.Device(local_rank)
    # the critical-path stream gets high priority; the library's side stream (bulk updates) is low
    hp = torch.cuda.Stream(device=tdev0, priority=-1) if os.environ.get('G3_BENCH_HIPRIO', '1') == '1' else None
    if hp is not None:
        torch.cuda.set_stream(hp)
    dev.set_stream(torch.cuda.current_stream().cuda_stream)   # HIP events below see this stream
    Np, Mp = _lib.roundup(N), _lib.roundup(M, _lib.G3_RHS_PAD)
    tdev = torch.device('cuda', local_rank)

    def tens(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=npdt)).to(tdev)

    def wrap(t, rows, cols):
        return dev.wrap(t.data_ptr(), rows, cols, t.stride(0) if t.dim() == 2 else cols, npdt, keep=t)

    Xt, Xst, dt_ = tens(X), tens(Xs), tens(delta[None, :])
    Xd, Xsd, dd = wrap(Xt, N, d), wrap(Xst, M, d), wrap(dt_, 1, N)

    # several ranks: a step -- or the driver's own start-up handshake -- that never returns (a collective one rank did not
    # enter) must not hold the node until the launcher's limit: every rank carries a watchdog from BEFORE the driver is
    # built to the end of its run and leaves with a line that says so
    watchdog = None
    if world > 1:
        import threading
        limit = float(os.environ.get('G3_BENCH_WATCHDOG_S', '900'))

        def _expired():
            sys.stderr.write('bench.py: rank %d made no progress for %.0f s (a collective some rank never entered?); '
                             'G3_DIST_DRIVER=python selects the torch.distributed driver\n' % (rank, limit))
            if rank == 0:
                _emit_line({'metric': 'GP logp+predict end-to-end', 'value': None, 'n_gpus': world,
                            'error': 'watchdog: no progress for %.0f s' % limit})
            sys.stderr.flush()
            os._exit(3)
        watchdog = threading.Timer(limit, _expired)
        watchdog.daemon = True
        watchdog.start()
    # G3_FORCE_DIST=1: run the multi-GPU driver with one rank (development: its overhead over the in-library sweep)
    use_dist = world > 1 or os.environ.get('G3_FORCE_DIST', '0') == '1'
    if not use_dist:
        Kt = torch.empty((Np + 128 + Mp, Np), dtype=tdt, device=tdev)   # covariance + right-hand-side rows
        at = torch.empty((1, Np), dtype=tdt, device=tdev)
        Vt = torch.empty((Mp, Np), dtype=tdt, device=tdev)
        mut = torch.empty((1, Mp), dtype=tdt, device=tdev)
        sst = torch.empty((1, Mp), dtype=tdt, device=tdev)
        Wt_ = torch.empty((Np, 128), dtype=tdt, device=tdev)
        Wd = wrap(Wt_, Np, 128)
        Kd, ad, Vd, mud, ssd = wrap(Kt, Np + 128 + Mp, Np), wrap(at, 1, Np), wrap(Vt, Mp, Np), wrap(mut, 1, Mp), wrap(sst, 1, Mp)
        prog_n, prog_f = compile_spec(spec_n, d), compile_spec(spec_f, d)
        result = {}

        if S > 0:
            Ksst = torch.empty((Mp, Mp), dtype=tdt, device=tdev)
            Lst = torch.zeros((Mp, Mp), dtype=tdt, device=tdev)
            Kssd, Lsd = wrap(Ksst, Mp, Mp), wrap(Lst, Mp, Mp)
            Vd = dev.wrap(Kt.data_ptr() + (Np + 128) * Np * Kt.element_size(), Mp, Np, Np, npdt, keep=Kt)

        def step():
            if S > 0:     # warped GP: delta = T^-1(y) - m(X) is part of every evaluation (gaussian.py:208)
                dl = np.asarray(warp.inv(yw), dtype=npdt)
                dt_.copy_(torch.from_numpy(dl[None, :]))
            # Gram + tall Cholesky: the delta row and the K(Xs, X) rows ride through the factorisation
            st = dev.gp_factor_predict(prog_n, prog_f, Xd, N, d, dd, Xsd, M, Kd, Wd, ad, mud, ssd)
            lp = -0.5 * N * np.log(2 * np.pi) - 0.5 * st['quad'] - st['logdet']
            if S > 0:
                lp += float(warp.logdet_dinv(yw))                                   # gaussian.py:225
                # posterior covariance K_f(Xs, Xs) - V V^T (elliptical.py:90-92), lower triangle
                dev.gram(prog_f, Xsd, None, d, Kssd, Mp, Mp, 0)
                dev.gemm_nt(Kssd, Vd, Vd, Mp, Mp, Np, alpha=-1.0, beta=1.0, lower_only=True)
                Lst.zero_()
                tries, fb, _ = dev.potrf_robust(wrap(Ksst, M, M), wrap(Lst, M, M), M)   # cholesky_robust
                loc = mut[0, :M].double().cpu().numpy()
                g = dev.gp_sample(Lsd, M, loc.astype(npdt), Z.astype(npdt))         # loc + L Z (gaussian.py:92-95)
                result['draws'] = np.asarray(warp(g))                               # mapping, vectorised
                result['cov_tries'] = tries
            dev.sync()
            result['logp'] = lp
            result['stats'] = st
        parallelism = '1gpu'
    else:
        from g3py_amd.distributed import DistributedGP, NativeDistributedGP
        if args.nb <= 0:
            args.nb = 1024      # measured (replay transport, profiles/r04_replay_*): 1024-row blocks are faster than 512 at every P for configs 4 and 5
        dgp = None
        driver_fallback = None       # reason when the native driver was asked for and could not be used
        result = {}

        def make_driver():
            """the driver object of this pass; the native driver reads G3_DIST_SERIAL_COLL when it is created"""
            nonlocal driver, driver_fallback
            d_ = None
            if driver == 'native':
                why = ''
                try:
                    with _StdoutToStderr():
                        d_ = NativeDistributedGP(dev, dist, rank, world, N, d, M, nb=args.nb, dtype=npdt, transport=native_transport)
                except Exception as e:       # noqa: BLE001 -- any failure on any rank sends ALL ranks to the python driver
                    why = '%s: %s' % (type(e).__name__, e)
                ok = torch.tensor([0 if d_ is None else 1], dtype=torch.int32)
                if world > 1:
                    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    if d_ is not None:
                        d_.close()
                        d_ = None
                    driver = 'python (native driver unavailable%s)' % ((': ' + why) if why else ' on another rank')
                    driver_fallback = why or 'the native driver could not be created on another rank'
                    if rank == 0:
                        print('bench.py: falling back to the torch.distributed driver -- ' + driver, file=sys.stderr)
            if d_ is None:
                gb = 'nccl' if (world > 1 and dist.get_backend() == 'gloo' and backend == 'nccl') else None
                d_ = DistributedGP(dev, dist if (world > 1 or solo_pg) else None, rank, world, N, d, M, nb=args.nb, torch_device=tdev,
                                   dtype=npdt, group_backend=gb)
            return d_

        def step():
            if S > 0:
                dl = np.asarray(warp.inv(yw), dtype=npdt)
                dt_.copy_(torch.from_numpy(dl[None, :]))
                lp = dgp.step(spec_n, spec_f, Xd, Xsd, dd, Z=Z) + float(warp.logdet_dinv(yw))
                result['draws'] = np.asarray(warp(dgp.last['draws'].astype(npdt)))
                result['cov_tries'] = dgp.last.get('cov_tries', 0)
                result['logp'] = lp
            else:
                result['logp'] = dgp.step(spec_n, spec_f, Xd, Xsd, dd)

    # ---- the timed region.  Several ranks through the native driver: TWO passes in the same processes --
    # first with the two communicators never in flight together (G3_DIST_SERIAL_COLL=1: the conservative schedule, which
    # cannot stall on the order of concurrent collectives), its line kept; then the driver is destroyed, the overlapped
    # (default) one created and the region repeated.  ONE line comes out, with both (`schedules`), the headline the faster
    # pass that reproduced the oracle pin.  If the overlapped pass runs into its limit the serial line is printed with
    # `overlap_timeout` and every rank exits non-zero: the first real multi-GPU run cannot come back empty.
    schedules = ['default']
    if use_dist and world > 1 and driver == 'native':
        schedules = [v for v in os.environ.get('G3_BENCH_SCHEDULES', 'serial,overlapped').split(',') if v in ('serial', 'overlapped')] \
            or ['overlapped']
    passes = {}
    overlap_wd = None
    serial_line = [None]
    for sched in schedules:
        if use_dist:
            if sched != 'default':
                os.environ['G3_DIST_SERIAL_COLL'] = '1' if sched == 'serial' else '0'
            dgp = make_driver()
            native = isinstance(dgp, NativeDistributedGP)
            if not native:
                schedules = [sched]                      # the torch.distributed driver has one schedule
            parallelism = 'row-block-cyclic x%d (nb=%d): diagonal-factor broadcast + panel all-gather (%s), look-ahead; driver: %s' \
                % (world, args.nb, ('RCCL' if (native_transport == 'rccl' if native else backend == 'nccl') else 'gloo, host-staged: rehearsal'),
                   'libg3hip g3_dist_* (C++ loop, library-owned communicators)' if native else 'torch.distributed (' + str(driver) + ')')
            if sched == 'overlapped' and 'serial' in passes and watchdog is not None:
                import threading
                lim2 = min(float(os.environ.get('G3_BENCH_WATCHDOG_S', '900')),
                           max(60.0, 30.0 * passes['serial']['wall_s']))
                lim2 = float(os.environ.get('G3_BENCH_OVERLAP_LIMIT_S', lim2))

                def _overlap_expired(lim2=lim2):
                    sys.stderr.write('bench.py: rank %d: the overlapped schedule made no progress for %.0f s; the line of the serial-'
                                     'collective pass stands (G3_DIST_SERIAL_COLL=1 selects it)\n' % (rank, lim2))
                    if rank == 0 and serial_line[0] is not None:
                        ln = dict(serial_line[0])
                        ln['overlap_timeout'] = True
                        ln['overlap_limit_s'] = lim2
                        _emit_line(ln)
                    sys.stderr.flush()
                    os._exit(4)
                overlap_wd = threading.Timer(lim2, _overlap_expired)
                overlap_wd.daemon = True
                overlap_wd.start()
        t_wall0 = time.perf_counter()
        with _StdoutToStderr():             # (the first collective of a communicator prints RCCL's banner)
            for _ in range(args.warmup):
                step()
        # one GPU: HIP events around the bulk GEMM launches only; several GPUs: around every 16th MFMA GEMM launch of rank 0
        dev.prof_enable(0 if args.no_prof else (3 if use_dist else 1))
        dev.prof_reset()
        if use_dist:
            dgp.comm_stats()         # reset: the warm-up's lazy RCCL connection setup is not part of the timed steps
        if use_dist and native and not args.no_prof:
            dgp.prof_enable(2)       # every MFMA GEMM launch of the driver's bulk stream (two staircase launches per step)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step_s = []
        for _ in range(args.steps):
            ts0 = time.perf_counter()
            step()                   # (ends with a host synchronisation: the scalars of the evaluation come back)
            step_s.append(time.perf_counter() - ts0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        prof = dev.prof_collect()
        dev.prof_enable(False)
        if use_dist and native and not args.no_prof:
            pb = dgp.prof_collect()
            for k_ in pb:            # the bulk stream's launches are where a rank's flops are
                for f_ in ('count', 'ms', 'work'):
                    prof[k_][f_] += pb[k_][f_]
        if world > 1:
            t = torch.tensor([elapsed] + step_s, dtype=torch.float64, device=tdev if dist.get_backend() == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)           # the slowest rank, for the whole region and step by step
            elapsed, step_s = float(t[0].item()), [float(v) for v in t[1:]]
        comm_all = None
        if use_dist:
            # per-rank collective counts, bytes over the fabric (sent + received) and host seconds spent waiting
            mine = dgp.comm_stats()
            if world > 1:
                comm_all = [None] * world
                dist.all_gather_object(comm_all, mine)
            else:
                comm_all = [mine]
        if overlap_wd is not None:
            overlap_wd.cancel()
            overlap_wd = None
        passes[sched] = dict(elapsed=elapsed, step_s=step_s, prof=prof, comm_all=comm_all, logp=float(result['logp']),
                             draws=result.get('draws'), cov_tries=result.get('cov_tries', 0), parallelism=parallelism,
                             wall_s=time.perf_counter() - t_wall0)
        if sched != schedules[-1]:
            # the serial pass is complete: its line is ready should the overlapped pass never come back; then the driver goes
            # (communicators, streams, buffers) and the next schedule's is created in the same processes
            if rank == 0:
                serial_line[0] = _pass_line(args, world, N, d, M, S, passes[sched], sched)
            dgp.close()
            dgp = None
            if world > 1:
                dist.barrier()
        if sched == schedules[-1]:
            break
    # the headline pass: the faster of those that reproduced the pin (decided on rank 0's numbers, which are the maxima
    # over ranks; every rank holds the same step times)
    pin = _golden_logp(N, d, M, seed, args.kernel) if not (args.f32 or S > 0) else None

    def _ok(pd):
        return pin is None or abs(pd['logp'] - pin) <= 1e-8 * abs(pin)
    order = sorted(passes, key=lambda k_: (not _ok(passes[k_]), float(np.median(passes[k_]['step_s']))))
    chosen = order[0]
    pd = passes[chosen]
    elapsed, step_s, prof, comm_all, parallelism = pd['elapsed'], pd['step_s'], pd['prof'], pd['comm_all'], pd['parallelism']
    result['logp'] = pd['logp']
    if pd['draws'] is not None:
        result['draws'], result['cov_tries'] = pd['draws'], pd['cov_tries']

    dist_grad = None
    if use_dist and native and args.grad and S == 0:
        # dlogp on the distributed covariance (collective: every rank): one step in gradient mode -- the identity rides
        # through the factorisation as N / world more right-hand-side rows per rank -- then g3_dist_gp_dlogp
        dgp.set_grad(True)
        dgp.step(spec_n, spec_f, Xd, Xsd, dd)              # the re-planned buffers' first touch
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        lp_g = dgp.step(spec_n, spec_f, Xd, Xsd, dd)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        _, gmap_g, slots_g, alpha_g = dgp.dlogp(spec_n, Xd)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        dgp.set_grad(False)
        tt = torch.tensor([t2 - t1, t3 - t2], dtype=torch.float64, device=tdev if (world > 1 and dist.get_backend() == 'nccl') else 'cpu')
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist_grad = {'step_grad_mode_ms': float(tt[0]) * 1e3, 'dlogp_ms': float(tt[1]) * 1e3, 'nslots': int(gmap_g.nslots),
                     'logp_grad_mode': float(lp_g), 'grad_natural': [float(v) for v in slots_g],
                     'alpha_norm': float(np.linalg.norm(alpha_g)),
                     'note': 'gradient mode: the factorisation also turns the identity into the rank\'s rows of L^-T (N^3/3 more '
                             'flops over all ranks); dlogp = alpha + K^-1 row blocks (N^3/3, gathered panels + staircase GEMMs) + '
                             'g3_gram_grad_rows + one all-reduce; after the timed region, not part of value'}
    failed = None
    if rank == 0:
        sec = elapsed / args.steps
        sec_med = float(np.median(step_s)) if step_s else sec     # median of the timed steps
        flops = step_flops(N, M, S)
        out = {
            'metric': 'GP logp+predict end-to-end (Gram+Cholesky+solves), N=%d %s: algorithmic TFLOP/s' % (N, 'fp32' if args.f32 else 'fp64'),
            'value': flops / sec_med / 1e12, 'unit': 'TFLOP/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': sec * 1e3, 'ms_per_step_median': sec_med * 1e3,
            'value_mean': flops / sec / 1e12,
            'value_note': 'value = algorithmic flops / MEDIAN step time (SURVEY.md 8d); ms_per_step = whole timed region / steps '
                          '(the mean), value_mean the figure that goes with it (rounds 1-3 quoted the mean: compare those with '
                          'value_mean / ms_per_step).  Inputs X, y, Xs are resident in HBM when the timed region starts; SURVEY.md '
                          '8d\'s logp metric starts at their host-to-device copy (1.3 MB here): api_ms is that host-to-host figure '
                          'through the public API',
            'higher_is_better': True, 'scaling': 'strong',
            'vs_baseline': None, 'dtype': 'f32' if args.f32 else 'f64', 'data': 'synthetic',
            'config': {'workload': args.kernel.upper() + '-kernel GaussianProcess, N=%d d=%d, M=%d test points: Gram + blocked '
                                   'Cholesky + L^-1 y (logp) + cross-Gram + %d-rhs trsm + posterior mean/variance'
                                   % (N, d, M, M), 'N': N, 'd': d, 'M': M, 'parallelism': parallelism},
            'e2e_sec': sec_med, 'logp': float(result['logp']),
        }
        if comm_all is not None:
            def per(v):
                o = {'calls_per_step': v['calls'] / args.steps, 'bytes_per_step': v['bytes'] / args.steps}
                if 'device_ms' in v:
                    o['device_ms_per_step'] = v['device_ms'] / args.steps
                else:
                    o['host_wait_s_per_step'] = v['wait_s'] / args.steps
                return o
            if driver_fallback:
                # a downgrade is never silent: it is in the line, and the run fails unless it was allowed explicitly
                out['driver_fallback'] = driver_fallback
            out['comm'] = {'per_rank': [{k: per(v) for k, v in r.items()} for r in comm_all],
                           'driver': 'native' if native else 'python', 'collectives_forced_at_world_1': bool(solo_pg or (native and world == 1)),
                           'transport': ('rccl (library-owned communicators)' if native_transport == 'rccl' else 'host callbacks over gloo (rehearsal)') if native else backend,
                           'note': 'bytes = sent + received by the rank; device_ms = HIP-event time of the collective calls on '
                                   'the stream each ran on (native driver); host_wait = host time inside work.wait() / blocking '
                                   'all-reduces (python driver: RCCL waits are stream dependencies, ~0 unless the host is the bottleneck)'}
        if len(passes) > 1 or chosen != 'default':
            out['schedule'] = chosen
            out['schedules'] = {k_: _pass_summary(args, N, M, S, v_, pin) for k_, v_ in passes.items()}
            out['schedule_note'] = ('serial: the diagonal-factor broadcast waits for the previous panel all-gather on every rank '
                                    '(G3_DIST_SERIAL_COLL=1: the two communicators are never in flight together); overlapped: the '
                                    'default, broadcast of block j+1 beside the all-gather of panel j.  Both passes ran in the same '
                                    'processes, serial first; the headline is the faster pass that reproduced the oracle pin')
        if S > 0:
            dr = result['draws']
            out['metric'] += ' (+ posterior covariance, its Cholesky and %d draws)' % S
            out['config']['workload'] = ('warped ' + out['config']['workload'] + ' + posterior covariance + its Cholesky + %d draws '
                                         'L Z + BoxCoxLinear mapping' % S)
            out['config']['draws'] = S
            out['draws'] = {'shape': list(dr.shape), 'finite': bool(np.all(np.isfinite(dr))), 'mean': float(np.mean(dr)),
                            'std': float(np.std(dr)), 'cov_jitter_tries': int(result.get('cov_tries', 0))}
        ph = {k: v['ms'] / args.steps for k, v in prof.items() if v['count']}
        out['phases_ms'] = ph
        if prof['gram']['count'] and prof['gram']['ms'] > 0:
            # algorithmic bytes of the lower-triangle Gram (N d s read + N(N+1)/2 s written) per second
            out['gram_gbps'] = prof['gram']['work'] / (prof['gram']['ms'] * 1e-3) / 1e9
        if prof['potrf']['count']:
            # the factorisation phase also carries the 1 + M right-hand-side rows; cholesky_tflops counts
            # only N^3/3 over that phase (conservative), factor_solve_tflops counts the solves too
            t_ph = prof['potrf']['ms'] / prof['potrf']['count'] * 1e-3
            out['cholesky_tflops'] = (N ** 3 / 3.0) / t_ph / 1e12
            out['factor_solve_tflops'] = prof['potrf']['work'] / prof['potrf']['count'] / t_ph / 1e12
        g = prof['gemm_bulk']
        kern = 'gemm_nt_kernel<%s,128,128,64,32> (eight waves per 128 x 128 tile), launches with >= 4096 tiles ' % ('float' if args.f32 else 'double') + \
               '(bulk panel updates of the blocked Cholesky and of the trsm)'
        if use_dist:
            # the row-block layout issues per-block updates (m = nb rows): a 1-in-16 sample of the MFMA GEMM launches of rank 0
            g = {k: sum(prof[t][k] for t in ('gemm_bulk', 'gemm_mid', 'gemm_small')) for k in ('count', 'ms', 'work')}
            kern = ('gemm_nt_kernel<%s,*>: rank 0, every staircase launch of the bulk stream + a 1-in-16 sample of the chain ' \
                    "stream's MFMA GEMM launches" if native else 'gemm_nt_kernel<%s,*>: 1-in-16 sample of the MFMA GEMM launches of ' \
                    'rank 0 (trailing updates and panel solves of its row blocks)') % ('float' if args.f32 else 'double')
        launches_div = args.steps
        if not use_dist and not g['count'] and not args.no_prof:
            # a size without a single >= 4096-tile launch (N < ~12000): one more, UNTIMED pass with HIP events around every
            # MFMA GEMM launch -- the dominant kernel class is then all of them (their events would perturb a timed small step)
            dev.prof_enable(2)
            dev.prof_reset()
            step()
            torch.cuda.synchronize()
            p2 = dev.prof_collect()
            dev.prof_enable(False)
            g = {k: sum(p2[t][k] for t in ('gemm_bulk', 'gemm_mid', 'gemm_small')) for k in ('count', 'ms', 'work')}
            launches_div = 1
            kern = 'gemm_nt_kernel<%s,*>: every MFMA GEMM launch of one extra, untimed pass (no launch of this size reaches ' \
                   '4096 tiles; panel updates and stripe solves)' % ('float' if args.f32 else 'double')
        if g['count'] and g['ms'] > 0:
            ach = g['work'] / (g['ms'] * 1e-3) / 1e12
            peak = FP32_MATRIX_PEAK_TFLOPS if args.f32 else FP64_MATRIX_PEAK_TFLOPS
            out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                               'frac': ach / peak,
                               'traffic': _traffic() if (world == 1 and not args.f32 and N == 32768) else None,
                               'traffic_source': 'profiles/r*_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate '
                                                 'passes of this command (not measured in this run)',
                               'kernel': kern,
                               'launches_per_step': g['count'] / launches_div,
                               'avg_launch_ms': g['ms'] / g['count'], 'avg_launch_flops': g['work'] / g['count'],
                               'note': 'HIP events per launch on the launching stream; launches on the two '
                                       'look-ahead streams overlap, so summed launch time exceeds wall time'}
        if dist_grad is not None:
            out['dlogp'] = dist_grad
        if not use_dist and args.grad:
            gmap = dev.grad_layout(prog_n)
            Yt = torch.empty((Np, Np), dtype=tdt, device=tdev)
            Kit = torch.empty((Np, Np), dtype=tdt, device=tdev)
            alt = torch.empty((1, Np), dtype=tdt, device=tdev)
            Yd, Kid, ald = wrap(Yt, Np, Np), wrap(Kit, Np, Np), wrap(alt, 1, Np)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                slots = dev.gp_dlogp(prog_n, gmap, Xd, N, d, Kd, Wd, ad, Yd, Kid, ald)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            tg = min(ts[1:])
            out['dlogp'] = {'ms': tg * 1e3, 'potri_tflops': (2.0 * N ** 3 / 3.0) / tg / 1e12, 'nslots': int(gmap.nslots),
                            'grad_natural': [float(v) for v in slots],
                            'note': 'K^-1 (2N^3/3 flops) + alpha + one pass over K^-1 for the kernel-parameter sums; '
                                    'after the timed region, not part of value'}
        if args.cpu_n < 0:
            args.cpu_n = cpu_default_n(N)
        if world == 1 and args.cpu_n > 0:
            cb, lp_cpu = cpu_baseline(N, d, M, seed, min(args.cpu_n, N))
            out['cpu_baseline'] = cb
            if min(args.cpu_n, N) == N:      # the sample is the whole workload: the CPU logp must agree too
                out['cpu_baseline']['logp'] = float(lp_cpu)
        # full-size pin: the CPU oracle's logp at this exact configuration (tests/golden/fullsize.json,
        # written once by oracle/gen_fullsize.py in the build container; data, not code)
        ref = _golden_logp(N, d, M, seed, args.kernel) if not (args.f32 or S > 0) else None
        tol = 1e-8
        if ref is not None:
            out['logp_ref'] = ref
            out['logp_rel_err'] = abs(out['logp'] - ref) / abs(ref)
        elif S > 0:
            # config 5 (warped GP + draws): the fp64 oracle's full-size pin, compared at the fp32 tolerance
            g5 = _golden_c5(N, d, M, seed, S)
            if g5 is not None:
                ref, tol = float(g5['logp']), (1e-4 if args.f32 else 1e-8)
                out['logp_ref'] = ref
                out['logp_rel_err'] = abs(out['logp'] - ref) / abs(ref)
                rows = g5['draw_rows']
                got = np.asarray(result['draws'])[rows]
                want = np.asarray(g5['draw_values'])
                out['draws']['max_abs_err_vs_oracle_rows'] = float(np.max(np.abs(got - want)))
                out['draws']['oracle_rows'] = rows
        if args.measure_traffic is None:      # default: the headline line carries a traffic figure measured in the run itself
            profiled = 'rocprof' in os.environ.get('LD_PRELOAD', '') or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ)
            args.measure_traffic = (world == 1 and not args.f32 and N == 32768 and not args.no_prof and not profiled
                                    and shutil_which('rocprofv3'))
        if args.measure_traffic and world == 1 and 'roofline' in out:
            wl = ['--points', str(N), '--dims', str(d), '--queries', str(M), '--kernel', args.kernel] + (['--f32'] if args.f32 else []) \
                + (['--draws', str(args.draws)] if args.draws >= 0 else [])
            tb, note = _measure_traffic(wl)
            if tb is not None:
                out['roofline']['traffic'] = tb
                out['roofline']['traffic_source'] = note
            else:                             # keep the committed figure, say why
                out['roofline']['traffic_source'] += '; live measurement unavailable: ' + note
        if args.api and world == 1 and not use_dist and S == 0:
            out['api_ms'] = _api_timing(g3, X, y, Xs, d, args.kernel, npdt)
        print(json.dumps(out), flush=True)
        if out.get('driver_fallback') and os.environ.get('G3_DIST_ALLOW_FALLBACK', '0') != '1':
            failed = ('bench.py: the native multi-GPU driver was not used (%s); the line above was measured with the '
                      'torch.distributed driver -- set G3_DIST_ALLOW_FALLBACK=1 to accept that' % out['driver_fallback'])
        if ref is not None and not out['logp_rel_err'] <= tol:
            failed = 'bench.py: logp %.12f differs from the oracle pin %.12f by more than %g relative' % (out['logp'], ref, tol)
    # every rank learns the verdict BEFORE the group is torn down, so a failed pin ends all ranks at once
    # instead of leaving the others to the launcher's timeout
    if world > 1:
        flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=tdev if dist.get_backend() == 'nccl' else 'cpu')
        dist.broadcast(flag, src=0)
        if int(flag.item()) and not failed:
            failed = 'bench.py: rank 0 reported a failed oracle pin'
    if watchdog is not None:
        # the line is out (or the verdict known): from here on a stuck teardown must neither print a second line nor hold
        # the node -- the run watchdog is replaced by one that just ends the process with the status already decided
        watchdog.cancel()
        import threading
        sys.stdout.flush()
        td = threading.Timer(120.0, lambda: os._exit(1 if failed else 0))
        td.daemon = True
        td.start()
    if use_dist and native:
        dgp.close()                       # communicators and driver buffers, before the contexts they live on
    if world > 1 or solo_pg:
        dist.destroy_process_group()
    # explicit teardown while the HIP runtime is alive (streams, events, pinned buffers, workspaces)
    g3.Device.close_all()
    if failed:
        raise SystemExit(failed)


if __name__ == '__main__':
    main()
