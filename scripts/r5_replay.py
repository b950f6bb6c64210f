"""P = 2 / 4 / 8 on ONE GPU, compute side (round 5): every rank of a P-rank evaluation replayed alone (replay transport,
include/g3hip.h::g3_dist_create_replay), each collective a device-to-device copy of the bytes the rank would receive.
Variants are environment settings read when a context / driver is created (a fresh g3.Device per variant):
  legacy   G3_DIST_FULLINV=0             the round-4 schedule: (L, block inverses) broadcast, stripe-recursion panel solve
  fullinv  (default)                     V = L^-1 broadcast, a rank's panel solve is one K-triangular MFMA product
  bigK<n>  G3_GEMM_BIG_MIN_K=<n>         the 128 x 128 tile from <n> tiles on when K >= 1024 (default 1024)
Writes gpurun_out/replay/r05_replay_<config>_P<P>_<variant>.json: per-rank step time, bulk-stream MFMA time and rate, the rank's
turns on the diagonal chain, its panel solves, bytes, the bandwidth that hides its receives under its compute.
usage: python scripts/r5_replay.py [c4|c5] [P ...]     env R5_VARIANTS=legacy,fullinv,bigK512  R5_NB=1024"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import g3py_amd as g3
from g3py_amd.distributed import NativeDistributedGP
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
Ps = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
nb = int(os.environ.get('R5_NB', '1024'))
variants = os.environ.get('R5_VARIANTS', 'legacy,fullinv').split(',')
if cfg == 'c4':
    N, d, M, seed, dt, tdt = 32768, 4, 1024, 1004, np.float64, torch.float64
else:                     # config 5's shape (the warping and the draws are host / M x M work outside the factorisation)
    N, d, M, seed, dt, tdt = 65536, 16, 4096, 1005, np.float32, torch.float32
N = int(os.environ.get('R5_N', N))
X, y, Xs = bench.synth(N, d, M, seed)
spec_f = ('SE', 1.0, np.ones(d), None)
from oracle import g3_oracle as orc          # (only for the noise wrapper of the kernel spec: no arithmetic)
spec_n = orc.with_noise(spec_f, 0.1)
flops = bench.step_flops(N, M)
KNOBS = ('G3_DIST_FULLINV', 'G3_GEMM_BIG_MIN_K', 'G3_GEMM_BIG_MIN', 'G3_TRSM_THIN_MAX', 'G3_SIDE_LDS', 'G3_DIST_DEAL', 'R5_IDLE', 'G3_DIST_OWN_CHAIN', 'G3_DIST_PAD_STREAMS', 'G3_DIST_BULK_MASK', 'G3_DIST_BULK_MASK_MODE')


def env_of(name):
    e = {}
    for part in name.split('+'):
        if part == 'legacy':
            e['G3_DIST_FULLINV'] = '0'
        elif part.startswith('bigK'):
            e['G3_GEMM_BIG_MIN_K'] = part[4:]
        elif part.startswith('big'):
            e['G3_GEMM_BIG_MIN'] = part[3:]
        elif part.startswith('lds'):
            e['G3_SIDE_LDS'] = part[3:]
        elif part == 'userchain':
            e['G3_DIST_OWN_CHAIN'] = '0'
        elif part.startswith('pad'):
            e['G3_DIST_PAD_STREAMS'] = part[3:]
        elif part in ('maskalways', 'masknever'):   # which bulk stream a plan takes (default: by plan, g3_dist_plan)
            e['G3_DIST_BULK_MASK_MODE'] = part[4:]
        elif part.startswith('mask'):
            e['G3_DIST_BULK_MASK'] = part[4:]
        elif part == 'snake':
            e['G3_DIST_DEAL'] = 'snake'
        elif part.startswith('idle'):           # idle<n>: n idle high- and n idle low-priority streams created before the drivers
            e['R5_IDLE'] = part[4:]
    return e


dev0 = g3.Device(0)
def tens(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
Xt, Xst, dlt = tens(X), tens(Xs), tens(y[None, :])
# the one-GPU baseline: the in-library sweep (what bench.py times at N = 1)
from g3py_amd import _lib
from g3py_amd.device import compile_spec
def wrap(dev, t, r, c): return dev.wrap(t.data_ptr(), r, c, t.stride(0) if t.dim() == 2 else c, dt, keep=t)
Np_, Mp_ = _lib.roundup(N), _lib.roundup(M, _lib.G3_RHS_PAD)
Kt = torch.empty((Np_ + 128 + Mp_, Np_), dtype=tdt, device='cuda'); at = torch.empty((1, Np_), dtype=tdt, device='cuda')
mut = torch.empty((1, Mp_), dtype=tdt, device='cuda'); sst = torch.empty((1, Mp_), dtype=tdt, device='cuda'); Wt = torch.empty((Np_, 128), dtype=tdt, device='cuda')
pn, pf = compile_spec(spec_n, d), compile_spec(spec_f, d)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev0.gp_factor_predict(pn, pf, wrap(dev0, Xt, N, d), N, d, wrap(dev0, dlt, 1, N), wrap(dev0, Xst, M, d), M, wrap(dev0, Kt, Np_ + 128 + Mp_, Np_),
                           wrap(dev0, Wt, Np_, 128), wrap(dev0, at, 1, Np_), wrap(dev0, mut, 1, Mp_), wrap(dev0, sst, 1, Mp_))
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
lib_ms = min(ts) * 1e3
del Kt
torch.cuda.empty_cache()
print('%s: in-library sweep on one GPU %.2f ms' % (cfg, lib_ms), flush=True)
outdir = os.path.join(R, 'gpurun_out', 'replay'); os.makedirs(outdir, exist_ok=True)
for var in variants:
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env_of(var))
    idle = [torch.cuda.Stream(priority=pr) for _ in range(int(os.environ.get('R5_IDLE', '0'))) for pr in (-1, 0)]
    dev = g3.Device(0)                               # the knobs are read when a context is created
    Xd, Xsd, dd = wrap(dev, Xt, N, d), wrap(dev, Xst, M, d), wrap(dev, dlt, 1, N)
    ref = NativeDistributedGP(dev, None, 0, 1, N, d, M, nb=nb, dtype=dt, transport='callbacks', keep=True)
    lp_ref = ref.step(spec_n, spec_f, Xd, Xsd, dd); ref.comm_stats()
    ref_last = dict(ref.last)
    print('%s nb %d %s: reference pass logp %.9f' % (cfg, nb, var, lp_ref), flush=True)
    for P in Ps:
        ranks = []
        tot = dict(logdet=0.0, quad=0.0, mean=np.zeros(M), ss=np.zeros(M))
        for r in range(P):
            rp = NativeDistributedGP(dev, None, r, P, N, d, M, nb=nb, dtype=dt, transport='replay', reference=ref)
            rp.step(spec_n, spec_f, Xd, Xsd, dd); rp.comm_stats()          # first touch of its buffers
            rp.prof_enable(2)
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); rp.step(spec_n, spec_f, Xd, Xsd, dd); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            cs, ph, pr = rp.comm_stats(), rp.phase_stats(), rp.prof_collect()
            nrep = len(ts)
            gemm_ms = sum(v['ms'] for v in pr.values()) / nrep
            gemm_work = sum(v['work'] for v in pr.values()) / nrep
            ms = min(ts) * 1e3
            by = {k: cs[k]['bytes'] / nrep for k in cs}
            recv = by['bcast'] + by['allgather'] / 2.0       # (all-gather bytes are sent + received)
            ranks.append({'rank': r, 'ms_per_step': ms, 'bulk_gemm_ms': gemm_ms, 'bulk_gemm_tflops': gemm_work / max(gemm_ms, 1e-9) / 1e9,
                          'bulk_by_tile': {k: {'ms': v['ms'] / nrep, 'tflops': v['work'] / max(v['ms'], 1e-9) / 1e9, 'launches': v['count'] / nrep}
                                           for k, v in pr.items() if v['count']},
                          'diag_blocks': ph['diag']['calls'] / nrep, 'diag_ms': ph['diag']['device_ms'] / nrep,
                          'solve_ms': ph['solve']['device_ms'] / nrep, 'copies_ms': sum(cs[k]['device_ms'] for k in ('bcast', 'allgather')) / nrep,
                          'bytes_sent_plus_received': sum(by.values()), 'bytes_received': recv,
                          'fabric_GBps_to_hide_receives_under_compute': recv / (ms * 1e-3) / 1e9,
                          'collective_calls': {k: cs[k]['calls'] / nrep for k in cs}})
            tot['logdet'] += rp.last['logdet']; tot['quad'] += rp.last['quad']; tot['mean'] += rp.last['mean']; tot['ss'] += rp.last['ss']
            rp.close()
        worst = max(x['ms_per_step'] for x in ranks)
        chain_ms = sum(x['diag_ms'] for x in ranks)           # every diagonal block's update + factorisation (+ inversion), one after the other
        lp = -0.5 * N * np.log(2 * np.pi) - 0.5 * tot['quad'] - tot['logdet']
        line = {'what': 'replay of every rank of a %d-rank evaluation on ONE MI355X: compute side measured, fabric not' % P,
                'config': cfg, 'N': N, 'd': d, 'M': M, 'dtype': np.dtype(dt).name, 'nb': nb, 'world': P, 'variant': var, 'env': env_of(var),
                'one_gpu_in_library_ms': lib_ms, 'ms_per_step_slowest_rank': worst, 'speedup_compute_side': lib_ms / worst,
                'serial_diagonal_chain_ms': chain_ms,
                'speedup_if_chain_bound': lib_ms / max(worst, chain_ms),
                'note': 'ms_per_step of a rank = its whole step with the other ranks infinitely fast (their factors and panels arrive '
                        'by device copy); serial_diagonal_chain_ms = sum over ALL diagonal blocks of the owner\'s update + factorisation '
                        '(+ inversion) time, which no rank can overlap with the next block\'s (the critical chain of the P-rank run, fabric '
                        'latency excluded); the P-rank step is at least max(slowest rank, chain)',
                'logp_from_the_ranks_contributions': lp, 'logp_one_rank': lp_ref, 'logp_rel_err': abs(lp - lp_ref) / abs(lp_ref),
                'mean_max_abs_err': float(np.abs(tot['mean'] - ref_last['mean']).max()),
                'ss_max_rel_err': float((np.abs(tot['ss'] - ref_last['ss']) / np.abs(ref_last['ss'])).max()),
                'algorithmic_tflops_if_compute_bound': flops / (max(worst, chain_ms) * 1e-3) / 1e12,
                'per_rank': ranks}
        fn = os.path.join(outdir, 'r05_replay_%s_P%d_%s.json' % (cfg, P, var))
        json.dump(line, open(fn, 'w'), indent=1)
        print('  %-14s P %d: slowest rank %.2f ms (%.2fx), serial chain %.2f ms; bulk %.1f-%.1f ms, solves %.1f-%.1f ms, diag turns %.1f-%.1f ms; '
              'receives %.2f GB per rank -> %.0f GB/s to hide; logp rel err %.1e'
              % (var, P, worst, lib_ms / worst, chain_ms, min(x['bulk_gemm_ms'] for x in ranks), max(x['bulk_gemm_ms'] for x in ranks),
                 min(x['solve_ms'] for x in ranks), max(x['solve_ms'] for x in ranks), min(x['diag_ms'] for x in ranks), max(x['diag_ms'] for x in ranks),
                 ranks[0]['bytes_received'] / 1e9, max(x['fabric_GBps_to_hide_receives_under_compute'] for x in ranks), line['logp_rel_err']), flush=True)
    ref.close()
    dev.close()
