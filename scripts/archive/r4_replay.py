"""P = 2 / 4 / 8 on ONE GPU, compute side: every rank of a P-rank evaluation replayed alone (replay transport,
include/g3hip.h::g3_dist_create_replay), each collective a device-to-device copy of the bytes the rank would receive.
Writes profiles/r04_replay_<config>_P<P>_nb<nb>.json: per-rank step time, bulk-stream MFMA time, the rank's turns on the
diagonal chain, its panel solves, the bytes it would send + receive and the bandwidth that hides them under its compute.
usage: python scripts/r4_replay.py [c4|c5] [P ...]   (env R4_NB=512,1024)"""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import g3py_amd as g3
from g3py_amd.distributed import NativeDistributedGP
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c4'
Ps = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
nbs = [int(a) for a in os.environ.get('R4_NB', '512,1024').split(',')]
if cfg == 'c4':
    N, d, M, seed, dt, tdt = 32768, 4, 1024, 1004, np.float64, torch.float64
else:                     # config 5's shape (the warping and the draws are host / M x M work outside the factorisation)
    N, d, M, seed, dt, tdt = 65536, 16, 4096, 1005, np.float32, torch.float32
N = int(os.environ.get('R4_N', N))
X, y, Xs = bench.synth(N, d, M, seed)
dev = g3.Device(0)
spec_f = ('SE', 1.0, np.ones(d), None)
from oracle import g3_oracle as orc          # (only for the noise wrapper of the kernel spec: no arithmetic)
spec_n = orc.with_noise(spec_f, 0.1)
def tens(a): return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
Xt, Xst, dlt = tens(X), tens(Xs), tens(y[None, :])
def wrap(t, r, c): return dev.wrap(t.data_ptr(), r, c, t.stride(0) if t.dim() == 2 else c, dt, keep=t)
Xd, Xsd, dd = wrap(Xt, N, d), wrap(Xst, M, d), wrap(dlt, 1, N)
flops = bench.step_flops(N, M)
# the one-GPU baseline: the in-library sweep (what bench.py times at N = 1)
from g3py_amd import _lib
from g3py_amd.device import compile_spec
Np_, Mp_ = _lib.roundup(N), _lib.roundup(M, _lib.G3_RHS_PAD)
Kt = torch.empty((Np_ + 128 + Mp_, Np_), dtype=tdt, device='cuda'); at = torch.empty((1, Np_), dtype=tdt, device='cuda')
mut = torch.empty((1, Mp_), dtype=tdt, device='cuda'); sst = torch.empty((1, Mp_), dtype=tdt, device='cuda'); Wt = torch.empty((Np_, 128), dtype=tdt, device='cuda')
pn, pf = compile_spec(spec_n, d), compile_spec(spec_f, d)
ts = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev.gp_factor_predict(pn, pf, Xd, N, d, dd, Xsd, M, wrap(Kt, Np_ + 128 + Mp_, Np_), wrap(Wt, Np_, 128), wrap(at, 1, Np_), wrap(mut, 1, Mp_), wrap(sst, 1, Mp_))
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
lib_ms = min(ts) * 1e3
del Kt
torch.cuda.empty_cache()
print('%s: in-library sweep on one GPU %.2f ms' % (cfg, lib_ms), flush=True)
outdir = os.path.join(R, 'gpurun_out', 'replay'); os.makedirs(outdir, exist_ok=True)
for nb in nbs:
    ref = NativeDistributedGP(dev, None, 0, 1, N, d, M, nb=nb, dtype=dt, transport='callbacks', keep=True)
    lp_ref = ref.step(spec_n, spec_f, Xd, Xsd, dd); ref.comm_stats()
    ts = []
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ref.step(spec_n, spec_f, Xd, Xsd, dd); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    one_ms = min(ts) * 1e3
    cs1 = ref.comm_stats(); ph1 = ref.phase_stats()
    ref_last = dict(ref.last)
    print('%s nb %d: reference pass (world-1 driver, host-callback transport) %.2f ms, logp %.9f; diagonal chain %d blocks %.2f ms (%.3f ms each), panel solves %.2f ms'
          % (cfg, nb, one_ms, lp_ref, ph1['diag']['calls'] / 2, ph1['diag']['device_ms'] / 2, ph1['diag']['device_ms'] / max(ph1['diag']['calls'], 1),
             ph1['solve']['device_ms'] / 2), flush=True)
    for P in Ps:
        ranks = []
        tot = dict(logdet=0.0, quad=0.0, mean=np.zeros(M), ss=np.zeros(M))
        for r in range(P):
            rp = NativeDistributedGP(dev, None, r, P, N, d, M, nb=nb, dtype=dt, transport='replay', reference=ref)
            rp.step(spec_n, spec_f, Xd, Xsd, dd); rp.comm_stats()          # first touch of its buffers
            rp.prof_enable(2)
            ts = []
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter(); rp.step(spec_n, spec_f, Xd, Xsd, dd); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            cs, ph, pr = rp.comm_stats(), rp.phase_stats(), rp.prof_collect()
            gemm_ms = sum(v['ms'] for v in pr.values()) / 2
            gemm_work = sum(v['work'] for v in pr.values()) / 2
            ms = min(ts) * 1e3
            by = {k: cs[k]['bytes'] / 2 for k in cs}
            recv = by['bcast'] + by['allgather'] / 2.0       # (all-gather bytes are sent + received)
            ranks.append({'rank': r, 'ms_per_step': ms, 'bulk_gemm_ms': gemm_ms, 'bulk_gemm_tflops': gemm_work / max(gemm_ms, 1e-9) / 1e9,
                          'diag_blocks': ph['diag']['calls'] / 2, 'diag_ms': ph['diag']['device_ms'] / 2,
                          'solve_ms': ph['solve']['device_ms'] / 2, 'copies_ms': sum(cs[k]['device_ms'] for k in ('bcast', 'allgather')) / 2,
                          'bytes_sent_plus_received': sum(by.values()), 'bytes_received': recv,
                          'fabric_GBps_to_hide_receives_under_compute': recv / (ms * 1e-3) / 1e9,
                          'collective_calls': {k: cs[k]['calls'] / 2 for k in cs}})
            tot['logdet'] += rp.last['logdet']; tot['quad'] += rp.last['quad']; tot['mean'] += rp.last['mean']; tot['ss'] += rp.last['ss']
            rp.close()
        worst = max(x['ms_per_step'] for x in ranks)
        chain_ms = sum(x['diag_ms'] for x in ranks)           # every diagonal block's update + factorisation, one after the other
        lp = -0.5 * N * np.log(2 * np.pi) - 0.5 * tot['quad'] - tot['logdet']
        line = {'what': 'replay of every rank of a %d-rank evaluation on ONE MI355X: compute side measured, fabric not' % P,
                'config': cfg, 'N': N, 'd': d, 'M': M, 'dtype': np.dtype(dt).name, 'nb': nb, 'world': P,
                'one_gpu_in_library_ms': lib_ms, 'ms_per_step_slowest_rank': worst, 'speedup_compute_side': lib_ms / worst,
                'serial_diagonal_chain_ms': chain_ms,
                'speedup_if_chain_bound': lib_ms / max(worst, chain_ms),
                'note': 'ms_per_step of a rank = its whole step with the other ranks infinitely fast (their factors and panels arrive '
                        'by device copy); serial_diagonal_chain_ms = sum over ALL diagonal blocks of the owner\'s update + factorisation '
                        'time, which no rank can overlap with the next block\'s (the critical chain of the P-rank run, fabric latency '
                        'excluded); the P-rank step is at least max(slowest rank, chain)',
                'logp_from_the_ranks_contributions': lp, 'logp_one_rank': lp_ref, 'logp_rel_err': abs(lp - lp_ref) / abs(lp_ref),
                'mean_max_abs_err': float(np.abs(tot['mean'] - ref_last['mean']).max()), 'ss_max_rel_err': float((np.abs(tot['ss'] - ref_last['ss']) / np.abs(ref_last['ss'])).max()),
                'algorithmic_tflops_if_compute_bound': flops / (max(worst, chain_ms) * 1e-3) / 1e12,
                'per_rank': ranks}
        fn = os.path.join(outdir, 'r04_replay_%s_P%d_nb%d.json' % (cfg, P, nb))
        json.dump(line, open(fn, 'w'), indent=1)
        print('  P %d: slowest rank %.2f ms (%.2fx), serial chain %.2f ms -> >= %.2f ms (%.2fx); receives %.2f GB per rank -> %.0f GB/s to hide; logp rel err %.1e'
              % (P, worst, lib_ms / worst, chain_ms, max(worst, chain_ms), lib_ms / max(worst, chain_ms), ranks[0]['bytes_received'] / 1e9,
                 max(x['fabric_GBps_to_hide_receives_under_compute'] for x in ranks), line['logp_rel_err']), flush=True)
    ref.close()
