"""Timeline of the chain server (measurement build libg3hip_ctrace.so): per panel, what workgroup 0 and two workers did when.
usage: python scripts/r4_chain_trace.py [n]"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault('G3_LIB_PATH', os.path.join(R, 'g3py_amd', 'lib', 'libg3hip_ctrace.so'))
import numpy as np, torch
import g3py_amd as g3
lib = C.CDLL(os.environ['G3_LIB_PATH'])
MAXP = 254
buf = (C.c_ulonglong * (MAXP * 16 * 3))()
if len(sys.argv) > 1 and sys.argv[1] == 'bench':
    # the timeline of the LAST sweep of a bench.py run (arguments after 'bench' go to bench.py)
    sys.argv = ['bench.py', '--cpu-n', '0', '--no-api', '--no-measure-traffic', '--skip-events'] + sys.argv[2:]
    import bench
    bench.main()
else:
    dev = g3.Device(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    rng = np.random.default_rng(0)
    B = rng.standard_normal((n, n // 4)); K = B @ B.T / n + np.eye(n)
    for rep in range(3):
        t = torch.from_numpy(K).cuda()
        lib.g3x_chain_trace(buf, 1)
        torch.cuda.synchronize()
        assert dev.potrf(dev.wrap(t.data_ptr(), n, n, n, np.float64), n) == 0
        torch.cuda.synchronize()
lib.g3x_chain_trace(buf, 0)
a = np.array(buf, dtype=np.uint64).reshape(MAXP, 16, 3).astype(np.float64)
t0 = a[a > 0].min()
a = np.where(a > 0, (a - t0) * 0.01, np.nan)   # us
names0 = ['D0 start', 'D0 end', 'D1 start', 'D1 end', 'D2 start', 'D2 end', 'D3 start', 'D3 end']
namesw = {0: 'saw D0', 1: 'T0 own', 2: 'T0 all', 3: 'saw D1', 4: 'T1 own', 5: 'T1 all', 6: 'saw D2', 7: 'T2 own', 8: 'T2 all',
          9: 'S0 all', 10: 'S1 all', 11: 'panel done', 12: 'rest ok', 13: 'Tex all', 14: 'bulk ok', 15: 'st3 own'}
for s in range(MAXP):
    if np.all(np.isnan(a[s])): break
    if s > 5 and s % 6: continue
    ev = []
    for k in range(8):
        if not np.isnan(a[s, k, 0]): ev.append((a[s, k, 0], 'wg0 ' + names0[k]))
    for k in range(16):
        for r, nm in ((1, 'w0 '), (2, 'wL ')):
            if not np.isnan(a[s, k, r]): ev.append((a[s, k, r], nm + namesw[k]))
    ev.sort()
    print('--- panel %d' % s)
    prev = ev[0][0]
    for (tt, nm) in ev:
        print('  %9.1f us  (+%6.1f)  %s' % (tt, tt - prev, nm))
        prev = tt
