"""Phase timing of the 128 x 128 diagonal-block kernel (development aid; needs the -DG3_DIAG_TIMING build:
G3_LIB_PATH=g3py_amd/lib/exp_diagts.so python scripts/diag_phases.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
dev = g3.Device(0)
rng = np.random.default_rng(0)
B = rng.standard_normal((128, 64)); K = B @ B.T / 64 + np.eye(128)
acc = np.zeros((20, 128))
for it in range(20):
    Kt = torch.from_numpy(K).cuda()
    dev.potrf(dev.wrap(Kt.data_ptr(), 128, 128, 128, np.float64), 128)
    torch.cuda.synchronize()
    ts = (C.c_ulonglong * 128)()
    assert dev.lib.g3_dbg_read(ts) == 0
    acc[it] = np.array(ts[:], dtype=np.float64)
t = np.median(acc[5:] - acc[5:, :1], axis=0) * 0.01      # us from the start of step 0 (100 MHz counter)
print('step: start  after_b1  panel_done  after_b2  trailing_done | diag16 (wave k): begin end (us)')
for k in range(8):
    print('%d: %7.2f %7.2f %7.2f %7.2f %7.2f | %7.2f %7.2f  (%.2f)' % ((k,) + tuple(t[8 * k:8 * k + 5]) + (t[64 + 2 * k], t[64 + 2 * k + 1], t[64 + 2 * k + 1] - t[64 + 2 * k])))
c = np.median(acc[5:, 96:128], axis=0)
print('16 x 16 tile routine, first tile, shader cycles (20-bit counter): block step: factor+inverse, panel element, mfma+store, inverse mfma, | total')
for q in range(4):
    d = [(c[8 * q + i + 1] - c[8 * q + i])  for i in range(4)]
    nxt = (c[8 * (q + 1)] - c[8 * q + 4])  if q < 3 else 0
    print(q, d, 'to next step', nxt, '| total', sum(d))

# fused 256-wide kernel (G3_NB=256 makes potrf(256) one call of it): phases seen by wave 7
if os.environ.get('G3_NB') == '256':
    B2 = rng.standard_normal((256, 128)); K2 = B2 @ B2.T / 128 + np.eye(256)
    acc2 = np.zeros((20, 128))
    for it in range(20):
        Kt = torch.from_numpy(K2).cuda()
        dev.potrf(dev.wrap(Kt.data_ptr(), 256, 256, 256, np.float64), 256)
        torch.cuda.synchronize()
        ts = (C.c_ulonglong * 128)()
        assert dev.lib.g3_dbg_read(ts) == 0
        acc2[it] = np.array(ts[:], dtype=np.float64)
    u = np.median(acc2[5:, 88:93] - acc2[5:, 88:89], axis=0) * 0.01
    print('potrf256 (us): first diagonal block %.2f | L10 product %.2f | SYRK %.2f | second diagonal block %.2f | total %.2f'
          % (u[1], u[2] - u[1], u[3] - u[2], u[4] - u[3], u[4]))
