"""Time the fused 128x128 diagonal kernel alone (development aid): potrf of 128x128 blocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import g3py_amd as g3
dev = g3.Device(0); st = torch.cuda.Stream(); torch.cuda.set_stream(st); dev.set_stream(st.cuda_stream)
rng = np.random.default_rng(0)
B = rng.standard_normal((128, 64)); K = B @ B.T / 64 + np.eye(128)
Kt = torch.from_numpy(K).cuda()
Kd = dev.wrap(Kt.data_ptr(), 128, 128, 128, np.float64)
Ks = [torch.from_numpy(K).cuda() for _ in range(200)]
for t in Ks[:5]:
    dev.potrf(dev.wrap(t.data_ptr(), 128, 128, 128, np.float64), 128)
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in Ks[5:]:
    dev.potrf(dev.wrap(t.data_ptr(), 128, 128, 128, np.float64), 128)
torch.cuda.synchronize()
print('potrf(128) incl. host round trip: %.1f us per call' % ((time.perf_counter() - t0) / 195 * 1e6))
