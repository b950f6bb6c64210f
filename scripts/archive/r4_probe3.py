"""How many resident kernels (each on a stream of its own) can run beside the chain and bulk streams before the bulk
updates slow down?  (measurement build libg3hip_probe.so, mode 6 of g3x_probe)"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault('G3_LIB_PATH', os.path.join(R, 'g3py_amd', 'lib', 'libg3hip_probe.so'))
import torch
import g3py_amd as g3
dev = g3.Device(0)
if os.environ.get('PROBE_TORCH_STREAM'):
    hp = torch.cuda.Stream(priority=-1); torch.cuda.set_stream(hp); dev.set_stream(hp.cuda_stream)
lib = C.CDLL(os.environ['G3_LIB_PATH'])
lib.g3x_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
out = (C.c_double * 8)()
for prio in (2, 1, 0):
    for E in (0, 1, 2, 3, 4, 6):
        rc = lib.g3x_probe(dev.ctx, 6, E, prio, 2000, 6, out)
        print('torch stream %s | %d resident kernels, priority %d: rc %d  load %.2f ms, trickle %.2f us per tiny kernel' % (
            os.environ.get('PROBE_TORCH_STREAM', '0'), E, prio, rc, out[4], out[0]), flush=True)
