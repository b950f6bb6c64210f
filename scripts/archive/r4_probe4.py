"""Does the LDS request of ONE resident workgroup slow the bulk updates?  (libg3hip_probe.so, mode 5 of g3x_probe)"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault('G3_LIB_PATH', os.path.join(R, 'g3py_amd', 'lib', 'libg3hip_probe.so'))
import g3py_amd as g3
dev = g3.Device(0)
lib = C.CDLL(os.environ['G3_LIB_PATH'])
lib.g3x_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
out = (C.c_double * 8)()
for rep in range(2):
    lib.g3x_probe(dev.ctx, 2, 0, 0, 0, 6, out)
    print('no resident workgroup: load %.2f ms' % out[4], flush=True)
for lds in (16384, 40960, 65536, 66048, 80000, 98304, 131072, 163000):
    for E in (1, 16):
        rc = lib.g3x_probe(dev.ctx, 5, E, lds, 3500, 6, out)
        print('E %2d resident (s_sleep only), lds %6d: rc %d load %.2f ms' % (E, lds, rc, out[4]), flush=True)
