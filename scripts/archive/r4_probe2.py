"""What do resident workgroups cost the bulk updates?  (measurement build libg3hip_probe.so, modes 2-5 of g3x_probe)"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault('G3_LIB_PATH', os.path.join(R, 'g3py_amd', 'lib', 'libg3hip_probe.so'))
import g3py_amd as g3
dev = g3.Device(0)
lib = C.CDLL(os.environ['G3_LIB_PATH'])
lib.g3x_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
out = (C.c_double * 8)()
names = {2: 'poll', 3: 'poll + acquire fence / 10 us', 4: 'poll + release fence / 10 us', 5: 's_sleep only'}
for rep in range(2):
    rc = lib.g3x_probe(dev.ctx, 2, 0, 0, 0, 6, out)
    print('no resident workgroups              : load %.2f ms' % out[4], flush=True)
for mode in (5, 2, 3, 4):
    for E in (1, 8, 32):
        for lds in (40960, 163000):
            rc = lib.g3x_probe(dev.ctx, mode, E, lds, 3500, 6, out)
            print('%-30s E %2d lds %6d: rc %d load %.2f ms' % (names[mode], E, lds, rc, out[4]), flush=True)
