"""Round-4 feasibility probe (measurement build libg3hip_probe.so, scripts/build_variant.sh probe g3_potrf.hip -DG3_PROBE):
(a) potrf256 on a workgroup that owns its CU vs one that shares it, with and without a bulk update streaming beside it;
(b) flag round trip stream -> resident kernel -> stream."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('G3_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'g3py_amd', 'lib', 'libg3hip_probe.so'))
import numpy as np
import g3py_amd as g3
dev = g3.Device(0)
lib = C.CDLL(os.environ['G3_LIB_PATH'])
lib.g3x_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
out = (C.c_double * 8)()
def run(mode, E, lds, reps, load):
    for i in range(8): out[i] = 0
    rc = lib.g3x_probe(dev.ctx, mode, E, lds, reps, load, out)
    return rc, list(out)
for (E, lds, load) in [(1, 0, 0), (1, 160 * 1024, 0), (1, 0, 6), (1, 160 * 1024, 6), (4, 160 * 1024, 6), (8, 160 * 1024, 6), (16, 160 * 1024, 6), (4, 100 * 1024, 6)]:
    rc, o = run(0, E, lds, 150, load)
    print('potrf256 loop  E %2d lds %6d load %d: rc %d  median %.1f us  min %.1f  max %.1f  mean %.1f | load %.2f ms, loop span %.0f us, info %d'
          % (E, lds, load, rc, o[0], o[1], o[2], o[3], o[4], o[5], int(o[6])), flush=True)
for (lds, load) in [(0, 0), (160 * 1024, 0), (160 * 1024, 6), (0, 6)]:
    rc, o = run(1, 1, lds, 200, load)
    print('flag round trip lds %6d load %d: rc %d  %.2f us per post+echo+wait  err %d | load %.2f ms' % (lds, load, rc, o[0], int(o[1]), o[4]), flush=True)
