import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
from sympgpr_amd import _lib as L
lib = L.load_library()
o = np.zeros(4)
for bits in (0, 8, 16, 24, 2, 10):
    lib.sgpr_probe_gemm_debug(bits)
    m, n, k, lower = 8192, 8192, 8192, 0
    L.check(lib.sgpr_probe_gemm(m, n, k, lower, L.dptr(o)))
    print("dbg=%d: %.2f TFLOP/s; %.1f cyc/k-step (ideal 8192); clock %.3f GHz" % (bits, o[0], o[1] / o[3], o[2]))
lib.sgpr_probe_gemm_debug(0)
