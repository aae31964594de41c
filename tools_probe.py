import ctypes as C, sys
import numpy as np
sys.path.insert(0, '.')
from sympgpr_amd import _lib as L
lib = L.load_library()
o = np.zeros(4)
for rep in range(2):
  for bits in (0,):
    lib.sgpr_probe_gemm_debug(bits)
    for (m, n, k, lower) in [(8192, 8192, 8192, 0), (32768, 32768, 8192, 1)]:
        L.check(lib.sgpr_probe_gemm(m, n, k, lower, L.dptr(o)))
        print("dbg=%d m=%d n=%d k=%d lower=%d: %.2f TFLOP/s; %.1f cyc/k-step; clock %.3f GHz" % (bits, m, n, k, lower, o[0], o[1] / o[3], o[2]))
lib.sgpr_probe_gemm_debug(0)
