import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sympgpr_amd import _lib as L
lib = L.load_library()
o = np.zeros(4)
for (m, n, k, lower) in [(16384, 16384, 8192, 0), (32768, 32768, 8192, 1)]:
    L.check(lib.sgpr_probe_gemm(m, n, k, lower, L.dptr(o)))
    print("pad=%s m=%d n=%d k=%d lower=%d: %.2f TFLOP/s; %.1f cyc/k-step" % (os.environ.get("SGPR_PROBE_PAD"), m, n, k, lower, o[0], o[1] / o[3]))
