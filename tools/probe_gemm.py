"""One synthetic C -= A B^T per (m n k lower) quadruple on the command line: python tools/probe_gemm.py 15360 15360 512 1 ..."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sympgpr_amd import _lib as L
lib = L.load_library()
o = np.zeros(4)
a = [int(v) for v in sys.argv[1:]]
cases = [tuple(a[i:i + 4]) for i in range(0, len(a) - 3, 4)] or [(16384, 16384, 8192, 0), (32768, 32768, 8192, 1)]
L.load_probe_library().sgpr_probe_gemm_debug(int(os.environ.get("SGPR_PROBE_DBG", "0")))   # 8 = 128x128 tiles, 16 = register-staged
for (m, n, k, lower) in cases:
    L.check(L.load_probe_library().sgpr_probe_gemm(m, n, k, lower, L.dptr(o)))
    print("pad=%s m=%d n=%d k=%d lower=%d: %.2f TFLOP/s; %.1f cyc/k-step" % (os.environ.get("SGPR_PROBE_PAD"), m, n, k, lower, o[0], o[1] / o[3]))
