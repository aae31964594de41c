"""Fixed cost per tile of the 256x128 MFMA kernel: C -= A B^T at 16384 x 16384 (8192 tiles = 32 full waves
of 256 workgroups) for a range of k; time per tile = a + b k.  python tools/probe_gemm_k.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L
lib = L.load_library()
o = np.zeros(4)
ks, ts = [], []
for k in (128, 256, 512, 1024, 2048, 4096, 8192):
    L.check(L.load_probe_library().sgpr_probe_gemm(16384, 16384, k, 0, L.dptr(o)))
    t_tile = 2.0 * 256 * 128 * k / (o[0] * 1e12 / 256) * 1e6      # us per tile (one tile per CU at a time)
    loop_us = o[1] / (o[2] * 1e3) if o[2] > 0 else float("nan")
    ks.append(k); ts.append(t_tile)
    print("k=%5d: %6.2f TFLOP/s  %8.1f us/tile  (k-loop %8.1f us at %.2f GHz -> outside the loop %5.1f us)" % (k, o[0], t_tile, loop_us, o[2], t_tile - loop_us))
b, a = np.polyfit(ks, ts, 1)
print("fit: %.1f us + %.4f us * k  (%.2f us per k-step of 16)" % (a, b, 16 * b))
