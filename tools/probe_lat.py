"""Cycles per step of the building blocks of the leaf's 16 x 16 diagonal-block factorisation (one wave, dependent steps)."""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sympgpr_amd import _lib as L
L.load_library()
o = np.zeros(12)
L.check(L.load_probe_library().sgpr_probe_lat(L.dptr(o)))
names = ["1 dependent MFMA f64 16x16x4 (acc -> mul -> operand)", "2 MFMAs (2nd independent)", "pivot chain (2 readlane pairs, fma, rsq, 2 Newton)",
         "v_rsq_f64 dependent", "8 dependent v_fma_f64", "readlane pair -> VALU", "LDS write -> read (same wave)", "MFMA -> readlane -> operand",
         "leaf loop: D + identity tile + chain + LDS writes", "  without the identity tile", "  without the LDS writes", "  without either"]
for n, v in zip(names, o):
    print("%-55s %7.1f cycles / step" % (n, v))
