import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sympgpr_amd import _lib as L
lib = L.load_library()
o = np.zeros(8)
L.check(L.load_probe_library().sgpr_probe_leaf(L.dptr(o)))
names = ["load", "diag(A)", "panel(B)", "update(C)", "writeback", "inv diag", "inv rows", "tail store"]
tot = o.sum()
for n, v in zip(names, o):
    print("%-12s %9.0f cyc  %5.1f us  %4.1f%%" % (n, v, v / 2.39e3, 100 * v / tot))
print("total %.1f us" % (tot / 2.39e3))
