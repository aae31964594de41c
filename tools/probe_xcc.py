import ctypes as C, sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sympgpr_amd import _lib as L
lib = L.load_library()
n = 2048
out = (C.c_int * n)()
L.check(L.load_probe_library().sgpr_probe_xcc(n, out))
x = np.array(out[:])
print("first 32:", x[:32])
print("ids seen:", sorted(set(x.tolist())))
print("id %% 8 == const per xcc? ", all(len(set(x[i::8].tolist())) == 1 for i in range(8)))
for i in range(8):
    print(i, np.bincount(x[i::8], minlength=8))
