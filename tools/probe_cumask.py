import ctypes as C, sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sympgpr_amd import _lib as L
lib = L.load_library()
def run(words, nb=512):
    m = (C.c_uint * len(words))(*words)
    out = (C.c_int * (2 * nb))()
    L.check(L.load_probe_library().sgpr_probe_cumask(m, len(words), nb, out))
    a = np.array(out[:]).reshape(nb, 2)
    xcc = a[:, 0]; hw = a[:, 1]
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    uniq = sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
    print("mask", [hex(w) for w in words], "-> distinct (xcc,se,sh,cu):", len(uniq), " per xcc:", np.bincount(xcc, minlength=8))
    return uniq
run([0xFFFFFFFF] * 8)
run([0xFFFFFFFF] + [0] * 7)
run([0x0000FFFF] + [0] * 7)
run([0x000000FF] + [0] * 7)
run([0xFFFFFF00] + [0xFFFFFFFF] * 7)
run([0x01010101] * 8)
