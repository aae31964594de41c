"""Per-launch breakdown of the MFMA GEMM kernel inside one factorisation: python tools_launches.py N"""
import sys
import numpy as np
sys.path.insert(0, ".")
from sympgpr_amd import _lib as L
from sympgpr_amd.fit import SympFit
from bench import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
lib = L.load_library()
q, P, z, hyp, s2 = synth(N)
f = SympFit("A", q, P, z, hyp, s2, lower_only=False)
f.run()
f.build()
L.check(lib.sgpr_profile_begin())
f.factor()
o = np.zeros(8)
L.check(lib.sgpr_profile_end(L.dptr(o)))
n = lib.sgpr_profile_launches(None, 0)
buf = np.zeros(6 * n)
lib.sgpr_profile_launches(L.dptr(buf), n)
r = buf.reshape(n, 6)
print("factor ms", f.stage_ms()[1], "launches", n, "sum gemm ms", r[:, 5].sum())
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0, 0.0])
for m, nn, k, lower, big, ms in r:
    key = (int(k), int(lower), int(big), "m<=1k" if m <= 1024 else ("m<=4k" if m <= 4096 else "m>4k"))
    flop = k * (m * (m + 1) if lower else 2 * m * nn)
    a = agg[key]; a[0] += 1; a[1] += ms; a[2] += flop
for key in sorted(agg, key=lambda k: -agg[k][1]):
    c, ms, fl = agg[key]
    print("k=%5d lower=%d big=%d %-6s: %4d launches %8.2f ms (%5.1f us each) %6.2f TFLOP/s" % (*key, c, ms, 1e3 * ms / c, fl / ms / 1e9))
