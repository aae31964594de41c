"""Per-launch breakdown of the MFMA GEMM kernel inside one factorisation:
python tools/gemm_launches.py N [--json out.json]"""
import json
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L
from sympgpr_amd.fit import SympFit
from bench import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
lib = L.load_library()
q, P, z, hyp, s2 = synth(N)
f = SympFit("A", q, P, z, hyp, s2, lower_only=False)
f.run()
f.build()
L.check(lib.sgpr_profile_begin())
f.factor()
o = np.zeros(12)
L.check(lib.sgpr_profile_end(L.dptr(o)))
n = lib.sgpr_profile_launches(None, 0)
buf = np.zeros(7 * n)
lib.sgpr_profile_launches(L.dptr(buf), n)
r = buf.reshape(n, 7)
print("factor ms", f.stage_ms()[1], "launches", n, "sum gemm ms", r[:, 5].sum())
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0, 0.0])
for m, nn, k, lower, big, ms, ovl in r:
    key = (int(k), int(lower), int(big), int(ovl), "m<=1k" if m <= 1024 else ("m<=4k" if m <= 4096 else "m>4k"))
    nl = min(m, nn)
    flop = 2 * k * ((nl * m - nl * (nl - 1) / 2) if lower else m * nn)
    a = agg[key]; a[0] += 1; a[1] += ms; a[2] += flop
for key in sorted(agg, key=lambda k: -agg[k][1]):
    c, ms, fl = agg[key]
    print("k=%5d lower=%d big=%d beside-panel-stream=%d %-6s: %4d launches %8.2f ms (%5.1f us each) %6.2f TFLOP/s" % (*key, c, ms, 1e3 * ms / c, fl / ms / 1e9))

if json_out:
    big = r[r[:, 4] == 1]
    nn_ = np.minimum(big[:, 1], big[:, 0])          # lower: columns j < min(m, n) hold m - j entries
    elems = np.where(big[:, 3] == 1, nn_ * big[:, 0] - nn_ * (nn_ - 1) / 2, big[:, 0] * big[:, 1])
    comp = 8.0 * (big[:, 0] * big[:, 2] + big[:, 1] * big[:, 2] + 2.0 * elems)
    json.dump({"n_pts": N, "big_launches": int(len(big)), "big_ms": float(big[:, 5].sum()),
               "big_flop": float((2.0 * big[:, 2] * elems).sum()), "big_compulsory_bytes": float(comp.sum())},
              open(json_out, "w"))
