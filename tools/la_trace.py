"""One factorisation at n = 2N for a kernel timeline: rocprofv3 --kernel-trace -- python3 tools/la_trace.py [name=value ...] N
(name=value: experiment knobs of the library, set through libsympgpr_probe.so before anything runs)"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L
for kv in [a for a in sys.argv[1:] if "=" in a]:
    L.check(L.load_probe_library().sgpr_probe_tune(kv.split("=")[0].encode(), float(kv.split("=")[1])))
from sympgpr_amd.fit import SympFit
from bench import synth
N = int([a for a in sys.argv[1:] if "=" not in a][0])
q, P, z, hyp, s2 = synth(N)
with SympFit("A", q, P, z, hyp, s2, lower_only=False) as f:
    f.run()                    # warm-up (module load, side stream)
    f.build(); f.factor()
    print("factor ms", f.stage_ms()[1])
