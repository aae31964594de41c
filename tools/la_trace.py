"""One factorisation at n = 2N for a kernel timeline: rocprofv3 --kernel-trace -- python3 tools/la_trace.py N"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd.fit import SympFit
from bench import synth
N = int(sys.argv[1])
q, P, z, hyp, s2 = synth(N)
with SympFit("A", q, P, z, hyp, s2, lower_only=False) as f:
    f.run()                    # warm-up (module load, side stream)
    f.build(); f.factor()
    print("factor ms", f.stage_ms()[1])
