"""Cost of one nll_chol call (the optimiser objective) split into handle set-up and the fit itself:
python tools/nll_call_cost.py N [N...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import func
from sympgpr_amd.fit import SympFit
from bench import synth
for N in [int(a) for a in sys.argv[1:]]:
    q, P, z, hyp, s2 = synth(N)
    x = np.hstack((q, P)); hyp4 = np.hstack((hyp, [s2]))
    func.nll_chol(hyp4, x, z, 2 * N)
    t = []
    for _ in range(4):
        t0 = time.perf_counter(); v = func.nll_chol(hyp4, x, z, 2 * N); t.append(time.perf_counter() - t0)
    with SympFit("A", q, P, z, hyp, s2) as f:
        f.run()
        r = []
        for _ in range(4):
            t0 = time.perf_counter(); f.set_hyp(hyp, s2); f.run(); w = f.nll(); r.append(time.perf_counter() - t0)
    print("n=%d: nll_chol call %.2f ms; resident handle set_hyp+run+nll %.2f ms" % (2 * N, 1e3 * min(t), 1e3 * min(r)))
