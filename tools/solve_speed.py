"""Solve-stage time (alpha = L^-T L^-1 z + nll) after one factorisation: python tools/solve_speed.py N [N...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd.fit import SympFit
from bench import synth
for N in [int(a) for a in sys.argv[1:]]:
    q, P, z, hyp, s2 = synth(N)
    with SympFit("A", q, P, z, hyp, s2) as f:
        f.run()
        ts = []
        for _ in range(4):
            f.solve(); ts.append(f.stage_ms()[2])
        a = f.alpha()
        op, oq = f.predict_rows(q[:256], P[:256])
        r = np.concatenate([op + s2 * a[:256] - z[:256], oq + s2 * a[N:N + 256] - z[N:N + 256]])
    n = 2 * N
    print("n=%d: solve %.3f ms (L read twice = %.2f GB -> %.0f GB/s)  resid %.1e" % (n, min(ts), 8.0 * n * n / 1e9, 8.0 * n * n / min(ts) / 1e6, np.linalg.norm(r) / np.linalg.norm(z[:512])))
