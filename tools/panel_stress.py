"""Soak of the look-ahead driver's persistent panel kernel (helper workgroups claimed at run time, rows below drawn tile by tile,
early-hand-off solves): many factorisations back to back at orders below the queue's window, every factor compared BIT FOR BIT with
the first one of its order (the products a tile takes are applied in the same order whoever applies them, so a race or a lost
hand-off shows as a differing bit or as a give-up) and checked by a residual:  python tools/panel_stress.py [N reps ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd.fit import SympFit
from bench import synth
args = [int(a) for a in sys.argv[1:]] or [512, 600, 1024, 600, 2048, 500, 3072, 300, 4096, 300, 6144, 150]
bad = 0
for N, reps in zip(args[0::2], args[1::2]):
    q, P, z, hyp, s2 = synth(N)
    with SympFit("A", q, P, z, hyp, s2, lower_only=False) as f:
        f.build(); f.factor()
        d0 = f.ldiag().copy()
        f.solve(); a0 = f.alpha().copy()
        t0, ts, ndiff = time.time(), [], 0
        for r in range(reps):
            f.build(); f.factor()
            ts.append(f.stage_ms()[1])
            if r % 10 == 0:                       # (the downloads cost more than the factorisation: every tenth)
                d = f.ldiag()
                f.solve()
                a = f.alpha()
                if not (np.array_equal(d, d0) and np.array_equal(a, a0)):
                    ndiff += 1
        f.solve()
        op, oq = f.predict_rows(q[:256], P[:256])
        res = np.linalg.norm(np.concatenate([op + s2 * a0[:256] - z[:256], oq + s2 * a0[N:N + 256] - z[N:N + 256]])) / np.linalg.norm(z[:512])
        bad += ndiff
        print("n=%5d: %4d factorisations in %5.1f s, none gave up; %d of %d sampled results differ from the first bit for bit; factor ms median %.3f "
              "min %.3f max %.3f; residual %.1e" % (2 * N, reps, time.time() - t0, ndiff, (reps + 9) // 10, np.median(ts), min(ts), max(ts), res), flush=True)
sys.exit(1 if bad else 0)
