"""Factor-stage time of the recursive vs the blocked look-ahead driver: python tools/potrf_modes.py N [N...]
(run once per mode: SGPR_POTRF=rec|la, SGPR_POTRF_NB=...)"""
import os, sys
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from sympgpr_amd.fit import SympFit
from bench import synth
# name=value arguments set experiment knobs of the library through libsympgpr_probe.so before anything runs (q_hoist=0 ...)
for kv in [a for a in sys.argv[1:] if "=" in a]:
    from sympgpr_amd import _lib as L
    L.check(L.load_probe_library().sgpr_probe_tune(kv.split("=")[0].encode(), float(kv.split("=")[1])))
TAG = " ".join(a for a in sys.argv[1:] if "=" in a)
sys.argv = [a for a in sys.argv if "=" not in a]
for N in [int(a) for a in sys.argv[1:]]:
    q, P, z, hyp, s2 = synth(N)
    with SympFit("A", q, P, z, hyp, s2, lower_only=False) as f:
        try:
            f.run()
        except Exception:
            from sympgpr_amd import _lib as L
            L.load_probe_library().sgpr_probe_queue_postmortem(1)
            raise
        ts = []
        for _ in range(3):
            f.build(); f.factor(); ts.append(f.stage_ms()[1])
        f.solve(); a = f.alpha()
        op, oq = f.predict_rows(q[:256], P[:256])
        r = np.concatenate([op + s2 * a[:256] - z[:256], oq + s2 * a[N:N + 256] - z[N:N + 256]])
    n = 2 * N
    print("mode=%s nb=%s %s n=%d: factor %.2f ms = %.2f TFLOP/s  resid %.1e" % (os.environ.get("SGPR_POTRF"), os.environ.get("SGPR_POTRF_NB"), TAG,
          n, min(ts), n**3 / 3 / min(ts) / 1e9, np.linalg.norm(r) / np.linalg.norm(z[:512])))
