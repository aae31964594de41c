"""The task-queue Cholesky against SciPy through the host entry point: python tools/queue_check.py n [n ...]"""
import sys, os, time
import numpy as np
import scipy.linalg
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import ops, _lib as L

# name=value arguments set experiment knobs of the library through libsympgpr_probe.so before anything runs (q_tail=6144 ...)
for kv in [a for a in sys.argv[1:] if "=" in a]:
    k, v = kv.split("=")
    L.check(L.load_probe_library().sgpr_probe_tune(k.encode(), float(v)))
sys.argv = [a for a in sys.argv if "=" not in a]

rng = np.random.default_rng(7)
for n in [int(a) for a in sys.argv[1:]]:
    B = rng.standard_normal((n, n + 3))
    A = B @ B.T / n + 0.5 * np.eye(n)
    t0 = time.perf_counter()
    Lg = ops.cholesky(A)
    dt = time.perf_counter() - t0
    Lr = scipy.linalg.cholesky(A, lower=True)
    err = np.abs(Lg - Lr).max() / np.abs(Lr).max()
    print("n=%d  max|L - L_scipy| / max|L| = %.2e  (%.2f s incl. transfers)  %s" % (n, err, dt, "ok" if err < 1e-11 else "FAIL"), flush=True)
    assert err < 1e-11
