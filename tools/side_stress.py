"""Repeatability of the two other paths that hand data between workgroups inside a launch: the batched mid-size fits (their
solves are strip solves over all problems) and the map application with teams of workgroups per orbit.  Every call's result is
compared bit for bit with the first one; a hand-off that timed out would be an error of the call.
    python tools/side_stress.py [--reps 200]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import maps
from sympgpr_amd.fit import SympFit, fit_batch

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
rng = np.random.default_rng(11)
for n, B in ((384, 128), (512, 64), (1024, 64), (2048, 16)):
    Np = n // 2
    x, y = rng.uniform(0, 2 * np.pi, (B, Np)), rng.uniform(-3, 3, (B, Np))
    z = rng.standard_normal((B, n))
    l = 2.0 * np.sqrt(12 * np.pi / Np)
    hyp = np.tile([l, l, 1.0], (B, 1))
    s2 = np.full(B, 1e-2 / l**2)
    al0, nll0, info = fit_batch("A", x, y, z, hyp, s2)
    assert not info.any()
    bad, t0 = 0, time.perf_counter()
    reps = a.reps if n <= 1024 else max(a.reps // 4, 1)
    for _ in range(reps):
        al, nll, info = fit_batch("A", x, y, z, hyp, s2)
        bad += int(not (np.array_equal(al, al0) and np.array_equal(nll, nll0)) or info.any())
    print("batched fits, order %d x %d: %d calls in %.1f s, %d differ from the first" % (n, B, reps, time.perf_counter() - t0, bad), flush=True)
    assert bad == 0
for N0, ntest in ((2048, 37), (16384, 37), (1500, 20)):
    q, pn = rng.uniform(0, 2 * np.pi, N0), rng.uniform(-1, 1, N0)
    p_old = pn + 0.3 * np.sin(q); Q = q + 0.3 * pn
    ztrain = np.hstack((p_old - pn, Q - q))
    l = max(0.15, 2.0 * np.sqrt(4 * np.pi / N0))
    hyp, hypp, s2 = np.array([l, l, 1.0]), np.array([l, l, 1.0]), 1e-6
    with SympFit("A", q, pn, ztrain, hyp, s2) as f:
        alpha = f.run().alpha()
    with SympFit("A", q, p_old, pn, hypp, s2, reg=True) as f:
        alphap = f.run().alpha()
    Q0, P0 = rng.uniform(0.5, 5.5, ntest), rng.uniform(-0.5, 0.5, ntest)
    run = lambda: maps.run_map_alpha(maps.WRAP_Q, 21, ntest, hyp, Q0, P0, q, pn, alpha, hypp, q, p_old, alphap, family="A")
    q0, p0 = run()
    bad, t0 = 0, time.perf_counter()
    for _ in range(a.reps):
        qm, pm = run()
        bad += int(not (np.array_equal(qm, q0, equal_nan=True) and np.array_equal(pm, p0, equal_nan=True)))
    print("map, N0 = %d, %d orbits x 20 steps: %d calls in %.1f s, %d differ from the first" % (N0, ntest, a.reps, time.perf_counter() - t0, bad),
          flush=True)
    assert bad == 0
