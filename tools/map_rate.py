"""Map application on the device (sgpr_applymap_host: every time step of every orbit in one launch) beside the reference's own
calcP / calcq (python/05_tokamak/SympGPR/sympgpr.f90:75-125, compiled: oracle/_ref): steps x orbits per second and K*-row pair
evaluations per second at N0 = 80 (the drivers' size), 2048 and 16384 training points, Ntest = 37 orbits.
    python tools/map_rate.py [--ntest 37] [--steps 50] [N0 ...]       -> a markdown table on stdout"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import _lib as L, maps
from sympgpr_amd.fit import SympFit

ap = argparse.ArgumentParser()
ap.add_argument("--ntest", type=int, default=37)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--tune", action="append", default=[], help="name=value experiment knobs (libsympgpr_probe.so)")
ap.add_argument("n0", type=int, nargs="*", default=[80, 2048, 16384])
a = ap.parse_args()
probe = L.load_probe_library()
for kv in a.tune:
    L.check(probe.sgpr_probe_tune(kv.split('=')[0].encode(), float(kv.split('=')[1])))
try:
    from oracle.oracle import Ref
    ref = Ref() if Ref.available() else None
except Exception:
    ref = None
print("| N0 | workgroups per orbit | device: %d steps x %d orbits (+ fixed cost of a call) | steps x orbits / s | K*-row evaluations | G pair evaluations / s | "
      "reference calcP + calcq (compiled Fortran, 1 thread) | device / reference |" % (a.steps, a.ntest))
print("|---|---|---|---|---|---|---|---|")
for N0 in a.n0:
    rng = np.random.default_rng(3)
    q, pn = rng.uniform(0, 2 * np.pi, N0), rng.uniform(-1, 1, N0)
    p_old = pn + 0.3 * np.sin(q); Q = q + 0.3 * pn
    ztrain = np.hstack((p_old - pn, Q - q))
    l = max(0.15, 2.0 * np.sqrt(4 * np.pi / N0))
    hyp, hypp, s2 = np.array([l, l, 1.0]), np.array([l, l, 1.0]), 1e-6
    with SympFit("A", q, pn, ztrain, hyp, s2) as f:
        alpha = f.run().alpha()
    with SympFit("A", q, p_old, pn, hypp, s2, reg=True) as f:
        alphap = f.run().alpha()
    Q0, P0 = rng.uniform(0.5, 5.5, a.ntest), rng.uniform(-0.5, 0.5, a.ntest)
    # two lengths: the difference is device time per step (uploads, allocations and the copy back are the same for both)
    def run(steps):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            qm, pm = maps.run_map_alpha(maps.WRAP_Q, steps + 1, a.ntest, hyp, Q0, P0, q, pn, alpha, hypp, q, p_old, alphap, family="A")
            ts.append(time.perf_counter() - t0)
        return min(ts), pm, probe.sgpr_probe_map_calls()
    run(2)
    t_short, _, calls_short = run(a.steps)
    t_long, pm, calls_long = run(5 * a.steps)
    t = (t_long - t_short) / 4.0                          # seconds per a.steps steps on the device
    fixed = t_short - t
    calls = (calls_long - calls_short) // 4
    lost = int(np.isnan(pm[-1]).sum())
    pairs = calls * N0 + a.steps * a.ntest * N0            # K*-rows on N0 pairs each + one first-guess row per step
    refcol, ratio = "n/a (Kyinv = %.1f GB)" % (8.0 * (2 * N0) ** 2 / 1e9), ""
    if ref is not None and N0 <= 2048:
        # the reference needs the explicit inverses (it multiplies Kyinv with ztrain inside every residual)
        from oracle.oracle import Oracle
        orc = Oracle()
        Kyinv = np.asfortranarray(np.linalg.inv(orc.build_K("A", q, pn, q, pn, hyp, threads=8) + s2 * np.eye(2 * N0)))
        Kyinvp = np.asfortranarray(np.linalg.inv(orc.buildKreg("A", q, p_old, q, p_old, hypp, threads=8) + s2 * np.eye(N0)))
        nref = 8 if N0 > 200 else 200
        t0 = time.perf_counter()
        for k in range(nref):
            qq, pp = Q0[k % a.ntest], P0[k % a.ntest]
            Pn = ref.calcP("A", qq, pp, hyp, hypp, q, p_old, pn, Kyinvp, q, pn, ztrain, Kyinv)
            ref.calcQ("A", qq, Pn, q, pn, hyp, Kyinv, ztrain)
        tr = (time.perf_counter() - t0) / nref
        refcol = "%.3g s per step and orbit (%d timed)" % (tr, nref)
        ratio = "%.0fx" % (tr * a.steps * a.ntest / t)
    print("| %d | %d | %.2f ms (+ %.2f ms)%s | %.3g | %d (%.1f per step and orbit) | %.1f | %s | %s |"
          % (N0, probe.sgpr_probe_map_team(a.ntest, N0), t * 1e3, fixed * 1e3, " (%d orbits lost)" % lost if lost else "", a.steps * a.ntest / t,
             calls, calls / (a.steps * a.ntest), pairs / t / 1e9, refcol, ratio), flush=True)
