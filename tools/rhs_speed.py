"""Block-of-right-hand-sides solve (X = L^-T L^-1 B, sgpr_fit_solve_rhs) after one factorisation: device time of the two
triangular solves, against the HBM-read floor (L read once per solve per 64 columns) and the fp64-MFMA floor (2 n^2 nrhs flop).
    python tools/rhs_speed.py [--d D] [--nrhs 64] N [N ...]        SGPR_TRSM=rec selects round 3's recursion over the GEMM kernel"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd.fit import SympFit
from bench import synth, synth_pairs
ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=1)
ap.add_argument("--nrhs", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("npts", type=int, nargs="+")
ap.add_argument("--tune", action="append", default=[], help="name=value experiment knobs (libsympgpr_probe.so), before anything runs")
a = ap.parse_args()
if a.tune:   # TUNE
    from sympgpr_amd import _lib as L
    for kv in a.tune:
        k, v = kv.split("=")
        L.check(L.load_probe_library().sgpr_probe_tune(k.encode(), float(v)))
for N in a.npts:
    if a.d == 1:
        q, P, z, hyp, s2 = synth(N)
        f = SympFit("A", q, P, z, hyp, s2)
    else:
        X, z, hyp, s2 = synth_pairs(N, a.d)
        f = SympFit.pairs("A", X, z, hyp, s2)
    with f:
        f.run()
        n = f.n
        rng = np.random.default_rng(5)
        B = rng.standard_normal((n, a.nrhs))
        B[:, 0] = z
        ts = []
        for _ in range(a.reps):
            Xs = f.solve_rhs(B)
            ts.append(f.solve_rhs_ms())
        al = f.alpha()
        err = np.linalg.norm(Xs[:, 0] - al) / np.linalg.norm(al)
    t = min(ts)
    passes = (a.nrhs + 63) // 64
    gb = 8.0 * n * n * passes / 1e9
    fl = 2.0 * n * n * a.nrhs
    print("n=%d nrhs=%d: %.3f ms (all: %s) | L read %.2f GB -> %.0f GB/s (floor at 8 TB/s %.2f ms) | %.2f TFLOP -> %.1f TFLOP/s (floor at 78.6 "
          "%.2f ms) | column 0 vs alpha %.1e" % (n, a.nrhs, t, " ".join("%.2f" % v for v in ts), gb, gb / t * 1e3, gb / 8.0, fl / 1e12,
                                                   fl / t / 1e9, fl / 78.6e9, err), flush=True)
