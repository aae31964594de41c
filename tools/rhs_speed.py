"""Block-of-right-hand-sides solve (X = L^-T L^-1 B, sgpr_fit_solve_rhs) after one factorisation: device time of the two
triangular solves, against the HBM-read floor (L read once per solve per 64 columns) and the fp64-MFMA floor (2 n^2 nrhs flop).
    python tools/rhs_speed.py [--d D] [--nrhs 64] [--host] N [N ...]        SGPR_TRSM=rec selects round 3's recursion over the GEMM kernel
Default: right-hand sides resident on the device (sgpr_fit_solve_rhs_dev), solves back to back; --host: the host-buffer entry."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd.fit import SympFit
from bench import synth, synth_pairs
ap = argparse.ArgumentParser()
ap.add_argument("--d", type=int, default=1)
ap.add_argument("--nrhs", type=int, default=64)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--host", action="store_true", help="time the host-buffer entry (sgpr_fit_solve_rhs) instead of the device-resident one")
ap.add_argument("--zero", action="store_true", help="experiment: all-zero right-hand sides (what the clock does without operand toggling)")
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
        if a.zero:
            B[:] = 0.0
        ts = []
        if a.host:
            for _ in range(a.reps):
                Xs = f.solve_rhs(B)
                ts.append(f.solve_rhs_ms())
            x0 = Xs[:, 0]
        else:   # right-hand sides resident on the device, solves back to back (one untimed call first: the first heavy launch
                # behind an idle device runs ~13 % slow)
            import torch
            dev = torch.device("cuda", torch.cuda.current_device())
            B0 = torch.from_numpy(np.ascontiguousarray(B.T)).to(dev)
            Bd = torch.empty_like(B0)
            for r in range(a.reps + 1):
                Bd.copy_(B0)
                torch.cuda.synchronize()
                f.solve_rhs_dev(Bd.data_ptr(), a.nrhs)
                if r:
                    ts.append(f.solve_rhs_ms())
            x0 = Bd[0].cpu().numpy()
            del B0, Bd
        al = f.alpha()
        err = np.linalg.norm(x0 - al) / np.linalg.norm(al)
    t = min(ts)
    passes = (a.nrhs + 63) // 64
    gb = 8.0 * n * n * passes / 1e9
    fl = 2.0 * n * n * a.nrhs
    print("n=%d nrhs=%d: %.3f ms (all: %s) | L read %.2f GB -> %.0f GB/s (floor at 8 TB/s %.2f ms) | %.2f TFLOP -> %.1f TFLOP/s (floor at 78.6 "
          "%.2f ms) | column 0 vs alpha %.1e" % (n, a.nrhs, t, " ".join("%.2f" % v for v in ts), gb, gb / t * 1e3, gb / 8.0, fl / 1e12,
                                                   fl / t / 1e9, fl / 78.6e9, err), flush=True)
