"""Wall time of one batched call (sgpr_fit_batch) at ORDER,COUNT shapes, with experiment knobs:
    python tools/batch_speed.py [--tune name=value ...] [--reps 9] 1024,64 [2048,64 ...]
Prints the median and minimum call time and the TFLOP/s that n^3/3 flop per fit make of it (what bench.py --batch reports)."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd.fit import fit_batch
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--tune", action="append", default=[], help="name=value experiment knobs (libsympgpr_probe.so), before anything runs")
ap.add_argument("shapes", nargs="+")
a = ap.parse_args()
if a.tune:
    from sympgpr_amd import _lib as L
    for kv in a.tune:
        k, v = kv.split("=")
        L.check(L.load_probe_library().sgpr_probe_tune(k.encode(), float(v)))
rng = np.random.default_rng(3)
for sh in a.shapes:
    n, B = (int(v) for v in sh.split(","))
    Np = n // 2
    x, y = rng.uniform(0, 2 * np.pi, (B, Np)), rng.uniform(-3, 3, (B, Np))
    z = rng.standard_normal((B, n))
    l = 2.0 * np.sqrt(12 * np.pi / Np)
    hyp = np.tile([l, l, 1.0], (B, 1))
    s2 = np.full(B, 1e-2 / l**2)
    al0, nll0, info = fit_batch("A", x, y, z, hyp, s2)
    assert not info.any()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        al, nll, info = fit_batch("A", x, y, z, hyp, s2)
        ts.append(time.perf_counter() - t0)
    assert np.array_equal(nll, nll0) and np.array_equal(al, al0), "results differ from call to call"
    med, mn = float(np.median(ts)), min(ts)
    fl = B * (n**3 / 3.0 + 2.0 * n * n)
    print(f"order {n} x {B} {' '.join(a.tune)}: median {med * 1e3:.3f} ms, min {mn * 1e3:.3f} ms -> {fl / med / 1e12:.2f} TFLOP/s "
          f"({fl / med / 78.6e12 * 100:.1f} % of 78.6)")
