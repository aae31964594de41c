"""Fits per second of the batched small-fit launch (sgpr_fit_batch) at the drivers' own sizes, beside (a) one
nll_chol at a time as the unmodified drivers call it (a batch of one: one launch), (a') one device-resident handle
per call (round 1's path) and (b) the reference's CPU path for the same call
(its compiled Fortran build_K from oracle/_ref + the SciPy cholesky / solve_triangular of
python/functions/func.py:189-196, one thread, as the reference runs it).
Orders above 256 take the mid-size path (three launches per batch: build, W workgroups per problem running the leaf chain
side by side, one workgroup per problem for the solves); their batch sizes are a CMA-ES generation's (4, 16, 64).
python tools/batch_rate.py [orders...]   (default 80 160 512 1024 2048)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sympgpr_amd import func
from sympgpr_amd.fit import SympFit, fit_batch


def cpu_nll(ref, x, y, z, hyp, s2):
    import scipy.linalg
    K = ref.build_K("A", x, y, x, y, hyp)
    K[np.diag_indices(len(z))] += abs(s2)
    Lf = scipy.linalg.cholesky(K, lower=True, check_finite=False)
    a = scipy.linalg.solve_triangular(Lf.T, scipy.linalg.solve_triangular(Lf, z, lower=True, check_finite=False),
                                      lower=False, check_finite=False)
    return 0.5 * z @ a + np.sum(np.log(Lf.diagonal()))


def main():
    orders = [int(a) for a in sys.argv[1:]] or [80, 160, 512, 1024, 2048]
    try:
        from threadpoolctl import threadpool_limits
        limit = threadpool_limits(limits=1)
    except Exception:
        limit = None
    from oracle.oracle import Oracle, Ref
    ref = Ref() if Ref.available() else None
    rng = np.random.default_rng(3)
    func.set_family("A")
    print("| order n | batch size | batched launch | one call at a time (func.nll_chol) | one handle per call (SympFit) | reference CPU path (1 thread) |")
    print("|---|---|---|---|---|---|")
    for n in orders:
        Np = n // 2
        small = n <= 256
        for B in ((4, 64, 1024) if small else (4, 16, 64)):
            x, y = rng.uniform(0, 2 * np.pi, (B, Np)), rng.uniform(-3, 3, (B, Np))
            z = rng.standard_normal((B, n))
            l = 2.0 * np.sqrt(12 * np.pi / Np)
            hyp = np.tile([l, l, 1.0], (B, 1))
            s2 = np.full(B, 1e-2 / l**2)
            fit_batch("A", x, y, z, hyp, s2)
            reps = max(1, 2000 // B) if small else 7
            tbs = []
            for _ in range(1 if small else reps):
                t0 = time.perf_counter()
                for _ in range(reps if small else 1):
                    _, nll, info = fit_batch("A", x, y, z, hyp, s2, want_alpha=False)
                tbs.append((time.perf_counter() - t0) / ((reps if small else 1) * B))
            tb = float(np.median(tbs))          # (mid-size path: the median of 7 calls; a call is 1 - 15 ms)
            m = min(B, 64 if small else 8)
            t0 = time.perf_counter()
            one = [func.nll_chol(np.append(hyp[b], s2[b]), np.hstack((x[b], y[b])), z[b], n) for b in range(m)]
            t1c = (time.perf_counter() - t0) / m
            assert np.allclose(one, nll[:m], rtol=1e-10)
            t0 = time.perf_counter()
            for b in range(m):
                with SympFit("A", x[b], y[b], z[b], hyp[b], s2[b]) as f:
                    f.run().nll()
            th = (time.perf_counter() - t0) / m
            if ref is not None:
                t0 = time.perf_counter()
                cpu = [cpu_nll(ref, x[b], y[b], z[b], hyp[b], s2[b]) for b in range(m)]
                tc = (time.perf_counter() - t0) / m
                assert np.allclose(cpu, nll[:m], rtol=1e-9)
                cpu_s = "%.0f fits/s (%.0f us)" % (1 / tc, tc * 1e6)
            else:
                cpu_s = "n/a"
            print("| %d | %d | %.0f fits/s (%.1f us) | %.0f fits/s (%.0f us) | %.0f fits/s (%.0f us) | %s |" % (
                n, B, 1 / tb, tb * 1e6, 1 / t1c, t1c * 1e6, 1 / th, th * 1e6, cpu_s))


if __name__ == "__main__":
    main()
