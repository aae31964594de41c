#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json from the REFERENCE itself (run in the build container only).

Sources of truth used here:
  * oracle/_ref/*.so  -- the reference's own Fortran (sympgpr.f90, kernels*.f90) compiled from
    /root/reference by `make -C oracle ref`; called through oracle.oracle.Ref.
  * scipy.linalg.cholesky / solve_triangular -- the very calls the reference makes at
    python/functions/func.py:165-196 for the factor/solve (third-party LAPACK, not vendored).
  * a higher-precision alpha (longdouble iterative refinement) for the tolerance budget.

The fixtures hold inputs and expected outputs only (data, no reference source text).
Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np
import scipy
import scipy.linalg

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.oracle import Ref  # noqa: E402

HYP = {"A": [0.5, 2.0, 0.4], "B": [0.5, 2.0, 0.4], "C": [0.5, 2.0, 0.4], "D": [0.5, 2.0, 0.7, 0.4]}


def ref_solve(Ky, z):
    """functions/func.py:165-177 verbatim semantics."""
    L = scipy.linalg.cholesky(Ky, lower=True)
    a = scipy.linalg.solve_triangular(
        L.T, scipy.linalg.solve_triangular(L, z, lower=True, check_finite=False),
        lower=False, check_finite=False)
    return L, a


def refine_hp(Ky, z, L, a0, iters=6):
    """alpha to ~longdouble accuracy: residual in longdouble, correction with the fp64 factor."""
    Kl = Ky.astype(np.longdouble)
    zl = z.astype(np.longdouble)
    a = a0.astype(np.longdouble)
    for _ in range(iters):
        r = (zl - Kl @ a).astype(np.float64)
        d = scipy.linalg.solve_triangular(
            L.T, scipy.linalg.solve_triangular(L, r, lower=True), lower=False)
        a = a + d.astype(np.longdouble)
    return a.astype(np.float64)


def main():
    ref = Ref()
    meta = {"numpy": np.__version__, "scipy": scipy.__version__,
            "source": "oracle/_ref (amdflang build of /root/reference Fortran) + SciPy LAPACK"}

    # ---- 1. the reference's own test inputs (05_tokamak/SympGPR/test_sympgpr.py:7-10,19-75)
    x = np.array([1.0, 2.0, 3.0]); y = np.array([0.0, 3.0, 2.0])
    x0 = np.array([1.0, 2.0]); y0 = np.array([0.0, 3.0])
    hyp = np.array([0.5, 2.0, 0.4]); hypp = np.array([0.6, 1.9, 0.3])
    Kyinvp = np.array([[0.9, -0.3], [0.3, 0.9]], order="F")
    ztrainp = np.cos(x0 + y0)
    Kyinv = np.reshape(np.arange(16.0), (4, 4), order="F")
    ztrain = np.hstack((np.cos(x0 + y0), np.sin(x0 + y0)))
    K66 = ref.build_K("A", x, y, x, y, hyp)
    Ky = K66 + 1e-3 * np.eye(6)
    z6 = np.cos(np.arange(6.0))
    L6, a6 = ref_solve(Ky, z6)
    ka = {
        "meta": meta,
        "x": x.tolist(), "y": y.tolist(), "x0": x0.tolist(), "y0": y0.tolist(),
        "hyp": hyp.tolist(), "hypp": hypp.tolist(),
        "build_K_6x4": ref.build_K("A", x, y, x0, y0, hyp).tolist(),
        "buildKreg_3x2": ref.buildKreg("A", x, y, x0, y0, hyp).tolist(),
        "buildKreg_1x2": ref.buildKreg("A", x[:1], y[:1], x0, y0, hyp).tolist(),
        "Kyinvp": Kyinvp.tolist(), "ztrainp": ztrainp.tolist(),
        "Kyinv": Kyinv.tolist(), "ztrain": ztrain.tolist(),
        "guessP": ref.guessP("A", x[0], y[0], hypp, x0, y0, ztrainp, Kyinvp),
        "calcQ": ref.calcQ("A", x[0], y[0], x0, y0, hyp, Kyinv, ztrain),
        "calcP": ref.calcP("A", x[0], y[0], hyp, hypp, x0, y0, ztrainp, Kyinvp, x0, y0, ztrain, Kyinv),
        "fit6": {"sig2n": 1e-3, "z": z6.tolist(), "alpha": a6.tolist(),
                 "nll": float(0.5 * z6 @ a6 + np.sum(np.log(L6.diagonal()))),
                 "cond": float(np.linalg.cond(Ky))},
    }
    with open(os.path.join(HERE, "known_answer.json"), "w") as f:
        json.dump(ka, f, indent=1)

    # ---- 2. Gram matrices, all four families, square + ragged shapes, seeded
    out = {}
    for fam in "ABCD":
        rng = np.random.default_rng(1234 + ord(fam))
        for tag, (n, n0) in {"sq8": (8, 8), "sq64": (64, 64), "rect5x7": (5, 7), "row1x9": (1, 9),
                             "col9x1": (9, 1)}.items():
            xx = rng.uniform(0, 2 * np.pi, n); yy = rng.uniform(-3, 3, n)
            if tag.startswith("sq"):
                xx0, yy0 = xx, yy
            else:
                xx0 = rng.uniform(0, 2 * np.pi, n0); yy0 = rng.uniform(-3, 3, n0)
            h = np.array(HYP[fam])
            if tag == "sq64":
                h = h.copy(); h[0] = 0.31; h[1] = 0.77
            out[f"{fam}_{tag}_x"] = xx; out[f"{fam}_{tag}_y"] = yy
            out[f"{fam}_{tag}_x0"] = xx0; out[f"{fam}_{tag}_y0"] = yy0
            out[f"{fam}_{tag}_hyp"] = h
            out[f"{fam}_{tag}_K"] = ref.build_K(fam, xx, yy, xx0, yy0, h)
            out[f"{fam}_{tag}_Kreg"] = ref.buildKreg(fam, xx, yy, xx0, yy0, h)
    np.savez_compressed(os.path.join(HERE, "gram.npz"), **out)

    # ---- 3. fits: synthetic inputs of SURVEY 8(d) (seed 1234, q~U(0,2pi), P~U(-3,3),
    #         l = 2 sqrt(12 pi / N), sig = 1, sig2n = 1e-2/l^2), families A and C; K not stored.
    fits = {}
    for fam in "AC":
        for N in (32, 128, 512):
            rng = np.random.default_rng(1234)
            q = rng.uniform(0, 2 * np.pi, N); P = rng.uniform(-3, 3, N)
            z = rng.standard_normal(2 * N)
            l = 2.0 * np.sqrt(12 * np.pi / N)
            h = np.array([l, l, 1.0]); s2 = 1e-2 / l**2
            K = ref.build_K(fam, q, P, q, P, h)
            Kyy = K + abs(s2) * np.diag(np.ones(2 * N))
            L, a = ref_solve(Kyy, z)
            ahp = refine_hp(Kyy, z, L, a)
            t = f"{fam}_N{N}"
            fits[t + "_q"] = q; fits[t + "_P"] = P; fits[t + "_z"] = z; fits[t + "_hyp"] = h
            fits[t + "_sig2n"] = s2; fits[t + "_alpha"] = a; fits[t + "_alpha_hp"] = ahp
            fits[t + "_nll"] = 0.5 * z @ a + np.sum(np.log(L.diagonal()))
            fits[t + "_Ldiag"] = L.diagonal().copy()
            fits[t + "_cond"] = np.linalg.cond(Kyy)
            print(t, "cond %.3g" % fits[t + "_cond"], "|a-ahp|/|ahp| %.2e" %
                  (np.linalg.norm(a - ahp) / np.linalg.norm(ahp)))
    # driver-like conditioning (l ~ O(1), tiny noise) at N=20: 01_pendulum/implicit/main.py:58-62
    rng = np.random.default_rng(7)
    N = 20
    q = rng.uniform(0, 2 * np.pi, N); P = rng.uniform(-3, 3, N); z = rng.standard_normal(2 * N)
    h = np.array([1.3, 2.1, 0.9]); s2 = 1e-8
    K = ref.build_K("A", q, P, q, P, h); Kyy = K + s2 * np.eye(2 * N)
    L, a = ref_solve(Kyy, z)
    for k, v in dict(q=q, P=P, z=z, hyp=h, sig2n=s2, alpha=a, alpha_hp=refine_hp(Kyy, z, L, a),
                     nll=0.5 * z @ a + np.sum(np.log(L.diagonal())), Ldiag=L.diagonal().copy(),
                     cond=np.linalg.cond(Kyy)).items():
        fits["A_driver20_" + k] = v
    print("A_driver20 cond %.3g" % fits["A_driver20_cond"])
    np.savez_compressed(os.path.join(HERE, "fits.npz"), **fits)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
