"""CPU suite: pins the oracle (oracle/sympgpr_oracle.c) against the reference's own test
inputs (python/05_tokamak/SympGPR/test_sympgpr.py:7-10,19-45) and against golden vectors
generated from the reference's compiled Fortran + SciPy (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

FAMS = "ABCD"


@pytest.fixture(scope="module")
def ka(golden_dir):
    return json.load(open(os.path.join(golden_dir, "known_answer.json")))


@pytest.fixture(scope="module")
def gram(golden_dir):
    return np.load(os.path.join(golden_dir, "gram.npz"))


@pytest.fixture(scope="module")
def fits(golden_dir):
    return np.load(os.path.join(golden_dir, "fits.npz"))


def test_known_answer_build_K(oracle, ka):
    # same tolerance as the reference's own equivalence test (test_sympgpr.py:26,35,44)
    K = oracle.build_K("A", ka["x"], ka["y"], ka["x0"], ka["y0"], ka["hyp"])
    assert K.shape == (6, 4) and K.flags.f_contiguous
    np.testing.assert_allclose(K, np.array(ka["build_K_6x4"]), rtol=1e-12, atol=1e-12)
    # sanity rows quoted in SURVEY.md 8(c): diag of q-block sig/4/lx^2, of P-block sig/ly^2
    assert K[0, 0] == pytest.approx(0.4, abs=1e-15) and K[3, 2] == pytest.approx(0.1, abs=1e-15)
    G = oracle.buildKreg("A", ka["x"], ka["y"], ka["x0"], ka["y0"], ka["hyp"])
    np.testing.assert_allclose(G, np.array(ka["buildKreg_3x2"]), rtol=1e-12, atol=1e-12)
    G1 = oracle.buildKreg("A", ka["x"][:1], ka["y"][:1], ka["x0"], ka["y0"], ka["hyp"])
    np.testing.assert_allclose(G1, np.array(ka["buildKreg_1x2"]), rtol=1e-12, atol=1e-12)


def test_known_answer_fit(oracle, ka):
    f = ka["fit6"]
    alpha, nll, L = oracle.fit("A", ka["x"], ka["y"], f["z"], ka["hyp"], f["sig2n"])
    np.testing.assert_allclose(alpha, f["alpha"], rtol=1e-12)
    assert nll == pytest.approx(f["nll"], rel=1e-13)
    assert np.all(np.triu(L, 1) == 0.0)


def test_known_answer_predict_rows(oracle, ka):
    """calcq / guessP of the reference recompute alpha = Kyinv @ ztrain per call
    (sympgpr.f90:72,85); with that alpha handed in, the row forms must agree."""
    alpha = np.array(ka["Kyinv"]) @ np.array(ka["ztrain"])
    op, oq = oracle.predict_rows("A", ka["x"][:1], ka["y"][:1], ka["x0"], ka["y0"], ka["hyp"], alpha)
    assert oq[0] == pytest.approx(ka["calcQ"], rel=1e-12)
    alphap = np.array(ka["Kyinvp"]) @ np.array(ka["ztrainp"])
    g = oracle.predict_reg("A", ka["x"][:1], ka["y"][:1], ka["x0"], ka["y0"], ka["hypp"], alphap)
    assert g[0] == pytest.approx(ka["guessP"], rel=1e-12)
    # calcP's fixed point: f(P) = pGP(q, P) - p + P = 0 (sympgpr.f90:112-124), hybrd1 tol 1e-13
    P = ka["calcP"]
    op, _ = oracle.predict_rows("A", ka["x"][:1], [P], ka["x0"], ka["y0"], ka["hyp"], alpha)
    assert abs(op[0] - ka["y"][0] + P) < 1e-11


@pytest.mark.parametrize("fam", FAMS)
@pytest.mark.parametrize("tag", ["sq8", "sq64", "rect5x7", "row1x9", "col9x1"])
def test_gram_golden(oracle, gram, fam, tag):
    g = lambda k: gram[f"{fam}_{tag}_{k}"]
    K = oracle.build_K(fam, g("x"), g("y"), g("x0"), g("y0"), g("hyp"))
    scale = np.abs(g("K")).max()
    assert np.abs(K - g("K")).max() <= 4e-16 * scale
    G = oracle.buildKreg(fam, g("x"), g("y"), g("x0"), g("y0"), g("hyp"))
    assert np.abs(G - g("Kreg")).max() <= 4e-16 * np.abs(g("Kreg")).max()
    if tag.startswith("sq") and fam != "B":
        # exact symmetry; family B's exp((-ya^2/2 + ya*yb - yb^2/2)/ly^2)
        # (kernels_sum.f90:9) is not rounding-symmetric under a<->b
        assert np.abs(K - K.T).max() == 0.0
    if fam == "B":
        n, n0 = len(g("x")), len(g("x0"))
        assert np.all(K[n:, :n0] == 0.0) and np.all(K[:n, n0:] == 0.0)


def test_gram_empty(oracle):
    K = oracle.build_K("A", [], [], [], [], [0.5, 2.0, 0.4])
    assert K.shape == (0, 0)


def test_threads_same_result(oracle, gram):
    g = lambda k: gram[f"A_sq64_{k}"]
    K1 = oracle.build_K("A", g("x"), g("y"), g("x0"), g("y0"), g("hyp"), threads=1)
    K4 = oracle.build_K("A", g("x"), g("y"), g("x0"), g("y0"), g("hyp"), threads=4)
    assert np.array_equal(K1, K4)


@pytest.mark.parametrize("tag", ["A_N32", "A_N128", "A_N512", "C_N32", "C_N128", "C_N512", "A_driver20"])
def test_fit_golden(oracle, fits, tag):
    g = lambda k: fits[f"{tag}_{k}"]
    alpha, nll, L = oracle.fit(tag[0], g("q"), g("P"), g("z"), g("hyp"), float(g("sig2n")))
    cond = float(g("cond"))
    # tolerance budget: both solves are backward stable -> relative difference <~ cond * eps * c
    tol = max(1e-10, 50 * cond * 2.2e-16)
    rel = np.linalg.norm(alpha - g("alpha_hp")) / np.linalg.norm(g("alpha_hp"))
    assert rel < tol, (rel, cond)
    rel2 = np.linalg.norm(alpha - g("alpha")) / np.linalg.norm(g("alpha"))
    assert rel2 < tol
    assert nll == pytest.approx(float(g("nll")), rel=max(1e-11, tol))
    np.testing.assert_allclose(np.diag(L), g("Ldiag"), rtol=1e-10)


def test_not_positive_definite(oracle):
    with pytest.raises(np.linalg.LinAlgError):
        oracle.cholesky(np.array([[1.0, 2.0], [2.0, 1.0]]))


def test_multi_rhs_solve(oracle, fits):
    g = lambda k: fits[f"A_N32_{k}"]
    K = oracle.build_K("A", g("q"), g("P"), g("q"), g("P"), g("hyp"))
    Ky = K + float(g("sig2n")) * np.eye(64)
    L = oracle.cholesky(Ky)
    B = np.random.default_rng(0).standard_normal((64, 5))
    X = oracle.solve_cholesky(L, B)
    assert np.abs(Ky @ X - B).max() < 1e-10


def test_length_scale_derivative_kernels_vs_reference_fortran(oracle):
    """the eight d../dl functions build_dK / build_dKreg use (kernels.f90:135-231,
    kernels_sq.f90:124-217, kernels_sum.f90:120-208), restated in the oracle, against the reference's
    compiled Fortran."""
    from oracle.oracle import Ref, DL_NAMES
    if not Ref.available():
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    ref = Ref()
    rng = np.random.default_rng(3)
    for fam in "ABC":
        for _ in range(100):
            a = rng.uniform(-3, 3, 4)
            lx, ly = rng.uniform(0.3, 2, 2)
            for w, name in DL_NAMES.items():
                v, r = oracle.scalar_dl(fam, w, *a, lx, ly), ref.scalar(fam, name, *a, lx, ly)
                assert abs(v - r) <= 1e-12 * max(abs(r), 1e-3)


def test_nll_grad_is_the_gradient(oracle):
    """functions/func.py:148-162 restated in the oracle: nlp_grad against central differences
    of nlp_val in (lx, ly)."""
    rng = np.random.default_rng(5)
    Np = 12
    x = np.hstack((rng.uniform(0, 2 * np.pi, Np), rng.uniform(-2, 2, Np)))
    y = rng.standard_normal(2 * Np)
    hyp = np.array([0.9, 1.1, 0.7, 1e-2])
    for fam in "ABC":
        val, g = oracle.nll_grad(fam, hyp, x, y, 2 * Np)
        for i in (0, 1):
            h = 1e-6
            hp, hm = hyp.copy(), hyp.copy()
            hp[i] += h; hm[i] -= h
            fd = (oracle.nll_grad(fam, hp, x, y, 2 * Np)[0] - oracle.nll_grad(fam, hm, x, y, 2 * Np)[0]) / (2 * h)
            assert g[i] == pytest.approx(fd, rel=1e-6, abs=1e-7)
        alpha, nll, _ = oracle.fit(fam, x[:Np], x[Np:], y, hyp[:3], hyp[3])
        assert val == pytest.approx(nll, rel=1e-12)


def test_build_K_nd_reduces_to_build_K_for_one_pair(oracle, gram):
    for fam in "AC":
        g = lambda k: gram[f"{fam}_rect5x7_{k}"]
        X = np.column_stack((g("x"), g("y")))
        X0 = np.column_stack((g("x0"), g("y0")))
        K = oracle.build_K_nd(fam, X, X0, g("hyp"))
        assert np.abs(K - g("K")).max() <= 4e-16 * np.abs(g("K")).max()


@pytest.mark.parametrize("fam,d", [("A", 2), ("C", 2), ("A", 3), ("B", 2), ("D", 2), ("D", 1), ("B", 1)])
def test_build_K_nd_against_sympy(oracle, fam, d):
    """d > 1 is not in the reference: the oracle is pinned on a symbolic differentiation of the
    kernel, the technique of the reference's own generator (01_pendulum/implicit/
    init_func.py:24-52: define k, differentiate, evaluate).  A, C, D: product of the per-coordinate
    factors (D with a free period per q); B: their sum."""
    import sympy as sp
    D = 2 * d
    xa = sp.symbols("xa0:%d" % D)
    xb = sp.symbols("xb0:%d" % D)
    ls = sp.symbols("l0:%d" % D, positive=True)
    ps = sp.symbols("p0:%d" % d, positive=True)
    fac = []
    for m in range(D):
        if m < d and fam in "AB":
            fac.append(sp.exp(-sp.sin((xa[m] - xb[m]) / 2) ** 2 / (2 * ls[m] ** 2)))
        elif m < d and fam == "D":
            fac.append(sp.exp(-sp.sin(ps[m] * (xa[m] - xb[m])) ** 2 / (2 * ls[m] ** 2)))
        else:
            fac.append(sp.exp(-(xa[m] - xb[m]) ** 2 / (2 * ls[m] ** 2)))
    k = sum(fac) if fam == "B" else sp.prod(fac)
    args = xa + xb + ls + (ps if fam == "D" else ())
    H = [[sp.lambdify(args, sp.diff(k, xa[a], xb[b]), "mpmath") for b in range(D)] for a in range(D)]
    rng = np.random.default_rng(100 * d + ord(fam))
    n, n0 = 3, 2
    X = rng.uniform(-1.5, 1.5, (n, D)); X0 = rng.uniform(-1.5, 1.5, (n0, D))
    l = rng.uniform(0.6, 1.4, D); sig = 0.7
    pv = rng.uniform(0.4, 0.9, d)
    hyp = np.concatenate((l, pv, [sig])) if fam == "D" else np.append(l, sig)
    K = oracle.build_K_nd(fam, X, X0, hyp)
    extra = tuple(pv) if fam == "D" else ()
    for a in range(D):
        for b in range(D):
            for i in range(n):
                for j in range(n0):
                    ref = sig * float(H[a][b](*X0[j], *X[i], *l, *extra))   # a = column ("0") point, as in build_K
                    assert K[a * n + i, b * n0 + j] == pytest.approx(ref, rel=1e-12, abs=1e-14)
    if d == 1:   # one pair: the family's own build_K (restated generated Fortran)
        hyp1 = [l[0], l[1], pv[0], sig] if fam == "D" else [l[0], l[1], sig]
        K1 = oracle.build_K(fam, X[:, 0], X[:, 1], X0[:, 0], X0[:, 1], hyp1)
        assert np.abs(K - K1).max() <= 1e-14 * np.abs(K1).max()


@pytest.mark.parametrize("fam", "ABCD")
def test_all_generated_scalars_vs_reference_fixture(oracle, golden_dir, fam):
    """tests/golden/scalars.json: all 19 functions of each kernels*.f90, evaluated by the
    reference's compiled Fortran (make_scalar_golden.py).  The oracle restates the 4 Gram
    functions (all families), the 8 length-scale derivatives (A, C) and the 7 unused ones."""
    import json
    from oracle.oracle import DL_NAMES, X_NAMES
    g = json.load(open(os.path.join(golden_dir, "scalars.json")))[fam]
    a = g["args"]
    m = len(a["x_a"])
    close = lambda v, r: abs(v - r) <= 1e-13 * max(abs(r), 1e-3)
    for i in range(m):
        pt = (a["x_a"][i], a["y_a"][i], a["x_b"][i], a["y_b"][i], a["lx"][i], a["ly"][i])
        p = a["p"][i] if fam == "D" else 0.0
        for w, name in enumerate(("kern_num", "d2kdxdx0_num", "d2kdydy0_num", "d2kdxdy0_num")):
            assert close(oracle.scalar(fam, w, *pt, p), g["values"][name][i]), name
        for w, name in X_NAMES.items():
            assert close(oracle.scalar_x(fam, w, *pt, p), g["values"][name][i]), name
        if fam in "ABC":
            for w, name in DL_NAMES.items():
                assert close(oracle.scalar_dl(fam, w, *pt), g["values"][name][i]), name
