"""GPU tests of a driver FLOW end to end against values recorded from the reference's own code
(tests/golden/make_driver_golden.py: python/01_pendulum/implicit/func.py over the reference's compiled
kernels.f90; tests/golden/make_tok_golden.py: the compiled `sympgpr.applymap_tok`).  What north_star
calls "the drivers call it unchanged": the sequence of python/01_pendulum/implicit/main.py:116-175 --
optimiser objective, regular-GP matrices, final build_K + inverse, applymap -- on the GPU surface."""
import os

import numpy as np
import pytest
from scipy.optimize import minimize

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def drv(golden_dir):
    return np.load(os.path.join(golden_dir, "driver_pendulum.npz"))


def test_pendulum_objective_at_recorded_points(drv):
    """nll_chol at the recorded hyper-parameter points of both optimiser stages (main.py:126-130 with the
    N/2-slice quirk of SURVEY 3.5, and main.py:143-149)."""
    from sympgpr_amd.examples import pendulum_implicit as pi
    N = int(drv["N"])
    q, p, Q, P = (drv[k] for k in "qpQP")
    xtrainp, ztrainp = np.hstack((q, p)), P
    xtrain, ztrain = np.hstack((q, P)), np.concatenate((p - P, Q - q))
    for h, want in zip(drv["step1_log10l"], drv["step1_nll"]):
        got = pi.nll_chol(np.hstack((10.0**h, drv["sigp"], [drv["sig2n_p"]])), xtrainp, ztrainp, N)
        assert got == pytest.approx(float(want), rel=1e-9)
    for h, want in zip(drv["step2_log10l"], drv["step2_nll"]):
        got = pi.nll_chol(np.hstack((10.0**h, drv["sig"], [drv["sig2n"]])), xtrain, ztrain, 2 * N)
        assert got == pytest.approx(float(want), rel=1e-9)
    # the optimiser's own trace: every point the reference evaluated
    tr = drv["opt_trace"]
    for row in tr[:: max(1, len(tr) // 12)]:
        got = pi.nll_chol(np.hstack((10.0**row[:2], drv["sig"], [drv["sig2n"]])), xtrain, ztrain, 2 * N)
        assert got == pytest.approx(float(row[2]), rel=1e-8)


def test_pendulum_lbfgsb_reaches_recorded_optimum(drv):
    """minimize(L-BFGS-B) over the GPU nll_chol from the driver's start and bounds (main.py:146-149)"""
    from sympgpr_amd.examples import pendulum_implicit as pi
    N = int(drv["N"])
    q, p, Q, P = (drv[k] for k in "qpQP")
    xtrain, ztrain = np.hstack((q, P)), np.concatenate((p - P, Q - q))

    def nll_transform(log10hyp, sig, sig2n, x, y, n):
        return pi.nll_chol(np.hstack((10**log10hyp, sig, [sig2n])), x, y, n)
    res = minimize(nll_transform, np.array((-1.0, -1.0)), args=(float(drv["sig"]), float(drv["sig2n"]), xtrain, ztrain, 2 * N),
                   method="L-BFGS-B", bounds=((-10, 1), (-10, 1)))
    ref_fun = float(drv["opt_fun"])
    assert res.fun == pytest.approx(ref_fun, rel=1e-8)
    np.testing.assert_allclose(res.x, drv["opt_x"], atol=2e-4)
    # and exactly at the reference's optimum
    assert nll_transform(drv["opt_x"], float(drv["sig"]), float(drv["sig2n"]), xtrain, ztrain, 2 * N) == \
        pytest.approx(ref_fun, rel=1e-9)


def test_pendulum_final_matrices_and_map(drv):
    """buildKreg / build_K + inverse (main.py:136-138,156-159), training prediction (main.py:163-165) and
    applymap (main.py:170-172) against the reference's values"""
    from sympgpr_amd.examples import pendulum_implicit as pi
    from sympgpr_amd.fit import SympFit
    N = int(drv["N"])
    q, p, Q, P = (drv[k] for k in "qpQP")
    xtrainp, ztrainp = np.hstack((q, p)), P
    xtrain, ztrain = np.hstack((q, P)), np.concatenate((p - P, Q - q))
    hyp, hypp, s2, s2p = drv["hyp"], drv["hypp"], float(drv["sig2n"]), float(drv["sig2n_p"])
    Kp = np.zeros((N, N), order="F")
    pi.buildKreg(xtrainp, xtrainp, hypp, Kp)
    np.testing.assert_allclose(Kp[::32], drv["Kp_rows"], rtol=1e-12, atol=1e-12 * np.abs(drv["Kp_rows"]).max())
    K = np.empty((2 * N, 2 * N), order="F")
    pi.build_K(xtrain, xtrain, hyp, K)
    np.testing.assert_allclose(K[::64], drv["K_rows"], rtol=1e-12, atol=1e-12 * np.abs(drv["K_rows"]).max())
    tol = max(1e-10, 50 * float(drv["cond"]) * 2.2e-16)
    with SympFit("A", q, P, ztrain, hyp, s2) as f:
        f.run()
        alpha = f.alpha()
        Kyinv = f.inverse()
    assert np.linalg.norm(alpha - drv["alpha"]) / np.linalg.norm(drv["alpha"]) < tol
    assert np.linalg.norm(Kyinv @ ztrain - drv["alpha"]) / np.linalg.norm(drv["alpha"]) < 10 * tol
    assert np.linalg.norm(K @ alpha - drv["Eftrain"]) / np.linalg.norm(drv["Eftrain"]) < tol
    with SympFit("A", q, p, ztrainp, hypp, s2p, reg=True) as f:
        f.run()
        Kyinvp = f.inverse()
    tolp = max(1e-10, 50 * float(drv["cond_p"]) * 2.2e-16)
    assert np.linalg.norm(Kyinvp @ ztrainp - drv["alphap"]) / np.linalg.norm(drv["alphap"]) < 10 * tolp
    qmap, pmap = pi.applymap(int(drv["nm"]), int(drv["Ntest"]), hyp, hypp, drv["Q0map"], drv["P0map"], xtrainp, ztrainp,
                             Kyinvp, xtrain, ztrain, Kyinv)
    # the reference iterates scipy's secant at most 5 times per step (func.py:145); the device solves the same
    # equation to 1e-13: agreement to the reference's own applymap tolerance (test_sympgpr.py:92-94: 1e-8)
    np.testing.assert_allclose(pmap, drv["pmap"], rtol=1e-7, atol=1e-7)
    dq = np.abs(np.mod(qmap - drv["qmap"] + np.pi, 2 * np.pi) - np.pi)
    assert dq.max() < 1e-7


def test_applymap_tok_f2py_form(golden_dir):
    """sympgpr.applymap_tok as the f2py wrapper is called (test_sympgpr.py:83-90): qmap, pmap F-ordered
    [nm, Ntest, 1] in/out; lost orbits (NaN) are left untouched and an orbit whose next row holds a number
    carries on from it -- against the reference's compiled Fortran (sympgpr.f90:128-177)."""
    from sympgpr_amd import ops
    from sympgpr_amd.fortran.sympgpr import sympgpr
    g = np.load(os.path.join(golden_dir, "applymap_tok.npz"))
    qmap = np.array(g["qmap_in"], order="F")
    pmap = np.array(g["pmap_in"], order="F")
    with ops.family_scope("A"):
        sympgpr.applymap_tok(g["hyp"], g["hypp"], g["Q0map"], g["P0map"], g["xtrainp"], g["ytrainp"], g["ztrainp"],
                             g["Kyinvp"], g["xtrain"], g["ytrain"], g["ztrain"], g["Kyinv"], qmap, pmap)
    assert np.array_equal(np.isnan(pmap), np.isnan(g["pmap_out"]))
    assert np.array_equal(np.isnan(qmap), np.isnan(g["qmap_out"]))
    np.testing.assert_allclose(pmap, g["pmap_out"], rtol=1e-8, atol=1e-8, equal_nan=True)
    dq = np.abs(np.mod(qmap - g["qmap_out"] + np.pi, 2 * np.pi) - np.pi)
    assert np.nanmax(dq) < 1e-8
    # untouched entries are bit-identical to what came in
    assert np.array_equal(qmap[1:, 4, 0], g["qmap_in"][1:, 4, 0])


# ---- the other three drivers (tests/golden/make_flow_golden.py: the reference's own func.py files over its compiled Fortran) ----
class _Devs:
    """relative deviations against the recorded values, all checked at the end (the message lists every one)"""

    def __init__(self):
        self.rows = []

    def rel(self, name, got, want, tol):
        got, want = np.asarray(got, dtype=float), np.asarray(want, dtype=float)
        assert got.shape == want.shape, (name, got.shape, want.shape)
        assert np.array_equal(np.isnan(got), np.isnan(want)), name + ": NaN pattern differs"
        den = np.linalg.norm(np.nan_to_num(want))
        dev = np.linalg.norm(np.nan_to_num(got - want)) / (den if den > 0 else 1.0)
        self.rows.append((name, dev, tol))

    def ang(self, name, got, want, tol):
        assert np.array_equal(np.isnan(got), np.isnan(want)), name + ": NaN pattern differs"
        d = np.abs(np.mod(got - want + np.pi, 2 * np.pi) - np.pi)
        self.rows.append((name, float(np.nanmax(d)), tol))

    def check(self):
        rep = os.environ.get("SGPR_FLOW_REPORT")
        if rep:
            with open(rep, "a") as f:
                for r in self.rows:
                    f.write("%-40s %.3e (tol %.1e)\n" % r)
        bad = [r for r in self.rows if not r[1] <= r[2]]
        assert not bad, "\n".join("%s: %.3e > %.1e" % r for r in bad)


def _fit_all(family, q, y2, z, hyp, s2, reg=False):
    from sympgpr_amd.fit import SympFit
    with SympFit(family, q, y2, z, hyp, s2, reg=reg, lower_only=False) as f:
        f.run()
        return f.alpha(), f.inverse()


def test_henon_heiles_flow(golden_dir):
    """python/03_henon_heiles/main.py:118-182 on the GPU surface: both objectives at recorded points, nll_grad (value and the
    three-entry gradient as written there), L-BFGS-B from the driver's start and bounds, final matrices, build_dK rows,
    applymap_henon -- against the reference's func.py over kernels_sq.f90."""
    from sympgpr_amd.examples import henon_heiles as hh
    g = np.load(os.path.join(golden_dir, "driver_henon.npz"))
    N = int(g["N"])
    q, p, Q, P = (g[k] for k in "qpQP")
    xtrain, ztrain = np.hstack((q, P)), np.concatenate((p - P, Q - q))
    xtrainp, ztrainp = np.hstack((q, p)), P - p
    sig, s2, sigp, s2p = (float(g[k]) for k in ("sig", "sig2n", "sigp", "sig2n_p"))
    dv = _Devs()
    for h, want in zip(g["step1_log10l"], g["step1_nll"]):
        dv.rel("step1 nll %s" % h, hh.nll_chol(np.hstack((10.0**h, sigp, [s2p])), xtrainp, ztrainp, N), want, 1e-9)
    dv.rel("nll_chol_reg", hh.nll_chol_reg(np.hstack((g["hypp"], [s2p])), xtrainp, ztrainp, N), g["nll_reg"], 1e-9)
    for h, want, gw in zip(g["step2_log10l"], g["step2_nll"], g["step2_grad"]):
        v, gr = hh.nll_grad(np.hstack((10.0**h, sig, [s2])), xtrain, ztrain, 2 * N)
        dv.rel("nll_grad value %s" % h, v, want, 1e-9)
        dv.rel("nll_grad grad %s" % h, gr, gw, 1e-7)
    for row in g["opt_trace"][:: max(1, len(g["opt_trace"]) // 10)]:
        dv.rel("trace %s" % row[:2], hh.nll_chol(np.hstack((10.0**row[:2], sig, [s2])), xtrain, ztrain, 2 * N), row[2], 1e-8)

    def obj(h):       # main.py:155-159: out[0] of nll_grad; the same number from nll_chol (func.py:175-177)
        return hh.nll_chol(np.hstack((10.0**h, sig, [s2])), xtrain, ztrain, 2 * N)
    res = minimize(obj, np.array((-1.0, -1.0)), method="L-BFGS-B", tol=1e-8, bounds=((-2, 2), (-2, 2)))
    dv.rel("L-BFGS-B fun", res.fun, g["opt_fun"], 1e-7)
    np.testing.assert_allclose(res.x, g["opt_x"], atol=5e-4)
    hyp, hypp = g["hyp"], g["hypp"]
    Kp = np.zeros((N, N), order="F")
    hh.buildKreg(xtrainp, xtrainp, hypp, Kp)
    dv.rel("Kp rows", Kp[::16], g["Kp_rows"], 1e-12)
    K = np.empty((2 * N, 2 * N), order="F")
    hh.build_K(xtrain, xtrain, hyp, K)
    dv.rel("K rows", K[::32], g["K_rows"], 1e-12)
    sub = np.hstack((q[:24], P[:24]))
    dv.rel("dK (24 points)", np.array(hh.build_dK(sub, sub, hyp)), g["dK_sub"], 1e-11)
    tol, tolp = max(1e-10, 50 * float(g["cond"]) * 2.2e-16), max(1e-10, 50 * float(g["cond_p"]) * 2.2e-16)
    alpha, Kyinv = _fit_all("C", q, P, ztrain, hyp, s2)
    alphap, Kyinvp = _fit_all("C", q, p, ztrainp, hypp, s2p, reg=True)
    dv.rel("alpha", alpha, g["alpha"], tol)
    dv.rel("Kyinv ztrain", Kyinv @ ztrain, g["alpha"], 10 * tol)
    dv.rel("Eftrain", K @ alpha, g["Eftrain"], tol)
    dv.rel("alphap", Kyinvp @ ztrainp, g["alphap"], 10 * tolp)
    qmap, pmap = hh.applymap_henon(int(g["nm"]), int(g["Ntest"]), hyp, hypp, g["Q0map"], g["P0map"], xtrainp, ztrainp, Kyinvp,
                                   xtrain, ztrain, Kyinv)
    # scipy's secant stops at |dp| < 1.48e-8 (func.py:221: newton's default tol); the device solves to 1e-13
    dv.rel("pmap", pmap, g["pmap"], 1e-7)
    dv.rel("qmap", qmap, g["qmap"], 1e-7)
    dv.check()


def test_standard_map_flow(golden_dir):
    """python/04_standard_map/main.py:84-180: the implicit method over kernels.f90 (objectives, L-BFGS-B, final matrices,
    applymap with pdiff) and the explicit method over kernels_expl_per_q_sq_p.f90 (nll_expl per block, both 1-D L-BFGS-B
    runs, build_K, applymap_expl)."""
    from sympgpr_amd.examples import standard_map as sm
    g = np.load(os.path.join(golden_dir, "driver_standard_map.npz"))
    N = int(g["N"])
    q, p, Q, P = (g[k] for k in "qpQP")
    zqtrain, zptrain = Q - q, p - P
    xtrain, ztrain = np.hstack((q, P)), np.concatenate((zptrain, zqtrain))
    xtrainp, ztrainp = np.hstack((q, p)), P - p
    sig, s2, sigp, s2p = (float(g[k]) for k in ("sig", "sig2n", "sigp", "sig2n_p"))
    dv = _Devs()
    for h, want in zip(g["step1_log10l"], g["step1_nll"]):
        dv.rel("step1 nll %s" % h, sm.nll_chol(np.hstack((10.0**h, sigp, [s2p])), xtrainp, ztrainp, N), want, 1e-9)
    for h, want in zip(g["step2_log10l"], g["step2_nll"]):
        dv.rel("step2 nll %s" % h, sm.nll_chol(np.hstack((10.0**h, sig, [s2])), xtrain, ztrain, 2 * N), want, 1e-9)

    def nll_transform(log10hyp, sig, sig2n, x, y, n):            # main.py:112-115
        return sm.nll_chol(np.hstack((10**log10hyp, sig, [sig2n])), x, y, n)
    res = minimize(nll_transform, np.array((0.0, -1.0)), args=(sig, s2, xtrain, ztrain, 2 * N), method="L-BFGS-B", tol=1e-8,
                   bounds=((-2, 2), (-2, 2)))
    dv.rel("L-BFGS-B fun", res.fun, g["opt_fun"], 1e-7)
    np.testing.assert_allclose(res.x, g["opt_x"], atol=5e-4)
    hyp, hypp = g["hyp"], g["hypp"]
    Kp = np.zeros((N, N), order="F")
    sm.buildKreg(xtrainp, xtrainp, hypp, Kp)
    dv.rel("Kp rows", Kp[::16], g["Kp_rows"], 1e-12)
    K = np.empty((2 * N, 2 * N), order="F")
    sm.build_K(xtrain, xtrain, hyp, K)
    dv.rel("K rows", K[::32], g["K_rows"], 1e-12)
    tol, tolp = max(1e-10, 50 * float(g["cond"]) * 2.2e-16), max(1e-10, 50 * float(g["cond_p"]) * 2.2e-16)
    alpha, Kyinv = _fit_all("A", q, P, ztrain, hyp, s2)
    _, Kyinvp = _fit_all("A", q, p, ztrainp, hypp, s2p, reg=True)
    dv.rel("alpha", alpha, g["alpha"], tol)
    dv.rel("Eftrain", K @ alpha, g["Eftrain"], tol)
    dv.rel("alphap", Kyinvp @ ztrainp, g["alphap"], 10 * tolp)
    qmap, pmap, pdiff = sm.applymap(int(g["nm"]), int(g["Ntest"]), hyp, hypp, g["Q0map"], g["P0map"], xtrainp, ztrainp, Kyinvp,
                                    xtrain, ztrain, Kyinv)
    dv.ang("implicit qmap", qmap, g["qmap"], 1e-7)
    dv.ang("implicit pmap", pmap, g["pmap"], 1e-7)
    dv.rel("implicit pdiff", pdiff, g["pdiff"], 1e-7)

    # explicit method
    s2x = float(g["expl_sig2n"])
    for h, wq, wp in zip(g["expl_log10l"], g["expl_nll_q"], g["expl_nll_p"]):
        dv.rel("nll_expl q %g" % h, sm.nll_expl(np.hstack((10.0**h, sig, [s2x])), xtrain, zptrain, 2 * N, 0), wq, 1e-9)
        dv.rel("nll_expl p %g" % h, sm.nll_expl(np.hstack((10.0**h, sig, [s2x])), xtrain, zqtrain, 2 * N, 1), wp, 1e-9)

    def nll_transform_expl(log10hyp, sig, sig2n, x, y, n, ind):  # main.py:152-157
        return sm.nll_expl(np.hstack((10**np.ravel(log10hyp), sig, [sig2n])), x, y, n, ind)
    res_lq = minimize(nll_transform_expl, np.array((1.0,)), args=(sig, s2x, xtrain, zptrain, 2 * N, 0), method="L-BFGS-B")
    res_lp = minimize(nll_transform_expl, np.array((1.0,)), args=(sig, s2x, xtrain, zqtrain, 2 * N, 1), method="L-BFGS-B")
    dv.rel("expl fun q", res_lq.fun, g["expl_fun_q"], 1e-7)
    dv.rel("expl fun p", res_lp.fun, g["expl_fun_p"], 1e-7)
    np.testing.assert_allclose(res_lq.x, g["expl_opt_lq"], atol=5e-4)
    np.testing.assert_allclose(res_lp.x, g["expl_opt_lp"], atol=5e-4)
    hypx = g["expl_hyp"]
    Kx = np.empty((2 * N, 2 * N), order="F")
    sm.build_K_expl(xtrain, xtrain, hypx, Kx)
    dv.rel("expl K rows", Kx[::32], g["expl_K_rows"], 1e-12)
    tolx = max(1e-10, 50 * float(g["expl_cond"]) * 2.2e-16)
    alphax, Kyinvx = _fit_all("B", q, P, ztrain, hypx, s2x)
    dv.rel("expl alpha", alphax, g["expl_alpha"], tolx)
    qx, px, pdx = sm.applymap_expl(int(g["nm"]), int(g["Ntest"]), hypx, g["Q0map"], g["P0map"], xtrain, ztrain, Kyinvx)
    dv.rel("explicit qmap", qx, g["expl_qmap"], 1e-9)
    dv.ang("explicit pmap", px, g["expl_pmap"], 1e-9)
    dv.rel("explicit pdiff", pdx, g["expl_pdiff"], 1e-9)
    dv.check()


def _compute_r(z, rstart):
    """fieldlines.compute_r (tokamak physics beside the path): 20 Newton steps on p_th = A_th(r, th),
    A_th = B0 (r^2/2 - r^3 cos(th) / (3 R0)) with B0 = R0 = 1"""
    r = rstart
    for _ in range(20):
        y = z[0] - (r * r / 2 - r**3 / 3 * np.cos(z[1]))
        dy = -(r - r * r * np.cos(z[1]))
        r = r - y / dy
    return r


def test_tokamak_split_flow(golden_dir, capsys):
    """python/05_tokamak/Split_SympGPR/main.py:22-112: per toroidal section the regular GP (nll_chol_reg at recorded points,
    the L-BFGS-B run of section 0, buildKreg + inverse) and the symplectic GP (nll_chol incl. its eigen fallback, build_K +
    inverse), then applymap_tok over the four sections with the flux-surface test -- one orbit is lost."""
    from sympgpr_amd.examples import tokamak_split as ts
    g = np.load(os.path.join(golden_dir, "driver_tokamak_split.npz"))
    N, nphmap, s2 = int(g["N"]), int(g["nphmap"]), float(g["sig2n"])
    q, p, Q, P = (g[k] for k in "qpQP")
    ztrain, xtrain = np.vstack((p - P, Q - q)), np.vstack((q, P))
    dv = _Devs()
    dv.rel("compute_r", [_compute_r(z, 0.3) for z in g["compute_r_in"]], g["compute_r_out"], 1e-13)
    Kyinvp, Kyinv = np.zeros((nphmap, N, N)), np.zeros((nphmap, 2 * N, 2 * N))
    xtrainp, ztrainp = np.zeros((2 * N, nphmap)), np.zeros((N, nphmap))
    for i in range(nphmap):
        xp, zp = np.hstack((q[:, i], p[:, i])), P[:, i] - p[:, i]
        xtrainp[:, i], ztrainp[:, i] = xp, zp
        tolp, tol = max(1e-10, 50 * float(g["cond_p"][i]) * 2.2e-16), max(1e-10, 50 * float(g["cond"][i]) * 2.2e-16)
        for h, want in zip(g["reg_log10hyp"], g["reg_nll"][i]):
            dv.rel("sec %d nll_chol_reg %s" % (i, h), ts.nll_chol_reg(np.hstack((10.0**h, [s2])), xp, zp, N), want, 1e-8)
        for h, want in zip(g["gp_hyp"], g["gp_nll"][i]):
            dv.rel("sec %d nll_chol %s" % (i, h), ts.nll_chol(np.hstack((h, [s2])), xtrain[:, i], ztrain[:, i], 2 * N), want, 1e-8)
        Kp = np.zeros((N, N), order="F")
        ts.buildKreg(xp, xp, g["hypp"][i], Kp)
        dv.rel("sec %d Kp rows" % i, Kp[::10], g["Kp_rows"][i], 1e-12)
        K = np.empty((2 * N, 2 * N), order="F")
        ts.build_K(xtrain[:, i], xtrain[:, i], g["hyp"][i], K)
        dv.rel("sec %d K rows" % i, K[::20], g["K_rows"][i], 1e-12)
        a, Kyinv[i] = _fit_all("A", q[:, i], P[:, i], ztrain[:, i], g["hyp"][i], s2)
        ap, Kyinvp[i] = _fit_all("A", q[:, i], p[:, i], zp, g["hypp"][i], s2, reg=True)
        dv.rel("sec %d alpha" % i, a, g["alpha"][i], tol)
        dv.rel("sec %d alphap" % i, ap, g["alphap"][i], tolp)
    # the `except:` branch (func.py:158-165): sig < 0, Ky negative definite -> eigsh fallback, log of negative eigenvalues
    capsys.readouterr()
    with np.errstate(all="ignore"):
        v = ts.nll_chol(np.array([0.8, 4.0, -0.5, s2]), xtrain[:, 0], ztrain[:, 0], 2 * N)
    assert "Fallback to eig solver" in capsys.readouterr().out
    assert np.isnan(v) and np.isnan(float(g["gp_nll_negsig"]))
    # the regular GP's optimiser run of section 0 (main.py:31-35 with opt = 'lbfgs')
    xp, zp = xtrainp[:, 0], ztrainp[:, 0]
    tr = g["opt_reg_trace"]
    for row in tr[:: max(1, len(tr) // 10)]:
        dv.rel("trace %s" % row[:3], ts.nll_chol_reg(np.hstack((10.0**row[:3], [s2])), xp, zp, N), row[3], 1e-7)

    def nll_transform2(log10hyp, sig2n, x, y, n):
        return ts.nll_chol_reg(np.hstack((10**log10hyp, [sig2n])), x, y, n)
    res = minimize(nll_transform2, np.array((-1.0, 0.0, 1.0)), args=(s2, xp, zp, N), method="L-BFGS-B")
    dv.rel("L-BFGS-B fun", res.fun, g["opt_reg_fun"], 1e-5)
    qmap, pmap = ts.applymap_tok(nphmap, int(g["nm"]), int(g["Ntest"]), g["Q0map"], g["P0map"], xtrainp, ztrainp, Kyinvp,
                                 g["hypp"], xtrain, ztrain, Kyinv, g["hyp"], compute_r=_compute_r)
    assert np.isnan(g["pmap"][1:, 5]).all() and not np.isnan(g["pmap"][:, :5]).any()       # what the fixture holds
    # sympgpr.calcp solves with MINPACK hybrd1 at tol 1e-13 (sympgpr.f90:107-113); the device to 1e-13 as well
    dv.rel("pmap", pmap, g["pmap"], 1e-7)
    dv.ang("qmap", qmap, g["qmap"], 1e-7)
    dv.check()
