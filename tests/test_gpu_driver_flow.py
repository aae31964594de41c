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
