"""SGPR_FAM_USER: the kernel-family slot that has no hand-written code anywhere -- Gram build (d = 1, d > 1), batched fits,
predictors, maps, nll_grad and the 19-function `kernels` module all run what tools/gen_kernels.py printed from USER_FAMILY
(the step the reference performs with python/03_henon_heiles/init_func.py:24-81).  As shipped the slot holds the
Henon-Heiles kernel a second time, so it is held (a) against sympy's own numerical evaluation of the same expressions
(tests/golden/user_family.npz, written by the generator) and (b) against the hand-written family C through every path."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uf(golden_dir):
    return np.load(os.path.join(golden_dir, "user_family.npz"))


def _rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def test_user_19_functions_vs_sympy(uf):
    """every function of the `kernels` module (init_func.py:55-76) for the user slot against sympy.lambdify"""
    from sympgpr_amd import kernels, ops
    args = [uf[k] for k in ("xa", "ya", "xb", "yb")] + [float(uf["l"][0]), float(uf["l"][1])]
    if bool(uf["has_p"]):
        args.append(float(uf["p"]))
    with ops.family_scope("USER"):
        for name in kernels.__all__:
            got = getattr(kernels, name)(*args)
            assert _rel(got, uf["f_" + name]) < 4e-15, name


def test_user_gram_vs_sympy(uf):
    """build_K (d = 1) and the d = 2 Gram matrix against sympy's evaluation of d2k / dx_r dx'_c of the same definition"""
    from sympgpr_amd import ops
    has_p = bool(uf["has_p"])
    hyp = list(uf["l"]) + ([float(uf["p"])] if has_p else []) + [float(uf["g1_sig"])]
    n, n0 = len(uf["g1_x"]), len(uf["g1_x0"])
    K = np.full((2 * n, 2 * n0), np.nan, order="F")
    ops.build_k(uf["g1_x"], uf["g1_y"], uf["g1_x0"], uf["g1_y0"], hyp, K, family="USER")
    assert _rel(K, uf["g1_K"]) < 4e-15
    hyp2 = list(uf["g2_l"]) + (list(uf["g2_p"]) if has_p else []) + [float(uf["g2_sig"])]
    K2 = ops.build_k_nd(uf["g2_X"], uf["g2_X0"], hyp2, family="USER")
    assert _rel(K2, uf["g2_K"]) < 4e-15


def test_user_slot_equals_hand_written_family_c_everywhere(uf):
    """the same kernel through both slots: Gram (d = 1, 2, 3), length derivatives, fit, predictors, nll_grad, the batched
    paths (one workgroup per problem, and the mid-size chain), the implicit map"""
    if str(uf["definition"]) != "exp(-(x_a - x_b)**2/(2*lx**2))*exp(-(y_a - y_b)**2/(2*ly**2))":
        pytest.skip("USER_FAMILY has been edited: no hand-written twin to compare with")
    from sympgpr_amd import func, ops
    from sympgpr_amd.fit import SympFit, fit_batch
    from sympgpr_amd.predict import Predictor
    rng = np.random.default_rng(11)
    N, N0 = 700, 300
    x, y, x0, y0 = rng.uniform(-2, 2, N), rng.uniform(-2, 2, N), rng.uniform(-2, 2, N0), rng.uniform(-2, 2, N0)
    hyp = [0.6, 0.8, 1.4]
    Ku, Kc = (np.empty((2 * N, 2 * N0), order="F") for _ in range(2))
    ops.build_k(x, y, x0, y0, hyp, Ku, family="USER")
    ops.build_k(x, y, x0, y0, hyp, Kc, family="C")
    assert _rel(Ku, Kc) < 2e-15
    for du, dc in zip(ops.build_dk(x[:100], y[:100], x0[:80], y0[:80], hyp, family="USER"),
                      ops.build_dk(x[:100], y[:100], x0[:80], y0[:80], hyp, family="C")):
        assert _rel(du, dc) < 4e-15
    Ru, Rc = (np.empty((N, N0), order="F") for _ in range(2))
    ops.buildkreg(x, y, x0, y0, hyp, Ru, family="USER")
    ops.buildkreg(x, y, x0, y0, hyp, Rc, family="C")
    assert _rel(Ru, Rc) < 2e-15
    for d in (2, 3):
        X, X0 = rng.uniform(-1, 1, (90, 2 * d)), rng.uniform(-1, 1, (70, 2 * d))
        h = list(rng.uniform(0.7, 1.3, 2 * d)) + [1.2]
        assert _rel(ops.build_k_nd(X, X0, h, family="USER"), ops.build_k_nd(X, X0, h, family="C")) < 4e-15
    # fit, predictor, nll_grad
    z = rng.standard_normal(2 * N)
    l = 2.0 * np.sqrt(16.0 / N)
    hyp, s2 = [l, l, 1.0], 1e-3
    res = {}
    for fam in ("USER", "C"):
        with SympFit(fam, x, y, z, hyp, s2, lower_only=False) as f:
            f.run()
            res[fam] = (f.alpha(), f.nll(), f.nll_grad_terms())
        pr = Predictor(fam, x, y, hyp, res[fam][0])
        res[fam] += (np.array(pr(x0, y0)),)
    assert _rel(res["USER"][0], res["C"][0]) < 1e-10 and abs(res["USER"][1] - res["C"][1]) <= 1e-12 * abs(res["C"][1])
    assert _rel(np.array(res["USER"][2]), np.array(res["C"][2])) < 1e-9
    assert _rel(res["USER"][3], res["C"][3]) < 1e-10
    xin = np.hstack((x[:200], y[:200]))
    hv = np.array([l, l, 1.0, s2])
    with ops.family_scope("USER"):
        vu, gu = func.nll_grad(hv, xin, z[:400], 400)
    with ops.family_scope("C"):
        vc, gc = func.nll_grad(hv, xin, z[:400], 400)
    assert abs(vu - vc) <= 1e-12 * abs(vc) and _rel(gu, gc) < 1e-10
    # batched fits: one workgroup per problem (order 120) and the mid-size path (order 600)
    for n_pts in (60, 300):
        B = 5
        xs, ys = rng.uniform(-2, 2, (B, n_pts)), rng.uniform(-2, 2, (B, n_pts))
        zs = rng.standard_normal((B, 2 * n_pts))
        lb = 2.0 * np.sqrt(16.0 / n_pts)
        hs = np.column_stack((lb * rng.uniform(0.8, 1.2, B), lb * rng.uniform(0.8, 1.2, B), np.ones(B)))
        au, nu, iu = fit_batch("USER", xs, ys, zs, hs, 1e-2)
        ac, nc, ic = fit_batch("C", xs, ys, zs, hs, 1e-2)
        assert np.all(iu == 0) and np.all(ic == 0)
        assert _rel(au, ac) < 1e-10 and np.allclose(nu, nc, rtol=1e-12)
    # the implicit map of the Henon-Heiles driver (func.py:225-247) through both slots
    from sympgpr_amd.maps import run_map
    Nt = 150
    q, p = rng.uniform(-1, 1, Nt), rng.uniform(-1, 1, Nt)
    P = p - 0.05 * np.sin(2 * q)
    Q = q + 0.1 * P
    xtrain, ztrain = np.hstack((q, P)), np.concatenate((p - P, Q - q))
    xtrainp, ztrainp = np.hstack((q, p)), P - p
    hm, hp = [0.9, 0.9, 1.0], [0.9, 0.9, 1.0]
    out = {}
    for fam in ("USER", "C"):
        with SympFit(fam, q, P, ztrain, hm, 1e-4, lower_only=False) as f:
            Kyinv = f.run().inverse()
        with SympFit(fam, q, p, ztrainp, hp, 1e-4, reg=True, lower_only=False) as f:
            Kyinvp = f.run().inverse()
        out[fam] = run_map(0, 6, 5, hm, np.linspace(-0.5, 0.5, 5), np.linspace(0.4, -0.4, 5), xtrain, ztrain, Kyinv, hp, xtrainp,
                           ztrainp, Kyinvp, family=fam)
    # (150 SE-kernel points on [-1, 1]^2 with l = 0.9: cond(Ky) ~ 1e8 even with this noise; the two slots differ by rounding)
    assert _rel(out["USER"][0], out["C"][0]) < 1e-8 and _rel(out["USER"][1], out["C"][1]) < 1e-8
