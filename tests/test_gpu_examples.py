"""GPU parity tests of the per-driver call surfaces (sympgpr_amd/examples), the map variants, the
single-block fits behind nll_expl, the eigen fallback of nll_chol and the section replicas --
each against the reference recurrences / formulas written out with the CPU oracle."""
import numpy as np
import pytest
import scipy.linalg
import scipy.optimize

pytestmark = pytest.mark.gpu

TWO_PI = 2.0 * np.pi


@pytest.fixture(scope="module", autouse=True)
def _device():
    import sympgpr_amd
    if sympgpr_amd.device_count() < 1:
        pytest.fail("no HIP device visible: GPU tests need the MI355X")


def _training(oracle, fam, Nt=40, seed=11, eps=0.3):
    """A gentle symplectic map as training data: P' = p - eps sin q, Q = q + eps P'."""
    rng = np.random.default_rng(seed)
    q, pn = rng.uniform(0, TWO_PI, Nt), rng.uniform(-1, 1, Nt)
    p_old = pn + eps * np.sin(q)
    Q = q + eps * pn
    d = dict(q=q, pn=pn, p_old=p_old)
    d["xtrain"] = np.hstack((q, pn))
    d["ztrain"] = np.hstack((p_old - pn, Q - q))
    d["xtrainp"] = np.hstack((q, p_old))
    d["ztrainp"] = pn
    d["hyp"] = np.array([1.2, 1.5, 0.5, 1.0]) if fam == "D" else np.array([1.2, 1.5, 1.0])
    d["hypp"] = d["hyp"].copy()
    K = oracle.build_K(fam, q, pn, q, pn, d["hyp"]) + 1e-8 * np.eye(2 * Nt)
    Kp = oracle.buildKreg(fam, q, p_old, q, p_old, d["hypp"]) + 1e-8 * np.eye(Nt)
    d["Kyinv"], d["Kyinvp"] = np.linalg.inv(K), np.linalg.inv(Kp)
    d["alpha"], d["alphap"] = d["Kyinv"] @ d["ztrain"], d["Kyinvp"] @ d["ztrainp"]
    return d


def _ref_map(oracle, fam, d, nm, Q0, P0, explicit, wrap_q, wrap_p):
    """The reference double loops (functions/func.py:216-237; 04_standard_map/func.py:218-285;
    01_pendulum/explicit/func_expl.py:113-128) with the oracle's K* rows and MINPACK hybrd."""
    Ntest = len(Q0)
    qr, pr, pd = (np.zeros((nm, Ntest)) for _ in range(3))
    qr[0], pr[0], pd[0] = Q0, P0, P0
    rows = lambda q, P: oracle.predict_rows(fam, [q], [P], d["q"], d["pn"], d["hyp"], d["alpha"])
    for i in range(nm - 1):
        for k in range(Ntest):
            if explicit:
                Pn = pr[i, k] - rows(qr[i, k], pr[i, k])[0][0]
            else:
                g0 = oracle.predict_reg(fam, [qr[i, k]], [pr[i, k]], d["q"], d["p_old"], d["hypp"], d["alphap"])[0]
                f = lambda P: rows(qr[i, k], P[0])[0][0] - pr[i, k] + P[0]
                Pn = scipy.optimize.fsolve(f, [g0], xtol=1e-13)[0]
            pd[i + 1, k] = pd[i, k] + (Pn - pr[i, k])
            pr[i + 1, k] = np.mod(Pn, TWO_PI) if wrap_p else Pn
            dq = rows(qr[i, k], pr[i + 1, k])[1][0]
            qr[i + 1, k] = np.mod(dq + qr[i, k], TWO_PI) if wrap_q else dq + qr[i, k]
    return qr, pr, pd


TOL = dict(rtol=1e-8, atol=1e-8)   # the reference's own applymap tolerance (test_sympgpr.py:92-93)


def test_standard_map_applymap_variants(oracle):
    from sympgpr_amd.examples import standard_map as sm
    rng = np.random.default_rng(5)
    Ntest, nm = 6, 7
    Q0 = rng.uniform(0.5, 5.5, Ntest)
    P0 = rng.uniform(-0.6, 0.6, Ntest)      # negative momenta: P mod 2 pi really wraps
    d = _training(oracle, "A")
    q, p, pdiff = sm.applymap(nm, Ntest, d["hyp"], d["hypp"], Q0, P0, d["xtrainp"], d["ztrainp"], d["Kyinvp"],
                              d["xtrain"], d["ztrain"], d["Kyinv"])
    qr, pr, pdr = _ref_map(oracle, "A", d, nm, Q0, P0, explicit=False, wrap_q=True, wrap_p=True)
    # after the first wrap the orbit sits at P ~ 2 pi - 0.5, outside the training box: the GP
    # extrapolates there exactly like the reference's does; compare every finite step
    ok = np.isfinite(p)
    assert ok[:2].all()
    np.testing.assert_allclose(p[ok], pr[ok], **TOL)
    np.testing.assert_allclose(q[ok], qr[ok], **TOL)
    np.testing.assert_allclose(pdiff[ok], pdr[ok], **TOL)
    assert np.any(np.abs(pdiff[1] - p[1]) > 1.0)          # the wrap happened somewhere

    dB = _training(oracle, "B")
    q, p, pdiff = sm.applymap_expl(nm, Ntest, dB["hyp"], Q0, P0, dB["xtrain"], dB["ztrain"], dB["Kyinv"])
    qr, pr, pdr = _ref_map(oracle, "B", dB, nm, Q0, P0, explicit=True, wrap_q=False, wrap_p=True)
    np.testing.assert_allclose(p, pr, **TOL)
    np.testing.assert_allclose(q, qr, **TOL)
    np.testing.assert_allclose(pdiff, pdr, **TOL)
    # calcP_expl = -pGP[0] + y
    r1 = oracle.predict_rows("B", [Q0[0]], [P0[0]], dB["q"], dB["pn"], dB["hyp"], dB["alpha"])[0][0]
    assert sm.calcP_expl(Q0[0], P0[0], dB["hyp"], dB["xtrain"], dB["ztrain"], dB["Kyinv"]) == pytest.approx(
        -r1 + P0[0], rel=1e-11, abs=1e-12)


def test_explicit_pendulum_and_henon_maps(oracle):
    from sympgpr_amd.examples import henon_heiles as hh
    from sympgpr_amd.examples import pendulum_explicit as pe
    from sympgpr_amd.examples import pendulum_period_unknown as pu
    rng = np.random.default_rng(6)
    Ntest, nm = 5, 6
    Q0, P0 = rng.uniform(0.5, 5.5, Ntest), rng.uniform(-0.5, 0.5, Ntest)
    dB = _training(oracle, "B")
    q, p = pe.applymap(dB["hyp"], Q0, P0, dB["xtrain"], dB["ztrain"], dB["Kyinv"], Ntest, nm)
    qr, pr, _ = _ref_map(oracle, "B", dB, nm, Q0, P0, explicit=True, wrap_q=True, wrap_p=False)
    np.testing.assert_allclose(p, pr, **TOL)
    np.testing.assert_allclose(q, qr, **TOL)
    f, g = pe.calcQ(Q0[1], P0[1], dB["xtrain"], dB["hyp"], dB["Kyinv"], dB["ztrain"], Ntest)
    r1, r2 = oracle.predict_rows("B", [Q0[1]], [P0[1]], dB["q"], dB["pn"], dB["hyp"], dB["alpha"])
    assert (f, g) == (pytest.approx(r2[0], rel=1e-11), pytest.approx(r1[0], rel=1e-11))
    assert pe.calcP(Q0[1], P0[1], dB["hyp"], dB["xtrain"], dB["ztrain"], dB["Kyinv"], Ntest) == pytest.approx(-r1[0], rel=1e-11)
    # nll_chol(hyp, x, y): N = len(x)
    hyp4 = np.hstack((dB["hyp"], [1e-3]))
    a, nll, _ = oracle.fit("B", dB["q"], dB["pn"], dB["ztrain"], dB["hyp"], 1e-3)
    assert pe.nll_chol(hyp4, dB["xtrain"], dB["ztrain"]) == pytest.approx(nll, rel=1e-10)

    dC = _training(oracle, "C")
    q, p = hh.applymap_henon(nm, Ntest, dC["hyp"], dC["hypp"], Q0, P0, dC["xtrainp"], dC["ztrainp"], dC["Kyinvp"],
                             dC["xtrain"], dC["ztrain"], dC["Kyinv"])
    qr, pr, _ = _ref_map(oracle, "C", dC, nm, Q0, P0, explicit=False, wrap_q=False, wrap_p=False)
    np.testing.assert_allclose(p, pr, **TOL)
    np.testing.assert_allclose(q, qr, **TOL)

    dD = _training(oracle, "D")
    q, p = pu.applymap(nm, Ntest, dD["hyp"], dD["hypp"], Q0, P0, dD["xtrainp"], dD["ztrainp"], dD["Kyinvp"],
                       dD["xtrain"], dD["ztrain"], dD["Kyinv"])
    qr, pr, _ = _ref_map(oracle, "D", dD, nm, Q0, P0, explicit=False, wrap_q=True, wrap_p=False)
    np.testing.assert_allclose(p, pr, **TOL)
    np.testing.assert_allclose(q, qr, **TOL)


def test_python_predictor_surface(oracle):
    """guessP / calcQ / Pnewton / calcP with the per-example signatures
    (01_pendulum/implicit/func.py:119-147)."""
    from sympgpr_amd.examples import pendulum_implicit as pi
    d = _training(oracle, "A")
    x, y = 2.1, 0.3
    g = pi.guessP([x], [y], d["hypp"], d["xtrainp"], d["ztrainp"], d["Kyinvp"], 1)
    g0 = oracle.predict_reg("A", [x], [y], d["q"], d["p_old"], d["hypp"], d["alphap"])[0]
    assert g.shape == (1,) and g[0] == pytest.approx(g0, rel=1e-11)
    r1, r2 = oracle.predict_rows("A", [x], [y], d["q"], d["pn"], d["hyp"], d["alpha"])
    assert pi.calcQ(x, y, d["xtrain"], d["hyp"], d["Kyinv"], d["ztrain"]) == pytest.approx(r2[0], rel=1e-11)
    f = pi.Pnewton(np.array([0.25]), np.array([x]), np.array([y]), d["hyp"], d["xtrain"], d["Kyinv"], d["ztrain"])
    r1b = oracle.predict_rows("A", [x], [0.25], d["q"], d["pn"], d["hyp"], d["alpha"])[0][0]
    assert f.shape == (1,) and f[0] == pytest.approx(r1b - y + 0.25, rel=1e-10, abs=1e-12)
    P = pi.calcP(x, y, d["hyp"], d["hypp"], d["xtrainp"], d["ztrainp"], d["Kyinvp"], d["xtrain"], d["ztrain"],
                 d["Kyinv"], 1)
    res = lambda Pn: oracle.predict_rows("A", [x], [Pn[0]], d["q"], d["pn"], d["hyp"], d["alpha"])[0][0] - y + Pn[0]
    assert P[0] == pytest.approx(scipy.optimize.fsolve(res, [g0], xtol=1e-13)[0], rel=1e-9)


def test_tokamak_maps_with_loss_test(oracle):
    """applymap_tok of 05_tokamak/SympGPR/func.py:182-211 and the per-section form
    Split_SympGPR/func.py:184-219, with a stand-in for fieldlines.compute_r."""
    from sympgpr_amd.examples import tokamak as tk
    from sympgpr_amd.examples import tokamak_split as ts
    rng = np.random.default_rng(8)
    Ntest, nm = 8, 7
    Q0, P0 = rng.uniform(0.5, 5.5, Ntest), rng.uniform(0.3, 0.7, Ntest)
    d = _training(oracle, "A")
    qr, pr, _ = _ref_map(oracle, "A", d, nm, Q0, P0, explicit=False, wrap_q=True, wrap_p=False)
    thr = np.sort(pr.max(axis=0))[Ntest // 2] - 1e-6        # loses the orbits that climb highest in P
    compute_r = lambda zk, r0: 0.6 if zk[0] > thr * 1e-2 else 0.1
    q, p = tk.applymap_tok(nm, Ntest, d["hyp"], d["hypp"], Q0, P0, d["xtrainp"], d["ztrainp"], d["Kyinvp"],
                           d["xtrain"], d["ztrain"], d["Kyinv"], compute_r=compute_r)
    lost = np.zeros(Ntest, bool)
    for i in range(1, nm):
        lost |= (pr[i] < 0) | (pr[i] > thr)
        assert np.all(np.isnan(p[i, lost])) and np.all(np.isnan(q[i, lost]))
        np.testing.assert_allclose(p[i, ~lost], pr[i, ~lost], **TOL)
        np.testing.assert_allclose(q[i, ~lost], qr[i, ~lost], **TOL)
    assert lost.any() and (~lost).any()
    # several steps per launch with the callback applied afterwards: the same map, bit for bit (orbits are independent)
    q3, p3 = tk.applymap_tok(nm, Ntest, d["hyp"], d["hypp"], Q0, P0, d["xtrainp"], d["ztrainp"], d["Kyinvp"],
                             d["xtrain"], d["ztrain"], d["Kyinv"], compute_r=compute_r, steps_per_launch=3)
    assert np.array_equal(q3, q, equal_nan=True) and np.array_equal(p3, p, equal_nan=True)
    # without a callback the whole map is ONE launch and only P < 0 loses an orbit (SGPR_MAP_LOSS_NEGP): start some orbits
    # close to P = 0 so that the test bites
    P0n = P0.copy()
    P0n[::2] = 0.02
    qrn, prn, _ = _ref_map(oracle, "A", d, nm, Q0, P0n, explicit=False, wrap_q=True, wrap_p=False)
    qn, pn = tk.applymap_tok(nm, Ntest, d["hyp"], d["hypp"], Q0, P0n, d["xtrainp"], d["ztrainp"], d["Kyinvp"],
                             d["xtrain"], d["ztrain"], d["Kyinv"])
    lostn = np.zeros(Ntest, bool)
    for i in range(1, nm):
        lostn |= prn[i] < 0
        assert np.all(np.isnan(pn[i, lostn])) and np.all(np.isnan(qn[i, lostn]))
        np.testing.assert_allclose(pn[i, ~lostn], prn[i, ~lostn], **TOL)
        np.testing.assert_allclose(qn[i, ~lostn], qrn[i, ~lostn], **TOL)
    assert lostn.any() and (~lostn).any()

    # two sections with different GPs, alternating; nm = 8 leaves the last row untouched
    nm = 8
    d2 = _training(oracle, "A", seed=12, eps=0.2)
    st = lambda k: np.stack((d[k], d2[k]), axis=1)
    nph = 2
    q, p = ts.applymap_tok(nph, nm, Ntest, Q0, P0, st("xtrainp"), st("ztrainp"), np.stack((d["Kyinvp"], d2["Kyinvp"])),
                           np.stack((d["hypp"], d2["hypp"])), st("xtrain"), st("ztrain"),
                           np.stack((d["Kyinv"], d2["Kyinv"])), np.stack((d["hyp"], d2["hyp"])))
    qr, pr = np.zeros((nm, Ntest)), np.zeros((nm, Ntest))
    qr[0], pr[0] = Q0, P0
    steps = -(-(nm - nph) // nph) * nph                    # the while loop advances nphmap steps at a time
    for i in range(steps):
        dd = (d, d2)[i % nph]
        live = np.nonzero(~np.isnan(pr[i]))[0]
        qr[i + 1], pr[i + 1] = np.nan, np.nan
        if len(live):
            qq, pp, _ = _ref_map(oracle, "A", dd, 2, qr[i, live], pr[i, live], explicit=False, wrap_q=True, wrap_p=False)
            keep = pp[1] >= 0                               # P < 0: the orbit left the plasma (func.py:213-215)
            qr[i + 1, live[keep]], pr[i + 1, live[keep]] = qq[1, keep], pp[1, keep]
    alive = ~np.isnan(pr[:steps + 1])
    assert alive[-1].any() and (~alive).any()
    assert np.array_equal(np.isnan(p[:steps + 1]), ~alive) and np.array_equal(np.isnan(q[:steps + 1]), ~alive)
    np.testing.assert_allclose(p[:steps + 1][alive], pr[:steps + 1][alive], **TOL)
    np.testing.assert_allclose(q[:steps + 1][alive], qr[:steps + 1][alive], **TOL)
    assert np.all(p[steps + 1:] == 0) and np.all(q[steps + 1:] == 0)


@pytest.mark.parametrize("ind", [0, 1])
def test_nll_expl_single_block(oracle, ind):
    """04_standard_map/func.py:126-141: one diagonal block of the sum-kernel matrix."""
    from sympgpr_amd.examples import standard_map as sm
    rng = np.random.default_rng(31 + ind)
    Np = 150
    x = np.hstack((rng.uniform(0, TWO_PI, Np), rng.uniform(-3, 3, Np)))
    y = rng.standard_normal(Np)
    lq, sig, s2 = 0.35, 0.8, 1e-3
    hyp = np.array([lq, sig, s2])
    # the reference call, other length 0 exactly as nll_expl passes it; the block it slices
    hk = np.array([lq, 0.0, sig]) if ind == 0 else np.array([0.0, lq, sig])
    with np.errstate(all="ignore"):
        K = oracle.build_K("B", x[:Np], x[Np:], x[:Np], x[Np:], hk)
    Ky = (K + s2 * np.eye(2 * Np))[:Np, :Np] if ind == 0 else (K + s2 * np.eye(2 * Np))[Np:, Np:]
    assert np.all(np.isfinite(Ky))
    Lc = scipy.linalg.cholesky(Ky, lower=True)
    a = scipy.linalg.solve_triangular(Lc.T, scipy.linalg.solve_triangular(Lc, y, lower=True), lower=False)
    ref = 0.5 * y.dot(a) + np.sum(np.log(Lc.diagonal()))
    assert sm.nll_expl(hyp, x, y, 2 * Np, ind) == pytest.approx(ref, rel=1e-10)
    # build_K_expl = build_K with the explicit method's kernels
    Kb = np.empty((2 * Np, 2 * Np), order="F")
    sm.build_K_expl(x, x, np.array([0.4, 0.9, sig]), Kb)
    Ko = oracle.build_K("B", x[:Np], x[Np:], x[:Np], x[Np:], [0.4, 0.9, sig])
    assert np.abs(Kb - Ko).max() <= 4e-15 * np.abs(Ko).max()


@pytest.mark.parametrize("fam", ["A", "C", "D"])
def test_length_scale_derivative_scalars(oracle, fam):
    """kernels.dkdlx_num ... d3kdxdy0dly_num (kernels.f90:133-231) through sgpr_kernel_eval_host."""
    from sympgpr_amd import kernels, ops
    from oracle.oracle import DL_NAMES
    rng = np.random.default_rng(3)
    m = 200
    xa, ya, xb, yb = rng.uniform(0, TWO_PI, m), rng.uniform(-2, 2, m), rng.uniform(0, TWO_PI, m), rng.uniform(-2, 2, m)
    lx, ly, p = 0.7, 1.1, 0.45
    with ops.family_scope(fam):
        for which, name in DL_NAMES.items():
            if fam == "D":
                got = getattr(kernels, name)(xa, ya, xb, yb, lx, ly, p)
                h = 1e-6                                    # no generated reference for D: central differences
                base = getattr(kernels, {"dkdl": "kern_num", "d3kdxdx0": "d2kdxdx0_num", "d3kdydy0": "d2kdydy0_num",
                                         "d3kdxdy0": "d2kdxdy0_num"}[name[:-7] if name.startswith("d3k") else "dkdl"])
                if name.endswith("dlx_num") or name == "dkdlx_num":
                    fd = (base(xa, ya, xb, yb, lx + h, ly, p) - base(xa, ya, xb, yb, lx - h, ly, p)) / (2 * h)
                else:
                    fd = (base(xa, ya, xb, yb, lx, ly + h, p) - base(xa, ya, xb, yb, lx, ly - h, p)) / (2 * h)
                assert np.abs(got - fd).max() <= 1e-6 * max(1.0, np.abs(fd).max())
            else:
                got = getattr(kernels, name)(xa, ya, xb, yb, lx, ly)
                ref = np.array([oracle.scalar_dl(fam, which, xa[i], ya[i], xb[i], yb[i], lx, ly) for i in range(m)])
                assert np.abs(got - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())
                assert isinstance(getattr(kernels, name)(1.0, 0.2, 0.4, -0.3, lx, ly), float)


@pytest.mark.parametrize("fam,Np", [("C", 60), ("A", 300)])
def test_nll_grad_three_components(oracle, fam, Np):
    """03_henon_heiles/func.py:168-192 / 05_tokamak/SympGPR/func.py:152-168 restated with the oracle."""
    from sympgpr_amd.examples import henon_heiles as hh
    from sympgpr_amd.examples import tokamak as tk
    mod = hh if fam == "C" else tk
    rng = np.random.default_rng(77 + Np)
    x = np.hstack((rng.uniform(0, TWO_PI, Np), rng.uniform(-3, 3, Np)))
    y = rng.standard_normal(2 * Np)
    l = 2.0 * np.sqrt(12 * np.pi / Np)
    hyp = np.array([l, 1.3 * l, 0.8, 1e-2 / l**2])
    N = 2 * Np
    val, g = mod.nll_grad(hyp, x, y, N)
    K = oracle.build_K(fam, x[:Np], x[Np:], x[:Np], x[Np:], hyp[:-1])
    Ky = K + np.abs(hyp[-1]) * np.eye(N)
    Kyinv = np.linalg.inv(Ky)
    Lc = scipy.linalg.cholesky(Ky, lower=True)
    alpha = Kyinv.dot(y)
    dK = oracle.build_dK(fam, x[:Np], x[Np:], x[:Np], x[Np:], hyp[:-1]) + [K / hyp[-2]]
    ref = np.array([-0.5 * alpha.dot(dK[0].dot(alpha)) + 0.5 * np.trace(Kyinv.dot(dK[0])),
                    -0.5 * alpha.dot(dK[1].dot(alpha)) + 0.5 * np.trace(Kyinv.dot(dK[1])),
                    -0.5 * alpha.dot(dK[1].dot(alpha)) + 0.5 * np.trace(Kyinv.dot(dK[2]))])
    assert val == pytest.approx(0.5 * y.dot(alpha) + np.sum(np.log(Lc.diagonal())), rel=1e-10)
    np.testing.assert_allclose(g, ref, rtol=1e-8)
    d3 = mod.build_dK(x, x, hyp[:-1])
    assert len(d3) == 3
    for a, b in zip(d3, dK):
        assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max()


@pytest.mark.parametrize("n", [1, 2, 7, 64, 301])
def test_eigh_on_device(n):
    from sympgpr_amd import ops
    rng = np.random.default_rng(100 + n)
    A = rng.standard_normal((n, n))
    A = A + A.T
    A[np.triu_indices(n, 1)] = np.nan            # only the lower triangle may be read
    w, Q = ops.eigh(A)
    Af = np.tril(np.nan_to_num(A)) + np.tril(np.nan_to_num(A), -1).T
    wr = np.linalg.eigvalsh(Af)
    scale = max(1.0, np.abs(wr).max())
    assert np.all(np.diff(w) >= 0)
    assert np.abs(w - wr).max() <= 1e-12 * scale
    assert np.abs(Q.T @ Q - np.eye(n)).max() <= 1e-12
    assert np.abs(Q @ np.diag(w) @ Q.T - Af).max() <= 1e-11 * scale


def _fallback_reference(Ky, y, neig, nx, sig2n):
    """02_pert_pendulum/func.py:199-203 with a dense eigensolver (what eigsh computes)."""
    w, Q = np.linalg.eigh(Ky)
    if neig < len(w):
        sel = np.sort(np.argsort(np.abs(w))[len(w) - neig:])
        w, Q = w[sel], Q[:, sel]
    alpha = Q.dot(np.diag(1.0 / w).dot(Q.T.dot(y)))
    with np.errstate(invalid="ignore"):
        return 0.5 * y.dot(alpha) + 0.5 * (np.sum(np.log(w)) + (nx - neig) * np.log(np.abs(sig2n)))


def test_nll_chol_eigen_fallback(oracle, capsys):
    """Ky numerically singular (every training point twice, sig2n ~ 0): cholesky fails and the
    drivers' nll_chol takes the eigsh branch (05_tokamak/Split_SympGPR/func.py:148-166)."""
    from sympgpr_amd.examples import pert_pendulum as pp
    from sympgpr_amd.examples import tokamak_split as ts
    from sympgpr_amd.fit import SympFit
    rng = np.random.default_rng(9)
    M = 24
    q0, P0 = rng.uniform(0, TWO_PI, M), rng.uniform(-3, 3, M)
    q, P = np.hstack((q0, q0)), np.hstack((P0, P0))         # exact duplicates: rank(K) <= n/2
    x = np.hstack((q, P))
    Np = 2 * M
    N = 2 * Np
    y = rng.standard_normal(N)
    hyp = np.array([0.5, 0.7, 1.0, 1e-30])
    with SympFit("A", q, P, y, hyp[:3], hyp[3]) as f:
        with pytest.raises(np.linalg.LinAlgError):
            f.run()
        w, c = f.eig()
    Ky = oracle.build_K("A", q, P, q, P, hyp[:3]) + hyp[3] * np.eye(N)
    wr = np.linalg.eigvalsh(Ky)
    assert np.abs(w - wr).max() <= 1e-12 * np.abs(wr).max()
    got = ts.nll_chol(hyp, x, y, N)                        # neig = len(x)//2 = the positive half
    assert "Fallback to eig solver" in capsys.readouterr().out
    ref = _fallback_reference(Ky, y, len(x) // 2, len(x), hyp[3])
    assert np.isfinite(ref)
    assert got == pytest.approx(ref, rel=1e-7)
    # neig = len(x): every eigenpair, including the rounding-level ones -> the reference's value is
    # dominated by 1/w of numerically zero eigenvalues (or NaN from log of a negative one); only
    # finiteness class is comparable
    got_all = pp.nll_chol(hyp, x, y, N)
    assert "Fallback to eig solver" in capsys.readouterr().out
    ref_all = _fallback_reference(Ky, y, len(x), len(x), hyp[3])
    assert np.isnan(got_all) == np.isnan(ref_all)
    # the well-posed case never touches the fallback
    hyp_ok = np.array([0.5, 0.7, 1.0, 1e-2])
    a, nll, _ = oracle.fit("A", q, P, y, hyp_ok[:3], hyp_ok[3])
    assert pp.nll_chol(hyp_ok, x, y, N) == pytest.approx(nll, rel=1e-10)
    assert "Fallback" not in capsys.readouterr().out


def test_sections_replicas(oracle):
    """nphmap independent fits (05_tokamak/Split_SympGPR/main.py:96-112), one rank: every section."""
    from sympgpr_amd import sections
    rng = np.random.default_rng(13)
    nph, Np = 3, 50
    xtrain = np.vstack((rng.uniform(0, TWO_PI, (Np, nph)), rng.uniform(-3, 3, (Np, nph))))
    ztrain = rng.standard_normal((2 * Np, nph))
    hyp = np.array([[0.5, 0.8, 1.0], [0.6, 0.7, 0.9], [0.4, 0.9, 1.1]])
    loc = sections.fit_sections("A", xtrain, ztrain, hyp, 1e-3)
    alphas, nlls = sections.gather_sections(loc, nph)
    for m in range(nph):
        a, nll, _ = oracle.fit("A", xtrain[:Np, m], xtrain[Np:, m], ztrain[:, m], hyp[m], 1e-3)
        assert np.linalg.norm(alphas[m] - a) / np.linalg.norm(a) < 1e-10
        assert nlls[m] == pytest.approx(nll, rel=1e-11)
    assert sections.owned_sections(5, 1, 2) == [1, 3]


@pytest.mark.parametrize("fam", "ABCD")
def test_all_19_kernel_functions_vs_reference_fixture(golden_dir, fam):
    """The f2py `kernels` module, function for function, against the reference's compiled Fortran
    (tests/golden/scalars.json); the length-scale derivatives of all four families included (the sum
    kernel's, kernels_sum.f90:133-208, are served by the generated code of tools/gen_kernels.py)."""
    import json
    import os
    from sympgpr_amd import kernels, ops
    g = json.load(open(os.path.join(golden_dir, "scalars.json")))[fam]
    a = {k: np.array(v) for k, v in g["args"].items()}
    extra = (a["p"],) if fam == "D" else ()
    with ops.family_scope(fam):
        for name, ref in g["values"].items():
            ref = np.array(ref, dtype=float)
            got = np.array([getattr(kernels, name)(a["x_a"][i], a["y_a"][i], a["x_b"][i], a["y_b"][i], a["lx"][i],
                                                   a["ly"][i], *(e[i] for e in extra)) for i in range(len(ref))])
            # kernels_sum.f90:139 etc. write (y_a - y_b)^2 expanded (y_a**2 - 2 y_a y_b + y_b**2): for y_a ~ y_b
            # the REFERENCE value carries that cancellation (5e-16 absolute at the fixture's closest pair, where
            # the device value equals the exact one), so the sum kernel's dl-functions get an absolute 1e-14 on top
            slack = 1e-14 if (fam == "B" and "dl" in name) else 0.0
            assert np.all(np.abs(got - ref) <= 1e-13 * np.maximum(np.abs(ref), 1e-3) + slack), name
    assert len(g["values"]) == 19 and len(kernels.__all__) == 19
