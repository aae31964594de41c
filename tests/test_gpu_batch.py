"""GPU parity of the batched small fits (sgpr_fit_batch: one workgroup per problem, one launch) against
the CPU oracle, problem by problem -- the reference's only batch axis (Split_SympGPR's nphmap sections
and CMA-ES populations over nll_chol, python/05_tokamak/Split_SympGPR/main.py:36-41,63-66,96-112)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HYP = {"A": [0.55, 0.8, 1.3], "B": [0.55, 0.8, 1.3], "C": [0.9, 0.8, 1.3], "D": [0.55, 0.8, 0.7, 1.3]}


def _problems(rng, B, n_pts, fam):
    x = rng.uniform(0, 2 * np.pi, (B, n_pts))
    y = rng.uniform(-3, 3, (B, n_pts))
    hyp = np.array(HYP[fam]) * rng.uniform(0.8, 1.25, (B, len(HYP[fam])))
    s2 = rng.uniform(1e-3, 1e-2, B)
    return x, y, hyp, s2


@pytest.mark.parametrize("fam", ["A", "B", "C", "D"])
@pytest.mark.parametrize("n_pts", [1, 7, 20, 40, 64, 65, 80, 100, 128])
def test_fit_batch_vs_oracle(oracle, fam, n_pts):
    """orders 2 ... 256: one leaf (n <= 128) and the two-leaf path (128 < n <= 256), ragged sizes"""
    from sympgpr_amd.fit import fit_batch
    rng = np.random.default_rng(100 * n_pts + ord(fam))
    B = 5
    x, y, hyp, s2 = _problems(rng, B, n_pts, fam)
    z = rng.standard_normal((B, 2 * n_pts))
    al, nll, info = fit_batch(fam, x, y, z, hyp, s2)
    assert np.all(info == 0)
    for b in range(B):
        if fam == "D":
            Ko = oracle.build_K(fam, x[b], y[b], x[b], y[b], hyp[b])
            Lf = oracle.cholesky(Ko + s2[b] * np.eye(2 * n_pts))
            a_o = oracle.solve_cholesky(Lf, z[b])
            nll_o = oracle.nll(Lf, z[b], a_o)
        else:
            a_o, nll_o, _ = oracle.fit(fam, x[b], y[b], z[b], hyp[b], s2[b])
        cond = np.linalg.cond(oracle.build_K(fam, x[b], y[b], x[b], y[b], hyp[b]) + s2[b] * np.eye(2 * n_pts))
        assert np.linalg.norm(al[b] - a_o) / np.linalg.norm(a_o) < max(1e-10, 50 * cond * 2.2e-16)
        assert nll[b] == pytest.approx(nll_o, rel=1e-10, abs=1e-10)


@pytest.mark.parametrize("n_pts", [30, 128, 200])
def test_fit_batch_reg_vs_oracle(oracle, n_pts):
    """the scalar-kernel GP (nll_chol_reg, functions/func.py:180-187): order n = n_pts"""
    from sympgpr_amd.fit import fit_batch
    rng = np.random.default_rng(n_pts)
    B = 4
    x, y, hyp, s2 = _problems(rng, B, n_pts, "A")
    z = rng.standard_normal((B, n_pts))
    al, nll, info = fit_batch("A", x, y, z, hyp, s2, reg=True)
    assert np.all(info == 0)
    for b in range(B):
        Ky = oracle.buildKreg("A", x[b], y[b], x[b], y[b], hyp[b]) + s2[b] * np.eye(n_pts)
        Lf = oracle.cholesky(Ky)
        a_o = oracle.solve_cholesky(Lf, z[b])
        assert np.linalg.norm(al[b] - a_o) / np.linalg.norm(a_o) < max(1e-10, 50 * np.linalg.cond(Ky) * 2.2e-16)
        assert nll[b] == pytest.approx(oracle.nll(Lf, z[b], a_o), rel=1e-10)


def test_fit_batch_many_problems_and_not_pd(oracle):
    """more problems than workgroups in flight; one of them indefinite (sig < 0): its info is the LAPACK
    index dpotrf reports, the others are unaffected"""
    import scipy.linalg
    from sympgpr_amd.fit import fit_batch
    rng = np.random.default_rng(5)
    B, n_pts = 1500, 20
    x, y, hyp, s2 = _problems(rng, B, n_pts, "A")
    z = rng.standard_normal((B, 2 * n_pts))
    bad = 777
    hyp[bad, -1] = -4.0
    s2[bad] = 0.5
    al, nll, info = fit_batch("A", x, y, z, hyp, s2)
    Kbad = oracle.build_K("A", x[bad], y[bad], x[bad], y[bad], hyp[bad]) + s2[bad] * np.eye(2 * n_pts)
    expect = scipy.linalg.lapack.dpotrf(Kbad, lower=1)[1]
    assert expect > 0 and info[bad] == expect and np.isnan(nll[bad])
    assert np.count_nonzero(info) == 1
    for b in (0, 1, 776, 778, 1499):
        a_o, nll_o, _ = oracle.fit("A", x[b], y[b], z[b], hyp[b], s2[b])
        assert np.linalg.norm(al[b] - a_o) / np.linalg.norm(a_o) < 1e-9
        assert nll[b] == pytest.approx(nll_o, rel=1e-10)


def test_nll_chol_batch_is_nll_chol_per_row(oracle):
    """a CMA-ES generation: nll_chol for a population of hyper-parameter vectors over the same data"""
    from sympgpr_amd import func
    rng = np.random.default_rng(9)
    Np = 40
    x = np.hstack((rng.uniform(0, 2 * np.pi, Np), rng.uniform(-3, 3, Np)))
    yv = rng.standard_normal(2 * Np)
    hyps = np.column_stack((rng.uniform(0.3, 1.0, 12), rng.uniform(0.4, 1.2, 12), rng.uniform(0.5, 2.0, 12),
                            rng.uniform(1e-3, 1e-2, 12)))
    hyps[5, 2] = -3.0                                  # not positive definite -> +inf
    hyps[5, 3] = 0.1
    func.set_family("A")
    got = func.nll_chol_batch(hyps, x, yv, 2 * Np)
    for b, h in enumerate(hyps):
        if b == 5:
            assert np.isinf(got[b])
            continue
        _, nll_o, _ = oracle.fit("A", x[:Np], x[Np:], yv, h[:-1], abs(h[-1]))
        assert got[b] == pytest.approx(nll_o, rel=1e-10)
        assert got[b] == pytest.approx(func.nll_chol(h, x, yv, 2 * Np), rel=1e-12)
    # the quirk of the drivers' first stage: the symplectic objective on an N x N matrix
    got_half = func.nll_chol_batch(hyps[:3], x, yv, Np)
    for b in range(3):
        assert got_half[b] == pytest.approx(func.nll_chol(hyps[b], x, yv, Np), rel=1e-12)
    # scalar-kernel objective
    got_reg = func.nll_chol_batch(hyps[:3], x, yv[:Np], Np, reg=True)
    for b in range(3):
        assert got_reg[b] == pytest.approx(func.nll_chol_reg(hyps[b], x, yv[:Np], Np), rel=1e-12)


def test_sections_use_one_launch(oracle):
    """Split_SympGPR's nphmap sections through sections.fit_sections: batched on the device, equal to the
    per-section oracle fits"""
    from sympgpr_amd import sections
    rng = np.random.default_rng(13)
    nph, Np = 4, 80                                   # order 160: the drivers' largest
    xtrain = np.vstack((rng.uniform(0, 2 * np.pi, (Np, nph)), rng.uniform(-3, 3, (Np, nph))))
    ztrain = rng.standard_normal((2 * Np, nph))
    hyp = np.column_stack((rng.uniform(0.4, 0.7, nph), rng.uniform(0.6, 0.9, nph), np.ones(nph)))
    loc = sections.fit_sections("A", xtrain, ztrain, hyp, 1e-3)
    assert sorted(loc) == list(range(nph))
    for m in range(nph):
        a_o, nll_o, _ = oracle.fit("A", xtrain[:Np, m], xtrain[Np:, m], ztrain[:, m], hyp[m], 1e-3)
        cond = 2e5
        assert np.linalg.norm(loc[m][0] - a_o) / np.linalg.norm(a_o) < 50 * cond * 2.2e-16
        assert loc[m][1] == pytest.approx(nll_o, rel=1e-10)


def test_fit_batch_edges(oracle):
    """empty batch; an order above the one-launch limit is refused by the C ABI and looped by nll_chol_batch;
    a NaN target propagates to that problem only"""
    import sympgpr_amd
    from sympgpr_amd import func
    from sympgpr_amd.fit import batch_max_order, fit_batch
    assert batch_max_order() == 2048
    al, nll, info = fit_batch("A", np.zeros((0, 5)), np.zeros((0, 5)), np.zeros((0, 10)), np.zeros((0, 3)), np.zeros(0))
    assert al.shape == (0, 10) and nll.shape == (0,) and info.shape == (0,)
    rng = np.random.default_rng(4)
    with pytest.raises(sympgpr_amd.SympGPRError):
        fit_batch("A", rng.random((2, 1025)), rng.random((2, 1025)), rng.random((2, 2050)), np.tile([0.5, 0.5, 1.0], (2, 1)), 1e-2)
    # order 2100 > 2048: nll_chol_batch falls back to one handle per row
    Np = 1050
    x = np.hstack((rng.uniform(0, 2 * np.pi, Np), rng.uniform(-3, 3, Np)))
    yv = rng.standard_normal(2 * Np)
    hyps = np.array([[0.4, 0.6, 1.0, 1e-2], [0.5, 0.7, 1.2, 2e-2]])
    func.set_family("A")
    got = func.nll_chol_batch(hyps, x, yv, 2 * Np)
    for b, h in enumerate(hyps):
        _, nll_o, _ = oracle.fit("A", x[:Np], x[Np:], yv, h[:-1], h[-1])
        assert got[b] == pytest.approx(nll_o, rel=1e-10)
    # NaN in one problem's targets
    B, n_pts = 3, 12
    X, Y = rng.uniform(0, 2 * np.pi, (B, n_pts)), rng.uniform(-3, 3, (B, n_pts))
    Z = rng.standard_normal((B, 2 * n_pts))
    Z[1, 5] = np.nan
    al, nll, info = fit_batch("A", X, Y, Z, np.tile([0.6, 0.8, 1.0], (B, 1)), 1e-2)
    assert np.all(info == 0) and np.isnan(nll[1]) and np.isfinite(nll[0]) and np.isfinite(nll[2])
    assert np.all(np.isfinite(al[0])) and np.all(np.isfinite(al[2]))


@pytest.mark.parametrize("fam,n_pts,reg", [("A", 192, False), ("C", 250, False), ("A", 512, False), ("B", 700, False),
                                           ("A", 1024, False), ("A", 300, True), ("C", 1024, True), ("A", 2048, True)])
def test_fit_batch_mid_orders_vs_oracle(oracle, fam, n_pts, reg):
    """orders 257 ... 2048 (three launches for the whole batch: build, W = ceil(n / 128) workgroups per problem running the
    leaf chain side by side, one workgroup per problem for the solves): every problem against the oracle's fit; ragged
    orders are padded with an identity block"""
    from sympgpr_amd.fit import fit_batch
    rng = np.random.default_rng(7 * n_pts + ord(fam) + reg)
    n = n_pts if reg else 2 * n_pts
    B = 3 if n > 1024 else 6
    x, y, hyp, s2 = _problems(rng, B, n_pts, fam)
    l = 2.0 * np.sqrt(12 * np.pi / n_pts)                  # lengths that keep cond(Ky) moderate at this density
    hyp[:, :2] = l * rng.uniform(0.8, 1.25, (B, 2))
    z = rng.standard_normal((B, n))
    al, nll, info = fit_batch(fam, x, y, z, hyp, s2, reg=reg)
    assert np.all(info == 0)
    for b in range(B):
        if reg:
            Ky = oracle.buildKreg(fam, x[b], y[b], x[b], y[b], hyp[b]) + s2[b] * np.eye(n)
            Lf = oracle.cholesky(Ky)
            a_o = oracle.solve_cholesky(Lf, z[b])
            nll_o = oracle.nll(Lf, z[b], a_o)
        else:
            a_o, nll_o, _ = oracle.fit(fam, x[b], y[b], z[b], hyp[b], s2[b], threads=4)
        assert np.linalg.norm(al[b] - a_o) / np.linalg.norm(a_o) < 1e-9, (b, n)
        assert nll[b] == pytest.approx(nll_o, rel=1e-10, abs=1e-10)


def test_fit_batch_mid_population_and_failures(oracle):
    """a CMA-ES generation at N = 512 points (order 1024): 24 hyper-parameter vectors over the same data in one call of
    func.nll_chol_batch; a vector whose Ky is not positive definite (sig < 0) comes back +inf (info > 0, the LAPACK-style
    index of the failing minor) and leaves its neighbours alone; more problems than the chip holds strips at once"""
    from sympgpr_amd import func
    from sympgpr_amd.fit import fit_batch
    rng = np.random.default_rng(99)
    Np = 512
    x = np.hstack((rng.uniform(0, 2 * np.pi, Np), rng.uniform(-3, 3, Np)))
    yv = rng.standard_normal(2 * Np)
    l = 2.0 * np.sqrt(12 * np.pi / Np)
    hyps = np.column_stack((l * rng.uniform(0.7, 1.4, 24), l * rng.uniform(0.7, 1.4, 24), rng.uniform(0.5, 2.0, 24),
                            rng.uniform(1e-3, 1e-2, 24)))
    hyps[7, 2] = -1.0                                    # Ky negative definite
    func.set_family("A")
    got = func.nll_chol_batch(hyps, x, yv, 2 * Np)
    assert np.isposinf(got[7])
    for b in (0, 6, 8, 23):
        _, nll_o, _ = oracle.fit("A", x[:Np], x[Np:], yv, hyps[b, :-1], hyps[b, -1], threads=4)
        assert got[b] == pytest.approx(nll_o, rel=1e-10)
    # 80 problems of order 512 = 320 strips > 256 CUs: tickets keep every wait pointed at a resident workgroup
    B, n_pts = 80, 256
    X, Y = rng.uniform(0, 2 * np.pi, (B, n_pts)), rng.uniform(-3, 3, (B, n_pts))
    Z = rng.standard_normal((B, 2 * n_pts))
    l = 2.0 * np.sqrt(12 * np.pi / n_pts)
    H = np.column_stack((l * rng.uniform(0.8, 1.25, B), l * rng.uniform(0.8, 1.25, B), np.ones(B)))
    H[11, 2] = -1.0
    al, nll, info = fit_batch("A", X, Y, Z, H, 1e-2)
    assert info[11] == 1 and np.isnan(nll[11]) and np.all(np.delete(info, 11) == 0)
    for b in (0, 10, 12, 79):
        a_o, nll_o, _ = oracle.fit("A", X[b], Y[b], Z[b], H[b], 1e-2, threads=4)
        assert np.linalg.norm(al[b] - a_o) / np.linalg.norm(a_o) < 1e-9
        assert nll[b] == pytest.approx(nll_o, rel=1e-10)


def test_fit_batch_mid_more_problems_than_one_chunk(oracle):
    """200 problems of order 2048 = 6.7 GB of images: more than the 6 GiB one chunk of the mid-size path holds, so the batch
    runs in two passes over the same scratch; problems on both sides of the chunk boundary against the oracle"""
    from sympgpr_amd.fit import fit_batch
    rng = np.random.default_rng(2048)
    B, n_pts = 200, 1024
    x, y = rng.uniform(0, 2 * np.pi, (B, n_pts)), rng.uniform(-3, 3, (B, n_pts))
    z = rng.standard_normal((B, 2 * n_pts))
    l = 2.0 * np.sqrt(12 * np.pi / n_pts)
    hyp = np.column_stack((l * rng.uniform(0.8, 1.25, B), l * rng.uniform(0.8, 1.25, B), rng.uniform(0.7, 1.4, B)))
    s2 = rng.uniform(2e-3, 1e-2, B)
    al, nll, info = fit_batch("A", x, y, z, hyp, s2)
    assert np.all(info == 0) and np.all(np.isfinite(nll))
    chunk = (6 << 30) // (2048 * 2048 * 8)
    assert 0 < chunk < B
    for b in (0, chunk - 1, chunk, B - 1):
        a_o, nll_o, _ = oracle.fit("A", x[b], y[b], z[b], hyp[b], s2[b], threads=4)
        assert np.linalg.norm(al[b] - a_o) / np.linalg.norm(a_o) < 1e-9, b
        assert nll[b] == pytest.approx(nll_o, rel=1e-10)
