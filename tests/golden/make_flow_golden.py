"""Generates the flow fixtures of the other three drivers -- values recorded from the REFERENCE'S OWN Python running over
the reference's own compiled Fortran (tests/golden/_refmods.py; libraries from `make -C oracle ref`):

  driver_henon.npz          python/03_henon_heiles/func.py over kernels_sq.f90          (main.py:118-170)
  driver_standard_map.npz   python/04_standard_map/func.py over kernels.f90 (implicit, main.py:84-141) and over
                            kernels_expl_per_q_sq_p.f90 (explicit, main.py:143-180)
  driver_tokamak_split.npz  python/05_tokamak/Split_SympGPR/func.py over sympgpr.f90 + fieldlines.f90 + kernels.f90
                            (calc_fieldlines.py:22-62, main.py:25-112)

Build container only (needs /root/reference):   python tests/golden/make_flow_golden.py [henon|stdmap|split ...]
The .npz files hold data only (training sets, hyper-parameters, recorded values); nothing of the reference travels.

Settings that differ from the drivers, on purpose (as for driver_pendulum.npz): more training points (256 / 200 / 70
instead of 55 / 20 / 70) and a noise term of 1e-4 sig resp. 1e-5 instead of 1e-12 .. 1e-14, so that cond(K + sig2n I)
stays where the recorded numbers are reproducible to ~1e-9 by any correct solver; Halton points from scipy.stats.qmc
(ghalton is not in this image); training flows of the Henon-Heiles system integrated by scipy (the driver's VODE
extension is tokamak-free physics outside the path); L-BFGS-B where a driver offers it beside CMA-ES (cma is not in this
image: main.py:31-35,62-66 of Split_SympGPR)."""
import os
import sys
import time

import numpy as np
import scipy.linalg
from scipy.integrate import solve_ivp
from scipy.optimize import minimize
from scipy.stats import qmc

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refmods as R  # noqa: E402

REF = "/root/reference/python"
OUT = HERE


def halton(n, d):
    return qmc.Halton(d, scramble=False).random(n + 1)[1:]


# ---------------------------------------------------------------------------------------------------------------------
def henon():
    t0 = time.time()
    ker = R.kernels_module("libkernels_C.so", REF + "/03_henon_heiles/kernels_sq.f90", "kernels_sq")
    ref = R.load_reference(REF + "/03_henon_heiles/func.py", "ref_henon_func", {"kernels_sq": ker})
    N, lam, E_bound = 256, 1.0, 0.01
    # main.py:27-49: points of the section q1 = 0 below the energy bound, p1 from the energy
    allp = halton(6 * N, 2) * np.array([0.3, 0.3]) + np.array([-0.15, -0.15])
    ebound = 0.5 * allp[:, 1]**2 + 0.5 * allp[:, 0]**2 - lam / 3 * allp[:, 0]**3
    smp = allp[ebound < E_bound][:N]
    assert len(smp) == N
    qdot = np.sqrt(2 * E_bound - smp[:, 1]**2 - smp[:, 0]**2 + lam * 2 / 3 * smp[:, 0]**3)
    z0 = np.stack((np.zeros(N), smp[:, 0], qdot, smp[:, 1])).T            # (q1, q2, p1, p2)

    def rhs(t, z):       # henon.f90:18-31 with w1 = w2 = 1
        return np.array([z[2], z[3], -z[0] - 2 * lam * z[0] * z[1], -z[1] - lam * (z[0]**2 - z[1]**2)])
    cut = lambda t, z: z[0]
    cut.direction = 1.0                                                   # q1 = 0 with p1 > 0 (henon.f90:84)
    nxt = np.empty((N, 4))
    for i in range(N):
        r = solve_ivp(rhs, [0.0, 40.0], z0[i], method="DOP853", rtol=1e-12, atol=1e-14, events=cut)
        k = np.nonzero(r.t_events[0] > 1e-3)[0][0]
        nxt[i] = r.y_events[0][k]
    q, p, Q, P = z0[:, 1] * 1e2, z0[:, 3] * 1e2, nxt[:, 1] * 1e2, nxt[:, 3] * 1e2      # main.py:95-98
    xtrain, ztrain = np.hstack((q, P)), np.concatenate((p - P, Q - q))
    xtrainp, ztrainp = np.hstack((q, p)), P - p                                        # main.py:129-130
    out = dict(N=N, q=q, p=p, Q=Q, P=P)
    print("henon: training flows done %.0f s" % (time.time() - t0), flush=True)

    # step 1 (main.py:132-147): the objective the driver hands to L-BFGS-B -- nll_chol (its buildK argument is never
    # used, func.py:159-161), i.e. the symplectic matrix on the first N/2 points of the regular GP's data
    sigp = 2 * np.amax(np.abs(ztrainp))**2
    s2p = 1e-4 * sigp
    pts1 = np.array([[0.0, 0.0], [0.4, 0.7], [0.8, 0.5]])
    out["step1_log10l"] = pts1
    out["step1_nll"] = np.array([ref.nll_chol(np.hstack((10.0**h, sigp, [s2p])), xtrainp, ztrainp, N) for h in pts1])
    hypp = np.hstack((10.0**pts1[1], sigp))
    Kp = np.zeros((N, N))
    ref.buildKreg(xtrainp, xtrainp, hypp, Kp)
    Kyp = Kp + s2p * np.eye(N)
    Kyinvp = scipy.linalg.inv(Kyp)
    out.update(sigp=sigp, sig2n_p=s2p, hypp=hypp, Kp_rows=Kp[::16].copy(), alphap=Kyinvp @ ztrainp, cond_p=np.linalg.cond(Kyp),
               nll_reg=ref.nll_chol_reg(np.hstack((hypp, [s2p])), xtrainp, ztrainp, N))
    print("henon: step 1 done %.0f s" % (time.time() - t0), flush=True)

    # step 2 (main.py:152-163): nll_grad's value is the objective (its gradient is computed and dropped)
    sig = 2 * np.amax(np.abs(ztrain))**2
    s2 = 1e-4 * sig
    pts2 = np.array([[-1.0, -1.0], [0.5, 0.6], [0.9, 0.8]])
    vals, grads = [], []
    for h in pts2:
        v, g = ref.nll_grad(np.hstack((10.0**h, sig, [s2])), xtrain, ztrain, 2 * N)
        vals.append(v)
        grads.append(g)
    out.update(step2_log10l=pts2, step2_nll=np.array(vals), step2_grad=np.array(grads))
    print("henon: nll_grad points done %.0f s" % (time.time() - t0), flush=True)
    trace = []

    def obj(h):
        v = ref.nll_chol(np.hstack((10.0**h, sig, [s2])), xtrain, ztrain, 2 * N)    # = nll_grad(...)[0], func.py:175-177
        trace.append(np.hstack((h, v)))
        return v
    res = minimize(obj, np.array((-1.0, -1.0)), method="L-BFGS-B", tol=1e-8, bounds=((-2, 2), (-2, 2)))
    out.update(opt_x=res.x, opt_fun=res.fun, opt_nfev=res.nfev, opt_trace=np.array(trace))
    print("henon: L-BFGS-B x = %s fun = %.12g nfev = %d, %.0f s" % (res.x, res.fun, res.nfev, time.time() - t0), flush=True)

    hyp = np.hstack((np.abs(10.0**res.x), sig))                                         # main.py:164-172
    K = np.empty((2 * N, 2 * N))
    ref.build_K(xtrain, xtrain, hyp, K)
    Ky = K + s2 * np.eye(2 * N)
    Kyinv = scipy.linalg.inv(Ky)
    alpha = Kyinv @ ztrain
    out.update(sig=sig, sig2n=s2, hyp=hyp, K_rows=K[::32].copy(), alpha=alpha, Eftrain=K @ alpha, cond=np.linalg.cond(Ky))
    sub = np.hstack((q[:24], P[:24]))                  # func.py:70-134 (its third block only indexes right for equal sets)
    out.update(dK_sub=np.array(ref.build_dK(sub, sub, hyp)))
    print("henon: final build done, cond %.3g / %.3g, %.0f s" % (out["cond"], out["cond_p"], time.time() - t0), flush=True)

    Ntest, nm = 8, 6                                                                     # main.py:181-182
    rng = np.random.default_rng(3)
    Q0map = rng.permutation(np.linspace(-0.1, 0.1, Ntest)) * 1e2
    P0map = rng.permutation(np.linspace(-0.1, 0.1, Ntest)) * 1e2
    qmap, pmap = ref.applymap_henon(nm, Ntest, hyp, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv)
    out.update(Q0map=Q0map, P0map=P0map, qmap=qmap, pmap=pmap, nm=nm, Ntest=Ntest)
    np.savez_compressed(os.path.join(OUT, "driver_henon.npz"), **out)
    print("henon: written, %.0f s" % (time.time() - t0), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
def stdmap():
    t0 = time.time()
    kerA = R.kernels_module("libkernels_A.so", REF + "/04_standard_map/kernels.f90")
    kerB = R.kernels_module("libkernels_Bsq.so", REF + "/04_standard_map/kernels_expl_per_q_sq_p.f90")
    ref = R.load_reference(REF + "/04_standard_map/func.py", "ref_stdmap_func", {"kernels": kerA})
    # "use kernels_expl_per_q_sq_p.pyd for sum kernel in func.py" (main.py:144): the same file over the other kernels
    refx = R.load_reference(REF + "/04_standard_map/func.py", "ref_stdmap_func_expl", {"kernels": kerB})
    N, kk = 200, 2.0
    X0 = halton(N, 2) * 2 * np.pi                                         # main.py:42-53
    q, p = X0[:, 0].copy(), X0[:, 1].copy()
    P = p + kk * np.sin(q)
    Q = q + P
    zqtrain, zptrain = Q - q, p - P
    xtrain, ztrain = np.hstack((q, P)), np.concatenate((zptrain, zqtrain))
    xtrainp, ztrainp = np.hstack((q, p)), P - p                            # main.py:90-91
    out = dict(N=N, k=kk, q=q, p=p, Q=Q, P=P)

    sigp = 2 * np.amax(np.abs(ztrainp))**2                                 # step 1, main.py:93-106
    s2p = 1e-4 * sigp
    pts1 = np.array([[-1.0, -1.0], [-0.1, 0.3], [0.2, 0.1]])
    out["step1_log10l"] = pts1
    out["step1_nll"] = np.array([ref.nll_chol(np.hstack((10.0**h, sigp, [s2p])), xtrainp, ztrainp, N) for h in pts1])
    hypp = np.hstack((10.0**pts1[1], sigp))
    Kp = np.zeros((N, N))
    ref.buildKreg(xtrainp, xtrainp, hypp, Kp)
    Kyp = Kp + s2p * np.eye(N)
    Kyinvp = scipy.linalg.inv(Kyp)
    out.update(sigp=sigp, sig2n_p=s2p, hypp=hypp, Kp_rows=Kp[::16].copy(), alphap=Kyinvp @ ztrainp, cond_p=np.linalg.cond(Kyp))
    print("stdmap: step 1 done %.0f s" % (time.time() - t0), flush=True)

    sig = 2 * np.amax(np.abs(ztrain))**2                                   # step 2, main.py:110-121
    s2 = 1e-4 * sig
    pts2 = np.array([[0.0, -1.0], [-0.2, 0.4], [0.1, 0.7]])
    out["step2_log10l"] = pts2
    out["step2_nll"] = np.array([ref.nll_chol(np.hstack((10.0**h, sig, [s2])), xtrain, ztrain, 2 * N) for h in pts2])
    trace = []

    def obj(h):
        v = ref.nll_chol(np.hstack((10.0**h, sig, [s2])), xtrain, ztrain, 2 * N)
        trace.append(np.hstack((h, v)))
        return v
    res = minimize(obj, np.array((0.0, -1.0)), method="L-BFGS-B", tol=1e-8, bounds=((-2, 2), (-2, 2)))
    out.update(opt_x=res.x, opt_fun=res.fun, opt_nfev=res.nfev, opt_trace=np.array(trace))
    print("stdmap: L-BFGS-B x = %s fun = %.12g nfev = %d, %.0f s" % (res.x, res.fun, res.nfev, time.time() - t0), flush=True)

    hyp = np.hstack((np.abs(10.0**res.x), sig))                            # main.py:125-134
    K = np.empty((2 * N, 2 * N))
    ref.build_K(xtrain, xtrain, hyp, K)
    Ky = K + s2 * np.eye(2 * N)
    Kyinv = scipy.linalg.inv(Ky)
    alpha = Kyinv @ ztrain
    out.update(sig=sig, sig2n=s2, hyp=hyp, K_rows=K[::32].copy(), alpha=alpha, Eftrain=K @ alpha, cond=np.linalg.cond(Ky))
    Ntest, nm = 8, 6                                                       # main.py:137-140
    rng = np.random.default_rng(4)
    Q0map = rng.permutation(np.linspace(0.0, 2 * np.pi, Ntest))
    P0map = rng.permutation(np.linspace(0.0, 2 * np.pi, Ntest))
    qmap, pmap, pdiff = ref.applymap(nm, Ntest, hyp, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv)
    out.update(Q0map=Q0map, P0map=P0map, qmap=qmap, pmap=pmap, pdiff=pdiff, nm=nm, Ntest=Ntest)
    print("stdmap: implicit flow done, cond %.3g / %.3g, %.0f s" % (out["cond"], out["cond_p"], time.time() - t0), flush=True)

    # explicit method (main.py:143-180): one length per diagonal block of the sum kernel's matrix, trained separately
    s2x = 1e-4 * sig                                                       # (the driver: 1e-8)
    ptsx = np.array([1.0, 0.3, -0.2])
    out["expl_log10l"] = ptsx
    with np.errstate(all="ignore"):                                        # nll_expl builds the unused block with a length of 0
        out["expl_nll_q"] = np.array([refx.nll_expl(np.hstack((10.0**h, sig, [s2x])), xtrain, zptrain, 2 * N, 0) for h in ptsx])
        out["expl_nll_p"] = np.array([refx.nll_expl(np.hstack((10.0**h, sig, [s2x])), xtrain, zqtrain, 2 * N, 1) for h in ptsx])
        fq = lambda h: refx.nll_expl(np.hstack((10.0**np.ravel(h), sig, [s2x])), xtrain, zptrain, 2 * N, 0)
        fp = lambda h: refx.nll_expl(np.hstack((10.0**np.ravel(h), sig, [s2x])), xtrain, zqtrain, 2 * N, 1)
        res_lq = minimize(fq, np.array((1.0,)), method="L-BFGS-B")
        res_lp = minimize(fp, np.array((1.0,)), method="L-BFGS-B")
    print("stdmap: explicit lengths %s %s (nfev %d, %d), %.0f s" % (res_lq.x, res_lp.x, res_lq.nfev, res_lp.nfev,
                                                                    time.time() - t0), flush=True)
    lx = np.hstack((np.abs(10.0**res_lq.x), np.abs(10.0**res_lp.x)))
    hypx = np.hstack((lx, sig))
    Kx = np.empty((2 * N, 2 * N))
    refx.build_K(xtrain, xtrain, hypx, Kx)
    Kyx = Kx + s2x * np.eye(2 * N)
    Kyinvx = scipy.linalg.inv(Kyx)
    qx, px, pdx = refx.applymap_expl(nm, Ntest, hypx, Q0map, P0map, xtrain, ztrain, Kyinvx)
    out.update(expl_sig2n=s2x, expl_opt_lq=res_lq.x, expl_opt_lp=res_lp.x, expl_fun_q=res_lq.fun, expl_fun_p=res_lp.fun,
               expl_hyp=hypx, expl_K_rows=Kx[::32].copy(), expl_alpha=Kyinvx @ ztrain, expl_cond=np.linalg.cond(Kyx),
               expl_qmap=qx, expl_pmap=px, expl_pdiff=pdx)
    np.savez_compressed(os.path.join(OUT, "driver_standard_map.npz"), **out)
    print("stdmap: written, cond expl %.3g, %.0f s" % (out["expl_cond"], time.time() - t0), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
def split():
    t0 = time.time()
    ker = R.kernels_module("libkernels_A.so", REF + "/05_tokamak/Split_SympGPR/kernels.f90")
    flm, spm = R.fieldlines_module(), R.sympgpr_module()
    ref = R.load_reference(REF + "/05_tokamak/Split_SympGPR/func.py", "ref_split_func",
                           {"kernels": ker, "fieldlines": flm, "sympgpr": spm})
    fl = flm.fieldlines
    N, nphmap, nturn, nph, mod_m, mod_n, eps = 70, 4, 2, 100, -3, 2, 0.001          # calc_fieldlines.py:11-18
    X0 = halton(N, 3) * np.array([0.38, 2 * np.pi, 0]) + np.array([0.1, 0, 0])
    yint = np.zeros([nph * nturn + 1, 3, N])
    for ipart in range(N):                                                            # calc_fieldlines.py:22-40
        r0, th0, ph0 = X0[ipart]
        fl.init(nph=nph, am=mod_m, an=mod_n, aeps=eps, aphase=0.0, arlast=r0)
        z = np.zeros([nph * nturn + 1, 3])
        z[0, :] = [fl.ath(r0, th0, ph0), th0, 0.0]
        for kph in range(nph * nturn):
            z[kph + 1, :] = z[kph, :]
            fl.timestep(z[kph + 1, :])
        yint[:, :, ipart] = z
    ind = nph // nphmap
    q, Q, p, P = (np.zeros([N, nphmap]) for _ in range(4))                             # calc_fieldlines.py:47-62
    for i in range(nphmap):
        q[:, i] = yint[i * ind, 1]
        p[:, i] = yint[i * ind, 0] * 1e2
        Q[:, i] = yint[(i + 1) * ind, 1]
        P[:, i] = yint[(i + 1) * ind, 0] * 1e2
    ztrain = np.vstack((p - P, Q - q))
    xtrain = np.vstack((q, P))
    out = dict(N=N, nphmap=nphmap, q=q, p=p, Q=Q, P=P)
    print("split: field lines traced %.0f s" % (time.time() - t0), flush=True)

    # (main.py:18: 1e-14.)  The map iterates K* (Ky^-1 z): with 1e-8 the reference's OWN map moved by 6e-5 in q over these 24
    # steps when its inverse came from a Cholesky factor instead of scipy.linalg.inv's LU (1e-6: 4e-7, 1e-5: 3e-9)
    s2 = 1e-5
    Kyinvp, hypp = np.zeros((nphmap, N, N)), np.zeros((nphmap, 3))
    xtrainp, ztrainp = np.zeros((2 * N, nphmap)), np.zeros((N, nphmap))
    Kyinv, hyp = np.zeros((nphmap, 2 * N, 2 * N)), np.zeros((nphmap, 3))
    reg_pts = np.array([[-1.0, 0.0, 1.0], [-0.2, 0.5, 0.3], [0.0, 0.8, -0.5]])         # log10 (lq, lp, sigp), main.py:26-29
    gp_pts = np.array([[0.5, 2.5, 2.0], [0.8, 4.0, 0.5], [1.2, 6.0, 1.5]])             # (lq, lp, sig) itself, main.py:48-51
    reg_nll, gp_nll, condp, cond, alphap, alpha, Kp_rows, K_rows = [], [], [], [], [], [], [], []
    opt_reg = None
    for i in range(nphmap):
        xp, zp = np.hstack((q[:, i], p[:, i])), P[:, i] - p[:, i]                     # regGP, main.py:22-46
        reg_nll.append([ref.nll_chol_reg(np.hstack((10.0**h, [s2])), xp, zp, N) for h in reg_pts])
        if i == 0:
            tr = []

            def obj(h):
                v = ref.nll_chol_reg(np.hstack((10.0**h, [s2])), xp, zp, N)
                tr.append(np.hstack((h, v)))
                return v
            opt_reg = minimize(obj, np.array((-1.0, 0.0, 1.0)), method="L-BFGS-B")     # main.py:31-35 (opt == 'lbfgs')
            out.update(opt_reg_x=opt_reg.x, opt_reg_fun=opt_reg.fun, opt_reg_trace=np.array(tr))
            print("split: regular-GP L-BFGS-B x = %s fun = %.10g nfev = %d" % (opt_reg.x, opt_reg.fun, opt_reg.nfev), flush=True)
        hp = 10.0**(reg_pts[1] + 0.03 * i)
        Kp = np.zeros((N, N), order="F")
        ref.buildKreg(xp, xp, hp, Kp)
        Kyp = Kp + s2 * np.eye(N)
        Kyinvp[i], hypp[i], xtrainp[:, i], ztrainp[:, i] = scipy.linalg.inv(Kyp), hp, xp, zp
        condp.append(np.linalg.cond(Kyp))
        alphap.append(Kyinvp[i] @ zp)
        Kp_rows.append(Kp[::10].copy())
        gp_nll.append([ref.nll_chol(np.hstack((h, [s2])), xtrain[:, i], ztrain[:, i], 2 * N) for h in gp_pts])   # GP, main.py:53-83
        h = gp_pts[1] * (1.0 + 0.02 * i)
        K = np.empty((2 * N, 2 * N), order="F")
        ref.build_K(xtrain[:, i], xtrain[:, i], h, K)
        Ky = K + s2 * np.eye(2 * N)
        Kyinv[i], hyp[i] = scipy.linalg.inv(Ky), h
        cond.append(np.linalg.cond(Ky))
        alpha.append(Kyinv[i] @ ztrain[:, i])
        K_rows.append(K[::20].copy())
    # the `except:` branch of nll_chol (func.py:158-165): sig < 0 makes Ky negative definite beyond doubt
    with np.errstate(all="ignore"):
        out["gp_nll_negsig"] = ref.nll_chol(np.array([0.8, 4.0, -0.5, s2]), xtrain[:, 0], ztrain[:, 0], 2 * N)
    out.update(sig2n=s2, reg_log10hyp=reg_pts, reg_nll=np.array(reg_nll), gp_hyp=gp_pts, gp_nll=np.array(gp_nll),
               hypp=hypp, hyp=hyp, cond_p=np.array(condp), cond=np.array(cond), alphap=np.array(alphap), alpha=np.array(alpha),
               Kp_rows=np.array(Kp_rows), K_rows=np.array(K_rows))
    print("split: sections fitted, cond %.3g / %.3g, negsig nll %r, %.0f s" % (max(cond), max(condp), out["gp_nll_negsig"],
                                                                              time.time() - t0), flush=True)
    Ntest, nm = 8, 4 * 6 + 1                                                          # main.py:104-106
    rng = np.random.default_rng(5)
    r0t = rng.permutation(np.linspace(0.16, 0.31, Ntest))                             # calc_fieldlines.py:66-81
    th0t = rng.permutation(np.linspace(0.0, 2 * np.pi, Ntest))
    P0map = np.array([fl.ath(r, th, 0.0) for r, th in zip(r0t, th0t)]) * 1e2
    Q0map = th0t.copy()
    # one orbit that is LOST (compute_r > r_cut or P < 0, func.py:209-214): the first start near the outer flux surface that
    # the reference's own map throws out within its first nphmap steps
    lost = None
    for P0c in np.linspace(11.0, 16.0, 11):
        for Q0c in (0.0, 0.3, 0.6, 5.9):
            qc, pc = ref.applymap_tok(nphmap, 2 * nphmap, 1, np.array([Q0c]), np.array([P0c]), xtrainp, ztrainp, Kyinvp, hypp,
                                      xtrain, ztrain, Kyinv, hyp)
            if lost is None and np.isnan(pc[1:nphmap + 1, 0]).any():
                lost = (Q0c, P0c)
    assert lost is not None
    Q0map[5], P0map[5] = lost
    qmap, pmap = ref.applymap_tok(nphmap, nm, Ntest, Q0map, P0map, xtrainp, ztrainp, Kyinvp, hypp, xtrain, ztrain, Kyinv, hyp)
    out.update(Q0map=Q0map, P0map=P0map, qmap=qmap, pmap=pmap, nm=nm, Ntest=Ntest,
               compute_r_in=np.array([[0.02, 1.0, 0.0], [0.08, 2.5, 1.0], [0.135, 0.3, 0.0]]),
               compute_r_out=np.array([fl.compute_r(np.array(z), 0.3) for z in ([0.02, 1.0, 0.0], [0.08, 2.5, 1.0], [0.135, 0.3, 0.0])]))
    np.savez_compressed(os.path.join(OUT, "driver_tokamak_split.npz"), **out)
    print("split: written (%d lost entries), %.0f s" % (int(np.isnan(pmap).sum()), time.time() - t0), flush=True)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["henon", "stdmap", "split"]
    for name in todo:
        {"henon": henon, "stdmap": stdmap, "split": split}[name]()
