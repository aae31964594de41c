"""Generates tests/golden/driver_pendulum.npz: inputs and outputs of the sequence the reference's
pendulum driver performs (python/01_pendulum/implicit/main.py:116-175), produced by the REFERENCE's
own library python/01_pendulum/implicit/func.py running over the reference's own generated Fortran
kernels (oracle/_ref/libkernels_A.so, built from /root/reference by `make -C oracle ref`).

Run in the build container only (needs /root/reference):   python tests/golden/make_driver_golden.py
Nothing of the reference travels: the .npz holds data only (training set, hyper-parameters, values).

Sequence recorded (N = 512 training points = BASELINE config `01_pendulum`):
  step 1  nll_chol(hyp, xtrainp, ztrainp, N) at a few hyper-parameter points      main.py:126-130
          [the symplectic objective on the regular GP's data: build_K slices N/2 "points", SURVEY 3.5]
          buildKreg(xtrainp, xtrainp, hypp, Kp); Kyinvp = inv(Kp + sig2n I)       main.py:136-138
  step 2  nll_chol(hyp, xtrain, ztrain, 2N) at a few points, and the L-BFGS-B run   main.py:143-149
          over log10(l) from the driver's start (-1, -1) with its bounds
  final   build_K(xtrain, xtrain, hyp, K); Kyinv = inv(K + sig2n I);               main.py:156-165
          Kyinv ztrain; K Kyinv ztrain
  map     applymap(nm, Ntest, ...) for 10 orbits, 5 steps                            main.py:170-172
Settings that differ from the driver, on purpose: N = 512 instead of 20 and sig2_n = 1e-3 sig instead of
1e-16, so that cond(K + sig2n I) stays near 1e6 and the recorded values are reproducible to ~1e-9 by any
correct solver (the driver's own setting is numerically singular at this N: SURVEY 7, hard parts).
"""
import ctypes as C
import importlib.util
import os
import sys
import time
import types

import numpy as np
import scipy.linalg
from scipy.integrate import solve_ivp
from scipy.optimize import minimize
from scipy.stats import qmc

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_FUNC = "/root/reference/python/01_pendulum/implicit/func.py"
KLIB = os.path.join(ROOT, "oracle", "_ref", "libkernels_A.so")


def kernels_module():
    """the f2py module `kernels` the reference's func.py star-imports (func.py:15), over the reference's
    compiled kernels.f90: name_num(x_a, y_a, x_b, y_b, lx, ly) -> float"""
    lib = C.CDLL(KLIB)
    mod = types.ModuleType("kernels")
    names = ["kern_num", "dkdx_num", "dkdy_num", "dkdx0_num", "dkdy0_num", "d2kdxdx0_num", "d2kdydy0_num",
             "d2kdxdy0_num", "d3kdxdx0dy0_num", "d3kdydy0dy0_num", "d3kdxdy0dy0_num", "dkdlx_num", "dkdly_num",
             "d3kdxdx0dlx_num", "d3kdydy0dlx_num", "d3kdxdy0dlx_num", "d3kdxdx0dly_num", "d3kdydy0dly_num",
             "d3kdxdy0dly_num"]
    dp = C.POINTER(C.c_double)

    def bind(name):
        f = getattr(lib, name + "_")
        f.restype = C.c_double
        f.argtypes = [dp] * 6

        def call(xa, ya, xb, yb, lx, ly):
            a = [C.c_double(float(np.ravel(v)[0])) for v in (xa, ya, xb, yb, lx, ly)]
            return f(*[C.byref(v) for v in a])
        call.__name__ = name
        return call
    for n in names:
        setattr(mod, n, bind(n))
    mod.__all__ = names
    return mod


def main():
    sys.modules["kernels"] = kernels_module()
    spec = importlib.util.spec_from_file_location("ref_pendulum_func", REF_FUNC)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    N, Nm, dtsymp = 512, 200, 0.001
    qmin, qmax, pmin, pmax = 0.0, 2 * np.pi, -3.0, 3.0
    samples = qmc.Halton(2, scramble=False).random(N + 1)[1:] * np.array([qmax - qmin, pmax - pmin]) + np.array([qmin, pmin])
    q, p = samples[:, 0].copy(), samples[:, 1].copy()
    t = np.linspace(0.0, dtsymp * Nm, Nm)
    Q, P = np.empty(N), np.empty(N)
    for i in range(N):       # the flow the driver integrates (main.py:24-37)
        r = solve_ivp(lambda tt, y: np.array([y[1], -np.sin(y[0] + np.pi)]), [t[0], t[-1]], [q[i], p[i]], t_eval=t,
                      method="LSODA", rtol=1e-13, atol=1e-16)
        Q[i], P[i] = r.y[0, -1], r.y[1, -1]
    xtrain = np.hstack((q, P))
    ztrain = np.concatenate((p - P, Q - q))
    xtrainp = np.hstack((q, p))
    ztrainp = P.copy()
    out = dict(N=N, q=q, p=p, Q=Q, P=P)

    # ---- step 1
    sigp = 2 * np.amax(np.abs(ztrainp))**2
    s2p = 1e-3 * sigp
    pts1 = np.array([[0.0, 0.0], [-0.3, 0.2], [0.25, -0.15]])
    t0 = time.time()
    out["step1_log10l"] = pts1
    out["step1_nll"] = np.array([ref.nll_chol(np.hstack((10.0**h, sigp, [s2p])), xtrainp, ztrainp, N) for h in pts1])
    lp = 10.0**pts1[1]
    hypp = np.hstack((lp, sigp))
    Kp = np.zeros((N, N), order="F")
    ref.buildKreg(xtrainp, xtrainp, hypp, Kp)
    Kyinvp = scipy.linalg.inv(Kp + s2p * np.eye(N))
    out.update(sigp=sigp, sig2n_p=s2p, hypp=hypp, Kp_rows=Kp[::32].copy(), alphap=Kyinvp @ ztrainp,
               cond_p=np.linalg.cond(Kp + s2p * np.eye(N)))
    print("step 1 done %.0f s" % (time.time() - t0), flush=True)

    # ---- step 2
    sig = 2 * np.amax(np.abs(ztrain))**2
    s2 = 1e-3 * sig
    pts2 = np.array([[-1.0, -1.0], [-0.2, 0.1], [0.1, 0.3], [-0.5, -0.3]])
    out["step2_log10l"] = pts2
    out["step2_nll"] = np.array([ref.nll_chol(np.hstack((10.0**h, sig, [s2])), xtrain, ztrain, 2 * N) for h in pts2])
    print("step 2 points done %.0f s" % (time.time() - t0), flush=True)
    trace = []

    def obj(h):
        v = ref.nll_chol(np.hstack((10.0**h, sig, [s2])), xtrain, ztrain, 2 * N)
        trace.append(np.hstack((h, v)))
        return v
    res = minimize(obj, np.array((-1.0, -1.0)), method="L-BFGS-B", bounds=((-10, 1), (-10, 1)))
    out.update(opt_x=res.x, opt_fun=res.fun, opt_nfev=res.nfev, opt_nit=res.nit, opt_trace=np.array(trace))
    print("L-BFGS-B done: x = %s fun = %.12g nfev = %d, %.0f s" % (res.x, res.fun, res.nfev, time.time() - t0), flush=True)

    # ---- final matrices
    l = np.abs(10.0**res.x)
    hyp = np.hstack((l, sig))
    K = np.empty((2 * N, 2 * N), order="F")
    ref.build_K(xtrain, xtrain, hyp, K)
    Ky = K + s2 * np.eye(2 * N)
    Kyinv = scipy.linalg.inv(Ky)
    alpha = Kyinv @ ztrain
    out.update(sig=sig, sig2n=s2, hyp=hyp, K_rows=K[::64].copy(), alpha=alpha, Eftrain=K @ alpha, cond=np.linalg.cond(Ky))
    print("final build done, cond %.3g, %.0f s" % (out["cond"], time.time() - t0), flush=True)

    # ---- map application: 10 orbits, 5 steps
    Ntest, nm = 10, 6
    rng = np.random.default_rng(1)
    Q0map = rng.permutation(np.linspace(np.pi - 2.8, np.pi + 1.5, Ntest))
    P0map = rng.permutation(np.linspace(-2.3, 1.8, Ntest))
    qmap, pmap = ref.applymap(nm, Ntest, hyp, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv)
    out.update(Q0map=Q0map, P0map=P0map, qmap=qmap, pmap=pmap, nm=nm, Ntest=Ntest)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "driver_pendulum.npz"), **out)
    print("written, %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
