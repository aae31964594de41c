"""Generates tests/golden/applymap_tok.npz: inputs and outputs of the reference's Fortran
`sympgpr.applymap_tok` (python/05_tokamak/SympGPR/sympgpr.f90:128-177) in the form its f2py wrapper is
called (test_sympgpr.py:83-90: qmap, pmap F-ordered [nm, Ntest, 1], in/out), produced by the reference's
own compiled module (oracle/_ref/libsympgpr_ref_A.so via the bind(C) shim oracle/ref_shim.f90).
Build container only:  python tests/golden/make_tok_golden.py
Columns: 0..3 ordinary orbits; 4: P0 = NaN and NaN everywhere in the incoming pmap (stays lost, qmap keeps
its incoming values); 5: P0 = NaN but a finite incoming pmap(2): the Fortran's `continue` leaves that
value in place, computes q from it and carries on -- the mirror has to do the same."""
import ctypes as C
import os
import sys

import numpy as np
import scipy.linalg

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Ref  # noqa: E402

dp = C.POINTER(C.c_double)
p_ = lambda a: a.ctypes.data_as(dp)


def main():
    ref = Ref()
    lib = ref.mod["A"]
    lib.ref_applymap_tok.restype = None
    lib.ref_applymap_tok.argtypes = [C.c_int, C.c_int, dp, dp, dp, dp, C.c_int, dp, dp, dp, dp, C.c_int, dp, dp, dp, dp, dp, dp]
    rng = np.random.default_rng(2026)
    Nt, Np_ = 24, 24
    # a smooth map to learn: the standard map with small kick, q in [0, 2 pi), p in [0.1, 0.9] (tokamak-like p > 0)
    q = rng.uniform(0, 2 * np.pi, Nt)
    p = rng.uniform(0.15, 0.85, Nt)
    kk = 0.05
    P = p - kk * np.sin(q)
    Q = q + P
    xtrain, ytrain = q.copy(), P.copy()
    ztrain = np.concatenate((p - P, Q - q))
    xtrainp, ytrainp, ztrainp = q.copy(), p.copy(), P.copy()
    hyp = np.array([1.1, 0.9, 2.0 * np.max(np.abs(ztrain))**2])
    hypp = np.array([1.2, 1.0, 2.0 * np.max(np.abs(ztrainp))**2])
    s2 = 1e-4   # keeps |Kyinv| ~ 1e4: Kyinv @ ztrain is then reproducible to ~1e-12 in any summation order
    K = ref.build_K("A", xtrain, ytrain, xtrain, ytrain, hyp)
    Kyinv = np.asfortranarray(scipy.linalg.inv(K + s2 * np.eye(2 * Nt)))
    Kp = ref.buildKreg("A", xtrainp, ytrainp, xtrainp, ytrainp, hypp)
    Kyinvp = np.asfortranarray(scipy.linalg.inv(Kp + s2 * np.eye(Np_)))
    nm, Ntest = 5, 6
    Q0 = rng.uniform(0.5, 5.5, Ntest)
    P0 = rng.uniform(0.3, 0.7, Ntest)
    P0[4] = np.nan
    P0[5] = np.nan
    qmap = np.asfortranarray(rng.uniform(1, 2, (nm, Ntest, 1)))
    pmap = np.asfortranarray(rng.uniform(0.3, 0.6, (nm, Ntest, 1)))
    pmap[:, 4, 0] = np.nan
    pmap[1, 5, 0] = 0.45
    qin, pin = qmap.copy(order="F"), pmap.copy(order="F")
    lib.ref_applymap_tok(nm, Ntest, p_(hyp), p_(hypp), p_(Q0), p_(P0), Np_, p_(xtrainp), p_(ytrainp), p_(ztrainp),
                         p_(Kyinvp), Nt, p_(xtrain), p_(ytrain), p_(ztrain), p_(Kyinv), p_(qmap), p_(pmap))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "applymap_tok.npz"), hyp=hyp, hypp=hypp, Q0map=Q0, P0map=P0,
                        xtrainp=xtrainp, ytrainp=ytrainp, ztrainp=ztrainp, Kyinvp=Kyinvp, xtrain=xtrain, ytrain=ytrain,
                        ztrain=ztrain, Kyinv=Kyinv, qmap_in=qin, pmap_in=pin, qmap_out=qmap, pmap_out=pmap)
    print(np.round(pmap[:, :, 0], 4))
    print(np.round(qmap[:, :, 0], 4))


if __name__ == "__main__":
    main()
