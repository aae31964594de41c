"""The f2py object `sympgpr` of the reference (module sympgpr, python/05_tokamak/SympGPR/
sympgpr.f90), lower-case names and argument order as f2py exposes them, backed by
libsympgpr_hip.so.  `K`, `qmap`, `pmap` are intent(inout): float64, Fortran-contiguous."""
import numpy as np

from .. import ops
from ..predict import Predictor, solve_implicit_P


class _Sympgpr:
    @staticmethod
    def build_k(x, y, x0, y0, hyp, K):
        """sympgpr.f90:12-38"""
        ops.build_k(x, y, x0, y0, hyp, K)

    @staticmethod
    def buildkreg(x, y, x0, y0, hyp, K):
        """sympgpr.f90:40-60"""
        ops.buildkreg(x, y, x0, y0, hyp, K)

    @staticmethod
    def guessp(x, y, hypp, xtrainp, ytrainp, ztrainp, Kyinvp):
        """sympgpr.f90:62-73: dot(Kstar(1,:), matmul(Kyinvp, ztrainp))"""
        alpha = np.asarray(Kyinvp, dtype=np.float64) @ np.asarray(ztrainp, dtype=np.float64)
        pr = Predictor(ops.get_family(), xtrainp, ytrainp, hypp, alpha, reg=True)
        return float(pr(x, y)[0][0])

    @staticmethod
    def calcq(x, y, xtrain, ytrain, hyp, Kyinv, ztrain):
        """sympgpr.f90:75-86: dot(Kstar(2,:), matmul(Kyinv, ztrain))"""
        alpha = np.asarray(Kyinv, dtype=np.float64) @ np.asarray(ztrain, dtype=np.float64)
        pr = Predictor(ops.get_family(), xtrain, ytrain, hyp, alpha)
        return float(pr(x, y)[1][0])

    @staticmethod
    def calcp(x, y, hyp, hypp, xtrainp, ytrainp, ztrainp, Kyinvp, xtrain, ytrain, ztrain, Kyinv):
        """sympgpr.f90:88-125: root of pGP(x, P) - y + P from the regular-GP guess"""
        fam = ops.get_family()
        pr = Predictor(fam, xtrain, ytrain, hyp,
                       np.asarray(Kyinv, dtype=np.float64) @ np.asarray(ztrain, dtype=np.float64))
        prp = Predictor(fam, xtrainp, ytrainp, hypp,
                        np.asarray(Kyinvp, dtype=np.float64) @ np.asarray(ztrainp, dtype=np.float64), reg=True)
        return float(solve_implicit_P(pr, prp, x, y)[0])

    @staticmethod
    def applymap_tok(hyp, hypp, Q0map, P0map, xtrainp, ytrainp, ztrainp, Kyinvp, xtrain, ytrain, ztrain,
                     Kyinv, qmap, pmap, compute_r=None):
        """sympgpr.f90:128-177 with qmap, pmap [nm, Ntest, 1] in/out.  The tokamak loss test of the
        Fortran calls fieldlines.compute_r (out of scope physics, SURVEY #8); pass it as
        `compute_r(zk, r0)` to reproduce it.  Like the Fortran (whose `continue` is a no-op),
        a lost orbit is only ever recognised through NaN."""
        nm, Ntest = qmap.shape[0], qmap.shape[1]
        fam = ops.get_family()
        pr = Predictor(fam, xtrain, ytrain, hyp,
                       np.asarray(Kyinv, dtype=np.float64) @ np.asarray(ztrain, dtype=np.float64))
        prp = Predictor(fam, xtrainp, ytrainp, hypp,
                        np.asarray(Kyinvp, dtype=np.float64) @ np.asarray(ztrainp, dtype=np.float64), reg=True)
        pmap[0, :, 0] = P0map
        qmap[0, :, 0] = Q0map
        for i in range(nm - 1):
            ok = ~np.isnan(pmap[i, :, 0])
            if ok.any():
                pmap[i + 1, ok, 0] = solve_implicit_P(pr, prp, qmap[i, ok, 0], pmap[i, ok, 0])
                if compute_r is not None:
                    for k in np.nonzero(ok)[0]:
                        compute_r(np.array([pmap[i + 1, k, 0] * 1e-2, qmap[i, k, 0], 0.0]), 0.3)
            ok2 = ~np.isnan(pmap[i + 1, :, 0])
            if ok2.any():
                dq = pr(qmap[i, ok2, 0], pmap[i + 1, ok2, 0])[1]
                qmap[i + 1, ok2, 0] = np.mod(dq + qmap[i, ok2, 0], 2.0 * np.pi)


sympgpr = _Sympgpr()
