"""python/05_tokamak/SympGPR/func.py -- kernel family A, the f2py-backed copy of the library."""
import numpy as np

from . import _common as _c
from ..func import calcP, calcQ, guessP  # noqa: F401  (func.py:49-52,170-180: the f2py wrappers)

FAMILY = "A"
_c.python_surface(FAMILY, globals())
for _n in ("calcP", "calcQ", "guessP"):
    globals()[_n] = _c.with_family(FAMILY)(globals()[_n])


def build_dK(xin, x0in, hyp):
    """func.py:54-118 -> [dK/dlx, dK/dly, K/sig]"""
    return _c.build_dK3(FAMILY, xin, x0in, hyp)


def nll_chol_reg(hyp, x, y, N):
    """func.py:134-141"""
    return _c.nll_fit(FAMILY, hyp, x, y, N, reg=True)


def nll_chol(hyp, x, y, N):
    """func.py:143-150"""
    return _c.nll_fit(FAMILY, hyp, x, y, N)


def nll_grad(hyp, x, y, N):
    """func.py:152-168 -> (nlp_val, nlp_grad[3])"""
    return _c.nll_grad3(FAMILY, hyp, x, y, N)


def applymap_tok(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv,
                 compute_r=None):
    """func.py:182-211.  Per step: the implicit P of all live orbits in one batched device solve,
    the flux-surface test, then q.  `compute_r(zk, r0)` is the reference's fieldlines.compute_r
    (tokamak physics, not part of this path); an orbit with compute_r > 0.5 or P < 0 is lost.
    Without it only the P < 0 test applies."""
    pr, prp = _c.predictor_pair(FAMILY, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv)
    pmap = np.zeros([nm, Ntest])
    qmap = np.zeros([nm, Ntest])
    pmap[0, :] = P0map
    qmap[0, :] = Q0map
    for i in range(0, nm - 1):
        pmap[i + 1, :] = np.nan
        ok = ~np.isnan(pmap[i, :])
        if ok.any():
            pmap[i + 1, ok] = _c.solve_implicit_P(pr, prp, qmap[i, ok], pmap[i, ok])
            for k in np.nonzero(ok)[0]:
                if np.isnan(pmap[i + 1, k]):
                    continue
                lost = pmap[i + 1, k] < 0.0
                if compute_r is not None and not lost:
                    lost = compute_r(np.array([pmap[i + 1, k] * 1e-2, qmap[i, k], 0]), 0.3) > 0.5
                if lost:
                    pmap[i + 1, k] = np.nan
        qmap[i + 1, :] = np.nan
        ok2 = ~np.isnan(pmap[i + 1, :])
        if ok2.any():
            dq = pr(qmap[i, ok2], pmap[i + 1, ok2])[1]
            qmap[i + 1, ok2] = np.mod(dq + qmap[i, ok2], 2.0 * np.pi)
    return qmap, pmap
