"""python/04_standard_map/func.py -- the implicit map with kernel family A (kernels.f90) and the
explicit map with the sum kernel (kernels_expl_per_q_sq_p.f90 = family B).  The reference picks
one at import time by editing the `from kernels import *` line (func.py:15-16); here the
implicit functions use A and the *_expl functions use B."""
import numpy as np

from . import _common as _c
from ..fit import SympFit

FAMILY = "A"
FAMILY_EXPL = "B"
_c.python_surface(FAMILY, globals())


def nll_chol_reg(hyp, x, y, N):
    """func.py:86-93"""
    return _c.nll_fit(FAMILY, hyp, x, y, N, reg=True)


def nll_chol(hyp, x, y, N):
    """func.py:95-102"""
    return _c.nll_fit(FAMILY, hyp, x, y, N)


@_c.with_family(FAMILY_EXPL)
def build_K_expl(xin, x0in, hyp, K):
    """func.py:104-124: build_K with the explicit method's kernels; K in place."""
    from ..func import build_K as _bk
    _bk(xin, x0in, hyp, K)


def nll_expl(hyp, x, y, N, ind):
    """func.py:126-141: negative log-posterior of ONE diagonal block of the sum-kernel matrix --
    ind = 0: the qq block with hyp = (lq, sig, sig2n), ind = 1: the PP block with (lp, sig, sig2n).
    The reference builds the whole 2x2-block matrix with the other length set to 0 and slices; the
    block alone is built here (its entries do not depend on the other length)."""
    hyp = np.asarray(hyp, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    npts = N // 2
    if len(y) != npts:
        raise ValueError("nll_expl: y must hold one target per training point")
    l = (hyp[0], 1.0, hyp[1]) if ind == 0 else (1.0, hyp[0], hyp[1])
    with SympFit(FAMILY_EXPL, x[0:npts], x[npts:2 * npts], y, l, np.abs(hyp[-1]), block="qq" if ind == 0 else "PP") as f:
        return f.run().nll()


def guessP(x, y, hypp, xtrainp, ztrainp, Kyinvp, N):
    """func.py:143-148"""
    return _c.guessP_py(FAMILY, x, y, hypp, xtrainp, ztrainp, Kyinvp)


def calcQ(x, y, xtrain, l, Kyinv, ztrain):
    """func.py:150-156"""
    return _c.calcQ_py(FAMILY, x, y, xtrain, l, Kyinv, ztrain)


def Pnewton(P, x, y, l, xtrain, Kyinv, ztrain):
    """func.py:158-164"""
    return _c.Pnewton_py(FAMILY, P, x, y, l, xtrain, Kyinv, ztrain)


def calcP(x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv, Ntest):
    """func.py:166-172"""
    return _c.calcP_py(FAMILY, x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv)


def calcP_expl(x, y, l, xtrain, ztrain, Kyinv):
    """func.py:174-179: -pGP[0] + y"""
    r1, _ = _c.rows_py(FAMILY_EXPL, x, y, xtrain, l, Kyinv, ztrain)
    return -r1 + y


def applymap(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv):
    """func.py:218-254 -> (qmap, pmap, pdiff): implicit map, P and q both mod 2 pi, pdiff the
    unwrapped momentum."""
    return _c.run_map(_c.WRAP_Q | _c.WRAP_P, nm, Ntest, l, Q0map, P0map, xtrain, ztrain, Kyinv, hypp, xtrainp,
                      ztrainp, Kyinvp, want_pdiff=True, family=FAMILY)


def applymap_expl(nm, Ntest, l, Q0map, P0map, xtrain, ztrain, Kyinv):
    """func.py:256-285 -> (qmap, pmap, pdiff): explicit map, P mod 2 pi, q not wrapped."""
    return _c.run_map(_c.EXPLICIT | _c.WRAP_P, nm, Ntest, l, Q0map, P0map, xtrain, ztrain, Kyinv, want_pdiff=True,
                      family=FAMILY_EXPL)
