"""python/01_pendulum/implicit_period_unknown/func.py -- kernel family D (the period of the
q-kernel is a third length parameter: l = (lx, ly, p), hyp = (lx, ly, p, sig))."""
from . import _common as _c
from ..func import quality  # noqa: F401  (func.py:178-189)

FAMILY = "D"
_c.python_surface(FAMILY, globals())


def nll_chol(hyp, x, y, N, buildK=None):
    """func.py:98-105 (the buildK argument is ignored there as well: the body calls build_K)."""
    return _c.nll_fit(FAMILY, hyp, x, y, N)


def guessP(x, y, hypp, xtrainp, ztrainp, Kyinvp, N):
    """func.py:110-115"""
    return _c.guessP_py(FAMILY, x, y, hypp, xtrainp, ztrainp, Kyinvp)


def calcQ(x, y, xtrain, l, Kyinv, ztrain):
    """func.py:117-123"""
    return _c.calcQ_py(FAMILY, x, y, xtrain, l, Kyinv, ztrain)


def Pnewton(P, x, y, l, xtrain, Kyinv, ztrain):
    """func.py:125-130"""
    return _c.Pnewton_py(FAMILY, P, x, y, l, xtrain, Kyinv, ztrain)


def calcP(x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv, Ntest):
    """func.py:132-138"""
    return _c.calcP_py(FAMILY, x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv)


def applymap(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv):
    """func.py:140-161: implicit map, q mod 2 pi."""
    return _c.run_map(_c.WRAP_Q, nm, Ntest, l, Q0map, P0map, xtrain, ztrain, Kyinv, hypp, xtrainp, ztrainp, Kyinvp,
                      family=FAMILY)
