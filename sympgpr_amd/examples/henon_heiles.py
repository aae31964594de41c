"""python/03_henon_heiles/func.py -- kernel family C (squared-exponential in q and P,
kernels_sq.f90)."""
from . import _common as _c
from ..func import quality  # noqa: F401  (func.py:249-260)

FAMILY = "C"
_c.python_surface(FAMILY, globals())


def build_dK(xin, x0in, hyp):
    """func.py:70-134 -> [dK/dlx, dK/dly, K/sig]"""
    return _c.build_dK3(FAMILY, xin, x0in, hyp)


def nll_chol_reg(hyp, x, y, N):
    """func.py:150-157"""
    return _c.nll_fit(FAMILY, hyp, x, y, N, reg=True)


def nll_chol(hyp, x, y, N, buildK=None):
    """func.py:159-166"""
    return _c.nll_fit(FAMILY, hyp, x, y, N)


def nll_grad(hyp, x, y, N):
    """func.py:168-192 -> (nlp_val, nlp_grad[3])"""
    return _c.nll_grad3(FAMILY, hyp, x, y, N)


def guessP(x, y, hypp, xtrainp, ztrainp, Kyinvp, N):
    """func.py:194-199"""
    return _c.guessP_py(FAMILY, x, y, hypp, xtrainp, ztrainp, Kyinvp)


def calcQ(x, y, xtrain, l, Kyinv, ztrain):
    """func.py:201-207"""
    return _c.calcQ_py(FAMILY, x, y, xtrain, l, Kyinv, ztrain)


def Pnewton(P, x, y, l, xtrain, Kyinv, ztrain):
    """func.py:209-215"""
    return _c.Pnewton_py(FAMILY, P, x, y, l, xtrain, Kyinv, ztrain)


def calcP(x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv, Ntest):
    """func.py:217-223"""
    return _c.calcP_py(FAMILY, x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv)


def applymap_henon(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv):
    """func.py:225-247: implicit map, q not wrapped."""
    return _c.run_map(0, nm, Ntest, l, Q0map, P0map, xtrain, ztrain, Kyinv, hypp, xtrainp, ztrainp, Kyinvp,
                      family=FAMILY)
