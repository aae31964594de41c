"""python/01_pendulum/implicit/func.py -- kernel family A (periodic x squared-exponential)."""
import numpy as np

from . import _common as _c
from ..func import quality  # noqa: F401  (func.py:187-198, same as functions/func.py:262-272)

FAMILY = "A"
_c.python_surface(FAMILY, globals())


def nll_chol(hyp, x, y, N):
    """func.py:99-114: with the eigen fallback when Ky is not positive definite (neig = len(x))."""
    return _c.nll_fit(FAMILY, hyp, x, y, N, neig=len(x))


def guessP(x, y, hypp, xtrainp, ztrainp, Kyinvp, N):
    """func.py:119-124 (N is unused there too)."""
    return _c.guessP_py(FAMILY, x, y, hypp, xtrainp, ztrainp, Kyinvp)


def calcQ(x, y, xtrain, l, Kyinv, ztrain):
    """func.py:126-132"""
    return _c.calcQ_py(FAMILY, x, y, xtrain, l, Kyinv, ztrain)


def Pnewton(P, x, y, l, xtrain, Kyinv, ztrain):
    """func.py:134-139"""
    return _c.Pnewton_py(FAMILY, P, x, y, l, xtrain, Kyinv, ztrain)


def calcP(x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv, Ntest):
    """func.py:141-147"""
    return _c.calcP_py(FAMILY, x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv)


def applymap(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv):
    """func.py:149-170: implicit map, q mod 2 pi; all steps of all orbits in one launch."""
    return _c.run_map(_c.WRAP_Q, nm, Ntest, l, Q0map, P0map, xtrain, ztrain, Kyinv, hypp, xtrainp, ztrainp, Kyinvp,
                      family=FAMILY)
