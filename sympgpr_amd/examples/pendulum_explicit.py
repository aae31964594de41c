"""python/01_pendulum/explicit/func_expl.py -- kernel family B (sum kernel, kernels_sum.f90): the
explicit symplectic map, no implicit equation."""
from . import _common as _c

FAMILY = "B"
_c.python_surface(FAMILY, globals())
del globals()["buildKreg"], globals()["gpsolve"]      # func_expl.py has neither


def nll_chol(hyp, x, y):
    """func_expl.py:88-95: N = len(x)"""
    return _c.nll_fit(FAMILY, hyp, x, y, len(x))


def calcQ(x, y, xtrain, l, Kyinv, ztrain, Ntest):
    """func_expl.py:98-104 -> (qGP[1], qGP[0])"""
    r1, r2 = _c.rows_py(FAMILY, x, y, xtrain, l, Kyinv, ztrain)
    return r2, r1


def calcP(x, y, l, xtrain, ztrain, Kyinv, Ntest):
    """func_expl.py:106-111 -> -pGP[0]"""
    r1, _ = _c.rows_py(FAMILY, x, y, xtrain, l, Kyinv, ztrain)
    return -r1


def applymap(l, Q0map, P0map, xtrain, ztrain, Kyinv, Ntest, nm):
    """func_expl.py:113-128: p' = p - pGP[0](q, p), q' = (q + pGP[1](q, p')) mod 2 pi."""
    return _c.run_map(_c.EXPLICIT | _c.WRAP_Q, nm, Ntest, l, Q0map, P0map, xtrain, ztrain, Kyinv, family=FAMILY)
