"""python/02_pert_pendulum/func.py -- kernel family A; the library copy python/functions/func.py
plus the eigen fallback in nll_chol and a five-argument quality()."""
import numpy as np

from . import _common as _c
from ..func import (applymap, build_dK, build_dKreg, calcP, calcQ, guessP, nll_chol_reg, nll_grad,  # noqa: F401
                    nll_grad_reg)

FAMILY = "A"
_c.python_surface(FAMILY, globals())
for _n in ("applymap", "build_dK", "build_dKreg", "calcP", "calcQ", "guessP", "nll_chol_reg", "nll_grad", "nll_grad_reg"):
    globals()[_n] = _c.with_family(FAMILY)(globals()[_n])


def nll_chol(hyp, x, y, N):
    """func.py:189-204: eigen fallback with neig = len(x)."""
    return _c.nll_fit(FAMILY, hyp, x, y, N, neig=len(x))


def quality(qmap, pmap, H, ysint, Ntest):
    """func.py:248-258 (host arithmetic)."""
    gd = np.zeros([Ntest])
    for lk in range(0, Ntest):
        d = np.array([qmap[1, lk], pmap[1, lk]]) - np.asarray(ysint)[:, lk, 1]
        gd[lk] = np.mean(d * d)
    stdgd = np.std(gd[:])
    Eosc = np.zeros([Ntest])
    for lk in range(0, Ntest):
        Eosc[lk] = np.std(H[:, lk]) / np.mean(H[:, lk])
    return Eosc, gd, stdgd
