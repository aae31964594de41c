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
    """The five-argument diagnostics of 02_pert_pendulum/func.py:248-258 (host arithmetic): the reference orbit's
    first point is ysint[:, k, 1] there."""
    first = np.stack((np.asarray(qmap)[1, :Ntest], np.asarray(pmap)[1, :Ntest]))
    gd = np.mean((first - np.asarray(ysint)[:, :Ntest, 1]) ** 2, axis=0)
    Hk = np.asarray(H)[:, :Ntest]
    return np.std(Hk, axis=0) / np.mean(Hk, axis=0), gd, np.std(gd)
