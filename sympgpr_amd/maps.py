"""Symplectic-map application: every time step of every orbit inside one device launch.

One recurrence, the variants the reference's drivers carry in their own func.py copies:

    applymap          functions/func.py:216-237           implicit, q mod 2 pi
    applymap_henon    functions/func.py:239-260           implicit, no wrap
    applymap          04_standard_map/func.py:218-254     implicit, q and P mod 2 pi, + pdiff
    applymap_expl     04_standard_map/func.py:256-285     explicit, P mod 2 pi, + pdiff
    applymap          01_pendulum/explicit/func_expl.py:113-128   explicit, q mod 2 pi
    applymap_tok      05_tokamak/SympGPR/func.py:182-211  implicit, q mod 2 pi, an orbit with P < 0 is lost (LOSS_NEGP)

alpha = Kyinv ztrain is formed once (the reference re-multiplies Kyinv inside every calcP / calcQ
call); a residual of the implicit equation is one block-wide reduction over the training points.
"""
import numpy as np

from . import _lib as L
from .ops import get_family

WRAP_Q, WRAP_P, EXPLICIT, LOSS_NEGP = L.MAP_WRAP_Q, L.MAP_WRAP_P, L.MAP_EXPLICIT, L.MAP_LOSS_NEGP


def run_map_alpha(mode, nm, Ntest, l, Q0map, P0map, xt, yt, alpha, hypp=None, xp=None, yp=None, alphap=None, family=None):
    """The same iteration from the posterior weights themselves (alpha = Ky^-1 ztrain, 2 N0; alphap, N0p) instead of the explicit
    inverses the drivers carry around: for training sets where Kyinv (8 (2 N0)^2 bytes) is not something to form."""
    lib = L.load_library()
    f = L.f64
    family = get_family() if family is None else family
    xt, yt, alpha, hyp = f(xt), f(yt), f(alpha), f(l)
    if mode & EXPLICIT:
        xp, yp, alphap, hp = f([]), f([]), f([]), f([])
    else:
        xp, yp, alphap, hp = f(xp), f(yp), f(alphap), f(hypp)
    Q0, P0 = f(np.broadcast_to(Q0map, (Ntest,))), f(np.broadcast_to(P0map, (Ntest,)))
    pmap, qmap = np.zeros([nm, Ntest]), np.zeros([nm, Ntest])
    L.check(lib.sgpr_applymap_host(L.family_id(family), int(mode), nm, Ntest, L.dptr(hyp), len(hyp), len(xt),
                                   L.dptr(xt), L.dptr(yt), L.dptr(alpha), L.dptr(hp), len(hp), len(xp), L.dptr(xp),
                                   L.dptr(yp), L.dptr(alphap), L.dptr(Q0), L.dptr(P0), L.dptr(qmap), L.dptr(pmap), None),
            "sgpr_applymap_host")
    return qmap, pmap


def run_map(mode, nm, Ntest, l, Q0map, P0map, xtrain, ztrain, Kyinv, hypp=None, xtrainp=None, ztrainp=None,
            Kyinvp=None, want_pdiff=False, family=None):
    """-> (qmap, pmap) or (qmap, pmap, pdiff), each [nm, Ntest].  `l` = (lx, ly, sig) of the
    symplectic GP, xtrain = (q || P), alpha = Kyinv @ ztrain; the *p arguments describe the
    regular GP that supplies the first guess of the implicit solve (unused with EXPLICIT)."""
    lib = L.load_library()
    f = L.f64
    family = get_family() if family is None else family
    Ntrain = len(xtrain) // 2
    xt, yt = f(xtrain[:Ntrain]), f(xtrain[Ntrain:2 * Ntrain])
    alpha = f(np.asarray(Kyinv, dtype=np.float64) @ np.asarray(ztrain, dtype=np.float64))
    hyp = f(l)
    if mode & EXPLICIT:
        Ntrainp, xp, yp, alphap, hp = 0, f([]), f([]), f([]), f([])
    else:
        Ntrainp = len(xtrainp) // 2
        xp, yp = f(xtrainp[:Ntrainp]), f(xtrainp[Ntrainp:2 * Ntrainp])
        alphap = f(np.asarray(Kyinvp, dtype=np.float64) @ np.asarray(ztrainp, dtype=np.float64))
        hp = f(hypp)
    Q0, P0 = f(np.broadcast_to(Q0map, (Ntest,))), f(np.broadcast_to(P0map, (Ntest,)))
    pmap = np.zeros([nm, Ntest])
    qmap = np.zeros([nm, Ntest])
    pdiff = np.zeros([nm, Ntest]) if want_pdiff else None
    L.check(lib.sgpr_applymap_host(L.family_id(family), int(mode), nm, Ntest, L.dptr(hyp), len(hyp), Ntrain,
                                   L.dptr(xt), L.dptr(yt), L.dptr(alpha), L.dptr(hp), len(hp), Ntrainp, L.dptr(xp),
                                   L.dptr(yp), L.dptr(alphap), L.dptr(Q0), L.dptr(P0), L.dptr(qmap), L.dptr(pmap),
                                   L.dptr(pdiff) if want_pdiff else None), "sgpr_applymap_host")
    return (qmap, pmap, pdiff) if want_pdiff else (qmap, pmap)
