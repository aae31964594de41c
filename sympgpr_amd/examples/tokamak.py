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
                 compute_r=None, steps_per_launch=None):
    """func.py:182-211.  Every time step on the device: the implicit P of an orbit, the P < 0 half of the loss test
    (SGPR_MAP_LOSS_NEGP) and the q update run inside the map kernel -- without a callback the whole map is ONE launch, as
    `applymap` is.  `compute_r(zk, r0)` is the reference's fieldlines.compute_r (tokamak physics, not part of this path): an
    orbit with compute_r > 0.5 is lost as well.  With it the map runs `steps_per_launch` steps (default 1) per launch and the
    callback is applied to the new rows on the host, step by step and orbit by orbit in the reference's order; orbits are
    independent, so what a chunk computed for an orbit beyond the step at which the callback lost it is simply dropped."""
    from .. import maps
    Ntrain, Ntrainp = len(xtrain) // 2, len(xtrainp) // 2
    xt, yt = np.asarray(xtrain[:Ntrain], dtype=np.float64), np.asarray(xtrain[Ntrain:2 * Ntrain], dtype=np.float64)
    xp, yp = np.asarray(xtrainp[:Ntrainp], dtype=np.float64), np.asarray(xtrainp[Ntrainp:2 * Ntrainp], dtype=np.float64)
    alpha = np.asarray(Kyinv, dtype=np.float64) @ np.asarray(ztrain, dtype=np.float64)          # once, not per step
    alphap = np.asarray(Kyinvp, dtype=np.float64) @ np.asarray(ztrainp, dtype=np.float64)
    mode = maps.WRAP_Q | maps.LOSS_NEGP
    run = lambda steps, n, Q0, P0: maps.run_map_alpha(mode, steps + 1, n, l, Q0, P0, xt, yt, alpha, hypp, xp, yp, alphap,
                                                      family=FAMILY)
    if compute_r is None:
        return run(nm - 1, Ntest, Q0map, P0map) if nm > 1 else (np.array([np.broadcast_to(Q0map, (Ntest,))], dtype=float),
                                                                 np.array([np.broadcast_to(P0map, (Ntest,))], dtype=float))
    k = max(1, int(steps_per_launch or 1))
    pmap = np.full([nm, Ntest], np.nan)
    qmap = np.full([nm, Ntest], np.nan)
    pmap[0, :] = P0map
    qmap[0, :] = Q0map
    i = 0
    while i < nm - 1:
        kk = min(k, nm - 1 - i)
        idx = np.nonzero(~np.isnan(pmap[i, :]))[0]
        if len(idx):
            qq, pp = run(kk, len(idx), qmap[i, idx], pmap[i, idx])
            alive = np.ones(len(idx), dtype=bool)
            for s in range(1, kk + 1):
                for j, orbit in enumerate(idx):
                    if not alive[j]:
                        continue
                    if np.isnan(pp[s, j]):                  # the solve failed or P < 0: lost inside the kernel
                        alive[j] = False
                        continue
                    if compute_r(np.array([pp[s, j] * 1e-2, qq[s - 1, j], 0]), 0.3) > 0.5:
                        alive[j] = False
                        continue
                    pmap[i + s, orbit], qmap[i + s, orbit] = pp[s, j], qq[s, j]
        i += kk
    return qmap, pmap
