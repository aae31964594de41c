"""python/05_tokamak/Split_SympGPR/func.py -- kernel family A; nphmap independent GPs, one per
toroidal section, applied in turn (the sections are independent fits: "replicas only" across
GPUs, see sympgpr_amd/sections.py)."""
import numpy as np

from . import _common as _c
from ..func import calcP, calcQ, guessP  # noqa: F401  (func.py:167-182)

FAMILY = "A"
_c.python_surface(FAMILY, globals())
for _n in ("calcP", "calcQ", "guessP"):
    globals()[_n] = _c.with_family(FAMILY)(globals()[_n])


def build_dK(xin, x0in, hyp):
    """func.py:47-111 -> [dK/dlx, dK/dly, K/sig]"""
    return _c.build_dK3(FAMILY, xin, x0in, hyp)


def nll_chol_reg(hyp, x, y, N):
    """func.py:128-146: eigen fallback with neig = len(x)//2"""
    return _c.nll_fit(FAMILY, hyp, x, y, N, reg=True, neig=len(x) // 2)


def nll_chol(hyp, x, y, N):
    """func.py:148-166: eigen fallback with neig = len(x)//2"""
    return _c.nll_fit(FAMILY, hyp, x, y, N, neig=len(x) // 2)


def applymap_tok(nphmap, nm, Ntest, Q0map, P0map, xtrainp, ztrainp, Kyinvp, hypp, xtrain, ztrain, Kyinv, hyp,
                 compute_r=None):
    """func.py:184-219: section m's GP pair maps step i -> i+1 for i = m (mod nphmap).
    xtrainp (2N x nphmap), ztrainp (N x nphmap), Kyinvp (nphmap x N x N), hypp (nphmap x 3) and the
    same for the symplectic GP.  `compute_r(zk, r_gss)`: the reference's fieldlines.compute_r; an
    orbit with compute_r > 0.5 or P < 0 is lost (without it only P < 0)."""
    preds = [_c.predictor_pair(FAMILY, hyp[m, :], hypp[m, :], xtrainp[:, m], ztrainp[:, m], Kyinvp[m], xtrain[:, m],
                               ztrain[:, m], Kyinv[m]) for m in range(nphmap)]
    pmap = np.zeros([nm, Ntest])
    qmap = np.zeros([nm, Ntest])
    pmap[0, :] = P0map
    qmap[0, :] = Q0map
    i = 0
    r_gss = 0.3
    r_cut = 0.5
    while i < nm - nphmap:
        for m in range(0, nphmap):
            pr, prp = preds[m]
            pmap[i + 1, :] = np.nan
            qmap[i + 1, :] = np.nan
            ok = ~np.isnan(pmap[i, :])
            if ok.any():
                pmap[i + 1, ok] = _c.solve_implicit_P(pr, prp, qmap[i, ok], pmap[i, ok])
            ok2 = ~np.isnan(pmap[i + 1, :])
            if ok2.any():
                dq = pr(qmap[i, ok2], pmap[i + 1, ok2])[1]
                qmap[i + 1, ok2] = np.mod(dq + qmap[i, ok2], 2 * np.pi)
                ph = (2 * np.pi) / nphmap * np.mod(i + 1, nphmap)
                for k in np.nonzero(ok2)[0]:
                    lost = pmap[i + 1, k] < 0.0
                    if compute_r is not None and not lost:
                        lost = compute_r(np.array([pmap[i + 1, k] * 1e-2, qmap[i + 1, k], ph]), r_gss) > r_cut
                    if lost:
                        pmap[i + 1, k] = np.nan
                        qmap[i + 1, k] = np.nan
            i = i + 1
    return qmap, pmap


def quality(qmap, pmap, H, ysint, Ntest, Nm):
    """Diagnostics of Split_SympGPR/func.py:221-233 (host arithmetic): (p, q) against ysint[Nm, 0:2, k], H indexed
    [orbit, step]; like the reference it wraps ysint[:, 1] mod 2 pi IN PLACE."""
    ysint[:, 1] = np.mod(ysint[:, 1], 2 * np.pi)
    first = np.stack((np.asarray(pmap)[1, :Ntest], np.asarray(qmap)[1, :Ntest]))
    gd = np.mean((first - np.asarray(ysint)[Nm, 0:2, :Ntest]) ** 2, axis=0)
    Hk = np.asarray(H)[:Ntest, :]
    return np.std(Hk, axis=1) / np.mean(Hk, axis=1), gd, np.std(gd)
