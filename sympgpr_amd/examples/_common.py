"""Shared pieces of the per-driver modules."""
import numpy as np

from .. import func as _base
from ..fit import SympFit
from ..fortran.sympgpr import sympgpr as _f2py
from ..maps import EXPLICIT, WRAP_P, WRAP_Q, run_map  # noqa: F401
from ..ops import family_scope
from ..predict import Predictor, solve_implicit_P  # noqa: F401


def eig_fallback_value(w, c, neig, nx, sig2n):
    """The `except:` branch of the drivers' nll_chol (python/02_pert_pendulum/func.py:199-203):
        w, Q = eigsh(Ky, neig, ...); alpha = Q diag(1/w) Q^T y
        ret = y.alpha/2 + (sum log w + (len(x) - neig) log|sig2n|)/2
    from the full spectrum w and c = Q^T y: eigsh keeps the neig eigenvalues of largest
    magnitude (which='LM'), and hands a dense matrix to eigh (all pairs) when neig >= n."""
    n = len(w)
    if neig <= 0:
        raise ValueError("k must be greater than 0.")
    sel = np.arange(n) if neig >= n else np.sort(np.argsort(np.abs(w), kind="stable")[n - neig:])
    ws, cs = w[sel], c[sel]
    with np.errstate(invalid="ignore", divide="ignore"):
        return 0.5 * np.sum(cs * cs / ws) + 0.5 * (np.sum(np.log(ws)) + (nx - neig) * np.log(np.abs(sig2n)))


def nll_fit(family, hyp, x, y, N, reg=False, neig=None):
    """Objective of the hyper-parameter search: Gram build + Cholesky + alpha + nll in one
    device-resident pass.  neig is not None: the driver's eigen-fallback when Ky is not
    positive definite; otherwise LinAlgError propagates like scipy's."""
    hyp = np.asarray(hyp, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    npts = N if reg else N // 2
    if 0 < N <= _base._SMALL_ORDER and (reg or N % 2 == 0):
        # the drivers' own sizes: the whole objective in one single-workgroup launch (see func._nll_small)
        from ..fit import fit_batch
        _, nll, info = fit_batch(family, x[None, 0:npts], x[None, npts:2 * npts], y[None, :N], hyp[None, :-1],
                                 np.abs(hyp[-1:]), reg=reg, want_alpha=False)
        if not info[0]:
            return float(nll[0])
        if neig is None:
            raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % int(info[0]))
        # not positive definite and the driver has an eigen fallback: take it through the handle below
    with SympFit(family, x[0:npts], x[npts:2 * npts], y[:N], hyp[:-1], np.abs(hyp[-1]), reg=reg,
                 lower_only=neig is None) as f:
        if neig is None:
            return f.run().nll()
        try:
            return f.run().nll()
        except np.linalg.LinAlgError:
            print('Warning! Fallback to eig solver!')
            w, c = f.eig()
            return eig_fallback_value(w, c, neig, len(x), hyp[-1])


def nll_grad3(family, hyp, x, y, N):
    """nll_grad of 03_henon_heiles/func.py:168-192 and 05_tokamak/SympGPR/func.py:152-168:
    (nlp_val, nlp_grad[3]).  The third entry is reproduced as written there -- the quadratic term
    uses dK[1] (d/dly), the trace term dK[2] = K / sig."""
    hyp = np.asarray(hyp, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    npts = N // 2
    sig, s2 = hyp[-2], np.abs(hyp[-1])
    with SympFit(family, x[0:npts], x[npts:2 * npts], np.asarray(y, dtype=np.float64)[:2 * npts], hyp[:-1], s2,
                 lower_only=False) as f:
        f.run()
        val = f.nll()
        aKa_x, tr_x, aKa_y, tr_y, tr_inv = f.nll_grad_terms()
    tr_sig = (N - s2 * tr_inv) / sig            # tr(Ky^-1 K) / sig with K = Ky - |sig2n| I
    return val, np.array([-0.5 * aKa_x + 0.5 * tr_x, -0.5 * aKa_y + 0.5 * tr_y, -0.5 * aKa_y + 0.5 * tr_sig])


def build_dK3(family, xin, x0in, hyp):
    """build_dK of 03_henon_heiles/func.py:70-134: [dK/dlx, dK/dly, K/sig], the third block with the
    roles of the two point sets as in its loop (rows over xin)."""
    with family_scope(family):
        dK = _base.build_dK(xin, x0in, hyp)
        N, N0 = len(xin) // 2, len(x0in) // 2
        K = np.empty((2 * N, 2 * N0), order='F')
        _base.build_K(xin, x0in, np.hstack((np.asarray(hyp, dtype=np.float64)[:-1], [1.0])), K)
    return dK + [K]


def predictor_pair(family, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv):
    Ntrain, Ntrainp = len(xtrain) // 2, len(xtrainp) // 2
    pr = Predictor(family, xtrain[:Ntrain], xtrain[Ntrain:2 * Ntrain], l,
                   np.asarray(Kyinv, dtype=np.float64) @ np.asarray(ztrain, dtype=np.float64))
    prp = Predictor(family, xtrainp[:Ntrainp], xtrainp[Ntrainp:2 * Ntrainp], hypp,
                    np.asarray(Kyinvp, dtype=np.float64) @ np.asarray(ztrainp, dtype=np.float64), reg=True)
    return pr, prp


def with_family(fam):
    """Decorator: run the wrapped function with kernel family `fam` selected."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def inner(*a, **k):
            with family_scope(fam):
                return fn(*a, **k)
        return inner
    return deco


def python_surface(fam, ns):
    """The functions every per-example func.py shares with python/functions/func.py, bound to
    kernel family `fam`, into namespace `ns`."""
    for name in ("f_kern", "d2kdxdx0", "d2kdydy0", "d2kdxdy0", "d2kdydx0", "build_K", "buildKreg", "gpsolve",
                 "solve_cholesky"):
        ns[name] = with_family(fam)(getattr(_base, name))


# ---- the pure-Python predictors of the per-example files (x, y arrive as 1-element lists / arrays) ----
def guessP_py(fam, x, y, hypp, xtrainp, ztrainp, Kyinvp):
    """01_pendulum/implicit/func.py:119-124: Kstar(1 x Ntrainp) . (Kyinvp ztrainp) -> array (1,)."""
    with family_scope(fam):
        return np.array([_base.guessP(float(np.ravel(x)[0]), float(np.ravel(y)[0]), hypp, xtrainp, ztrainp, Kyinvp)])


def calcQ_py(fam, x, y, xtrain, l, Kyinv, ztrain):
    """01_pendulum/implicit/func.py:126-132 -> dq (scalar)."""
    with family_scope(fam):
        return _base.calcQ(float(np.ravel(x)[0]), float(np.ravel(y)[0]), xtrain, l, Kyinv, ztrain)


def rows_py(fam, x, y, xtrain, l, Kyinv, ztrain):
    """Kstar^T (Kyinv ztrain) for one test point -> (pGP[0], pGP[1])."""
    Ntrain = len(xtrain) // 2
    pr = Predictor(fam, xtrain[:Ntrain], xtrain[Ntrain:2 * Ntrain], l,
                   np.asarray(Kyinv, dtype=np.float64) @ np.asarray(ztrain, dtype=np.float64))
    r1, r2 = pr(np.ravel(x)[:1], np.ravel(y)[:1])
    return float(r1[0]), float(r2[0])


def Pnewton_py(fam, P, x, y, l, xtrain, Kyinv, ztrain):
    """01_pendulum/implicit/func.py:134-139: f = pGP[0] - y + P (arrays of one element)."""
    r1, _ = rows_py(fam, x, P, xtrain, l, Kyinv, ztrain)
    return r1 - np.asarray(y, dtype=np.float64) + np.asarray(P, dtype=np.float64)


def calcP_py(fam, x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv):
    """calcP of the per-example files (scipy `newton` from the regular-GP guess,
    01_pendulum/implicit/func.py:141-147): the same root, by the batched secant of
    predict.solve_implicit_P (tol 1e-13) -> array (1,)."""
    pr, prp = predictor_pair(fam, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv)
    return solve_implicit_P(pr, prp, np.ravel(x)[:1], np.ravel(y)[:1])
