"""Mirror of the reference's GP library python/functions/func.py -- same names, positional
order, in-place semantics and error behaviour -- with the hot path on the MI355X.

    reference                               here
    ------------------------------------    -----------------------------------------------
    sympgpr.build_k / buildkreg (f2py)      sgpr_build_k_host / sgpr_buildkreg_host
    scipy.linalg.cholesky(lower=True)       sgpr_potrf_host          (LinAlgError if not PD)
    solve_triangular x 2                    sgpr_potrs_host
    nll_chol / nll_chol_reg                 one device-resident sgpr_fit_* pass (K stays in HBM)
    guessP / calcQ / calcP / applymap       K* rows . alpha on the device, alpha cached

Select the kernel family (which kernels*.f90 the reference would have compiled) with
`set_family("A"|"B"|"C"|"D")`; default "A" (python/05_tokamak/SympGPR/kernels.f90).
build_dK / build_dKreg / nll_grad / nll_grad_reg are available for the product kernels (A, C, D).
"""
import numpy as np

from . import kernels as _kernels
from .fit import SympFit
from .fortran.sympgpr import sympgpr
from .kernels import *  # noqa: F401,F403  (kern_num, d2kdxdx0_num, ... like `from kernels import *`, func.py:15)
from . import ops as _ops
from .ops import cholesky as _cholesky
from .ops import get_family, set_family, solve_cholesky as _solve_cholesky  # noqa: F401
from .predict import Predictor, solve_implicit_P


_SMALL_ORDER = 256      # sgpr_fit_batch_max_order(): up to here one objective call is one single-workgroup launch


def _l(l):
    return tuple(float(v) for v in l)


def f_kern(x, y, x0, y0, l):            # functions/func.py:17-18
    return _kernels.kern_num(x, y, x0, y0, *_l(l))


def d2kdxdx0(x, y, x0, y0, l):          # :20-21
    return _kernels.d2kdxdx0_num(x, y, x0, y0, *_l(l))


def d2kdydy0(x, y, x0, y0, l):          # :23-24
    return _kernels.d2kdydy0_num(x, y, x0, y0, *_l(l))


def d2kdxdy0(x, y, x0, y0, l):          # :26-27
    return _kernels.d2kdxdy0_num(x, y, x0, y0, *_l(l))


def d2kdydx0(x, y, x0, y0, l):          # :29-30
    return d2kdxdy0(x, y, x0, y0, l)


def build_K(xin, x0in, hyp, K):
    """functions/func.py:32-40: covariance with derivative observations, Eq. (38); K in place."""
    N = K.shape[0] // 2
    N0 = K.shape[1] // 2
    x0 = x0in[0:N0]
    x = xin[0:N]
    y0 = x0in[N0:2 * N0]
    y = xin[N:2 * N]
    sympgpr.build_k(x, y, x0, y0, hyp, K)


def buildKreg(xin, x0in, hyp, K):
    """functions/func.py:42-50: scalar-kernel covariance on the regular (q,p) grid; K in place."""
    N = K.shape[0]
    N0 = K.shape[1]
    x0 = x0in[0:N0]
    x = xin[0:N]
    y0 = x0in[N0:2 * N0]
    y = xin[N:2 * N]
    sympgpr.buildkreg(x, y, x0, y0, hyp, K)


def gpsolve(Ky, ft):
    """functions/func.py:165-171 -> (L, alpha)."""
    Lf = _cholesky(Ky, lower=True)
    return Lf, _solve_cholesky(Lf, ft)


def solve_cholesky(L, b):
    """functions/func.py:174-177."""
    return _solve_cholesky(L, b)


def _nll_small(hyp, x, y, N, reg):
    """One objective value at the drivers' own sizes (matrix order <= 256): the whole body runs in ONE launch of
    one workgroup (sgpr_fit_batch with a batch of one) -- a device-resident handle costs five launches, an
    allocation per buffer and 230-390 us at these orders, more than the reference's CPU path needs at order 80."""
    from .fit import fit_batch
    npts = N if reg else N // 2
    _, nll, info = fit_batch(get_family(), x[None, 0:npts], x[None, npts:2 * npts], y[None, :N if reg else 2 * npts],
                             hyp[None, :-1], np.abs(hyp[-1:]), reg=reg, want_alpha=False)
    if info[0]:
        raise np.linalg.LinAlgError("%d-th leading minor of the array is not positive definite" % int(info[0]))
    return float(nll[0])


def nll_chol_reg(hyp, x, y, N):
    """functions/func.py:180-187: negative log-posterior of the scalar-kernel GP.  N = matrix
    order; x holds (q || p) and is sliced exactly as buildKreg does."""
    hyp = np.asarray(hyp, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if 0 < N <= _SMALL_ORDER:
        return _nll_small(hyp, x, y, N, True)
    with SympFit(get_family(), x[0:N], x[N:2 * N], y[:N], hyp[:-1],
                 np.abs(hyp[-1]), reg=True) as f:
        return f.run().nll()


def nll_chol(hyp, x, y, N):
    """functions/func.py:189-196: negative log-posterior of the symplectic GP.  N = matrix order
    (twice the number of points build_K slices out of x)."""
    hyp = np.asarray(hyp, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if 0 < N <= _SMALL_ORDER and N % 2 == 0:
        return _nll_small(hyp, x, y, N, False)
    npts = N // 2
    with SympFit(get_family(), x[0:npts], x[npts:2 * npts], y[:2 * npts],
                 hyp[:-1], np.abs(hyp[-1])) as f:
        return f.run().nll()


def nll_chol_batch(hyps, x, y, N, reg=False):
    """nll_chol (reg=True: nll_chol_reg) for a whole POPULATION of hyper-parameter vectors over the same
    data in one launch: what a CMA-ES generation costs the Split_SympGPR driver one call at a time
    (python/05_tokamak/Split_SympGPR/main.py:36-41,63-66: `cma.fmin(nll_transform, ...)`).
    hyps (B, nhyp + 1) rows like nll_chol's hyp (the last entry is sig2_n); N = matrix order, at most
    fit.batch_max_order() (larger orders: loop over nll_chol).  Rows whose Ky is not positive
    definite come back as +inf (an optimiser's usual penalty for an exception)."""
    from .fit import batch_max_order, fit_batch
    hyps = np.atleast_2d(np.asarray(hyps, dtype=np.float64))
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if N > batch_max_order():
        f = nll_chol_reg if reg else nll_chol
        out = np.empty(len(hyps))
        for b, h in enumerate(hyps):
            try:
                out[b] = f(h, x, y, N)
            except np.linalg.LinAlgError:
                out[b] = np.inf
        return out
    npts = N if reg else N // 2
    B = len(hyps)
    X = np.broadcast_to(x[0:npts], (B, npts))
    Y = np.broadcast_to(x[npts:2 * npts], (B, npts))
    Z = np.broadcast_to(y[:N if reg else 2 * npts], (B, N if reg else 2 * npts))
    _, nll, info = fit_batch(get_family(), X, Y, Z, hyps[:, :-1], np.abs(hyps[:, -1]), reg=reg, want_alpha=False)
    nll[info != 0] = np.inf
    return nll


def guessP(x, y, hypp, xtrainp, ztrainp, Kyinvp):
    """functions/func.py:198-201."""
    Ntrain = len(xtrainp) // 2
    return sympgpr.guessp(x, y, hypp, xtrainp[0:Ntrain], xtrainp[Ntrain:], ztrainp, Kyinvp)


def calcQ(x, y, xtrain, l, Kyinv, ztrain):
    """functions/func.py:204-207."""
    Ntrain = len(xtrain) // 2
    return sympgpr.calcq(x, y, xtrain[:Ntrain], xtrain[Ntrain:], l, Kyinv, ztrain)


def calcP(x, y, l, hypp, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv):
    """functions/func.py:209-213."""
    Ntrain = len(xtrain) // 2
    Ntrainp = len(xtrainp) // 2
    return sympgpr.calcp(x, y, l, hypp, xtrainp[:Ntrainp], xtrainp[Ntrainp:], ztrainp, Kyinvp,
                         xtrain[:Ntrain], xtrain[Ntrain:], ztrain, Kyinv)


def _applymap(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv, wrap):
    """The recurrences of the reference's double loop (functions/func.py:216-237) with all nm steps
    of all Ntest orbits inside one device launch (sgpr_applymap_host): alpha = Kyinv ztrain is
    formed once instead of inside every calcP / calcQ call."""
    from .maps import WRAP_Q, run_map
    return run_map(WRAP_Q if wrap else 0, nm, Ntest, l, Q0map, P0map, xtrain, ztrain, Kyinv, hypp, xtrainp,
                   ztrainp, Kyinvp)


def applymap(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv):
    """functions/func.py:216-237 (q taken mod 2 pi)."""
    return _applymap(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv, True)


def applymap_henon(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv):
    """functions/func.py:239-260 (no wrap of q)."""
    return _applymap(nm, Ntest, l, hypp, Q0map, P0map, xtrainp, ztrainp, Kyinvp, xtrain, ztrain, Kyinv, False)


def quality(qmap, pmap, H, ysint, Ntest, Nm):
    """Diagnostics the drivers print after applying the map (same name, arguments and return values as
    functions/func.py:262-272; host arithmetic on [nm, Ntest] arrays, outside the accelerated path):
    gd[k]   mean squared distance between the first mapped point (q_1, p_1) of orbit k and the reference
            orbit ysint[Nm, :, k];  stdgd = its standard deviation over the orbits;
    Eosc[k] relative energy oscillation std(H[:, k]) / mean(H[:, k])."""
    first = np.stack((np.asarray(qmap)[1, :Ntest], np.asarray(pmap)[1, :Ntest]))          # (2, Ntest)
    gd = np.mean((first - np.asarray(ysint)[Nm, :, :Ntest]) ** 2, axis=0)
    Hk = np.asarray(H)[:, :Ntest]
    return np.std(Hk, axis=0) / np.mean(Hk, axis=0), gd, np.std(gd)


def build_dKreg(xin, x0in, hyp):
    """functions/func.py:52-78 -> [dK/dlx, dK/dly], each (N x N0)."""
    N = len(xin) // 2
    N0 = len(x0in) // 2
    return _ops.build_dkreg(xin[0:N], xin[N:2 * N], x0in[0:N0], x0in[N0:2 * N0], hyp)


def build_dK(xin, x0in, hyp):
    """functions/func.py:80-129 -> [dK/dlx, dK/dly], each (2 N0 x 2 N), rows over the "0" points."""
    N = len(xin) // 2
    N0 = len(x0in) // 2
    return _ops.build_dk(xin[0:N], xin[N:2 * N], x0in[0:N0], x0in[N0:2 * N0], hyp)


def nll_grad_reg(hyp, x, y, N):
    """functions/func.py:132-146 -> (nlp_val, nlp_grad[2]).  The reference inverts Ky and takes
    slogdet; 0.5 slogdet(Ky) = sum log diag L, so the value is the one nll_chol_reg returns."""
    hyp = np.asarray(hyp, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    with SympFit(get_family(), x[0:N], x[N:2 * N], np.asarray(y, dtype=np.float64)[:N], hyp[:-1],
                 np.abs(hyp[-1]), reg=True, lower_only=False) as f:
        f.run()
        return f.nll(), f.nll_grad()


def nll_grad(hyp, x, y, N):
    """functions/func.py:148-162 -> (nlp_val, nlp_grad[2])."""
    hyp = np.asarray(hyp, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    npts = N // 2
    with SympFit(get_family(), x[0:npts], x[npts:2 * npts], np.asarray(y, dtype=np.float64)[:2 * npts],
                 hyp[:-1], np.abs(hyp[-1]), lower_only=False) as f:
        f.run()
        return f.nll(), f.nll_grad()
