"""Host-array front-ends of the C ABI (NumPy in, NumPy out).  Each call stages through HBM;
large problems should use ``fit.SympFit`` which keeps K on the device."""
import contextlib
import threading

import numpy as np

from . import _lib as L

_family = "A"                 # process-wide, like the compiled `kernels` module of the reference: set_family() changes it for everybody
_tls = threading.local()      # family_scope() shadows it for the CALLING thread only: two threads fitting different examples
                              # (a Henon-Heiles fit beside a tokamak fit) must not flip each other's kernels


def set_family(fam):
    """Select which generated kernels file of the reference is mirrored (default "A":
    periodic x SE, python/05_tokamak/SympGPR/kernels.f90).  The reference selects it by
    which kernels*.f90 was compiled into the `kernels` / `sympgpr` module -- process-wide, and so is this:
    a selection made in the main thread is what an optimiser's worker threads see.  Inside a
    ``family_scope`` of the calling thread it changes that scope's selection instead."""
    global _family
    if fam not in L.FAMILIES:
        raise ValueError("family must be one of %s" % sorted(L.FAMILIES))
    if getattr(_tls, "family", None) is not None:
        _tls.family = fam
    else:
        _family = fam


def get_family():
    """The calling thread's ``family_scope`` selection if it is inside one, else the process-wide selection."""
    fam = getattr(_tls, "family", None)
    return _family if fam is None else fam


@contextlib.contextmanager
def family_scope(fam):
    """Select a kernel family for the calling THREAD until the block ends (the per-example modules under
    sympgpr_amd/examples each correspond to one kernels*.f90 of the reference); other threads keep seeing
    the process-wide selection."""
    if fam not in L.FAMILIES:
        raise ValueError("family must be one of %s" % sorted(L.FAMILIES))
    old = getattr(_tls, "family", None)
    _tls.family = fam
    try:
        yield
    finally:
        _tls.family = old


def _check_inout(K, shape):
    # f2py semantics for intent(inout) (sympgpr.f90:18): float64, F-contiguous, exact shape
    if not isinstance(K, np.ndarray) or K.dtype != np.float64:
        raise ValueError("K must be a float64 numpy array (intent(inout))")
    if K.shape != shape:
        raise ValueError("K has shape %s, expected %s" % (K.shape, shape))
    if not (K.flags.f_contiguous and K.flags.writeable):
        raise ValueError("failed to initialize intent(inout) array -- input not fortran contiguous")


def build_k(x, y, x0, y0, hyp, K, family=None):
    """sympgpr.build_k(x, y, x0, y0, hyp, K): fills K (2N x 2N0) in place
    (python/05_tokamak/SympGPR/sympgpr.f90:12-38)."""
    lib = L.load_library()
    x, y, x0, y0, hyp = map(L.f64, (np.atleast_1d(x), np.atleast_1d(y), np.atleast_1d(x0), np.atleast_1d(y0), hyp))
    n, n0 = K.shape[0] // 2, K.shape[1] // 2
    _check_inout(K, (2 * n, 2 * n0))
    if len(x) < n or len(y) < n or len(x0) < n0 or len(y0) < n0:
        raise ValueError("coordinate arrays shorter than K's block size")
    L.check(lib.sgpr_build_k_host(L.family_id(family or get_family()), n, n0, L.dptr(x), L.dptr(y), L.dptr(x0),
                                  L.dptr(y0), L.dptr(hyp), len(hyp), L.dptr(K), max(K.shape[0], 1)),
            "sgpr_build_k_host")


def buildkreg(x, y, x0, y0, hyp, K, family=None):
    """sympgpr.buildkreg(x, y, x0, y0, hyp, K): fills K (N x N0) in place (sympgpr.f90:40-60)."""
    lib = L.load_library()
    x, y, x0, y0, hyp = map(L.f64, (np.atleast_1d(x), np.atleast_1d(y), np.atleast_1d(x0), np.atleast_1d(y0), hyp))
    n, n0 = K.shape
    _check_inout(K, (n, n0))
    if len(x) < n or len(y) < n or len(x0) < n0 or len(y0) < n0:
        raise ValueError("coordinate arrays shorter than K's shape")
    L.check(lib.sgpr_buildkreg_host(L.family_id(family or get_family()), n, n0, L.dptr(x), L.dptr(y), L.dptr(x0),
                                    L.dptr(y0), L.dptr(hyp), len(hyp), L.dptr(K), max(n, 1)),
            "sgpr_buildkreg_host")


def kernel_eval(which, x_a, y_a, x_b, y_b, l, family=None):
    """kernels.<name>_num(x_a, y_a, x_b, y_b, lx, ly[, p]) -- scalar or elementwise arrays."""
    lib = L.load_library()
    scalar = np.ndim(x_a) == 0 and np.ndim(y_a) == 0 and np.ndim(x_b) == 0 and np.ndim(y_b) == 0
    xa, ya, xb, yb = np.broadcast_arrays(*[np.asarray(v, dtype=np.float64) for v in (x_a, y_a, x_b, y_b)])
    shape = xa.shape
    xa, ya, xb, yb = (np.ascontiguousarray(v).ravel() for v in (xa, ya, xb, yb))
    l = L.f64(l)
    out = np.empty(len(xa))
    L.check(lib.sgpr_kernel_eval_host(L.family_id(family or get_family()), which, len(xa), L.dptr(xa), L.dptr(ya),
                                      L.dptr(xb), L.dptr(yb), L.dptr(l), len(l), L.dptr(out)),
            "sgpr_kernel_eval_host")
    return float(out[0]) if scalar else out.reshape(shape)


def cholesky(Ky, lower=True):
    """scipy.linalg.cholesky(Ky, lower=True) (python/functions/func.py:166,184,193):
    returns a new F-ordered L with a clean upper triangle; LinAlgError when not PD."""
    if not lower:
        raise ValueError("only lower=True is used by the reference and implemented")
    lib = L.load_library()
    A = np.array(Ky, dtype=np.float64, order="F")
    if A.ndim != 2 or A.shape[0] != A.shape[1]:
        raise ValueError("expected square matrix")
    L.check(lib.sgpr_potrf_host(A.shape[0], L.dptr(A), max(A.shape[0], 1)), "sgpr_potrf_host")
    return A


def eigh(A):
    """Dense symmetric eigen-decomposition on the device -> (w ascending, Q): what the drivers'
    `eigsh(Ky, neig, ...)` fallback reduces to (python/02_pert_pendulum/func.py:199); the lower
    triangle of A is read."""
    lib = L.load_library()
    Q = np.array(A, dtype=np.float64, order="F")
    if Q.ndim != 2 or Q.shape[0] != Q.shape[1]:
        raise ValueError("expected square matrix")
    w = np.empty(Q.shape[0])
    rc = lib.sgpr_syev_host(Q.shape[0], L.dptr(Q), max(Q.shape[0], 1), L.dptr(w))
    if rc > 0:
        raise np.linalg.LinAlgError("Jacobi eigen-solver did not converge")
    L.check(rc, "sgpr_syev_host")
    return w, Q


def solve_cholesky(Lfac, b):
    """solve_triangular(L.T, solve_triangular(L, b, lower=True), lower=False)
    (python/functions/func.py:174-177)."""
    lib = L.load_library()
    Lf = np.asfortranarray(Lfac, dtype=np.float64)
    B = np.array(b, dtype=np.float64, order="F")
    n = Lf.shape[0]
    if B.shape[0] != n:
        raise ValueError("shapes of L and b are incompatible")
    nrhs = 1 if B.ndim == 1 else B.shape[1]
    L.check(lib.sgpr_potrs_host(n, L.dptr(Lf), max(n, 1), L.dptr(B), max(n, 1), nrhs), "sgpr_potrs_host")
    return B


def build_dk(x, y, x0, y0, hyp, family=None):
    """[dK/dlx, dK/dly], each (2 N0 x 2 N): build_dK of functions/func.py:80-129."""
    lib = L.load_library()
    x, y, x0, y0, hyp = map(L.f64, (np.atleast_1d(x), np.atleast_1d(y), np.atleast_1d(x0), np.atleast_1d(y0), hyp))
    n, n0 = len(x), len(x0)
    out = []
    for which in (0, 1):
        D = np.empty((2 * n0, 2 * n), order="F")
        L.check(lib.sgpr_build_dk_host(L.family_id(family or get_family()), which, n, n0, L.dptr(x), L.dptr(y),
                                       L.dptr(x0), L.dptr(y0), L.dptr(hyp), len(hyp), L.dptr(D), max(2 * n0, 1)),
                "sgpr_build_dk_host")
        out.append(D)
    return out


def build_dkreg(x, y, x0, y0, hyp, family=None):
    """[dK/dlx, dK/dly], each (N x N0): build_dKreg of functions/func.py:52-78."""
    lib = L.load_library()
    x, y, x0, y0, hyp = map(L.f64, (np.atleast_1d(x), np.atleast_1d(y), np.atleast_1d(x0), np.atleast_1d(y0), hyp))
    n, n0 = len(x), len(x0)
    out = []
    for which in (0, 1):
        D = np.empty((n, n0), order="F")
        L.check(lib.sgpr_build_dkreg_host(L.family_id(family or get_family()), which, n, n0, L.dptr(x), L.dptr(y),
                                          L.dptr(x0), L.dptr(y0), L.dptr(hyp), len(hyp), L.dptr(D), max(n, 1)),
                "sgpr_build_dkreg_host")
        out.append(D)
    return out


def build_k_nd(X, X0, hyp, family=None):
    """d canonical pairs per point: X (n, 2d), X0 (n0, 2d), hyp = (lq.., lP.., sig) -> K (2dn, 2dn0),
    block (a, b) = sig d^2k/dx_a dx'_b.  d = 1 equals build_k."""
    lib = L.load_library()
    X = np.asfortranarray(X, dtype=np.float64)
    X0 = np.asfortranarray(X0, dtype=np.float64)
    hyp = L.f64(hyp)
    n, D = X.shape
    n0 = X0.shape[0]
    if X0.shape[1] != D or D % 2:
        raise ValueError("X and X0 must be (n, 2d) and (n0, 2d)")
    K = np.empty((D * n, D * n0), order="F")
    L.check(lib.sgpr_build_k_nd_host(L.family_id(family or get_family()), D // 2, n, n0, L.dptr(X), max(n, 1), L.dptr(X0),
                                     max(n0, 1), L.dptr(hyp), len(hyp), L.dptr(K), max(D * n, 1)),
            "sgpr_build_k_nd_host")
    return K
