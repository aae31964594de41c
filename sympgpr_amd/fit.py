"""Device-resident fit: Gram build + Cholesky + alpha without K ever leaving HBM.

Host-side owner of a ``sgpr_fit_t`` handle (include/sympgpr_hip.h); the same sequence the
reference runs in ``nll_chol`` / ``gpsolve`` (python/functions/func.py:165-171,189-196).
"""
import ctypes as C

import numpy as np

from . import _lib as L


class SympFit:
    def __init__(self, family, x, y, z, hyp, sig2n, lower_only=True, stream=None, reg=False, block=None):
        """reg=True: the scalar-kernel GP of buildKreg / nll_chol_reg (order n = len(x)).
        block="qq" | "PP": only that diagonal block of build_K (order n = len(x)), the systems
        nll_expl factors (04_standard_map/func.py:126-141)."""
        self._lib = L.load_library()
        self._h = C.c_void_p()
        x, y, hyp = L.f64(x), L.f64(y), L.f64(hyp)
        if x.shape != y.shape or x.ndim != 1:
            raise ValueError("x and y must be 1-D arrays of equal length")
        self.n_pts = len(x)
        if block not in (None, "qq", "PP") or (block and reg):
            raise ValueError("block must be None, 'qq' or 'PP' (and excludes reg)")
        self.n = self.n_pts if (reg or block) else 2 * self.n_pts
        z = L.f64(z) if z is not None else np.zeros(self.n)
        if z.shape != (self.n,):
            raise ValueError("z must have length %d" % self.n)
        flags = (L.FIT_LOWER_ONLY if lower_only else 0) | (L.FIT_REG if reg else 0)
        flags |= {None: 0, "qq": L.FIT_BLOCK_QQ, "PP": L.FIT_BLOCK_PP}[block]
        L.check(self._lib.sgpr_fit_create(L.family_id(family), self.n_pts, L.dptr(x), L.dptr(y), L.dptr(z),
                                          L.dptr(hyp), len(hyp), float(sig2n), flags,
                                          C.c_void_p(stream or 0), C.byref(self._h)), "sgpr_fit_create")

    @classmethod
    def pairs(cls, family, X, z, hyp, sig2n, stream=None):
        """d canonical pairs per point (BASELINE configs d = 2, 3): X is (n_pts, 2d) with columns
        (q_1..q_d, P_1..P_d), hyp = (lq_1..lq_d, lP_1..lP_d, sig), z has 2*d*n_pts entries ordered
        block by block like the rows of K.  d = 1 is the reference's layout."""
        self = cls.__new__(cls)
        self._lib = L.load_library()
        self._h = C.c_void_p()
        X = np.asfortranarray(X, dtype=np.float64)
        if X.ndim != 2 or X.shape[1] % 2:
            raise ValueError("X must be (n_pts, 2d)")
        self.n_pts, D = X.shape
        self.d = D // 2
        self.n = D * self.n_pts
        hyp = L.f64(hyp)
        z = L.f64(z) if z is not None else np.zeros(self.n)
        if z.shape != (self.n,):
            raise ValueError("z must have length %d" % self.n)
        L.check(self._lib.sgpr_fit_create_nd(L.family_id(family), self.d, self.n_pts, L.dptr(X), max(self.n_pts, 1),
                                             L.dptr(z), L.dptr(hyp), len(hyp), float(sig2n), 0,
                                             C.c_void_p(stream or 0), C.byref(self._h)), "sgpr_fit_create_nd")
        return self

    def predict_pairs(self, Xt):
        """K* . alpha for test points Xt (m, 2d) -> (m, 2d): column a = predicted dF/dx_a."""
        Xt = np.asfortranarray(np.atleast_2d(Xt), dtype=np.float64)
        m = Xt.shape[0]
        out = np.empty((m, Xt.shape[1]), order="F")
        L.check(self._lib.sgpr_fit_predict_nd(self._h, m, L.dptr(Xt), max(m, 1), L.dptr(out)), "sgpr_fit_predict_nd")
        return out

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.sgpr_fit_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_hyp(self, hyp, sig2n):
        hyp = L.f64(hyp)
        L.check(self._lib.sgpr_fit_set_hyp(self._h, L.dptr(hyp), len(hyp), float(sig2n)), "sgpr_fit_set_hyp")

    def set_targets(self, z):
        z = L.f64(z)
        if z.shape != (self.n,):
            raise ValueError("z must have length 2*n_pts")
        L.check(self._lib.sgpr_fit_set_targets(self._h, L.dptr(z)), "sgpr_fit_set_targets")

    def build(self):
        L.check(self._lib.sgpr_fit_build(self._h), "sgpr_fit_build")

    def factor(self):
        L.check(self._lib.sgpr_fit_factor(self._h), "sgpr_fit_factor")

    def solve(self):
        L.check(self._lib.sgpr_fit_solve(self._h), "sgpr_fit_solve")

    def run(self):
        L.check(self._lib.sgpr_fit_run(self._h), "sgpr_fit_run")
        return self

    def alpha(self):
        out = np.empty(self.n)
        L.check(self._lib.sgpr_fit_alpha(self._h, L.dptr(out)), "sgpr_fit_alpha")
        return out

    def nll(self):
        v = C.c_double()
        L.check(self._lib.sgpr_fit_nll(self._h, C.cast(C.byref(v), L._dp)), "sgpr_fit_nll")
        return v.value

    def nll_grad(self):
        """d nll / d(lx, ly) (functions/func.py:132-162: nlp_grad) on a solved fit."""
        g = np.empty(2)
        L.check(self._lib.sgpr_fit_nll_grad(self._h, L.dptr(g)), "sgpr_fit_nll_grad")
        return g

    def nll_grad_terms(self):
        """[alpha^T dK_lx alpha, tr(Ky^-1 dK_lx), alpha^T dK_ly alpha, tr(Ky^-1 dK_ly), tr(Ky^-1)]: the
        pieces of Rasmussen (5.9) the per-example nll_grad variants recombine."""
        t = np.empty(5)
        L.check(self._lib.sgpr_fit_nll_grad_terms(self._h, L.dptr(t)), "sgpr_fit_nll_grad_terms")
        return t

    def eig(self):
        """-> (w, c): eigenvalues of Ky = K + |sig2n| I (ascending) and c = Q^T z, computed on the
        device (parallel Jacobi).  The failure path of the drivers' nll_chol: their
        `except: eigsh(Ky, neig, ...)` branch (02_pert_pendulum/func.py:194-203).  The handle has to
        be run() again before other queries."""
        w, c = np.empty(self.n), np.empty(self.n)
        rc = self._lib.sgpr_fit_eig(self._h, L.dptr(w), L.dptr(c))
        if rc > 0:
            raise np.linalg.LinAlgError("Jacobi eigen-solver did not converge")
        L.check(rc, "sgpr_fit_eig")
        return w, c

    def inverse(self):
        """Ky^-1 as an F-ordered host array (the drivers' scipy.linalg.inv(K + sig2n I))."""
        A = np.empty((self.n, self.n), order="F")
        L.check(self._lib.sgpr_fit_inverse(self._h, L.dptr(A), self.n), "sgpr_fit_inverse")
        return A

    def ldiag(self):
        out = np.empty(self.n)
        L.check(self._lib.sgpr_fit_ldiag(self._h, L.dptr(out)), "sgpr_fit_ldiag")
        return out

    def matrix(self):
        """The factor L (after factor()) or Ky (after build()), as an F-ordered host array."""
        A = np.empty((self.n, self.n), order="F")
        L.check(self._lib.sgpr_fit_get_matrix(self._h, L.dptr(A), self.n), "sgpr_fit_get_matrix")
        return A

    def solve_rhs(self, B):
        B = np.array(B, dtype=np.float64, order="F")
        nrhs = 1 if B.ndim == 1 else B.shape[1]
        if B.shape[0] != self.n:
            raise ValueError("B must have 2*n_pts rows")
        L.check(self._lib.sgpr_fit_solve_rhs(self._h, L.dptr(B), self.n, nrhs), "sgpr_fit_solve_rhs")
        return B

    def solve_rhs_dev(self, dptr, nrhs, ldb=None):
        """solve_rhs for right-hand sides that are already on the fit's device: `dptr` = device address (int) of an n x nrhs
        column-major block of doubles -- e.g. `t.data_ptr()` of a contiguous torch.float64 tensor of shape (nrhs, n) --,
        overwritten with the solution.  No host copies; returns when the solve has finished.

        The solve runs on the FIT'S OWN stream and is not ordered against the stream that produced the block: the caller makes
        sure B is complete on the device before the call (e.g. `torch.cuda.current_stream().synchronize()`, as bench.py does).
        The scratch of the block solves (~0.8 GB at n = 98 304) stays with the handle until `release_scratch()` or close."""
        L.check(self._lib.sgpr_fit_solve_rhs_dev(self._h, C.c_void_p(int(dptr)), self.n if ldb is None else int(ldb), int(nrhs)),
                "sgpr_fit_solve_rhs_dev")

    def release_scratch(self):
        """give the block solves' device scratch back (it is allocated again by the next solve_rhs that needs it)"""
        L.check(self._lib.sgpr_fit_trim(self._h), "sgpr_fit_trim")

    def solve_rhs_ms(self):
        """Device time (ms) of the last solve_rhs: the triangular solves without the host copies of B."""
        v = C.c_double()
        L.check(self._lib.sgpr_fit_solve_rhs_ms(self._h, C.cast(C.byref(v), L._dp)), "sgpr_fit_solve_rhs_ms")
        return v.value

    def predict_rows(self, q, P):
        q, P = L.f64(np.atleast_1d(q)), L.f64(np.atleast_1d(P))
        m = len(q)
        op, oq = np.empty(m), np.empty(m)
        L.check(self._lib.sgpr_fit_predict_rows(self._h, m, L.dptr(q), L.dptr(P), L.dptr(op), L.dptr(oq)),
                "sgpr_fit_predict_rows")
        return op, oq

    def cond_estimate(self, iters=30):
        """cond_2(Ky) from below: power iteration for lambda_max (Ky v through the prediction kernel), inverse iteration with
        the cached factor for lambda_min.  Returns dict(lambda_max, lambda_min, cond, last_change, iters).  Not part of the
        reference's call surface (it never computes a condition number); the parity tolerances are multiples of cond * eps."""
        o = np.zeros(4)
        L.check(self._lib.sgpr_fit_cond_estimate(self._h, int(iters), L.dptr(o)), "sgpr_fit_cond_estimate")
        return {"lambda_max": float(o[0]), "lambda_min": float(o[1]), "cond": float(o[2]), "last_change": float(o[3]),
                "iters": int(iters), "method": "power iteration on Ky v (rows re-evaluated by the prediction kernel) / inverse "
                                                "iteration with the cached factor; Rayleigh quotients: a lower bound of cond_2"}

    def stage_ms(self):
        b, f, s = C.c_double(), C.c_double(), C.c_double()
        c = lambda v: C.cast(C.byref(v), L._dp)
        L.check(self._lib.sgpr_fit_stage_ms(self._h, c(b), c(f), c(s)), "sgpr_fit_stage_ms")
        return b.value, f.value, s.value

    def device_ptrs(self):
        dA, lda, dal = C.c_void_p(), C.c_size_t(), C.c_void_p()
        L.check(self._lib.sgpr_fit_device_ptrs(self._h, C.byref(dA), C.byref(lda), C.byref(dal)))
        return dA.value, lda.value, dal.value


def batch_max_order():
    """largest matrix order per problem sgpr_fit_batch takes (256)"""
    return L.load_library().sgpr_fit_batch_max_order()


def fit_batch(family, x, y, z, hyp, sig2n, reg=False, want_alpha=True):
    """Many small independent fits in ONE launch (one workgroup per problem): the body of nll_chol
    (python/functions/func.py:189-196; reg=True: nll_chol_reg) for every row b of
        x, y (B, n_pts);  z (B, n), n = 2 n_pts (n_pts with reg);  hyp (B, nhyp);  sig2n (B,) or scalar.
    -> (alpha (B, n) or None, nll (B,), info (B,)): info[b] > 0 where Ky_b is not positive definite
    (nll[b] is NaN there, like the LinAlgError scipy raises for that problem)."""
    x, y, z, hyp = (np.ascontiguousarray(np.atleast_2d(np.asarray(v, dtype=np.float64))) for v in (x, y, z, hyp))
    B, n_pts = x.shape
    n = n_pts if reg else 2 * n_pts
    if y.shape != (B, n_pts) or z.shape != (B, n) or hyp.shape[0] != B:
        raise ValueError("fit_batch: x, y (B, n_pts), z (B, n), hyp (B, nhyp)")
    s2 = np.ascontiguousarray(np.broadcast_to(np.asarray(sig2n, dtype=np.float64), (B,)))
    alpha = np.empty((B, n)) if want_alpha else None
    nll = np.empty(B)
    info = np.zeros(B, dtype=np.int32)
    L.check(L.load_library().sgpr_fit_batch(L.family_id(family), B, n_pts, L.dptr(x), L.dptr(y), L.dptr(z), L.dptr(hyp),
                                            hyp.shape[1], L.dptr(s2), L.FIT_REG if reg else 0,
                                            L.dptr(alpha) if want_alpha else None, L.dptr(nll),
                                            info.ctypes.data_as(C.POINTER(C.c_int))), "sgpr_fit_batch")
    nll[info != 0] = np.nan
    return alpha, nll, info
