"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Two checkers live here:

* ``Oracle``  -- ``liboracle.so``: our plain-C restatement (``sympgpr_oracle.c``), travels to
  the GPU box, is the thing ``tests/`` compare the HIP path with.
* ``Ref``     -- ``_ref/*.so``: the reference's own Fortran (compiled from /root/reference by
  ``make -C oracle ref``); used to pin the restatement, to generate ``tests/golden`` and, when
  present, as bench.py's ``cpu_baseline`` of kind "reference".

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing in sympgpr_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FAMILIES = {"A": 0, "B": 1, "C": 2, "D": 3}
W_KERN, W_DXDX0, W_DYDY0, W_DXDY0 = 0, 1, 2, 3
_dp = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def build(ref=True):
    """Compile liboracle.so (gcc) and, when /root/reference is present, oracle/_ref."""
    subprocess.run(["make", "-s", "-C", HERE, "oracle"], check=True)
    if ref:
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


class Oracle:
    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.lib = C.CDLL(path)
        L.orc_scalar.restype = C.c_double
        L.orc_scalar.argtypes = [C.c_int, C.c_int] + [C.c_double] * 7
        for f in (L.orc_build_k, L.orc_buildkreg):
            f.restype = C.c_int
            f.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, C.c_size_t, C.c_int]
        L.orc_scalar_dl.restype = C.c_double
        L.orc_scalar_dl.argtypes = [C.c_int, C.c_int] + [C.c_double] * 6
        L.orc_scalar_x.restype = C.c_double
        L.orc_scalar_x.argtypes = [C.c_int, C.c_int] + [C.c_double] * 7
        for f in (L.orc_build_dk, L.orc_build_dkreg):
            f.restype = C.c_int
            f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, C.c_size_t]
        L.orc_build_k_nd.restype = C.c_int
        L.orc_build_k_nd.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_size_t]
        L.orc_cholesky_lower.restype = C.c_int
        L.orc_cholesky_lower.argtypes = [C.c_int, _dp, C.c_size_t]
        L.orc_solve_cholesky.restype = None
        L.orc_solve_cholesky.argtypes = [C.c_int, _dp, C.c_size_t, _dp, C.c_size_t, C.c_int]
        L.orc_nll.restype = C.c_double
        L.orc_nll.argtypes = [C.c_int, _dp, C.c_size_t, _dp, _dp]
        L.orc_fit.restype = C.c_int
        L.orc_fit.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, _dp, _dp, _dp, C.c_int]
        L.orc_predict_rows.restype = None
        L.orc_predict_rows.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_predict_reg.restype = None
        L.orc_predict_reg.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]

    def scalar(self, fam, which, xa, ya, xb, yb, lx, ly, p=0.0):
        return self.lib.orc_scalar(FAMILIES[fam], which, xa, ya, xb, yb, lx, ly, p)

    def build_K(self, fam, x, y, x0, y0, hyp, threads=1):
        x, y, x0, y0, hyp = map(_f64, (x, y, x0, y0, hyp))
        n, n0 = len(x), len(x0)
        K = np.empty((2 * n, 2 * n0), order="F")
        self.lib.orc_build_k(FAMILIES[fam], n, n0, _p(x), _p(y), _p(x0), _p(y0), _p(hyp), _p(K),
                             max(2 * n, 1), threads)
        return K

    def buildKreg(self, fam, x, y, x0, y0, hyp, threads=1):
        x, y, x0, y0, hyp = map(_f64, (x, y, x0, y0, hyp))
        n, n0 = len(x), len(x0)
        K = np.empty((n, n0), order="F")
        self.lib.orc_buildkreg(FAMILIES[fam], n, n0, _p(x), _p(y), _p(x0), _p(y0), _p(hyp), _p(K),
                               max(n, 1), threads)
        return K

    def build_K_nd(self, fam, X, X0, hyp):
        """d canonical pairs: X (n x 2d), X0 (n0 x 2d), hyp = (lq.., lP.., sig) -> K (2dn x 2dn0)."""
        X = np.asfortranarray(X, dtype=np.float64)
        X0 = np.asfortranarray(X0, dtype=np.float64)
        hyp = _f64(hyp)
        n, D = X.shape
        n0 = X0.shape[0]
        K = np.empty((D * n, D * n0), order="F")
        rc = self.lib.orc_build_k_nd(FAMILIES[fam], D // 2, n, n0, _p(X), _p(X0), _p(hyp), _p(K), max(D * n, 1))
        if rc:
            raise ValueError("family not available for d > 1")
        return K

    def fit_nd(self, fam, X, z, hyp, sig2n):
        K = self.build_K_nd(fam, X, X, hyp)
        n = K.shape[0]
        Lf = self.cholesky(K + abs(sig2n) * np.eye(n))
        alpha = self.solve_cholesky(Lf, z)
        return alpha, self.nll(Lf, z, alpha), Lf

    def scalar_x(self, fam, which, xa, ya, xb, yb, lx, ly, p=0.0):
        """the seven functions of a kernels*.f90 no caller uses (X_NAMES)"""
        return self.lib.orc_scalar_x(FAMILIES[fam], which, xa, ya, xb, yb, lx, ly, p)

    def scalar_dl(self, fam, which, xa, ya, xb, yb, lx, ly):
        return self.lib.orc_scalar_dl(FAMILIES[fam], which, xa, ya, xb, yb, lx, ly)

    def build_dK(self, fam, x, y, x0, y0, hyp):
        """-> [dK/dlx, dK/dly], each (2 n0 x 2 n)  (functions/func.py:80-129)"""
        x, y, x0, y0, hyp = map(_f64, (x, y, x0, y0, hyp))
        n, n0 = len(x), len(x0)
        out = []
        for w in (0, 1):
            D = np.empty((2 * n0, 2 * n), order="F")
            self.lib.orc_build_dk(FAMILIES[fam], w, n, n0, _p(x), _p(y), _p(x0), _p(y0), _p(hyp), _p(D),
                                  max(2 * n0, 1))
            out.append(D)
        return out

    def build_dKreg(self, fam, x, y, x0, y0, hyp):
        """-> [dK/dlx, dK/dly], each (n x n0)  (functions/func.py:52-78)"""
        x, y, x0, y0, hyp = map(_f64, (x, y, x0, y0, hyp))
        n, n0 = len(x), len(x0)
        out = []
        for w in (0, 1):
            D = np.empty((n, n0), order="F")
            self.lib.orc_build_dkreg(FAMILIES[fam], w, n, n0, _p(x), _p(y), _p(x0), _p(y0), _p(hyp), _p(D),
                                     max(n, 1))
            out.append(D)
        return out

    def nll_grad(self, fam, hyp, x, y, N, reg=False):
        """functions/func.py:132-162 restated: explicit inverse, slogdet, trace terms."""
        hyp = np.asarray(hyp, dtype=np.float64)
        x = np.asarray(x, dtype=np.float64)
        if reg:
            K = self.buildKreg(fam, x[:N], x[N:2 * N], x[:N], x[N:2 * N], hyp[:-1])
            dK = self.build_dKreg(fam, x[:N], x[N:2 * N], x[:N], x[N:2 * N], hyp[:-1])
        else:
            h = N // 2
            K = self.build_K(fam, x[:h], x[h:N], x[:h], x[h:N], hyp[:-1])
            dK = self.build_dK(fam, x[:h], x[h:N], x[:h], x[h:N], hyp[:-1])
        Ky = K + np.abs(hyp[-1]) * np.diag(np.ones(N))
        Kyinv = np.linalg.inv(Ky)
        alpha = Kyinv.dot(y)
        val = 0.5 * y.T.dot(alpha) + 0.5 * np.linalg.slogdet(Ky)[1]
        grad = np.array([-0.5 * alpha.T.dot(dK[i].dot(alpha)) + 0.5 * np.trace(Kyinv.dot(dK[i])) for i in (0, 1)])
        return val, grad

    def cholesky(self, Ky):
        A = np.array(Ky, dtype=np.float64, order="F")
        info = self.lib.orc_cholesky_lower(A.shape[0], _p(A), max(A.shape[0], 1))
        if info:
            raise np.linalg.LinAlgError("%d-th leading minor not positive definite" % info)
        return A

    def solve_cholesky(self, L, b):
        L = np.asfortranarray(L, dtype=np.float64)
        B = np.array(b, dtype=np.float64, order="F")
        n = L.shape[0]
        nrhs = 1 if B.ndim == 1 else B.shape[1]
        self.lib.orc_solve_cholesky(n, _p(L), max(n, 1), _p(B), max(n, 1), nrhs)
        return B

    def nll(self, L, y, alpha):
        L = np.asfortranarray(L, dtype=np.float64)
        y, alpha = _f64(y), _f64(alpha)
        return self.lib.orc_nll(L.shape[0], _p(L), max(L.shape[0], 1), _p(y), _p(alpha))

    def fit(self, fam, x, y, z, hyp, sig2n, threads=1):
        """-> (alpha, nll, L)   as nll_chol (python/functions/func.py:189-196)."""
        x, y, z, hyp = map(_f64, (x, y, z, hyp))
        npts = len(x)
        n = 2 * npts
        K = np.empty((n, n), order="F")
        alpha = np.empty(n)
        nll = C.c_double()
        info = self.lib.orc_fit(FAMILIES[fam], npts, _p(x), _p(y), _p(z), _p(hyp), float(sig2n), _p(K),
                                _p(alpha), C.cast(C.byref(nll), _dp), threads)
        if info:
            raise np.linalg.LinAlgError("%d-th leading minor not positive definite" % info)
        return alpha, nll.value, K

    def predict_rows(self, fam, q, P, xtrain, ytrain, hyp, alpha):
        q, P, xtrain, ytrain, hyp, alpha = map(_f64, (q, P, xtrain, ytrain, hyp, alpha))
        m = len(q)
        op, oq = np.empty(m), np.empty(m)
        self.lib.orc_predict_rows(FAMILIES[fam], m, _p(q), _p(P), len(xtrain), _p(xtrain), _p(ytrain),
                                  _p(hyp), _p(alpha), _p(op), _p(oq))
        return op, oq

    def predict_reg(self, fam, q, P, xtrain, ytrain, hyp, alpha):
        q, P, xtrain, ytrain, hyp, alpha = map(_f64, (q, P, xtrain, ytrain, hyp, alpha))
        m = len(q)
        out = np.empty(m)
        self.lib.orc_predict_reg(FAMILIES[fam], m, _p(q), _p(P), len(xtrain), _p(xtrain), _p(ytrain),
                                 _p(hyp), _p(alpha), _p(out))
        return out


_SCALARS = ("kern_num", "d2kdxdx0_num", "d2kdydy0_num", "d2kdxdy0_num")
X_NAMES = {16: "dkdx_num", 17: "dkdy_num", 18: "dkdx0_num", 19: "dkdy0_num", 20: "d3kdxdx0dy0_num",
           21: "d3kdydy0dy0_num", 22: "d3kdxdy0dy0_num"}
DL_NAMES = {4: "dkdlx_num", 5: "dkdly_num", 6: "d3kdxdx0dlx_num", 7: "d3kdydy0dlx_num", 8: "d3kdxdy0dlx_num",
            9: "d3kdxdx0dly_num", 10: "d3kdydy0dly_num", 11: "d3kdxdy0dly_num"}


class Ref:
    """The reference's own Fortran (oracle/_ref).  ``Ref.available()`` is False where
    the libraries were never built (e.g. a checkout without /root/reference)."""

    @staticmethod
    def available():
        return os.path.exists(os.path.join(HERE, "_ref", "libsympgpr_ref_A.so"))

    def __init__(self):
        d = os.path.join(HERE, "_ref")
        self.mod = {f: C.CDLL(os.path.join(d, "libsympgpr_ref_%s.so" % f)) for f in "AC"}
        for lib in self.mod.values():
            for f in (lib.ref_build_k, lib.ref_buildkreg):
                f.restype = None
                f.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
            lib.ref_guessp.restype = C.c_double
            lib.ref_guessp.argtypes = [_dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp]
            lib.ref_calcq.restype = C.c_double
            lib.ref_calcq.argtypes = [_dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]
            lib.ref_calcp.restype = C.c_double
            lib.ref_calcp.argtypes = [_dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp]
        self.ker = {f: C.CDLL(os.path.join(d, "libkernels_%s.so" % f)) for f in ("A", "B", "Bsq", "C", "D")}

    def scalar(self, fam, name, xa, ya, xb, yb, lx, ly, p=None):
        """name_num_(x_a,y_a,x_b,y_b,lx,ly[,p]) of the family's generated kernels file."""
        f = getattr(self.ker[fam], name + "_")
        # family B's zero functions are INTEGER*4 (kernels_sum.f90:79,89,110)
        # family B's identically-zero functions are INTEGER*4 (kernels_sum.f90:79,89,110)
        int_zero = ("d2kdxdy0_num", "d3kdxdx0dy0_num", "d3kdxdy0dy0_num", "d3kdydy0dlx_num", "d3kdxdy0dlx_num",
                    "d3kdxdx0dly_num", "d3kdxdy0dly_num")
        f.restype = C.c_int if (fam == "B" and name in int_zero) else C.c_double
        args = [C.byref(C.c_double(v)) for v in (xa, ya, xb, yb, lx, ly)]
        if fam == "D":
            args.append(C.byref(C.c_double(p)))
        return float(f(*args))

    def build_K(self, fam, x, y, x0, y0, hyp):
        x, y, x0, y0, hyp = map(_f64, (x, y, x0, y0, hyp))
        n, n0 = len(x), len(x0)
        K = np.empty((2 * n, 2 * n0), order="F")
        if fam in self.mod:
            self.mod[fam].ref_build_k(n, n0, _p(x), _p(y), _p(x0), _p(y0), _p(hyp), _p(K))
            return K
        # B / D: scalar functions + the reference's pure-Python double loop
        # (01_pendulum/explicit/func_expl.py:53-72, implicit_period_unknown/func.py:43-63)
        l, sig = hyp[:-1], hyp[-1]
        p = l[2] if fam == "D" else None
        for k in range(n):
            for lk in range(n0):
                a = (x0[lk], y0[lk], x[k], y[k], l[0], l[1], p)
                K[k, lk] = self.scalar(fam, "d2kdxdx0_num", *a)
                K[n + k, lk] = self.scalar(fam, "d2kdxdy0_num", *a)
                K[k, n0 + lk] = self.scalar(fam, "d2kdxdy0_num", *a)
                K[n + k, n0 + lk] = self.scalar(fam, "d2kdydy0_num", *a)
        K[:, :] = sig * K[:, :]
        return K

    def buildKreg(self, fam, x, y, x0, y0, hyp):
        x, y, x0, y0, hyp = map(_f64, (x, y, x0, y0, hyp))
        n, n0 = len(x), len(x0)
        K = np.empty((n, n0), order="F")
        if fam in self.mod:
            self.mod[fam].ref_buildkreg(n, n0, _p(x), _p(y), _p(x0), _p(y0), _p(hyp), _p(K))
            return K
        l, sig = hyp[:-1], hyp[-1]
        p = l[2] if fam == "D" else None
        for k in range(n):
            for lk in range(n0):
                K[k, lk] = self.scalar(fam, "kern_num", x0[lk], y0[lk], x[k], y[k], l[0], l[1], p)
        K[:, :] = sig * K[:, :]
        return K

    def guessP(self, fam, x, y, hypp, xtrainp, ytrainp, ztrainp, Kyinvp):
        a = [_f64(v) for v in ([x], [y], hypp, xtrainp, ytrainp, ztrainp)]
        Ki = np.asfortranarray(Kyinvp, dtype=np.float64)
        return self.mod[fam].ref_guessp(_p(a[0]), _p(a[1]), _p(a[2]), len(a[3]), _p(a[3]), _p(a[4]),
                                        _p(a[5]), _p(Ki))

    def calcQ(self, fam, x, y, xtrain, ytrain, hyp, Kyinv, ztrain):
        a = [_f64(v) for v in ([x], [y], xtrain, ytrain, hyp, ztrain)]
        Ki = np.asfortranarray(Kyinv, dtype=np.float64)
        return self.mod[fam].ref_calcq(_p(a[0]), _p(a[1]), len(a[2]), _p(a[2]), _p(a[3]), _p(a[4]),
                                       _p(Ki), _p(a[5]))

    def calcP(self, fam, x, y, hyp, hypp, xtrainp, ytrainp, ztrainp, Kyinvp, xtrain, ytrain, ztrain, Kyinv):
        a = [_f64(v) for v in ([x], [y], hyp, hypp, xtrainp, ytrainp, ztrainp, xtrain, ytrain, ztrain)]
        Kip = np.asfortranarray(Kyinvp, dtype=np.float64)
        Ki = np.asfortranarray(Kyinv, dtype=np.float64)
        return self.mod[fam].ref_calcp(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), len(a[4]), _p(a[4]),
                                       _p(a[5]), _p(a[6]), _p(Kip), len(a[7]), _p(a[7]), _p(a[8]),
                                       _p(a[9]), _p(Ki))
