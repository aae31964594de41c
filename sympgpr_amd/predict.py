"""K* rows . alpha on the device for batches of test points -- the O(n) form of the reference's
`guessP`, `calcq` and of the Newton target inside `calcP` (python/05_tokamak/SympGPR/
sympgpr.f90:62-125).  The reference recomputes alpha = matmul(Kyinv, ztrain) on every call
(O(n^2)); here alpha is formed once and stays in HBM with the training inputs."""
import ctypes as C

import numpy as np

from . import _lib as L


class Predictor:
    """Device-resident (xtrain, ytrain, alpha, hyp) for one GP.  reg=False: the symplectic GP
    (rows 1 and 2 of Kstar = build_K(test, train)); reg=True: the scalar-kernel GP of guessP."""

    def __init__(self, family, xtrain, ytrain, hyp, alpha, reg=False):
        import torch  # device memory only
        self._torch = torch
        self.lib = L.load_library()
        if L.device_count() < 1:
            raise L.NoDeviceError("no HIP device: libsympgpr_hip.so has no CPU fallback")
        self.family = L.family_id(family)
        self.reg = reg
        self.hyp = L.f64(hyp)
        dev = torch.device("cuda", torch.cuda.current_device())
        put = lambda a: torch.as_tensor(L.f64(a)).to(dev)
        self.x, self.y, self.alpha = put(xtrain), put(ytrain), put(alpha)
        self.n0 = len(self.x)
        if len(self.alpha) != (self.n0 if reg else 2 * self.n0):
            raise ValueError("alpha has the wrong length for this training set")
        self.dev = dev

    def __call__(self, q, P):
        """-> (row1 . alpha, row2 . alpha) for the symplectic GP, or (k* . alpha,) for reg."""
        torch = self._torch
        q, P = L.f64(np.atleast_1d(q)), L.f64(np.atleast_1d(P))
        m = len(q)
        dq, dP = torch.as_tensor(q).to(self.dev), torch.as_tensor(P).to(self.dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())
        if self.reg:
            out = torch.empty(m, dtype=torch.float64, device=self.dev)
            L.check(self.lib.sgpr_predict_reg_dev(self.family, m, p(dq), p(dP), self.n0, p(self.x), p(self.y),
                                                  L.dptr(self.hyp), len(self.hyp), p(self.alpha), p(out), st),
                    "sgpr_predict_reg_dev")
            return (out.cpu().numpy(),)
        o1 = torch.empty(m, dtype=torch.float64, device=self.dev)
        o2 = torch.empty(m, dtype=torch.float64, device=self.dev)
        L.check(self.lib.sgpr_predict_rows_dev(self.family, m, p(dq), p(dP), self.n0, p(self.x), p(self.y),
                                               L.dptr(self.hyp), len(self.hyp), p(self.alpha), p(o1), p(o2), st),
                "sgpr_predict_rows_dev")
        return o1.cpu().numpy(), o2.cpu().numpy()


def solve_implicit_P(pred, pred_guess, q, p, tol=1e-13, maxiter=60):
    """Batched root of f(P) = pGP(q, P) - p + P (Eq. (42); `target` in sympgpr.f90:112-124) for
    all test points at once: the reference runs MINPACK hybrd1 (tol 1e-13) per point from the
    regular-GP guess (sympgpr.f90:103-108); this is the same fixed point found by a vectorised
    secant iteration whose every residual evaluation is one batched device call.
    Points that fail to converge come back NaN ("orbit lost", functions/func.py:231-232)."""
    q, p = L.f64(np.atleast_1d(q)), L.f64(np.atleast_1d(p))
    P0 = pred_guess(q, p)[0]                      # guessP
    f = lambda P: pred(q, P)[0] - p + P
    f0 = f(P0)
    P1 = P0 - f0                                  # first step: f'(P) ~ 1 (the map is near identity)
    f1 = f(P1)
    active = np.isfinite(f1)
    for _ in range(maxiter):
        done = np.abs(P1 - P0) <= tol * np.maximum(1.0, np.abs(P1))
        if np.all(done | ~active):
            break
        d = f1 - f0
        step = np.where((d != 0) & ~done, f1 * (P1 - P0) / np.where(d == 0, 1.0, d), 0.0)
        P0, f0 = P1, f1
        P1 = P1 - step
        f1 = f(P1)
        active &= np.isfinite(f1)
    bad = ~np.isfinite(f1) | (np.abs(f1) > 1e-8 * np.maximum(1.0, np.abs(p)))
    return np.where(bad, np.nan, P1)
