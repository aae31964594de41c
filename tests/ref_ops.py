"""NumPy/SciPy stand-in for sympgpr_amd.dist.HipOps -- TEST INFRASTRUCTURE ONLY.

Lets the multi-process CPU tests (gloo, world_size > 1) run the block-cyclic driver's
distribution logic without a GPU.  The Gram blocks come from the CPU oracle; the dense block
operations are the LAPACK/BLAS calls they stand for."""
import numpy as np
import scipy.linalg
import torch

from oracle.oracle import Oracle


def _mat(t, off, m, n, ld):
    a = t.numpy()
    return np.lib.stride_tricks.as_strided(a[off:], shape=(m, n), strides=(8, 8 * ld))


class RefOps:
    device_type = "cpu"
    device = torch.device("cpu")

    def __init__(self):
        self.oracle = Oracle()

    def empty(self, n, dtype=torch.float64):
        return torch.full((n,), float("nan"), dtype=dtype) if dtype == torch.float64 else torch.zeros(n, dtype=dtype)

    def zeros(self, n, dtype=torch.float64):
        return torch.zeros(n, dtype=dtype)

    def work_size(self, nb):
        return 8

    def gram_pairs(self, fam, mi, mj, xb, yb, xa, ya, hyp, A, offs, ld, flags):
        K = self.oracle.build_K(fam, xb.numpy(), yb.numpy(), xa.numpy(), ya.numpy(), hyp)
        for (r0, c0), off in zip(((0, 0), (mi, 0), (0, mj), (mi, mj)), offs):
            if off is not None:
                _mat(A, off, mi, mj, ld)[:, :] = K[r0:r0 + mi, c0:c0 + mj]

    def gram_nd(self, fam, d, mi, mj, Xb, Xa, hyp, A, ld):
        D = 2 * d
        K = self.oracle.build_K_nd(fam, Xb.numpy().reshape(D, mi).T, Xa.numpy().reshape(D, mj).T, hyp)
        _mat(A, 0, D * mi, D * mj, ld)[:, :] = K

    def gram_nd_sel(self, fam, d, mi, mj, Xb, Xa, hyp, A, ld, roff, coff):
        D = 2 * d
        K = self.oracle.build_K_nd(fam, Xb.numpy().reshape(D, mi).T, Xa.numpy().reshape(D, mj).T, hyp)
        for a in range(D):
            for b in range(D):
                if roff[a] >= 0 and coff[b] >= 0:
                    _mat(A, roff[a] + coff[b] * ld, mi, mj, ld)[:, :] = K[a * mi:(a + 1) * mi, b * mj:(b + 1) * mj]

    def potrf(self, nb, A, work, info):
        M = _mat(A, 0, nb, nb, nb)
        Lf, inf = scipy.linalg.lapack.dpotrf(np.tril(M), lower=1)
        info[0] = inf
        if inf == 0:
            il = np.tril_indices(nb)
            M[il] = Lf[il]

    def trsm(self, m, nb, Lkk, work, B, boff, ldb):
        Lm = np.tril(_mat(Lkk, 0, nb, nb, nb))
        Bv = _mat(B, boff, m, nb, ldb)
        Bv[:, :] = scipy.linalg.solve_triangular(Lm, Bv.T, lower=True).T

    def gemm_nt(self, m, n, k, alpha, A, aoff, lda, B, boff, ldb, beta, Cm, coff, ldc):
        Cv = _mat(Cm, coff, m, n, ldc)
        Cv[:, :] = beta * Cv + alpha * (_mat(A, aoff, m, k, lda) @ _mat(B, boff, n, k, ldb).T)

    def gemm_nn(self, m, n, k, alpha, A, aoff, lda, B, boff, ldb, beta, Cm, coff, ldc):
        Cv = _mat(Cm, coff, m, n, ldc)
        Cv[:, :] = beta * Cv + alpha * (_mat(A, aoff, m, k, lda) @ _mat(B, boff, k, n, ldb))

    def trsm_rows(self, m, nb, Lkk, work, B, boff, ldb, trans):
        Lm = np.tril(_mat(Lkk, 0, nb, nb, nb))
        Bv = _mat(B, boff, m, nb, ldb)
        # rows of B are right-hand sides: B L^-T solves L x = b, B L^-1 solves L^T x = b
        Bv[:, :] = scipy.linalg.solve_triangular(Lm, Bv.T, lower=True, trans=1 if trans else 0).T

    def copy_blocks(self, rows, cols, cnt, src, soff, lds, sstep, dst, doff, ldd, dstep):
        for i in range(cnt):
            _mat(dst, doff + i * dstep, rows, cols, ldd)[:, :] = _mat(src, soff + i * sstep, rows, cols, lds)

    def trsv(self, nb, Lkk, work, b, trans):
        Lm = np.tril(_mat(Lkk, 0, nb, nb, nb))
        v = b.numpy()
        v[:] = scipy.linalg.solve_triangular(Lm, v, lower=True, trans=1 if trans else 0)

    def gemv_sub(self, trans, m, k, A, aoff, lda, x, y):
        Av = _mat(A, aoff, m, k, lda)
        yv = y.numpy()
        if trans:
            yv[:k] -= Av.T @ x.numpy()[:m]
        else:
            yv[:m] -= Av @ x.numpy()[:k]

    def sync(self):
        pass
