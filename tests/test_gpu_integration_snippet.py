"""The binding INTEGRATION.md section 2 shows a maintainer (raw ctypes on the C ABI, no package
import) really works against the built library."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_binding(oracle):
    import torch  # noqa: F401  one HIP runtime per process: the ROCm torch wheel bundles its own
    _lib = C.CDLL(os.path.join(ROOT, "sympgpr_amd", "lib", "libsympgpr_hip.so"))
    _dp = C.POINTER(C.c_double)
    _sig = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_size_t]
    _lib.sgpr_build_k_host.argtypes = _sig
    _lib.sgpr_build_k_host.restype = C.c_int
    _lib.sgpr_buildkreg_host.argtypes = _sig
    _lib.sgpr_buildkreg_host.restype = C.c_int
    _lib.sgpr_last_error.restype = C.c_char_p
    _lib.sgpr_potrf_host.argtypes = [C.c_int, _dp, C.c_size_t]
    _lib.sgpr_potrs_host.argtypes = [C.c_int, _dp, C.c_size_t, _dp, C.c_size_t, C.c_int]
    FAMILY = 0

    def _p(a):
        return a.ctypes.data_as(_dp)

    def _f(a):
        return np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64)

    class sympgpr:
        @staticmethod
        def build_k(x, y, x0, y0, hyp, K):
            x, y, x0, y0, hyp = map(_f, (x, y, x0, y0, hyp))
            rc = _lib.sgpr_build_k_host(FAMILY, K.shape[0] // 2, K.shape[1] // 2, _p(x), _p(y), _p(x0), _p(y0),
                                        _p(hyp), len(hyp), _p(K), K.shape[0])
            if rc:
                raise RuntimeError(_lib.sgpr_last_error().decode())

        @staticmethod
        def buildkreg(x, y, x0, y0, hyp, K):
            x, y, x0, y0, hyp = map(_f, (x, y, x0, y0, hyp))
            rc = _lib.sgpr_buildkreg_host(FAMILY, K.shape[0], K.shape[1], _p(x), _p(y), _p(x0), _p(y0),
                                          _p(hyp), len(hyp), _p(K), K.shape[0])
            if rc:
                raise RuntimeError(_lib.sgpr_last_error().decode())

    def gpsolve(Ky, ft):
        L = np.array(Ky, order='F')
        rc = _lib.sgpr_potrf_host(L.shape[0], _p(L), L.shape[0])
        if rc > 0:
            raise np.linalg.LinAlgError("%d-th leading minor not positive definite" % rc)
        alpha = np.array(ft, order='F')
        _lib.sgpr_potrs_host(L.shape[0], _p(L), L.shape[0], _p(alpha), L.shape[0], 1)
        return L, alpha

    rng = np.random.default_rng(0)
    x, y, x0, y0 = rng.uniform(0, 6, 50), rng.uniform(-2, 2, 50), rng.uniform(0, 6, 30), rng.uniform(-2, 2, 30)
    hyp = np.array([0.5, 0.8, 1.1])
    K = np.empty((100, 60), order='F')
    sympgpr.build_k(x, y, x0, y0, hyp, K)
    Ko = oracle.build_K("A", x, y, x0, y0, hyp)
    assert np.abs(K - Ko).max() <= 4e-15 * np.abs(Ko).max()
    G = np.empty((50, 30), order='F')
    sympgpr.buildkreg(x, y, x0, y0, hyp, G)
    Go = oracle.buildKreg("A", x, y, x0, y0, hyp)
    assert np.abs(G - Go).max() <= 4e-15 * np.abs(Go).max()
    Ky = oracle.build_K("A", x, y, x, y, hyp) + 1e-2 * np.eye(100)
    z = rng.standard_normal(100)
    L, a = gpsolve(Ky, z)
    Lo = oracle.cholesky(Ky)
    ao = oracle.solve_cholesky(Lo, z)
    assert np.linalg.norm(a - ao) / np.linalg.norm(ao) < 1e-10
    with pytest.raises(np.linalg.LinAlgError):
        gpsolve(-Ky, z)
