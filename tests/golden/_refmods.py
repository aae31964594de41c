"""Build-container helpers of the fixture generators (make_flow_golden.py): module objects that stand where the
reference's Python expects its f2py extension modules, every function forwarding -- through ctypes -- into the
REFERENCE'S OWN compiled Fortran under oracle/_ref (built from /root/reference by `make -C oracle ref`).

  kernels_module(lib, src)   `kernels` / `kernels_sq`: name_num(x_a, y_a, x_b, y_b, lx, ly) of a generated kernels*.f90
  sympgpr_module()           `sympgpr.sympgpr`: build_k, buildkreg, guessp, calcq, calcp          (sympgpr.f90:12-126)
  fieldlines_module()        `fieldlines.fieldlines`: init, ath, timestep, compute_r              (fieldlines.f90)
  load_reference(path, name, modules)   import one of the reference's func.py files with those modules in sys.modules

Nothing here is imported by the tests or the product; nothing of the reference is copied -- its files are compiled and
imported where they lie."""
import ctypes as C
import importlib.util
import os
import re
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
_dp = C.POINTER(C.c_double)
_p = lambda a: a.ctypes.data_as(_dp)
_f = lambda a: np.ascontiguousarray(a, dtype=np.float64)

SCALAR_NAMES = ["kern_num", "dkdx_num", "dkdy_num", "dkdx0_num", "dkdy0_num", "d2kdxdx0_num", "d2kdydy0_num",
                "d2kdxdy0_num", "d3kdxdx0dy0_num", "d3kdydy0dy0_num", "d3kdxdy0dy0_num", "dkdlx_num", "dkdly_num",
                "d3kdxdx0dlx_num", "d3kdydy0dlx_num", "d3kdxdy0dlx_num", "d3kdxdx0dly_num", "d3kdydy0dly_num",
                "d3kdxdy0dly_num"]


def kernels_module(libname, src, modname="kernels"):
    """libname: oracle/_ref/libkernels_<X>.so; src: the .f90 it was compiled from (read as text only to learn which
    of its functions sympy's codegen declared INTEGER*4 -- the identically-zero ones of the sum kernel)."""
    lib = C.CDLL(os.path.join(REFDIR, libname))
    decl = dict((m.group(2).lower(), m.group(1).upper()) for m in
                re.finditer(r"^\s*(REAL\*8|INTEGER\*4)\s+function\s+(\w+)", open(src).read(), re.M | re.I))
    mod = types.ModuleType(modname)

    def bind(name):
        f = getattr(lib, name + "_")
        f.restype = C.c_int if decl.get(name, "REAL*8") == "INTEGER*4" else C.c_double
        f.argtypes = [_dp] * 6

        def call(xa, ya, xb, yb, lx, ly):
            a = [C.c_double(float(np.ravel(v)[0])) for v in (xa, ya, xb, yb, lx, ly)]
            return f(*[C.byref(v) for v in a])
        call.__name__ = name
        return call
    names = [n for n in SCALAR_NAMES if n in decl]
    for n in names:
        setattr(mod, n, bind(n))
    mod.__all__ = names
    return mod


def _fcol(a, cache={}):
    """a 2-D array as the f2py wrapper hands it to Fortran: element (i, j) = a[i, j], column-major"""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2 or a.flags.f_contiguous:
        return a
    key = (id(a), a.shape, a.__array_interface__["data"][0])
    hit = cache.get(key)
    if hit is None or not np.array_equal(hit[0], a):
        if len(cache) > 64:
            cache.clear()
        hit = cache[key] = (a.copy(), np.asfortranarray(a))
    return hit[1]


def sympgpr_module(which="A"):
    lib = C.CDLL(os.path.join(REFDIR, "libsympgpr_ref_%s.so" % which))
    for f in (lib.ref_build_k, lib.ref_buildkreg):
        f.restype = None
        f.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
    lib.ref_guessp.restype = C.c_double
    lib.ref_guessp.argtypes = [_dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp]
    lib.ref_calcq.restype = C.c_double
    lib.ref_calcq.argtypes = [_dp, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]
    lib.ref_calcp.restype = C.c_double
    lib.ref_calcp.argtypes = [_dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp]

    class _S:
        @staticmethod
        def _inout(K, shape):
            # f2py's intent(inout): float64, Fortran-contiguous, exact shape -- anything else is an error there too
            if not (isinstance(K, np.ndarray) and K.dtype == np.float64 and K.flags.f_contiguous and K.shape == shape):
                raise ValueError("failed in converting 6th argument `k' of sympgpr.build_k to C/Fortran array")

        @staticmethod
        def build_k(x, y, x0, y0, hyp, K):
            x, y, x0, y0, hyp = map(_f, (x, y, x0, y0, hyp))
            _S._inout(K, (2 * len(x), 2 * len(x0)))
            lib.ref_build_k(len(x), len(x0), _p(x), _p(y), _p(x0), _p(y0), _p(hyp), _p(K))

        @staticmethod
        def buildkreg(x, y, x0, y0, hyp, K):
            x, y, x0, y0, hyp = map(_f, (x, y, x0, y0, hyp))
            _S._inout(K, (len(x), len(x0)))
            lib.ref_buildkreg(len(x), len(x0), _p(x), _p(y), _p(x0), _p(y0), _p(hyp), _p(K))

        @staticmethod
        def guessp(x, y, hypp, xtrainp, ytrainp, ztrainp, Kyinvp):
            x, y, hypp, xtrainp, ytrainp, ztrainp = map(_f, (np.ravel(x), np.ravel(y), hypp, xtrainp, ytrainp, ztrainp))
            Ki = _fcol(Kyinvp)
            return lib.ref_guessp(_p(x), _p(y), _p(hypp), len(xtrainp), _p(xtrainp), _p(ytrainp), _p(ztrainp), _p(Ki))

        @staticmethod
        def calcq(x, y, xtrain, ytrain, l, Kyinv, ztrain):
            x, y, xtrain, ytrain, l, ztrain = map(_f, (np.ravel(x), np.ravel(y), xtrain, ytrain, l, ztrain))
            Ki = _fcol(Kyinv)
            return lib.ref_calcq(_p(x), _p(y), len(xtrain), _p(xtrain), _p(ytrain), _p(l), _p(Ki), _p(ztrain))

        @staticmethod
        def calcp(x, y, l, hypp, xtrainp, ytrainp, ztrainp, Kyinvp, xtrain, ytrain, ztrain, Kyinv):
            x, y, l, hypp, xtrainp, ytrainp, ztrainp, xtrain, ytrain, ztrain = map(
                _f, (np.ravel(x), np.ravel(y), l, hypp, xtrainp, ytrainp, ztrainp, xtrain, ytrain, ztrain))
            Kip, Ki = _fcol(Kyinvp), _fcol(Kyinv)
            return lib.ref_calcp(_p(x), _p(y), _p(l), _p(hypp), len(xtrainp), _p(xtrainp), _p(ytrainp), _p(ztrainp), _p(Kip),
                                 len(xtrain), _p(xtrain), _p(ytrain), _p(ztrain), _p(Ki))
    mod = types.ModuleType("sympgpr")
    mod.sympgpr = _S
    return mod


def fieldlines_module(which="A"):
    lib = C.CDLL(os.path.join(REFDIR, "libsympgpr_ref_%s.so" % which))
    lib.ref_fl_init.restype = None
    lib.ref_fl_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    lib.ref_fl_ath.restype = C.c_double
    lib.ref_fl_ath.argtypes = [C.c_double] * 3
    lib.ref_fl_timestep.restype = None
    lib.ref_fl_timestep.argtypes = [_dp]
    lib.ref_compute_r.restype = C.c_double
    lib.ref_compute_r.argtypes = [_dp, C.c_double]

    class _F:
        @staticmethod
        def init(nph, am, an, aeps, aphase, arlast):
            lib.ref_fl_init(int(nph), int(am), int(an), float(aeps), float(aphase), float(arlast))

        @staticmethod
        def ath(r, th, ph):
            return lib.ref_fl_ath(float(r), float(th), float(ph))

        @staticmethod
        def timestep(z):
            # intent(inout) z(3): a contiguous float64 view is updated in place, as through f2py
            assert isinstance(z, np.ndarray) and z.dtype == np.float64 and z.shape == (3,) and z.flags.c_contiguous
            lib.ref_fl_timestep(_p(z))

        @staticmethod
        def compute_r(z, rstart):
            z = _f(z)
            return lib.ref_compute_r(_p(z), float(rstart))
    mod = types.ModuleType("fieldlines")
    mod.fieldlines = _F
    return mod


def load_reference(path, name, modules):
    """import the reference file `path` as module `name` with `modules` (dict) visible to its import statements"""
    saved = {k: sys.modules.get(k) for k in modules}
    sys.modules.update(modules)
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ref
