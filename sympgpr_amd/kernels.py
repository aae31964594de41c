"""Mirror of the reference's f2py `kernels` module (generated kernels*.f90): the scalar
functions that enter the Gram matrix, `name_num(x_a, y_a, x_b, y_b, lx, ly[, p])`, evaluated by
the same device code the Gram kernels use (sgpr_kernel_eval_host).  Arguments may be scalars
(-> float, like f2py) or broadcastable arrays (-> array, one batched launch).

All 19 functions of a generated kernels*.f90 are here: the four above, the eight length-scale
derivatives build_dK / build_dKreg call (kernels.f90:133-231; for the sum kernel, kernels_sum.f90:133-208,
straight from the code generator tools/gen_kernels.py), and the seven no caller in the
reference uses (dkdx, dkdy, dkdx0, dkdy0 and the three d3k...dy0 functions, kernels.f90:12-57,95-132)."""
from . import _lib as L
from . import ops

__all__ = ["kern_num", "d2kdxdx0_num", "d2kdydy0_num", "d2kdxdy0_num", "dkdlx_num", "dkdly_num",
           "d3kdxdx0dlx_num", "d3kdydy0dlx_num", "d3kdxdy0dlx_num", "d3kdxdx0dly_num", "d3kdydy0dly_num",
           "d3kdxdy0dly_num", "dkdx_num", "dkdy_num", "dkdx0_num", "dkdy0_num", "d3kdxdx0dy0_num",
           "d3kdydy0dy0_num", "d3kdxdy0dy0_num"]


def _eval(which, x_a, y_a, x_b, y_b, lx, ly, p):
    fam = ops.get_family()
    if L.family_has_p(fam):
        if p is None:
            raise TypeError("family %s kernels take 7 arguments (x_a, y_a, x_b, y_b, lx, ly, p)" % fam)
        l = (lx, ly, p)
    else:
        if p is not None:
            raise TypeError("this kernel family takes 6 arguments (x_a, y_a, x_b, y_b, lx, ly)")
        l = (lx, ly)
    return ops.kernel_eval(which, x_a, y_a, x_b, y_b, l, family=fam)


def kern_num(x_a, y_a, x_b, y_b, lx, ly, p=None):          # kernels.f90:1-11
    return _eval(L.K_KERN, x_a, y_a, x_b, y_b, lx, ly, p)


def d2kdxdx0_num(x_a, y_a, x_b, y_b, lx, ly, p=None):      # kernels.f90:58-70
    return _eval(L.K_DXDX0, x_a, y_a, x_b, y_b, lx, ly, p)


def d2kdydy0_num(x_a, y_a, x_b, y_b, lx, ly, p=None):      # kernels.f90:71-82
    return _eval(L.K_DYDY0, x_a, y_a, x_b, y_b, lx, ly, p)


def d2kdxdy0_num(x_a, y_a, x_b, y_b, lx, ly, p=None):      # kernels.f90:83-94
    return _eval(L.K_DXDY0, x_a, y_a, x_b, y_b, lx, ly, p)


def dkdlx_num(x_a, y_a, x_b, y_b, lx, ly, p=None):         # kernels.f90:133-143
    return _eval(L.K_KERN | L.K_DLX, x_a, y_a, x_b, y_b, lx, ly, p)


def dkdly_num(x_a, y_a, x_b, y_b, lx, ly, p=None):         # kernels.f90:144-154
    return _eval(L.K_KERN | L.K_DLY, x_a, y_a, x_b, y_b, lx, ly, p)


def d3kdxdx0dlx_num(x_a, y_a, x_b, y_b, lx, ly, p=None):   # kernels.f90:155-168
    return _eval(L.K_DXDX0 | L.K_DLX, x_a, y_a, x_b, y_b, lx, ly, p)


def d3kdydy0dlx_num(x_a, y_a, x_b, y_b, lx, ly, p=None):   # kernels.f90:169-180
    return _eval(L.K_DYDY0 | L.K_DLX, x_a, y_a, x_b, y_b, lx, ly, p)


def d3kdxdy0dlx_num(x_a, y_a, x_b, y_b, lx, ly, p=None):   # kernels.f90:181-193
    return _eval(L.K_DXDY0 | L.K_DLX, x_a, y_a, x_b, y_b, lx, ly, p)


def d3kdxdx0dly_num(x_a, y_a, x_b, y_b, lx, ly, p=None):   # kernels.f90:194-206
    return _eval(L.K_DXDX0 | L.K_DLY, x_a, y_a, x_b, y_b, lx, ly, p)


def d3kdydy0dly_num(x_a, y_a, x_b, y_b, lx, ly, p=None):   # kernels.f90:207-218
    return _eval(L.K_DYDY0 | L.K_DLY, x_a, y_a, x_b, y_b, lx, ly, p)


def d3kdxdy0dly_num(x_a, y_a, x_b, y_b, lx, ly, p=None):   # kernels.f90:219-231
    return _eval(L.K_DXDY0 | L.K_DLY, x_a, y_a, x_b, y_b, lx, ly, p)


def dkdx_num(x_a, y_a, x_b, y_b, lx, ly, p=None):           # kernels.f90:12-23
    return _eval(L.K_DX, x_a, y_a, x_b, y_b, lx, ly, p)


def dkdy_num(x_a, y_a, x_b, y_b, lx, ly, p=None):           # kernels.f90:24-34
    return _eval(L.K_DY, x_a, y_a, x_b, y_b, lx, ly, p)


def dkdx0_num(x_a, y_a, x_b, y_b, lx, ly, p=None):          # kernels.f90:35-46
    return _eval(L.K_DX0, x_a, y_a, x_b, y_b, lx, ly, p)


def dkdy0_num(x_a, y_a, x_b, y_b, lx, ly, p=None):          # kernels.f90:47-57
    return _eval(L.K_DY0, x_a, y_a, x_b, y_b, lx, ly, p)


def d3kdxdx0dy0_num(x_a, y_a, x_b, y_b, lx, ly, p=None):    # kernels.f90:95-107
    return _eval(L.K_DXDX0DY0, x_a, y_a, x_b, y_b, lx, ly, p)


def d3kdydy0dy0_num(x_a, y_a, x_b, y_b, lx, ly, p=None):    # kernels.f90:108-119
    return _eval(L.K_DYDY0DY0, x_a, y_a, x_b, y_b, lx, ly, p)


def d3kdxdy0dy0_num(x_a, y_a, x_b, y_b, lx, ly, p=None):    # kernels.f90:120-132
    return _eval(L.K_DXDY0DY0, x_a, y_a, x_b, y_b, lx, ly, p)
