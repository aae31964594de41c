"""Mirror of the reference's f2py `kernels` module (generated kernels*.f90): the scalar
functions that enter the Gram matrix, `name_num(x_a, y_a, x_b, y_b, lx, ly[, p])`, evaluated by
the same device code the Gram kernels use (sgpr_kernel_eval_host).  Arguments may be scalars
(-> float, like f2py) or broadcastable arrays (-> array, one batched launch).

The 15 remaining functions of a kernels*.f90 (first / third derivatives, length-scale
derivatives; kernels.f90:12-57,95-231) are only used by build_dK / nll_grad and are not mirrored
yet."""
from . import _lib as L
from . import ops

__all__ = ["kern_num", "d2kdxdx0_num", "d2kdydy0_num", "d2kdxdy0_num"]


def _eval(which, x_a, y_a, x_b, y_b, lx, ly, p):
    fam = ops.get_family()
    if fam == "D":
        if p is None:
            raise TypeError("family D kernels take 7 arguments (x_a, y_a, x_b, y_b, lx, ly, p)")
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
