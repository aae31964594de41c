// pair_eval.h -- the per-pair formulas of the four kernel families (the generated scalar functions of
// the reference's kernels*.f90, re-expressed with one exp + one sincos per pair), shared by gram.hip and
// batch.hip.  generated/pair_generated.h (tools/gen_kernels.py, from the sympy definition) holds the
// machine-derived forms these are checked against (tests/test_gpu_generated.py).
#pragma once
#include "common.h"
#include "devmath.h"
#include "generated/pair_generated.h"

namespace sgpr {
namespace pairf {

// One pair: a = column ("0") point, b = row point -- the argument order build_K uses
// (sympgpr.f90:27-34: f(x0(j), y0(j), x(i), y(i), ...)).
template <int FAM, bool OCML>
__device__ __forceinline__ void pair_eval(double xa, double ya, double xb, double yb,
                                          const KConst &kc, double &kxx, double &kxy, double &kyy)
{
    if constexpr (FAM == SGPR_FAM_USER) {
        // the user's kernel: generated code only (tools/gen_kernels.py)
        double o[4];
        gen::pair<SGPR_FAM_USER>(xa, ya, xb, yb, kc.lx, kc.ly, kc.p, o);
        kxx = kc.sig * o[1];
        kyy = kc.sig * o[2];
        kxy = kc.sig * o[3];
        return;
    }
    const double dy = ya - yb;
    const double dy2 = dy * dy;
    if constexpr (FAM == SGPR_FAM_C) {
        // kernels_sq.f90:55-87
        const double dx = xa - xb;
        const double dx2 = dx * dx;
        const double E = exp_sel<OCML>(-0.5 * (dy2 * kc.inv_ly2) - 0.5 * (dx2 * kc.inv_lx2));
        kxx = kc.cxx * (kc.lx2 - dx2) * E;
        kyy = kc.cyy * (kc.ly2 - dy2) * E;
        kxy = kc.cxy * (dx * dy) * E;
    } else {
        // A: h = 0.5 x_a - 0.5 x_b (kernels.f90:66-69); D: h = p (x_a - x_b)
        // (implicit_period_unknown/kernels.f90:72-74).  Scaling by 0.5 is exact, so the two
        // forms round identically for A.
        const double h = kc.hscale * (xa - xb);
        double s, c;
        sincos_sel<OCML>(h, s, c);
        const double s2 = s * s;
        const double sc = s * c;
        const double cos2h = __builtin_fma(-2.0, s2, 1.0);  // cos(x_a - x_b) resp. cos(2p dx)
        if constexpr (FAM == SGPR_FAM_B) {
            // kernels_sum.f90:58-88: the q and P factors separate, mixed block is zero.
            const double Ex = exp_sel<OCML>(-0.5 * (s2 * kc.inv_lx2));
            // the reference writes the P exponent expanded (kernels_sum.f90:9,76); keep its
            // operation order (no contraction) so the cancellation error is the same one.
            const double t = __dadd_rn(__dadd_rn(__dmul_rn(-0.5, __dmul_rn(ya, ya)),
                                                 __dmul_rn(1.0, __dmul_rn(ya, yb))),
                                       -__dmul_rn(0.5, __dmul_rn(yb, yb)));
            const double Ey = exp_sel<OCML>(t * kc.inv_ly2);
            kxx = kc.cxx * (kc.lx2 * cos2h - sc * sc) * Ex;
            kyy = kc.cyy * (kc.ly2 - dy2) * Ey;
            kxy = 0.0;
        } else {
            const double E = exp_sel<OCML>(-0.5 * (dy2 * kc.inv_ly2) - 0.5 * (s2 * kc.inv_lx2));
            kxx = kc.cxx * (kc.lx2 * cos2h - sc * sc) * E;
            kyy = kc.cyy * (kc.ly2 - dy2) * E;
            kxy = kc.cxy * (dy * sc) * E;
        }
    }
}

// d/dlx (DL = 1) or d/dly (DL = 2) of the three Hessian entries: the third-derivative kernels
// d3kd..dl._num of kernels.f90:133-231 / kernels_sq.f90:146-217 in factored form.  With
// E = exp(-u/2lx^2 - v/2ly^2), u = sin^2 h (A, D) or dx^2 (C), v = dy^2:
//   d(g(l) E)/dlx = g'(lx) E + g E u/lx^3,   d(.)/dly = ... + g E v/ly^3.
template <int FAM, int DL>
__device__ __forceinline__ void pair_eval_d(double xa, double ya, double xb, double yb,
                                            const KConst &kc, double &dxx, double &dxy, double &dyy)
{
    if constexpr (FAM == SGPR_FAM_B || FAM == SGPR_FAM_USER) {
        // the sum kernel's dl-functions (kernels_sum.f90:133-208) come straight from the generator
        // (tools/gen_kernels.py): no driver differentiates this family, nothing to hand-optimise; the user slot has
        // nothing but generated code
        double o[4];
        if constexpr (DL == DERIV_LX) gen::pair_dlx<FAM>(xa, ya, xb, yb, kc.lx, kc.ly, kc.p, o);
        else                          gen::pair_dly<FAM>(xa, ya, xb, yb, kc.lx, kc.ly, kc.p, o);
        dxx = kc.sig * o[1];
        dyy = kc.sig * o[2];
        dxy = kc.sig * o[3];
        return;
    }
    const double dy = ya - yb;
    const double v = dy * dy;
    double u, E, kxx, kxy, kyy, gp;  // gp = g'(lx)/gxx for the xx entry
    if constexpr (FAM == SGPR_FAM_C) {
        const double dx = xa - xb;
        u = dx * dx;
        E = exp_fast(-0.5 * (v * kc.inv_ly2) - 0.5 * (u * kc.inv_lx2));
        kxx = kc.cxx * (kc.lx2 - u) * E;
        kxy = kc.cxy * (dx * dy) * E;
        gp = (-2.0 + 4.0 * u * kc.inv_lx2) * kc.inv_lx3;           // d/dlx (1/lx^2 - u/lx^4)
    } else {
        const double h = kc.hscale * (xa - xb);
        double s, c;
        sincos_fast(h, s, c);
        u = s * s;
        const double sc = s * c;
        const double cos2h = __builtin_fma(-2.0, u, 1.0);
        E = exp_fast(-0.5 * (v * kc.inv_ly2) - 0.5 * (u * kc.inv_lx2));
        kxx = kc.cxx * (kc.lx2 * cos2h - sc * sc) * E;
        kxy = kc.cxy * (dy * sc) * E;
        gp = (-2.0 * cos2h + 4.0 * (sc * sc) * kc.inv_lx2) * kc.inv_lx3;  // d/dlx (cos2h/lx^2 - sc^2/lx^4)
    }
    kyy = kc.cyy * (kc.ly2 - v) * E;
    if constexpr (DL == DERIV_LX) {
        const double w = u * kc.inv_lx3;
        dxx = __builtin_fma(kxx, w, kc.gxx * gp * E);
        dyy = kyy * w;
        dxy = kxy * (w - 2.0 * kc.inv_lx);
    } else {
        const double w = v * kc.inv_ly3;
        dxx = kxx * w;
        dyy = __builtin_fma(kyy, w, kc.sig * ((-2.0 + 4.0 * v * kc.inv_ly2) * kc.inv_ly3) * E);
        dxy = kxy * (w - 2.0 * kc.inv_ly);
    }
}

// dk/dlx, dk/dly (dkdlx_num, dkdly_num: kernels.f90:135-154), without sig
template <int FAM, int DL>
__device__ __forceinline__ double kern_eval_d(double xa, double ya, double xb, double yb, const KConst &kc)
{
    if constexpr (FAM == SGPR_FAM_B || FAM == SGPR_FAM_USER) {
        double o[4];
        if constexpr (DL == DERIV_LX) gen::pair_dlx<FAM>(xa, ya, xb, yb, kc.lx, kc.ly, kc.p, o);
        else                          gen::pair_dly<FAM>(xa, ya, xb, yb, kc.lx, kc.ly, kc.p, o);
        return o[0];
    }
    const double dy = ya - yb;
    const double v = dy * dy;
    double u;
    if constexpr (FAM == SGPR_FAM_C) {
        const double dx = xa - xb;
        u = dx * dx;
    } else {
        double s, c;
        sincos_fast(kc.hscale * (xa - xb), s, c);
        u = s * s;
    }
    const double E = exp_fast(-0.5 * (v * kc.inv_ly2) - 0.5 * (u * kc.inv_lx2));
    return DL == DERIV_LX ? E * u * kc.inv_lx3 : E * v * kc.inv_ly3;
}

// scalar kernel k(a, b) (kern_num): kernels.f90:1-11 and variants
template <int FAM, bool OCML>
__device__ __forceinline__ double kern_eval(double xa, double ya, double xb, double yb,
                                            const KConst &kc)
{
    if constexpr (FAM == SGPR_FAM_USER) {
        double o[4];
        gen::pair<SGPR_FAM_USER>(xa, ya, xb, yb, kc.lx, kc.ly, kc.p, o);
        return o[0];
    }
    const double dy = ya - yb;
    if constexpr (FAM == SGPR_FAM_C) {
        const double dx = xa - xb;
        return exp_sel<OCML>(-0.5 * (dy * dy * kc.inv_ly2) - 0.5 * (dx * dx * kc.inv_lx2));
    } else {
        const double h = kc.hscale * (xa - xb);
        double s, c;
        sincos_sel<OCML>(h, s, c);
        if constexpr (FAM == SGPR_FAM_B) {
            const double t = __dadd_rn(__dadd_rn(__dmul_rn(-0.5, __dmul_rn(ya, ya)),
                                                 __dmul_rn(1.0, __dmul_rn(ya, yb))),
                                       -__dmul_rn(0.5, __dmul_rn(yb, yb)));
            return exp_sel<OCML>(t * kc.inv_ly2) + exp_sel<OCML>(-0.5 * (s * s * kc.inv_lx2));
        } else {
            return exp_sel<OCML>(-0.5 * (dy * dy * kc.inv_ly2) - 0.5 * (s * s * kc.inv_lx2));
        }
    }
}

template <int FAM, bool OCML, int DL>
__device__ __forceinline__ void pair_any(double xa, double ya, double xb, double yb, const KConst &kc,
                                         double &kxx, double &kxy, double &kyy)
{
    if constexpr (DL == DERIV_NONE) pair_eval<FAM, OCML>(xa, ya, xb, yb, kc, kxx, kxy, kyy);
    else pair_eval_d<FAM, DL>(xa, ya, xb, yb, kc, kxx, kxy, kyy);
}
template <int FAM, bool OCML, int DL>
__device__ __forceinline__ double kern_any(double xa, double ya, double xb, double yb, const KConst &kc)
{
    if constexpr (DL == DERIV_NONE) return kern_eval<FAM, OCML>(xa, ya, xb, yb, kc);
    else return kern_eval_d<FAM, DL>(xa, ya, xb, yb, kc);
}


}  // namespace pairf
}  // namespace sgpr
