// devmath.h -- fp64 exp / sincos for the Gram kernels (gfx950 has no fp64 transcendental
// hardware; everything is FMA chains on the fp64 VALU, so instruction count is what matters).
//
// The reference evaluates exp/sin/cos three times per matrix ENTRY (kernels.f90:58-94); the
// Gram kernel needs one exp and one sincos per PAIR.  The in-house versions below are ~17 and
// ~30 fp64 ops against ~110 for the device-libs pair (measured from the .s), with < 1 ulp
// error on the reduced argument.  SGPR_G_OCML switches back to device-libs for A/B checks.
#pragma once
#ifdef SGPR_HOST_MATH_TEST  // tests/host_math_check.cpp compiles the same source with g++
#include <cmath>
#define SGPR_DEV static inline
#else
#include <hip/hip_runtime.h>
#define SGPR_DEV __device__ __forceinline__
#endif

namespace sgpr {

// exp(x) for the x <= 0 arguments of the kernels (works for any finite x; NaN stays NaN,
// x < -800 -> 0 through ldexp underflow).
SGPR_DEV double exp_fast(double x)
{
    x = (x < -800.0) ? -800.0 : x;  // NaN compares false and is kept
    const double L2E = 1.44269504088896338700e+00;
    const double LN2_HI = 6.93147180369123816490e-01;
    const double LN2_LO = 1.90821492927058770002e-10;
    const double n = __builtin_rint(x * L2E);
    double r = __builtin_fma(-n, LN2_HI, x);
    r = __builtin_fma(-n, LN2_LO, r);
    // Taylor to degree 13 on |r| <= ln2/2: truncation 0.347^14/14! = 4e-18
    double p = 1.60590438368216145994e-10;            // 1/13!
    p = __builtin_fma(p, r, 2.08767569878680989792e-09);  // 1/12!
    p = __builtin_fma(p, r, 2.50521083854417187751e-08);  // 1/11!
    p = __builtin_fma(p, r, 2.75573192239858906526e-07);  // 1/10!
    p = __builtin_fma(p, r, 2.75573192239858906526e-06);  // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873015873e-05);  // 1/8!
    p = __builtin_fma(p, r, 1.98412698412698412698e-04);  // 1/7!
    p = __builtin_fma(p, r, 1.38888888888888888889e-03);  // 1/6!
    p = __builtin_fma(p, r, 8.33333333333333333333e-03);  // 1/5!
    p = __builtin_fma(p, r, 4.16666666666666666667e-02);  // 1/4!
    p = __builtin_fma(p, r, 1.66666666666666666667e-01);  // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// sin / cos on |r| <= pi/4 (+ a little): the classic minimax kernels (Sun fdlibm k_sin / k_cos
// coefficient sets), evaluated with FMAs.
SGPR_DEV double ksin(double x)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x;
    double r = __builtin_fma(z, S6, S5);
    r = __builtin_fma(z, r, S4);
    r = __builtin_fma(z, r, S3);
    r = __builtin_fma(z, r, S2);
    r = __builtin_fma(z, r, S1);
    return __builtin_fma(z * x, r, x);
}
SGPR_DEV double kcos(double x)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    double r = __builtin_fma(z, C6, C5);
    r = __builtin_fma(z, r, C4);
    r = __builtin_fma(z, r, C3);
    r = __builtin_fma(z, r, C2);
    r = __builtin_fma(z, r, C1);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * (z * r));
}

// sin(h), cos(h).  Two-FMA Cody-Waite reduction by pi/2 (the FMA keeps n*PIO2_HI exact, the
// remaining error is |n| * 1.5e-33): fine up to |h| ~ 1e9; beyond that (any lane) the wave
// takes the device-libs path, which does full Payne-Hanek.
SGPR_DEV void sincos_fast(double h, double &s, double &c)
{
    if (__builtin_expect(!(__builtin_fabs(h) < 1.0e9), 0)) {  // also NaN / inf
        ::sincos(h, &s, &c);
        return;
    }
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_MID = 6.12323399573676603587e-17;
    const double n = __builtin_rint(h * TWO_OVER_PI);
    double r = __builtin_fma(-n, PIO2_HI, h);
    r = __builtin_fma(-n, PIO2_MID, r);
    const int q = (int)n;
    const double sr = ksin(r), cr = kcos(r);
    const double s0 = (q & 1) ? cr : sr;
    const double c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

template <bool OCML>
SGPR_DEV double exp_sel(double x)
{
    if constexpr (OCML) return ::exp(x);
    else return exp_fast(x);
}
template <bool OCML>
SGPR_DEV void sincos_sel(double h, double &s, double &c)
{
    if constexpr (OCML) ::sincos(h, &s, &c);
    else sincos_fast(h, s, c);
}

}  // namespace sgpr
