/*
 * oracle/sympgpr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C (CPU, fp64) restatement of the SympGPR training hot path, used only as the
 * parity checker in tests/, in __graft_entry__.smoke() and as bench.py's cpu_baseline leg.
 * Nothing under sympgpr_amd/ may import, link or call it.
 *
 * Pinning: every function below is checked in tests/test_oracle.py against
 *   (1) the known-answer inputs of the reference's own test
 *       (python/05_tokamak/SympGPR/test_sympgpr.py:7-10,19) and
 *   (2) golden vectors produced by the reference's own Fortran compiled from
 *       /root/reference with amdflang (oracle/Makefile `ref` target; generator
 *       tests/golden/make_golden.py) plus the same SciPy calls the reference makes for
 *       the factor/solve (python/functions/func.py:165-196).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/).  Column-major everywhere, like the Fortran.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

enum { FAM_A = 0, FAM_B = 1, FAM_C = 2, FAM_D = 3 };
/* which-codes of the scalar evaluator */
enum { W_KERN = 0, W_DXDX0 = 1, W_DYDY0 = 2, W_DXDY0 = 3,
       /* length-scale derivatives (build_dK / build_dKreg, functions/func.py:52-129) */
       W_DKDLX = 4, W_DKDLY = 5, W_DXDX0DLX = 6, W_DYDY0DLX = 7, W_DXDY0DLX = 8,
       W_DXDX0DLY = 9, W_DYDY0DLY = 10, W_DXDY0DLY = 11 };

/* ---- scalar kernels ------------------------------------------------------------------
 * Family A  periodic(q) x SE(P), product:
 *     python/05_tokamak/SympGPR/kernels.f90:1-11 (kern), :58-70 (d2kdxdx0),
 *     :71-82 (d2kdydy0), :83-94 (d2kdxdy0)   [5 md5-identical copies in the tree]
 * Family B  periodic(q) + SE(P), sum (explicit map):
 *     python/01_pendulum/explicit/kernels_sum.f90:1-11, :58-67, :68-78, :79-88 (== 0)
 * Family C  SE x SE:
 *     python/03_henon_heiles/kernels_sq.f90:1-10, :55-65, :66-76, :77-87
 * Family D  periodic with free period p (7-argument functions, l = (lx, ly, p)):
 *     python/01_pendulum/implicit_period_unknown/kernels.f90:1-12, :63-75, :76-88, :89-101
 * The expressions keep the operation order of the generated Fortran.
 */
static double sq(double v) { return v * v; }

double orc_scalar(int fam, int which, double x_a, double y_a, double x_b, double y_b,
                  double lx, double ly, double p)
{
    const double lx2 = lx * lx, ly2 = ly * ly;
    switch (fam) {
    case FAM_A: {
        const double s = sin(0.5 * x_a - 0.5 * x_b), c = cos(0.5 * x_a - 0.5 * x_b);
        const double dy = y_a - y_b;
        switch (which) {
        case W_KERN:
            return exp(-0.5 * sq(dy) / ly2 - 0.5 * sq(s) / lx2);
        case W_DXDX0:
            return 0.25 * (lx2 * cos(1.0 * x_a - 1.0 * x_b) - sq(s) * sq(c)) *
                   exp(0.5 * (-lx2 * sq(dy) - ly2 * sq(s)) / (lx2 * ly2)) / (lx2 * lx2);
        case W_DYDY0:
            return 1.0 * (ly2 - sq(dy)) *
                   exp(0.5 * (-lx2 * sq(dy) - ly2 * sq(s)) / (lx2 * ly2)) / (ly2 * ly2);
        case W_DXDY0:
            return -0.5 * dy * exp(-0.5 * (lx2 * sq(dy) + ly2 * sq(s)) / (lx2 * ly2)) * s * c /
                   (lx2 * ly2);
        }
        break;
    }
    case FAM_B: {
        const double s = sin(0.5 * x_a - 0.5 * x_b), c = cos(0.5 * x_a - 0.5 * x_b);
        const double ey = exp((-0.5 * sq(y_a) + 1.0 * y_a * y_b - 0.5 * sq(y_b)) / ly2);
        switch (which) {
        case W_KERN:
            return ey + exp(-0.5 * sq(s) / lx2);
        case W_DXDX0:
            return (0.25 * lx2 * cos(1.0 * x_a - 1.0 * x_b) - 0.25 * sq(s) * sq(c)) *
                   exp(-0.5 * sq(s) / lx2) / (lx2 * lx2);
        case W_DYDY0:
            return 1.0 * (ly2 - sq(y_a - y_b)) * ey / (ly2 * ly2);
        case W_DXDY0:
            return 0.0;
        }
        break;
    }
    case FAM_C: {
        const double dx = x_a - x_b, dy = y_a - y_b;
        switch (which) {
        case W_KERN:
            return exp(-0.5 * sq(dy) / ly2 - 0.5 * sq(dx) / lx2);
        case W_DXDX0:
            return 1.0 * (lx2 - sq(dx)) *
                   exp(0.5 * (-lx2 * sq(dy) - ly2 * sq(dx)) / (lx2 * ly2)) / (lx2 * lx2);
        case W_DYDY0:
            return 1.0 * (ly2 - sq(dy)) *
                   exp(0.5 * (-lx2 * sq(dy) - ly2 * sq(dx)) / (lx2 * ly2)) / (ly2 * ly2);
        case W_DXDY0:
            return -1.0 * dx * dy * exp(-0.5 * (lx2 * sq(dy) + ly2 * sq(dx)) / (lx2 * ly2)) /
                   (lx2 * ly2);
        }
        break;
    }
    case FAM_D: {
        const double s = sin(p * (x_a - x_b)), c = cos(p * (x_a - x_b));
        const double dy = y_a - y_b;
        switch (which) {
        case W_KERN:
            return exp(-0.5 * sq(dy) / ly2 - 0.5 * sq(s) / lx2);
        case W_DXDX0:
            return 1.0 * sq(p) * (lx2 * cos(2.0 * p * (x_a - x_b)) - sq(s) * sq(c)) *
                   exp(-0.5 * (lx2 * sq(dy) + ly2 * sq(s)) / (lx2 * ly2)) / (lx2 * lx2);
        case W_DYDY0:
            return 1.0 * (ly2 - sq(dy)) *
                   exp(0.5 * (-lx2 * sq(dy) - ly2 * sq(s)) / (lx2 * ly2)) / (ly2 * ly2);
        case W_DXDY0:
            return -1.0 * p * dy * exp(-0.5 * (lx2 * sq(dy) + ly2 * sq(s)) / (lx2 * ly2)) * s *
                   c / (lx2 * ly2);
        }
        break;
    }
    }
    return NAN;
}

/* Length-scale derivatives of the kernel and of its three Hessian entries, families A and C
 * (the ones whose drivers call nll_grad: 02_pert_pendulum/main.py:55, 05_tokamak/SympGPR/main.py,
 * 03_henon_heiles/main.py:155).  Expressions as generated:
 *   A: python/05_tokamak/SympGPR/kernels.f90:135-231 (dkdlx, dkdly, d3kdxdx0dlx, d3kdydy0dlx,
 *      d3kdxdy0dlx, d3kdxdx0dly, d3kdydy0dly, d3kdxdy0dly)
 *   C: python/03_henon_heiles/kernels_sq.f90:124-217
 *   B: python/01_pendulum/explicit/kernels_sum.f90:120-208 */
double orc_scalar_dl(int fam, int which, double x_a, double y_a, double x_b, double y_b,
                     double lx, double ly)
{
    const double lx2 = lx * lx, ly2 = ly * ly;
    const double dy = y_a - y_b;
    if (fam == FAM_A) {
        const double s = sin(0.5 * x_a - 0.5 * x_b), c = cos(0.5 * x_a - 0.5 * x_b);
        const double cd = cos(1.0 * x_a - 1.0 * x_b);
        const double Ek = exp(-0.5 * sq(dy) / ly2 - 0.5 * sq(s) / lx2);
        const double E = exp(-0.5 * (lx2 * sq(dy) + ly2 * sq(s)) / (lx2 * ly2));
        switch (which) {
        case W_DKDLX: return 1.0 * Ek * sq(s) / (lx2 * lx);
        case W_DKDLY: return 1.0 * sq(dy) * Ek / (ly2 * ly);
        case W_DXDX0DLX:
            return (-0.5 * lx2 * lx2 * cd + lx2 * (0.75 * cd + 0.5) * sq(s) - 0.25 * sq(sq(s)) * sq(c)) * E /
                   (lx2 * lx2 * lx2 * lx);
        case W_DYDY0DLX: return 1.0 * (ly2 - sq(dy)) * E * sq(s) / (lx2 * lx * ly2 * ly2);
        case W_DXDY0DLX: return (lx2 - 0.5 * sq(s)) * dy * E * s * c / (lx2 * lx2 * lx * ly2);
        case W_DXDX0DLY: return 0.25 * sq(dy) * (lx2 * cd - sq(s) * sq(c)) * E / (lx2 * lx2 * ly2 * ly);
        case W_DYDY0DLY:
            return (-2.0 * ly2 * ly2 + 5.0 * ly2 * sq(dy) - 1.0 * sq(sq(dy))) * E / (ly2 * ly2 * ly2 * ly);
        case W_DXDY0DLY: return (ly2 - 0.5 * sq(dy)) * dy * E * s * c / (lx2 * ly2 * ly2 * ly);
        }
    } else if (fam == FAM_C) {
        const double dx = x_a - x_b;
        const double Ek = exp(-0.5 * sq(dy) / ly2 - 0.5 * sq(dx) / lx2);
        const double E = exp(-0.5 * (lx2 * sq(dy) + ly2 * sq(dx)) / (lx2 * ly2));
        switch (which) {
        case W_DKDLX: return 1.0 * sq(dx) * Ek / (lx2 * lx);
        case W_DKDLY: return 1.0 * sq(dy) * Ek / (ly2 * ly);
        case W_DXDX0DLX:
            return (-2.0 * lx2 * lx2 + 5.0 * lx2 * sq(dx) - 1.0 * sq(sq(dx))) * E / (lx2 * lx2 * lx2 * lx);
        case W_DYDY0DLX: return 1.0 * (ly2 - sq(dy)) * sq(dx) * E / (lx2 * lx * ly2 * ly2);
        case W_DXDY0DLX: return (2.0 * lx2 - 1.0 * sq(dx)) * dx * dy * E / (lx2 * lx2 * lx * ly2);
        case W_DXDX0DLY: return 1.0 * (lx2 - sq(dx)) * sq(dy) * E / (lx2 * lx2 * ly2 * ly);
        case W_DYDY0DLY:
            return (-2.0 * ly2 * ly2 + 5.0 * ly2 * sq(dy) - 1.0 * sq(sq(dy))) * E / (ly2 * ly2 * ly2 * ly);
        case W_DXDY0DLY: return (2.0 * ly2 - 1.0 * sq(dy)) * dx * dy * E / (lx2 * ly2 * ly2 * ly);
        }
    } else if (fam == FAM_B) {
        /* python/01_pendulum/explicit/kernels_sum.f90:120-208: the sum kernel's factors separate; the
         * Fortran writes (y_a - y_b)^2 expanded (:139, :178-181) -- kept, so the cancellation is the same one */
        const double s = sin(0.5 * x_a - 0.5 * x_b), c = cos(0.5 * x_a - 0.5 * x_b);
        const double cd = cos(1.0 * x_a - 1.0 * x_b);
        const double ex = exp(-0.5 * sq(s) / lx2);
        const double ey = exp((-0.5 * sq(y_a) + 1.0 * y_a * y_b - 0.5 * sq(y_b)) / ly2);
        switch (which) {
        case W_DKDLX: return 1.0 * ex * sq(s) / (lx2 * lx);
        case W_DKDLY: return (sq(y_a) - 2.0 * y_a * y_b + sq(y_b)) * ey / (ly2 * ly);
        case W_DXDX0DLX:
            return (-0.5 * lx2 * lx2 * cd + lx2 * (0.75 * cd + 0.5) * sq(s) - 0.25 * sq(sq(s)) * sq(c)) * ex /
                   (lx2 * lx2 * lx2 * lx);
        case W_DYDY0DLY:
            return (-2.0 * ly2 * ly2 + ly2 * (1.0 * sq(y_a) - 2.0 * y_a * y_b + 1.0 * sq(y_b) + 4.0 * sq(dy)) +
                    sq(dy) * (-1.0 * sq(y_a) + 2.0 * y_a * y_b - 1.0 * sq(y_b))) * ey / (ly2 * ly2 * ly2 * ly);
        case W_DYDY0DLX: case W_DXDY0DLX: case W_DXDX0DLY: case W_DXDY0DLY: return 0.0;   /* INTEGER*4 ... = 0 */
        }
    }
    return NAN;
}

/* build_dK: python/functions/func.py:80-129.  which = 0 (d/dlx) or 1 (d/dly).  Output is
 * (2 n0 x 2 n), column-major: rows index the "0" points, [[k11,k12],[k21,k22]], each
 * sig * d3k(x0[k], y0[k], x[lk], y[lk]). */
int orc_build_dk(int fam, int which, int n, int n0, const double *x, const double *y, const double *x0,
                 const double *y0, const double *hyp, double *dK, size_t ld)
{
    const double lx = hyp[0], ly = hyp[1], sig = hyp[2];
    const int wxx = which ? W_DXDX0DLY : W_DXDX0DLX, wyy = which ? W_DYDY0DLY : W_DYDY0DLX,
              wxy = which ? W_DXDY0DLY : W_DXDY0DLX;
    for (int lk = 0; lk < n; ++lk)
        for (int k = 0; k < n0; ++k) {
            dK[k + (size_t)lk * ld] = sig * orc_scalar_dl(fam, wxx, x0[k], y0[k], x[lk], y[lk], lx, ly);
            dK[n0 + k + (size_t)lk * ld] = sig * orc_scalar_dl(fam, wxy, x0[k], y0[k], x[lk], y[lk], lx, ly);
            dK[k + (size_t)(n + lk) * ld] = sig * orc_scalar_dl(fam, wxy, x0[k], y0[k], x[lk], y[lk], lx, ly);
            dK[n0 + k + (size_t)(n + lk) * ld] = sig * orc_scalar_dl(fam, wyy, x0[k], y0[k], x[lk], y[lk], lx, ly);
        }
    return 0;
}

/* The seven generated functions no caller in the reference uses (first derivatives and the
 * third derivatives with respect to y_b), restated for the completeness of the `kernels` module:
 *   A: python/05_tokamak/SympGPR/kernels.f90:12-57,95-132
 *   B: python/01_pendulum/explicit/kernels_sum.f90:12-55,89-131 (its zero functions are INTEGER*4)
 *   C: python/03_henon_heiles/kernels_sq.f90:11-54,89-123
 *   D: python/01_pendulum/implicit_period_unknown/kernels.f90 (7-argument variants)
 * which: 16 dkdx, 17 dkdy, 18 dkdx0, 19 dkdy0, 20 d3kdxdx0dy0, 21 d3kdydy0dy0, 22 d3kdxdy0dy0 */
enum { X_DKDX = 16, X_DKDY = 17, X_DKDX0 = 18, X_DKDY0 = 19, X_DXDX0DY0 = 20, X_DYDY0DY0 = 21, X_DXDY0DY0 = 22 };
double orc_scalar_x(int fam, int which, double x_a, double y_a, double x_b, double y_b,
                    double lx, double ly, double p)
{
    const double lx2 = lx * lx, ly2 = ly * ly;
    const double dy = y_a - y_b;
    if (fam == FAM_A || fam == FAM_D) {
        const double h = fam == FAM_A ? 0.5 * x_a - 0.5 * x_b : p * (x_a - x_b);
        const double hs = fam == FAM_A ? 0.5 : 1.0 * p;      /* dh/dx_a */
        const double s = sin(h), c = cos(h);
        const double cd = fam == FAM_A ? cos(1.0 * x_a - 1.0 * x_b) : cos(2.0 * p * (x_a - x_b));
        const double E = exp(-0.5 * (lx2 * sq(dy) + ly2 * sq(s)) / (lx2 * ly2));
        switch (which) {
        case X_DKDX: return -hs * E * s * c / lx2;
        case X_DKDY: return 1.0 * (-y_a + y_b) * E / ly2;
        case X_DKDX0: return hs * E * s * c / lx2;
        case X_DKDY0: return 1.0 * (y_a - y_b) * E / ly2;
        case X_DXDX0DY0: return hs * hs * dy * (lx2 * cd - sq(s) * sq(c)) * E / (lx2 * lx2 * ly2);
        case X_DYDY0DY0: return (3.0 * ly2 - sq(dy)) * dy * E / (ly2 * ly2 * ly2);
        case X_DXDY0DY0: return hs * (1.0 * ly2 - sq(dy)) * E * s * c / (lx2 * ly2 * ly2);
        }
    } else if (fam == FAM_B) {
        const double s = sin(0.5 * x_a - 0.5 * x_b), c = cos(0.5 * x_a - 0.5 * x_b);
        const double ex = exp(-0.5 * sq(s) / lx2);
        const double ey = exp((-0.5 * sq(y_a) + 1.0 * y_a * y_b - 0.5 * sq(y_b)) / ly2);
        switch (which) {
        case X_DKDX: return -0.5 * ex * s * c / lx2;
        case X_DKDY: return 1.0 * (-y_a + y_b) * ey / ly2;
        case X_DKDX0: return 0.5 * ex * s * c / lx2;
        case X_DKDY0: return 1.0 * (y_a - y_b) * ey / ly2;
        case X_DXDX0DY0: return 0.0;
        case X_DYDY0DY0: return (3.0 * ly2 - 1.0 * sq(dy)) * dy * ey / (ly2 * ly2 * ly2);
        case X_DXDY0DY0: return 0.0;
        }
    } else if (fam == FAM_C) {
        const double dx = x_a - x_b;
        const double E = exp(-0.5 * (lx2 * sq(dy) + ly2 * sq(dx)) / (lx2 * ly2));
        switch (which) {
        case X_DKDX: return 1.0 * (-x_a + x_b) * E / lx2;
        case X_DKDY: return 1.0 * (-y_a + y_b) * E / ly2;
        case X_DKDX0: return 1.0 * (x_a - x_b) * E / lx2;
        case X_DKDY0: return 1.0 * (y_a - y_b) * E / ly2;
        case X_DXDX0DY0: return (lx2 - 1.0 * sq(dx)) * dy * E / (lx2 * lx2 * ly2);
        case X_DYDY0DY0: return (3.0 * ly2 - sq(dy)) * dy * E / (ly2 * ly2 * ly2);
        case X_DXDY0DY0: return (1.0 * ly2 - sq(dy)) * dx * E / (lx2 * ly2 * ly2);
        }
    }
    return NAN;
}

/* build_dKreg: python/functions/func.py:52-78.  Output (n x n0): Kp[k,lk] = sig dkdl(x0[lk],y0[lk],x[k],y[k]) */
int orc_build_dkreg(int fam, int which, int n, int n0, const double *x, const double *y, const double *x0,
                    const double *y0, const double *hyp, double *dK, size_t ld)
{
    const double lx = hyp[0], ly = hyp[1], sig = hyp[2];
    for (int lk = 0; lk < n0; ++lk)
        for (int k = 0; k < n; ++k)
            dK[k + (size_t)lk * ld] =
                sig * orc_scalar_dl(fam, which ? W_DKDLY : W_DKDLX, x0[lk], y0[lk], x[k], y[k], lx, ly);
    return 0;
}

/* hyp layout of the reference: (lx, ly, sig) for A/B/C (sympgpr.f90:17, func.py:47-48
 * `l = hyp[:-1]; sig = hyp[-1]`), (lx, ly, p, sig) for D
 * (01_pendulum/implicit_period_unknown/func.py:18-19,45-46). */
static void split_hyp(int fam, const double *hyp, double *lx, double *ly, double *p, double *sig)
{
    *lx = hyp[0];
    *ly = hyp[1];
    if (fam == FAM_D) { *p = hyp[2]; *sig = hyp[3]; }
    else              { *p = 0.0;    *sig = hyp[2]; }
}

/* build_K: python/05_tokamak/SympGPR/sympgpr.f90:12-38 (== pure-Python loop of
 * python/01_pendulum/implicit/func.py:44-64).  K is (2n x 2n0), column-major, leading
 * dimension ldk; a = column ("0") point, b = row point; K *= sig AFTER the fill.
 * `threads` <= 1 reproduces the reference's single-threaded j-outer/i-inner loop order. */
int orc_build_k(int fam, int n, int n0, const double *x, const double *y, const double *x0,
                const double *y0, const double *hyp, double *K, size_t ldk, int threads)
{
    double lx, ly, p, sig;
    split_hyp(fam, hyp, &lx, &ly, &p, &sig);
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int j = 0; j < n0; ++j) {
        for (int i = 0; i < n; ++i) {
            K[i + (size_t)j * ldk] = orc_scalar(fam, W_DXDX0, x0[j], y0[j], x[i], y[i], lx, ly, p);
            K[n + i + (size_t)j * ldk] = orc_scalar(fam, W_DXDY0, x0[j], y0[j], x[i], y[i], lx, ly, p);
            K[i + (size_t)(n0 + j) * ldk] = orc_scalar(fam, W_DXDY0, x0[j], y0[j], x[i], y[i], lx, ly, p);
            K[n + i + (size_t)(n0 + j) * ldk] = orc_scalar(fam, W_DYDY0, x0[j], y0[j], x[i], y[i], lx, ly, p);
        }
    }
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int j = 0; j < 2 * n0; ++j)
        for (int i = 0; i < 2 * n; ++i) K[i + (size_t)j * ldk] *= sig; /* sympgpr.f90:37 */
    return 0;
}

/* buildKreg: python/05_tokamak/SympGPR/sympgpr.f90:40-60. K is (n x n0). */
int orc_buildkreg(int fam, int n, int n0, const double *x, const double *y, const double *x0,
                  const double *y0, const double *hyp, double *K, size_t ldk, int threads)
{
    double lx, ly, p, sig;
    split_hyp(fam, hyp, &lx, &ly, &p, &sig);
    if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
    for (int j = 0; j < n0; ++j)
        for (int i = 0; i < n; ++i)
            K[i + (size_t)j * ldk] =
                sig * orc_scalar(fam, W_KERN, x0[j], y0[j], x[i], y[i], lx, ly, p); /* :54-59 */
    return 0;
}

/* Ky = K + |sig2n| I : python/functions/func.py:183,192 */
void orc_add_noise(int n, double *K, size_t ldk, double sig2n)
{
    for (int i = 0; i < n; ++i) K[i + (size_t)i * ldk] += fabs(sig2n);
}

/* Lower Cholesky, in place, column-major (what scipy.linalg.cholesky(Ky, lower=True)
 * -> LAPACK dpotrf computes at python/functions/func.py:166,184,193).  Left-looking
 * column (dpotf2-style) algorithm; the strict upper triangle is zeroed like SciPy does.
 * Returns 0, or k>0 if the leading minor of order k is not positive definite
 * (SciPy raises LinAlgError there). */
int orc_cholesky_lower(int n, double *A, size_t lda)
{
    for (int j = 0; j < n; ++j) {
        double *cj = A + (size_t)j * lda;
        for (int k = 0; k < j; ++k) {
            const double *ck = A + (size_t)k * lda;
            const double ljk = ck[j];
            if (ljk != 0.0)
                for (int i = j; i < n; ++i) cj[i] -= ck[i] * ljk;
        }
        const double d = cj[j];
        if (!(d > 0.0)) return j + 1;
        const double r = sqrt(d);
        cj[j] = r;
        for (int i = j + 1; i < n; ++i) cj[i] /= r;
    }
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i) A[i + (size_t)j * lda] = 0.0;
    return 0;
}

/* alpha = L^-T (L^-1 b): python/functions/func.py:174-177 (two dtrtrs). nrhs columns. */
void orc_solve_cholesky(int n, const double *L, size_t ldl, double *B, size_t ldb, int nrhs)
{
    for (int r = 0; r < nrhs; ++r) {
        double *b = B + (size_t)r * ldb;
        for (int j = 0; j < n; ++j) { /* forward, column-oriented */
            b[j] /= L[j + (size_t)j * ldl];
            const double bj = b[j];
            const double *cj = L + (size_t)j * ldl;
            for (int i = j + 1; i < n; ++i) b[i] -= cj[i] * bj;
        }
        for (int j = n - 1; j >= 0; --j) { /* backward with L^T: dot products */
            const double *cj = L + (size_t)j * ldl;
            double s = b[j];
            for (int i = j + 1; i < n; ++i) s -= cj[i] * b[i];
            b[j] = s / cj[j];
        }
    }
}

/* nll = 0.5 y.alpha + sum log diag L : python/functions/func.py:186,195 */
double orc_nll(int n, const double *L, size_t ldl, const double *y, const double *alpha)
{
    double q = 0.0, ld = 0.0;
    for (int i = 0; i < n; ++i) { q += y[i] * alpha[i]; ld += log(L[i + (size_t)i * ldl]); }
    return 0.5 * q + ld;
}

/* Whole fit as nll_chol does it (python/functions/func.py:189-196): build_K on (x,x),
 * add |sig2n| I, factor, solve.  x,y: the n_pts training inputs; z: 2*n_pts targets.
 * K (2n x 2n, ld 2n) is caller scratch and holds L on return.  Returns the potrf info. */
int orc_fit(int fam, int n_pts, const double *x, const double *y, const double *z,
            const double *hyp, double sig2n, double *K, double *alpha, double *nll, int threads)
{
    const int n = 2 * n_pts;
    orc_build_k(fam, n_pts, n_pts, x, y, x, y, hyp, K, (size_t)n, threads);
    orc_add_noise(n, K, (size_t)n, sig2n);
    int info = orc_cholesky_lower(n, K, (size_t)n);
    if (info) return info;
    memcpy(alpha, z, sizeof(double) * (size_t)n);
    orc_solve_cholesky(n, K, (size_t)n, alpha, (size_t)n, 1);
    if (nll) *nll = orc_nll(n, K, (size_t)n, z, alpha);
    return 0;
}

/* Prediction rows with cached alpha (the O(n) form of sympgpr.f90:75-86 `calcq` and of
 * `target` at :112-124; the reference recomputes matmul(Kyinv, ztrain) = alpha per call).
 * For m test points (q_k, P_k):  out_p[k] = Kstar(1,:).alpha, out_q[k] = Kstar(2,:).alpha
 * with Kstar = build_K(x=q_k, y=P_k, x0=xtrain, y0=ytrain)  (2 x 2N0). */
void orc_predict_rows(int fam, int m, const double *q, const double *P, int n0,
                      const double *xtrain, const double *ytrain, const double *hyp,
                      const double *alpha, double *out_p, double *out_q)
{
    double lx, ly, p, sig;
    split_hyp(fam, hyp, &lx, &ly, &p, &sig);
    for (int k = 0; k < m; ++k) {
        double r1 = 0.0, r2 = 0.0;
        for (int j = 0; j < n0; ++j) {
            const double kxx = sig * orc_scalar(fam, W_DXDX0, xtrain[j], ytrain[j], q[k], P[k], lx, ly, p);
            const double kxy = sig * orc_scalar(fam, W_DXDY0, xtrain[j], ytrain[j], q[k], P[k], lx, ly, p);
            const double kyy = sig * orc_scalar(fam, W_DYDY0, xtrain[j], ytrain[j], q[k], P[k], lx, ly, p);
            r1 += kxx * alpha[j] + kxy * alpha[n0 + j];
            r2 += kxy * alpha[j] + kyy * alpha[n0 + j];
        }
        out_p[k] = r1;
        out_q[k] = r2;
    }
}

/* Regular-GP prediction with cached alpha_p (sympgpr.f90:62-73 `guessP`). */
void orc_predict_reg(int fam, int m, const double *q, const double *P, int n0,
                     const double *xtrain, const double *ytrain, const double *hyp,
                     const double *alpha, double *out)
{
    double lx, ly, p, sig;
    split_hyp(fam, hyp, &lx, &ly, &p, &sig);
    for (int k = 0; k < m; ++k) {
        double r = 0.0;
        for (int j = 0; j < n0; ++j)
            r += sig * orc_scalar(fam, W_KERN, xtrain[j], ytrain[j], q[k], P[k], lx, ly, p) * alpha[j];
        out[k] = r;
    }
}


/* ---- d canonical pairs (SURVEY.md 8 preamble, 8(f) rank 4) ---------------------------------------
 * The reference implements one pair only (kernels.f90 takes scalar (x, y)); BASELINE's d = 2, 3
 * configs generalise it as SURVEY.md recommends: inputs x = (q_1..q_d, P_1..P_d), product kernel
 *     k(x, x') = prod_m f_m(x_m - x'_m),   f_m periodic (family A) or SE (family C) for the q's,
 *                                          SE for the P's,
 * covariance of the gradient observations K_ab = d^2 k / dx_a dx'_b, a, b = 1..2d, i.e.
 *     K_ab = sig k (a == b ? -f_a''/f_a : -(f_a'/f_a)(f_b'/f_b)).
 * d = 1 is exactly build_K (sympgpr.f90:12-38): -f''/f are d2kdxdx0 / d2kdydy0 over k and the
 * cross term is d2kdxdy0 (kernels.f90:58-94).  PARITY UNPINNED BY THE REFERENCE for d > 1: pinned by
 * the d = 1 reduction and by a sympy differentiation of the product kernel in tests/test_oracle.py
 * (the technique of the reference's own init_func.py:24-52).
 * X is (n x 2d) column-major (one column per coordinate); hyp = (lq_1..lq_d, lP_1..lP_d, sig);
 * K is (2 d n x 2 d n0), block (a, b) at rows a n, columns b n0. */
int orc_build_k_nd(int fam, int d, int n, int n0, const double *X, const double *X0, const double *hyp,
                   double *K, size_t ldk)
{
    /* families B (sum kernel: k = sum_m f_m, K_aa = -sig f_a'', other blocks zero) and D (free period p_m per
     * q, hyp = (lq.., lP.., p_1..p_d, sig)) generalise the same way; d = 1 is their build_K. */
    const int D = 2 * d;
    const double sig = hyp[fam == FAM_D ? 3 * d : D];
    if (fam < FAM_A || fam > FAM_D) return -1;
    for (int j = 0; j < n0; ++j)
        for (int i = 0; i < n; ++i) {
            double g[16], nh[16], ar[16], arg = 0.0;
            for (int m = 0; m < D; ++m) {
                const double l = hyp[m], l2 = l * l;
                const double dx = X0[j + (size_t)m * n0] - X[i + (size_t)m * n];   /* a = column point */
                if (m < d && fam != FAM_C) {
                    const double hs = fam == FAM_D ? hyp[D + m] : 0.5;
                    const double s = sin(hs * dx), c = cos(hs * dx);
                    ar[m] = -0.5 * s * s / l2;
                    g[m] = -hs * s * c / l2;
                    nh[m] = hs * hs * (l2 * cos(2.0 * hs * dx) - s * s * c * c) / (l2 * l2);
                } else {
                    ar[m] = -0.5 * dx * dx / l2;
                    g[m] = -dx / l2;
                    nh[m] = (l2 - dx * dx) / (l2 * l2);
                }
                arg += ar[m];
            }
            const double E = sig * exp(arg);
            for (int a = 0; a < D; ++a)
                for (int b = 0; b < D; ++b) {
                    double v;
                    if (fam == FAM_B) v = a == b ? sig * exp(ar[a]) * nh[a] : 0.0;
                    else v = E * (a == b ? nh[a] : -g[a] * g[b]);
                    K[(size_t)a * n + i + ((size_t)b * n0 + j) * ldk] = v;
                }
        }
    return 0;
}
