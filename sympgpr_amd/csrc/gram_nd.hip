// gram_nd.hip -- Gram build for d canonical pairs per training point (BASELINE configs d = 2, 3).
//
// The reference's kernels take one pair (x, y) = (q, P) (kernels.f90:1); SURVEY.md 8 generalises
// them the way its own generator would (init_func.py:24-52): inputs x = (q_1..q_d, P_1..P_d),
// product kernel k = prod_m f_m(x_m - x'_m) with f_m periodic (family A; D: with a free period p_m per q,
// hyp = (lq.., lP.., p_1..p_d, sig)) or SE (family C) on the q's and SE on the P's, and the covariance of
// the gradient observations
//     K_ab = d^2 k / dx_a dx'_b = sig k (a == b ? -f_a''/f_a : -(f_a'/f_a)(f_b'/f_b)),  a, b = 1..2d;
// family B, the SUM kernel k = sum_m f_m: K_aa = -sig f_a'', all other blocks zero (the explicit maps),
// stored as (2d)^2 blocks of N x N0, block (a, b) at rows a*N, columns b*N0.  d = 1 is build_K
// (sympgpr.f90:12-38) entry for entry.  One exp and d sincos per PAIR feed all (2d)^2 entries:
// 32 d^2 bytes written per pair, so the kernel is even more firmly HBM-write bound than d = 1.
#include "common.h"
#include "devmath.h"
#include "generated/pair_generated.h"

namespace sgpr {

namespace {

constexpr int NT = 256, NTI = 2 * NT, NTJ = 16;
typedef double double2_t __attribute__((ext_vector_type(2)));

struct NdArgs {
    int mi, mj;
    const double *Xb, *Xa;   // row points (mi x D), column points (mj x D), column-major
    size_t ldxb, ldxa;
    double *K;
    size_t ld, rstride, cstride;   // element distance between consecutive row / column blocks
    int sel;                       // 1: block (a, b) goes to K + roff[a] + coff[b] * ld instead, skipped when either is < 0
    long roff[6], coff[6];
    long diag_off;
    double noise, sig;
    double l[6], l2[6], inv_l2[6], inv_l4[6];
    double hs[6];            // periodic coordinates: sin(hs (x - x')), hs = 1/2 (A, B) or p_m (D)
};

// per coordinate: exponent of f_m, g = f'/f, nh = -f''/f
template <int FAM, int D, int M>
__device__ __forceinline__ void coord(const NdArgs &a, double dx, double &arg, double &g, double &nh)
{
    if constexpr (FAM == SGPR_FAM_USER) {
        // the user's kernel: its generated factor forms (tools/gen_kernels.py), the q's with hs = p_m when it has one
        double o[3];
        if constexpr (M < D / 2) gen::factor<SGPR_FAM_USER, 1>(dx, a.l[M], a.hs[M], o);
        else                     gen::factor<SGPR_FAM_USER, 0>(dx, a.l[M], 0.0, o);
        arg = o[0]; g = o[1]; nh = o[2];
    } else if constexpr (FAM != SGPR_FAM_C && M < D / 2) {
        double s, c;
        sincos_fast(a.hs[M] * dx, s, c);
        const double s2 = s * s, sc = s * c;
        arg = -0.5 * a.inv_l2[M] * s2;
        g = -a.hs[M] * sc * a.inv_l2[M];
        nh = (a.hs[M] * a.hs[M]) * (a.l2[M] * __builtin_fma(-2.0, s2, 1.0) - sc * sc) * a.inv_l4[M];
    } else {
        const double d2 = dx * dx;
        arg = -0.5 * a.inv_l2[M] * d2;
        g = -dx * a.inv_l2[M];
        nh = (a.l2[M] - d2) * a.inv_l4[M];
    }
}

template <int FAM, int D, int M = 0>
__device__ __forceinline__ void all_coords(const NdArgs &a, const double *xa, const double (&xb)[D],
                                           double (&arg)[D], double (&g)[D], double (&nh)[D])
{
    if constexpr (M < D) {
        coord<FAM, D, M>(a, xa[M] - xb[M], arg[M], g[M], nh[M]);
        all_coords<FAM, D, M + 1>(a, xa, xb, arg, g, nh);
    }
}

// E[m]: the factor that multiplies nh[m] on the diagonal blocks -- sig k for the product kernels (one exp
// per pair), sig f_m for the sum kernel (one exp per coordinate)
template <int FAM> constexpr bool is_sum() { return FAM == SGPR_FAM_B || (FAM == SGPR_FAM_USER && gen::user_is_sum); }

template <int FAM, int D>
__device__ __forceinline__ void weights(const NdArgs &a, const double (&arg)[D], double (&E)[D])
{
    if constexpr (is_sum<FAM>()) {
#pragma unroll
        for (int m = 0; m < D; ++m) E[m] = a.sig * exp_fast(arg[m]);
    } else {
        double t = 0.0;
#pragma unroll
        for (int m = 0; m < D; ++m) t += arg[m];
        const double e = a.sig * exp_fast(t);
#pragma unroll
        for (int m = 0; m < D; ++m) E[m] = e;
    }
}

template <int FAM, int D>
__global__ __launch_bounds__(NT) void gram_nd_kernel(const NdArgs a)
{
    __shared__ double sxa[NTJ][D];
    const int i0 = blockIdx.x * NTI, j0 = blockIdx.y * NTJ;
    const int t = threadIdx.x;
    const int nj = min(NTJ, a.mj - j0);
    for (int e = t; e < nj * D; e += NT) sxa[e / D][e % D] = a.Xa[(size_t)(j0 + e / D) + (size_t)(e % D) * a.ldxa];
    __syncthreads();
    const int i = i0 + 2 * t;
    const bool v0 = i < a.mi, v1 = i + 1 < a.mi;
    double xb0[D], xb1[D];
#pragma unroll
    for (int m = 0; m < D; ++m) {
        xb0[m] = v0 ? a.Xb[(size_t)i + (size_t)m * a.ldxb] : 0.0;
        xb1[m] = v1 ? a.Xb[(size_t)i + 1 + (size_t)m * a.ldxb] : 0.0;
    }
    bool even = ((a.ld | a.rstride | a.cstride) & 1) == 0;
    if (a.sel) {
#pragma unroll
        for (int c = 0; c < D; ++c) even = even && ((a.roff[c] & 1) == 0 || a.roff[c] < 0);
    }
    const bool vec = (i0 + NTI <= a.mi) && even && (((uintptr_t)a.K & 15) == 0);
    const long d0 = (long)i + a.diag_off;
    for (int jj = 0; jj < nj; ++jj) {
        double g0[D], nh0[D], g1[D], nh1[D], arg0[D], arg1[D], E0[D], E1[D];
        all_coords<FAM, D>(a, sxa[jj], xb0, arg0, g0, nh0);
        all_coords<FAM, D>(a, sxa[jj], xb1, arg1, g1, nh1);
        weights<FAM, D>(a, arg0, E0);
        weights<FAM, D>(a, arg1, E1);
        const long j = j0 + jj;
        const double n0 = (d0 == j) ? a.noise : 0.0, n1 = (d0 + 1 == j) ? a.noise : 0.0;
#pragma unroll
        for (int ca = 0; ca < D; ++ca) {
#pragma unroll
            for (int cb = 0; cb < D; ++cb) {
                const double off0 = is_sum<FAM>() ? 0.0 : -E0[ca] * (g0[ca] * g0[cb]);
                const double off1 = is_sum<FAM>() ? 0.0 : -E1[ca] * (g1[ca] * g1[cb]);
                const double k0 = (ca == cb) ? __builtin_fma(E0[ca], nh0[ca], n0) : off0;
                const double k1 = (ca == cb) ? __builtin_fma(E1[ca], nh1[ca], n1) : off1;
                if (a.sel && (a.roff[ca] < 0 || a.coff[cb] < 0)) continue;      // block not wanted by this call
                double *dst = a.sel ? a.K + (size_t)a.roff[ca] + (size_t)i + ((size_t)a.coff[cb] + (size_t)j) * a.ld
                                    : a.K + (size_t)ca * a.rstride + (size_t)i + ((size_t)cb * a.cstride + (size_t)j) * a.ld;
                if (vec) {
                    *reinterpret_cast<double2_t *>(dst) = double2_t{k0, k1};
                } else {
                    if (v0) dst[0] = k0;
                    if (v1) dst[1] = k1;
                }
            }
        }
    }
}

// K*(2d x 2d n0) . alpha for one test point per workgroup; out is (m x D) column-major
template <int FAM, int D>
__global__ __launch_bounds__(NT) void predict_nd_kernel(const NdArgs a, int m, const double *alpha, double *out)
{
    // here: Xb = test points (m x D), Xa = training points (mj = n0)
    const int k = blockIdx.x;
    double xb[D], acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) { xb[c] = a.Xb[(size_t)k + (size_t)c * a.ldxb]; acc[c] = 0.0; }
    for (int j = threadIdx.x; j < a.mj; j += NT) {
        double xa[D], g[D], nh[D], arg[D], E[D];
#pragma unroll
        for (int c = 0; c < D; ++c) xa[c] = a.Xa[(size_t)j + (size_t)c * a.ldxa];
        all_coords<FAM, D>(a, xa, xb, arg, g, nh);
        weights<FAM, D>(a, arg, E);
        double al[D], S = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) { al[c] = alpha[(size_t)c * a.mj + j]; S = __builtin_fma(g[c], al[c], S); }
#pragma unroll
        for (int c = 0; c < D; ++c) {  // sum_b K_cb alpha_b = E (nh_c al_c - g_c (S - g_c al_c)); sum kernel: E_c nh_c al_c
            const double cross = is_sum<FAM>() ? 0.0 : -g[c] * (S - g[c] * al[c]);
            acc[c] = __builtin_fma(E[c], __builtin_fma(nh[c], al[c], cross), acc[c]);
        }
    }
    __shared__ double sh[NT / 64][D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        double v = acc[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < D) {
        double v = 0.0;
        for (int w = 0; w < NT / 64; ++w) v += sh[w][threadIdx.x];
        out[(size_t)k + (size_t)threadIdx.x * m] = v;
    }
}

int fill_args(int family, int d, const double *hyp, int nhyp, NdArgs &a)
{
    if (d < 1 || d > 3) { set_error("d must be 1, 2 or 3"); return SGPR_E_ARG; }
    if (family < SGPR_FAM_A || family > SGPR_FAM_USER) { set_error("unknown kernel family"); return SGPR_E_ARG; }
    const bool has_p = family_has_p(family);
    const int need = has_p ? 3 * d + 1 : 2 * d + 1;
    if (!hyp || nhyp != need) {
        set_error("hyp must hold (lq_1..lq_d, lP_1..lP_d, sig) -- (lq.., lP.., p_1..p_d, sig) for family D");
        return SGPR_E_ARG;
    }
    for (int m = 0; m < 2 * d; ++m) {
        a.l[m] = hyp[m];
        a.l2[m] = hyp[m] * hyp[m];
        a.inv_l2[m] = 1.0 / a.l2[m];
        a.inv_l4[m] = a.inv_l2[m] * a.inv_l2[m];
        a.hs[m] = (has_p && m < d) ? hyp[2 * d + m] : 0.5;
    }
    a.sig = hyp[nhyp - 1];
    return 0;
}

template <typename F>
int dispatch_nd(int family, int d, F &&f)
{
#define SGPR_ND_CASE(FAMV, DV) if (family == FAMV && d == DV) return f(std::integral_constant<int, FAMV>(), std::integral_constant<int, 2 * DV>())
    SGPR_ND_CASE(SGPR_FAM_A, 1); SGPR_ND_CASE(SGPR_FAM_A, 2); SGPR_ND_CASE(SGPR_FAM_A, 3);
    SGPR_ND_CASE(SGPR_FAM_C, 1); SGPR_ND_CASE(SGPR_FAM_C, 2); SGPR_ND_CASE(SGPR_FAM_C, 3);
    SGPR_ND_CASE(SGPR_FAM_B, 1); SGPR_ND_CASE(SGPR_FAM_B, 2); SGPR_ND_CASE(SGPR_FAM_B, 3);
    SGPR_ND_CASE(SGPR_FAM_D, 1); SGPR_ND_CASE(SGPR_FAM_D, 2); SGPR_ND_CASE(SGPR_FAM_D, 3);
    SGPR_ND_CASE(SGPR_FAM_USER, 1); SGPR_ND_CASE(SGPR_FAM_USER, 2); SGPR_ND_CASE(SGPR_FAM_USER, 3);
#undef SGPR_ND_CASE
    set_error("unsupported (family, d)");
    return SGPR_E_ARG;
}

}  // namespace

int gram_nd(int family, int d, int mi, int mj, const double *Xb, size_t ldxb, const double *Xa, size_t ldxa,
            const double *hyp, int nhyp, double *K, size_t ld, size_t rstride, size_t cstride, long diag_off,
            double noise, hipStream_t st)
{
    NdArgs a{};
    int rc = fill_args(family, d, hyp, nhyp, a);
    if (rc) return rc;
    if (mi <= 0 || mj <= 0) return 0;
    a.mi = mi; a.mj = mj; a.Xb = Xb; a.Xa = Xa; a.ldxb = ldxb; a.ldxa = ldxa;
    a.K = K; a.ld = ld; a.rstride = rstride; a.cstride = cstride; a.diag_off = diag_off; a.noise = noise;
    const dim3 grid((mi + NTI - 1) / NTI, (mj + NTJ - 1) / NTJ);
    if (grid.y > 65535) { set_error("too many pair columns for one launch"); return SGPR_E_ARG; }
    return dispatch_nd(family, d, [&](auto fam, auto dd) {
        hipLaunchKernelGGL((gram_nd_kernel<decltype(fam)::value, decltype(dd)::value>), grid, dim3(NT), 0, st, a);
        SGPR_CHECK_LAUNCH();
        return 0;
    });
}

// the same pairs, but only the blocks (a, b) with roff[a] >= 0 and coff[b] >= 0, block (a, b) at K + roff[a] + coff[b] * ld
// (a block-cyclic rank whose coordinate blocks hold different points: sympgpr_amd/dist.py)
int gram_nd_sel(int family, int d, int mi, int mj, const double *Xb, size_t ldxb, const double *Xa, size_t ldxa,
                const double *hyp, int nhyp, double *K, size_t ld, const long *roff, const long *coff, hipStream_t st)
{
    NdArgs a{};
    int rc = fill_args(family, d, hyp, nhyp, a);
    if (rc) return rc;
    if (mi <= 0 || mj <= 0) return 0;
    if (!roff || !coff) { set_error("gram_nd_sel: null offsets"); return SGPR_E_ARG; }
    a.mi = mi; a.mj = mj; a.Xb = Xb; a.Xa = Xa; a.ldxb = ldxb; a.ldxa = ldxa;
    a.K = K; a.ld = ld; a.rstride = 0; a.cstride = 0; a.diag_off = 1L << 60; a.noise = 0.0;
    a.sel = 1;
    bool any = false;
    for (int c = 0; c < 2 * d; ++c) { a.roff[c] = roff[c]; a.coff[c] = coff[c]; }
    for (int c = 0; c < 2 * d; ++c)
        for (int e = 0; e < 2 * d; ++e) any = any || (roff[c] >= 0 && coff[e] >= 0);
    if (!any) return 0;
    const dim3 grid((mi + NTI - 1) / NTI, (mj + NTJ - 1) / NTJ);
    if (grid.y > 65535) { set_error("too many pair columns for one launch"); return SGPR_E_ARG; }
    return dispatch_nd(family, d, [&](auto fam, auto dd) {
        hipLaunchKernelGGL((gram_nd_kernel<decltype(fam)::value, decltype(dd)::value>), grid, dim3(NT), 0, st, a);
        SGPR_CHECK_LAUNCH();
        return 0;
    });
}

int predict_nd(int family, int d, int m, const double *Xt, size_t ldxt, int n0, const double *Xtr, size_t ldxtr,
               const double *hyp, int nhyp, const double *alpha, double *out, hipStream_t st)
{
    NdArgs a{};
    int rc = fill_args(family, d, hyp, nhyp, a);
    if (rc) return rc;
    if (m <= 0) return 0;
    a.mi = m; a.mj = n0; a.Xb = Xt; a.Xa = Xtr; a.ldxb = ldxt; a.ldxa = ldxtr;
    return dispatch_nd(family, d, [&](auto fam, auto dd) {
        hipLaunchKernelGGL((predict_nd_kernel<decltype(fam)::value, decltype(dd)::value>), dim3(m), dim3(NT), 0, st, a, m,
                           alpha, out);
        SGPR_CHECK_LAUNCH();
        return 0;
    });
}

}  // namespace sgpr
