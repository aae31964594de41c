// chol.hip -- lower Cholesky factor / solves on one MI355X.
//
// Replaces scipy.linalg.cholesky(Ky, lower=True) -> LAPACK dpotrf and the two
// solve_triangular -> dtrtrs calls of python/functions/func.py:165-177,184-186,193-195.
//
// Structure (right-looking, recursively blocked): at every level
//     factor the leading block  ->  panel solve A21 := A21 L11^-T  ->  trailing update
//     A22 -= A21 A21^T (SYRK, lower tiles only)  ->  factor A22,
// with the split at a multiple of LEAF near the middle, so almost all of the n^3/3 flop are
// large-k GEMM/SYRK calls on the fp64 MFMA kernel (gemm_f64.hip) and the trailing matrix is
// re-read log2(n/LEAF) times instead of n/nb times.  The recursion bottoms out in LEAF = 128
// diagonal blocks that ONE workgroup factors entirely in LDS (128 x 129 fp64 = 129 KiB of the
// CU's 160 KiB) and then inverts in place; the inverse (LEAF x LEAF, zero upper) is kept in
// the workspace so that every panel solve and every triangular solve below is a multiply with
// inv(L_leaf) -- i.e. GEMM work on the matrix cores instead of a substitution.
#include "common.h"

namespace sgpr {

namespace {

constexpr int LT = 256;            // threads of the leaf kernel
constexpr int LLD = LEAF + 2;      // LDS leading dimension: even (16-B aligned column pairs), 4*LLD mod 64 banks = 8
constexpr int PW = 16;             // panel width inside the leaf

enum { LEAF_FACTOR = 0, LEAF_INVERT_ONLY = 1 };

typedef double double2_t __attribute__((ext_vector_type(2)));

// A (nb x nb, lower, global) -> L in place (mode FACTOR) and inv(L) -> inv (LEAF x LEAF,
// ld LEAF, zero-filled outside the nb x nb lower triangle).
//
// One workgroup, the whole block in LDS (128 x 130 fp64 = 130 KiB), padded to 128 with an
// identity so every loop bound is a compile-time constant.  Both phases work on 16-column
// panels (8 panel steps, 3 barriers each) instead of one barrier-separated step per column:
//   factor : 16x16 diagonal block by one wave (row per lane, pivots/columns via shuffles) ->
//            panel rows solved one per thread against it -> rank-16 update of the trailing
//            lower triangle in 4x4 register tiles;
//   inverse: LAPACK dtrtri order (last panel first): X21 = -X22 L21 inv(L11) with one row per
//            thread, inv(L11) by 16 lanes of wave 0.
__global__ __launch_bounds__(LT) void leaf_kernel(int nb, double *A, size_t lda, double *inv,
                                                  int *dinfo, int goff, int mode)
{
    __shared__ double s[LEAF * LLD];
    __shared__ double sInv[PW * (PW + 1)];
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    for (int idx = tid; idx < LEAF * LEAF; idx += LT) {
        const int i = idx % LEAF, c = idx / LEAF;
        double v = (i == c) ? 1.0 : 0.0;               // identity padding beyond nb
        if (i < nb && c < nb) v = (i >= c) ? A[(size_t)i + (size_t)c * lda] : 0.0;
        s[c * LLD + i] = v;
    }
    __syncthreads();

    if (mode == LEAF_FACTOR) {
        for (int c0 = 0; c0 < LEAF; c0 += PW) {
            // ---- (A) 16x16 diagonal block, wave 0: lane r holds row r
            if (tid < 64) {
                double a[PW];
#pragma unroll
                for (int c = 0; c < PW; ++c)
                    a[c] = (lane < PW && c <= lane) ? s[(c0 + c) * LLD + c0 + lane] : 0.0;
                bool bad = false;
                int badj = 0;
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    const double d = __shfl(a[j], j, 64);
                    if (!(d > 0.0) && !bad) { bad = true; badj = j; }
                    const double l = sqrt(d);
                    a[j] = (lane == j) ? l : a[j] / l;
#pragma unroll
                    for (int c = j + 1; c < PW; ++c) {
                        const double lcj = __shfl(a[j], c, 64);
                        a[c] = __builtin_fma(-a[j], lcj, a[c]);
                    }
                }
                if (bad && tid == 0 && *dinfo == 0) *dinfo = goff + c0 + badj + 1;
                if (lane < PW) {
#pragma unroll
                    for (int c = 0; c < PW; ++c)
                        if (c <= lane) s[(c0 + c) * LLD + c0 + lane] = a[c];
                }
            }
            __syncthreads();
            const int r0 = c0 + PW;
            const int rem = LEAF - r0;
            // ---- (B) panel rows: r := r L11^-T, one row per thread
            if (tid < rem) {
                const int i = r0 + tid;
                double r[PW];
#pragma unroll
                for (int c = 0; c < PW; ++c) r[c] = s[(c0 + c) * LLD + i];
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    double acc = r[j];
#pragma unroll
                    for (int k = 0; k < j; ++k) acc = __builtin_fma(-r[k], s[(c0 + k) * LLD + c0 + j], acc);
                    r[j] = acc / s[(c0 + j) * LLD + c0 + j];
                }
#pragma unroll
                for (int c = 0; c < PW; ++c) s[(c0 + c) * LLD + i] = r[c];
            }
            __syncthreads();
            // ---- (C) trailing lower triangle -= L21 L21^T, 4x4 tiles
            const int nt = rem / 4;
            const int ntile = nt * (nt + 1) / 2;
            for (int idx = tid; idx < ntile; idx += LT) {
                int ti = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
                while (ti * (ti + 1) / 2 > idx) --ti;
                while ((ti + 1) * (ti + 2) / 2 <= idx) ++ti;
                const int tj = idx - ti * (ti + 1) / 2;
                const int i0 = r0 + 4 * ti, j0 = r0 + 4 * tj;
                double acc[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll
                for (int k = 0; k < PW; ++k) {
                    const double *col = s + (c0 + k) * LLD;
                    const double2_t a01 = *reinterpret_cast<const double2_t *>(col + i0);
                    const double2_t a23 = *reinterpret_cast<const double2_t *>(col + i0 + 2);
                    const double2_t b01 = *reinterpret_cast<const double2_t *>(col + j0);
                    const double2_t b23 = *reinterpret_cast<const double2_t *>(col + j0 + 2);
                    const double av[4] = {a01.x, a01.y, a23.x, a23.y};
                    const double bv[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_fma(av[a], bv[b], acc[a][b]);
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    double *col = s + (j0 + b) * LLD + i0;
                    double2_t c01 = *reinterpret_cast<double2_t *>(col);
                    double2_t c23 = *reinterpret_cast<double2_t *>(col + 2);
                    c01.x -= acc[0][b]; c01.y -= acc[1][b]; c23.x -= acc[2][b]; c23.y -= acc[3][b];
                    *reinterpret_cast<double2_t *>(col) = c01;
                    *reinterpret_cast<double2_t *>(col + 2) = c23;
                }
            }
            __syncthreads();
        }
        for (int idx = tid; idx < nb * nb; idx += LT) {
            const int i = idx % nb, c = idx / nb;
            if (i >= c) A[(size_t)i + (size_t)c * lda] = s[c * LLD + i];
        }
        __syncthreads();
    }

    // ---- inverse, last panel first; only the lower triangle of s is read
    for (int c0 = LEAF - PW; c0 >= 0; c0 -= PW) {
        // (a) inv(L11) by 16 lanes: column c solves L x = e_c
        if (tid < PW) {
            const int c = tid;
            double x[PW];
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                double acc = (i == c) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k) acc = __builtin_fma(-s[(c0 + k) * LLD + c0 + i], x[k], acc);
                x[i] = acc / s[(c0 + i) * LLD + c0 + i];
            }
#pragma unroll
            for (int i = 0; i < PW; ++i) sInv[c * (PW + 1) + i] = x[i];
        }
        __syncthreads();
        const int r0 = c0 + PW;
        const int rem = LEAF - r0;
        double out[PW];
        if (tid < rem) {
            // (b) T = X22 L21 (row i), then (c) X21 = -T inv(L11)
            const int i = r0 + tid;
            double t[PW];
#pragma unroll
            for (int c = 0; c < PW; ++c) t[c] = 0.0;
            for (int k = r0; k < LEAF; ++k) {
                const double xv = (k <= i) ? s[k * LLD + i] : 0.0;
#pragma unroll
                for (int c = 0; c < PW; ++c) t[c] = __builtin_fma(xv, s[(c0 + c) * LLD + k], t[c]);
            }
#pragma unroll
            for (int c = 0; c < PW; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int k = c; k < PW; ++k) acc = __builtin_fma(t[k], sInv[c * (PW + 1) + k], acc);
                out[c] = -acc;
            }
        }
        __syncthreads();
        if (tid < rem) {
            const int i = r0 + tid;
#pragma unroll
            for (int c = 0; c < PW; ++c) s[(c0 + c) * LLD + i] = out[c];
        }
        if (tid >= 64 && tid < 64 + PW) {  // another wave drops inv(L11) into the diagonal block
            const int c = tid - 64;
#pragma unroll
            for (int i = 0; i < PW; ++i)
                if (i >= c) s[(c0 + c) * LLD + c0 + i] = sInv[c * (PW + 1) + i];
        }
        __syncthreads();
    }
    for (int idx = tid; idx < LEAF * LEAF; idx += LT) {
        const int i = idx % LEAF, c = idx / LEAF;
        inv[idx] = (i >= c && i < nb && c < nb) ? s[c * LLD + i] : 0.0;
    }
}

// b(nb) := inv(L) b  or  inv(L)^T b   (inv: LEAF x LEAF lower, zero upper).
// 1024 threads: row i = t & 127, k-slice = t >> 7 (8 slices of 16): sixteen independent loads per
// thread instead of one 128-long dependent chain (25 us -> ~3 us per call).
constexpr int MV_T = 1024;
__global__ __launch_bounds__(MV_T) void leaf_matvec_kernel(int nb, const double *inv, double *b,
                                                           int trans)
{
    __shared__ double sb[LEAF];
    __shared__ double part[MV_T / LEAF][LEAF];
    const int t = threadIdx.x, i = t & (LEAF - 1), ks = t >> 7;
    if (t < LEAF) sb[t] = t < nb ? b[t] : 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int kk = 0; kk < LEAF / (MV_T / LEAF); ++kk) {
        const int k = ks * (LEAF / (MV_T / LEAF)) + kk;
        // inv is zero above the diagonal and outside nb x nb, so no triangular bounds are needed
        const double m = trans ? inv[k + i * LEAF] : inv[i + k * LEAF];
        acc = __builtin_fma(m, sb[k], acc);
    }
    part[ks][i] = acc;
    __syncthreads();
    if (t < nb) {
        double r = 0.0;
#pragma unroll
        for (int q = 0; q < MV_T / LEAF; ++q) r += part[q][t];
        b[t] = r;
    }
}

inline int split(int n)
{
    // first part: a multiple of LEAF close to n/2 (>= LEAF, < n)
    int n1 = ((n / 2 + LEAF - 1) / LEAF) * LEAF;
    if (n1 >= n) n1 -= LEAF;
    return n1;
}

struct Ctx {
    double *inv;   // leaf inverses: leaf t at inv + t * LEAF * LEAF
    int *dinfo;
    hipStream_t st;
};

int trsm_rec(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, int off, const Ctx &c)
{
    if (n <= LEAF) {
        // B := B inv(L)^T, in place: one column tile (n <= 128), every workgroup owns full rows
        return gemm_nt(m, n, n, 1.0, B, ldb, c.inv + (size_t)(off / LEAF) * LEAF * LEAF, LEAF, 0.0,
                       B, ldb, 0, 0, c.st);
    }
    const int n1 = split(n), n2 = n - n1;
    int rc = trsm_rec(m, n1, L, ldl, B, ldb, off, c);
    if (rc) return rc;
    // B2 -= B1 L21^T
    rc = gemm_nt(m, n2, n1, -1.0, B, ldb, L + n1, ldl, 1.0, B + (size_t)n1 * ldb, ldb, 0, 0, c.st);
    if (rc) return rc;
    return trsm_rec(m, n2, L + n1 + (size_t)n1 * ldl, ldl, B + (size_t)n1 * ldb, ldb, off + n1, c);
}

int potrf_rec(int n, double *A, size_t lda, int off, const Ctx &c)
{
    if (n <= LEAF) {
        hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(LT), 0, c.st, n, A, lda,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, c.dinfo, off, (int)LEAF_FACTOR);
        SGPR_CHECK_LAUNCH();
        return 0;
    }
    const int n1 = split(n), n2 = n - n1;
    double *A21 = A + n1, *A22 = A + n1 + (size_t)n1 * lda;
    int rc = potrf_rec(n1, A, lda, off, c);
    if (rc) return rc;
    rc = trsm_rec(n2, n1, A, lda, A21, lda, off, c);
    if (rc) return rc;
    rc = gemm_nt(n2, n2, n1, -1.0, A21, lda, A21, lda, 1.0, A22, lda, 1, 0, c.st);  // SYRK, lower
    if (rc) return rc;
    return potrf_rec(n2, A22, lda, off + n1, c);
}

int trsv_n_rec(int n, const double *L, size_t ldl, double *b, int off, const Ctx &c)
{
    if (n <= LEAF) {
        hipLaunchKernelGGL(leaf_matvec_kernel, dim3(1), dim3(MV_T), 0, c.st, n,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, b, 0);
        SGPR_CHECK_LAUNCH();
        return 0;
    }
    const int n1 = split(n), n2 = n - n1;
    int rc = trsv_n_rec(n1, L, ldl, b, off, c);
    if (rc) return rc;
    rc = gemv_n_sub(n2, n1, L + n1, ldl, b, b + n1, c.st);  // b2 -= L21 b1
    if (rc) return rc;
    return trsv_n_rec(n2, L + n1 + (size_t)n1 * ldl, ldl, b + n1, off + n1, c);
}

int trsv_t_rec(int n, const double *L, size_t ldl, double *b, int off, const Ctx &c)
{
    if (n <= LEAF) {
        hipLaunchKernelGGL(leaf_matvec_kernel, dim3(1), dim3(MV_T), 0, c.st, n,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, b, 1);
        SGPR_CHECK_LAUNCH();
        return 0;
    }
    const int n1 = split(n), n2 = n - n1;
    int rc = trsv_t_rec(n2, L + n1 + (size_t)n1 * ldl, ldl, b + n1, off + n1, c);
    if (rc) return rc;
    rc = gemv_t_sub(n2, n1, L + n1, ldl, b + n1, b, c.st);  // b1 -= L21^T b2
    if (rc) return rc;
    return trsv_t_rec(n1, L, ldl, b, off, c);
}

int leaves_invert_only(int n, double *L, size_t ldl, const Ctx &c)
{
    for (int off = 0; off < n; off += LEAF) {
        const int nb = n - off < LEAF ? n - off : LEAF;
        hipLaunchKernelGGL(leaf_kernel, dim3(1), dim3(LT), 0, c.st, nb, L + off + (size_t)off * ldl, ldl,
                           c.inv + (size_t)(off / LEAF) * LEAF * LEAF, c.dinfo, off,
                           (int)LEAF_INVERT_ONLY);
        SGPR_CHECK_LAUNCH();
    }
    return 0;
}

inline size_t inv_bytes(int n)
{
    return (size_t)((n + LEAF - 1) / LEAF) * LEAF * LEAF * sizeof(double);
}

}  // namespace

size_t potrf_workspace(int n) { return n <= 0 ? 256 : inv_bytes(n) + 256; }

int potrf(int n, double *A, size_t lda, void *work, size_t lwork, int *dinfo, hipStream_t st)
{
    if (n < 0 || (n > 0 && lda < (size_t)n)) { set_error("potrf: bad n / lda"); return SGPR_E_ARG; }
    if (lwork < potrf_workspace(n)) { set_error("potrf: workspace too small"); return SGPR_E_ARG; }
    SGPR_HIP(hipMemsetAsync(dinfo, 0, sizeof(int), st));
    if (n == 0) return 0;
    Ctx c{static_cast<double *>(work), dinfo, st};
    return potrf_rec(n, A, lda, 0, c);
}

int trsm_rlt(int m, int n, const double *L, size_t ldl, double *B, size_t ldb, const void *work,
             hipStream_t st)
{
    if (m <= 0 || n <= 0) return 0;
    Ctx c{const_cast<double *>(static_cast<const double *>(work)), nullptr, st};
    return trsm_rec(m, n, L, ldl, B, ldb, 0, c);
}

int potrs_vec(int n, const double *L, size_t ldl, const void *work, double *b, hipStream_t st)
{
    if (n <= 0) return 0;
    Ctx c{const_cast<double *>(static_cast<const double *>(work)), nullptr, st};
    int rc = trsv_n_rec(n, L, ldl, b, 0, c);
    if (rc) return rc;
    return trsv_t_rec(n, L, ldl, b, 0, c);
}

// one-sided solve: b := L^-1 b (trans = 0) or L^-T b (trans = 1)
int trsv(int n, const double *L, size_t ldl, const void *work, double *b, int trans, hipStream_t st)
{
    if (n <= 0) return 0;
    Ctx c{const_cast<double *>(static_cast<const double *>(work)), nullptr, st};
    return trans ? trsv_t_rec(n, L, ldl, b, 0, c) : trsv_n_rec(n, L, ldl, b, 0, c);
}

// leaf inverses of an existing factor (for solves against an L that was not produced by potrf())
int leaf_inverses(int n, const double *L, size_t ldl, void *work, int *dinfo, hipStream_t st)
{
    if (n <= 0) return 0;
    Ctx c{static_cast<double *>(work), dinfo, st};
    return leaves_invert_only(n, const_cast<double *>(L), ldl, c);
}

}  // namespace sgpr
