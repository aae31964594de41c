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
constexpr int LLD = LEAF + 1;      // LDS leading dimension (odd: conflict-free row walks)

enum { LEAF_FACTOR = 0, LEAF_INVERT_ONLY = 1 };

// A (nb x nb, lower, global) -> L in place (mode FACTOR) and inv(L) -> inv (LEAF x LEAF,
// ld LEAF, zero-filled outside the nb x nb lower triangle).
__global__ __launch_bounds__(LT) void leaf_kernel(int nb, double *A, size_t lda, double *inv,
                                                  int *dinfo, int goff, int mode)
{
    __shared__ double s[LEAF * LLD];
    __shared__ double sd[LEAF];
    const int tid = threadIdx.x;

    for (int idx = tid; idx < nb * nb; idx += LT) {
        const int i = idx % nb, c = idx / nb;
        s[c * LLD + i] = (i >= c) ? A[(size_t)i + (size_t)c * lda] : 0.0;
    }
    __syncthreads();

    if (mode == LEAF_FACTOR) {
        // Outer-product elimination on UNSCALED columns (one barrier per column):
        //   s(i,c) -= s(i,j) s(c,j) / d_j ,  d_j = s(j,j);  L(i,j) = s(i,j)/sqrt(d_j) at the end.
        const int ii = tid & 63, cc = tid >> 6;
        for (int j = 0; j < nb; ++j) {
            const double d = s[j * LLD + j];
            if (!(d > 0.0) && tid == 0 && *dinfo == 0) *dinfo = goff + j + 1;
            const double dinv = 1.0 / d;
            for (int c = j + 1 + cc; c < nb; c += LT / 64) {
                const double lcj = s[j * LLD + c] * dinv;
                for (int i = j + 1 + ii; i < nb; i += 64)
                    if (i >= c) s[c * LLD + i] = __builtin_fma(-s[j * LLD + i], lcj, s[c * LLD + i]);
            }
            __syncthreads();
        }
        if (tid < nb) sd[tid] = sqrt(s[tid * LLD + tid]);
        __syncthreads();
        for (int idx = tid; idx < nb * nb; idx += LT) {
            const int i = idx % nb, c = idx / nb;
            if (i > c) s[c * LLD + i] /= sd[c];
            else if (i == c) s[c * LLD + i] = sd[c];
        }
        __syncthreads();
        for (int idx = tid; idx < nb * nb; idx += LT) {
            const int i = idx % nb, c = idx / nb;
            if (i >= c) A[(size_t)i + (size_t)c * lda] = s[c * LLD + i];
        }
        __syncthreads();
    }

    // In-place inverse of the lower-triangular s (LAPACK dtrti2 order: last column first):
    //   X(j,j) = 1/L(j,j);  X(j+1:,j) = -X(j,j) * X(j+1:,j+1:) * L(j+1:,j)
    // two threads per row split the dot product.
    {
        const int i = tid >> 1, half = tid & 1;
        for (int j = nb - 1; j >= 0; --j) {
            const double ajj = 1.0 / s[j * LLD + j];
            double y = 0.0;
            if (i > j && i < nb)
                for (int k = j + 1 + half; k <= i; k += 2) y = __builtin_fma(s[k * LLD + i], s[j * LLD + k], y);
            y += __shfl_xor(y, 1, 64);
            __syncthreads();
            if (half == 0) {
                if (i > j && i < nb) s[j * LLD + i] = -ajj * y;
                else if (i == j) s[j * LLD + j] = ajj;
            }
            __syncthreads();
        }
    }
    for (int idx = tid; idx < LEAF * LEAF; idx += LT) {
        const int i = idx % LEAF, c = idx / LEAF;
        inv[idx] = (i >= c && i < nb && c < nb) ? s[c * LLD + i] : 0.0;
    }
}

// b(nb) := inv(L) b  or  inv(L)^T b   (inv: LEAF x LEAF lower, zero upper)
__global__ __launch_bounds__(LEAF) void leaf_matvec_kernel(int nb, const double *inv, double *b,
                                                           int trans)
{
    __shared__ double sb[LEAF];
    const int i = threadIdx.x;
    sb[i] = i < nb ? b[i] : 0.0;
    __syncthreads();
    double acc = 0.0;
    if (!trans) {
        for (int k = 0; k <= i && k < nb; ++k) acc = __builtin_fma(inv[i + k * LEAF], sb[k], acc);
    } else {
        for (int k = i; k < nb; ++k) acc = __builtin_fma(inv[k + i * LEAF], sb[k], acc);
    }
    if (i < nb) b[i] = acc;
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
        hipLaunchKernelGGL(leaf_matvec_kernel, dim3(1), dim3(LEAF), 0, c.st, n,
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
        hipLaunchKernelGGL(leaf_matvec_kernel, dim3(1), dim3(LEAF), 0, c.st, n,
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
