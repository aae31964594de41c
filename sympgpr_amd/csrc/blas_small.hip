// blas_small.hip -- the bandwidth-bound helpers around the factorisation: GEMV updates of the
// triangular solves (solve_triangular x2 at python/functions/func.py:174-177), the NLL
// reduction (func.py:186,195), triangle clean-up for SciPy-shaped outputs.
#include "common.h"

namespace sgpr {

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));

// strict upper triangle := 0   (scipy.linalg.cholesky returns a clean triangle)
__global__ void zero_upper_kernel(int n, double *A, size_t lda)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i < n && i < j) A[(size_t)i + (size_t)j * lda] = 0.0;
}

// A(i,j) := A(j,i) for i < j  (mirror the lower triangle up)
__global__ void sym_fill_kernel(int n, double *A, size_t lda)
{
    __shared__ double tile[32][33];
    const int bi = blockIdx.x, bj = blockIdx.y;  // tile (rows bi, cols bj) of the UPPER part
    if (bi > bj) return;
    // read lower tile (rows bj*32.., cols bi*32..), write transposed into (bi, bj)
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int gi = bj * 32 + threadIdx.x, gj = bi * 32 + r;
        if (gi < n && gj < n) tile[r][threadIdx.x] = A[(size_t)gi + (size_t)gj * lda];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int gi = bi * 32 + threadIdx.x, gj = bj * 32 + r;
        if (gi < n && gj < n && gi < gj) A[(size_t)gi + (size_t)gj * lda] = tile[threadIdx.x][r];
    }
}

__global__ __launch_bounds__(256) void nll_kernel(int n, const double *L, size_t ldl,
                                                  const double *z, const double *alpha, double *out)
{
    double q = 0.0, ld = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        q = __builtin_fma(z[i], alpha[i], q);
        ld += log(L[(size_t)i + (size_t)i * ldl]);
    }
    __shared__ double sq[4], sl[4];
    for (int o = 32; o > 0; o >>= 1) {
        q += __shfl_down(q, o, 64);
        ld += __shfl_down(ld, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sq[threadIdx.x >> 6] = q;
        sl[threadIdx.x >> 6] = ld;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = 0.5 * (sq[0] + sq[1] + sq[2] + sq[3]) + (sl[0] + sl[1] + sl[2] + sl[3]);
        out[1] = sl[0] + sl[1] + sl[2] + sl[3];
    }
}

__global__ void copy_diag_kernel(int n, const double *A, size_t lda, double *d)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = A[(size_t)i + (size_t)i * lda];
}

__global__ __launch_bounds__(256) void dot_kernel(int n, const double *a, size_t inca, const double *b, double *out)
{
    double q = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) q = __builtin_fma(a[(size_t)i * inca], b ? b[i] : 1.0, q);
    __shared__ double sq[4];
    for (int o = 32; o > 0; o >>= 1) q += __shfl_down(q, o, 64);
    if ((threadIdx.x & 63) == 0) sq[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = sq[0] + sq[1] + sq[2] + sq[3];
}

// sum of squares, two deterministic passes: per-workgroup partials, then one workgroup over them
constexpr int SS_WG = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(size_t count, const double *a, double *part)
{
    double q = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
        q = __builtin_fma(a[i], a[i], q);
    __shared__ double sq[4];
    for (int o = 32; o > 0; o >>= 1) q += __shfl_down(q, o, 64);
    if ((threadIdx.x & 63) == 0) sq[threadIdx.x >> 6] = q;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sq[0] + sq[1] + sq[2] + sq[3];
}

__global__ void transpose_kernel(int m, int n, const double *A, size_t lda, double *B, size_t ldb)
{
    __shared__ double tile[32][33];
    const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int i = i0 + threadIdx.x, j = j0 + r;
        if (i < m && j < n) tile[r][threadIdx.x] = A[(size_t)i + (size_t)j * lda];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int j = j0 + threadIdx.x, i = i0 + r;
        if (i < m && j < n) B[(size_t)j + (size_t)i * ldb] = tile[threadIdx.x][r];
    }
}

// y(m) -= A(m x k) x(k).  Workgroup = 512 rows (2 per thread), ALL k columns in chunks of KC staged through LDS: every
// element of y has one owner and one fixed order of summation, so the result is bitwise reproducible (round 2 split the
// columns over workgroups and combined them with fp64 atomics: alpha of the small-order fallback then differed in the last
// bits from run to run, which the reference's LAPACK path never does).  HBM-bound: A is read exactly once.
constexpr int GV_T = 256, GV_ROWS = 2 * GV_T, GV_KC = 128;
__global__ __launch_bounds__(GV_T) void gemv_n_kernel(int m, int k, const double *A, size_t lda,
                                                      const double *x, double *y)
{
    __shared__ double sx[GV_KC];
    const int i = blockIdx.x * GV_ROWS + 2 * threadIdx.x;
    const bool vec = (i + 1 < m) && ((lda & 1) == 0) && (((uintptr_t)A & 15) == 0);
    double s0 = 0.0, s1 = 0.0;
    for (int k0 = 0; k0 < k; k0 += GV_KC) {
        const int kn = min(GV_KC, k - k0);
        __syncthreads();
        if (threadIdx.x < kn) sx[threadIdx.x] = x[k0 + threadIdx.x];
        __syncthreads();
        if (i >= m) continue;
        const double *a = A + (size_t)i + (size_t)k0 * lda;
        if (vec) {
#pragma unroll 4
            for (int c = 0; c < kn; ++c) {
                const double2_t v = *reinterpret_cast<const double2_t *>(a + (size_t)c * lda);
                s0 = __builtin_fma(v.x, sx[c], s0);
                s1 = __builtin_fma(v.y, sx[c], s1);
            }
        } else {
            for (int c = 0; c < kn; ++c) {
                s0 = __builtin_fma(a[(size_t)c * lda], sx[c], s0);
                if (i + 1 < m) s1 = __builtin_fma(a[(size_t)c * lda + 1], sx[c], s1);
            }
        }
    }
    if (i >= m) return;
    y[i] -= s0;
    if (i + 1 < m) y[i + 1] -= s1;
}

// y(k) -= A(m x k)^T x(m).  One wave per column over ALL rows, then a wave reduction in a fixed order: no atomics,
// bitwise reproducible.
__global__ __launch_bounds__(256) void gemv_t_kernel(int m, int k, const double *A, size_t lda,
                                                     const double *x, double *y)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + wave;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < k) {
        const double *a = A + (size_t)c * lda;
        int i = lane;
        for (; i + 192 < m; i += 256) {                      // four independent chains per lane
            s0 = __builtin_fma(a[i], x[i], s0);
            s1 = __builtin_fma(a[i + 64], x[i + 64], s1);
            s2 = __builtin_fma(a[i + 128], x[i + 128], s2);
            s3 = __builtin_fma(a[i + 192], x[i + 192], s3);
        }
        for (; i < m; i += 64) s0 = __builtin_fma(a[i], x[i], s0);
    }
    double s = (s0 + s1) + (s2 + s3);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0 && c < k) y[c] -= s;
}

}  // namespace

int zero_strict_upper(int n, double *A, size_t lda, hipStream_t st)
{
    if (n <= 1) return 0;
    hipLaunchKernelGGL(zero_upper_kernel, dim3((n + 255) / 256, n), dim3(256), 0, st, n, A, lda);
    SGPR_CHECK_LAUNCH();
    return 0;
}

int sym_fill_upper(int n, double *A, size_t lda, hipStream_t st)
{
    if (n <= 1) return 0;
    const int t = (n + 31) / 32;
    hipLaunchKernelGGL(sym_fill_kernel, dim3(t, t), dim3(32, 8), 0, st, n, A, lda);
    SGPR_CHECK_LAUNCH();
    return 0;
}

int nll_reduce(int n, const double *L, size_t ldl, const double *z, const double *alpha,
               double *dout, hipStream_t st)
{
    hipLaunchKernelGGL(nll_kernel, dim3(1), dim3(256), 0, st, n, L, ldl, z, alpha, dout);
    SGPR_CHECK_LAUNCH();
    return 0;
}

int copy_diag(int n, const double *A, size_t lda, double *d, hipStream_t st)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(copy_diag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, A, lda, d);
    SGPR_CHECK_LAUNCH();
    return 0;
}

int dot(int n, const double *a, const double *b, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(256), 0, st, n, a, (size_t)1, b, out);
    SGPR_CHECK_LAUNCH();
    return 0;
}

// part: device scratch of SUMSQ_SCRATCH doubles
int sumsq(size_t count, const double *a, double *part, double *out, hipStream_t st)
{
    static_assert(SS_WG == SUMSQ_SCRATCH, "scratch size");
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SS_WG), dim3(256), 0, st, count, a, part);
    SGPR_CHECK_LAUNCH();
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(256), 0, st, SS_WG, part, (size_t)1, (const double *)nullptr, out);
    SGPR_CHECK_LAUNCH();
    return 0;
}

int trace(int n, const double *A, size_t lda, double *out, hipStream_t st)
{
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(256), 0, st, n, A, lda + 1, (const double *)nullptr, out);
    SGPR_CHECK_LAUNCH();
    return 0;
}

int transpose(int m, int n, const double *A, size_t lda, double *B, size_t ldb, hipStream_t st)
{
    if (m <= 0 || n <= 0) return 0;
    hipLaunchKernelGGL(transpose_kernel, dim3((m + 31) / 32, (n + 31) / 32), dim3(32, 8), 0, st, m, n,
                       A, lda, B, ldb);
    SGPR_CHECK_LAUNCH();
    return 0;
}

int gemv_n_sub(int m, int k, const double *A, size_t lda, const double *x, double *y, hipStream_t st)
{
    if (m <= 0 || k <= 0) return 0;
    hipLaunchKernelGGL(gemv_n_kernel, dim3((m + GV_ROWS - 1) / GV_ROWS), dim3(GV_T), 0, st, m, k, A, lda, x, y);
    SGPR_CHECK_LAUNCH();
    return 0;
}

int gemv_t_sub(int m, int k, const double *A, size_t lda, const double *x, double *y, hipStream_t st)
{
    if (m <= 0 || k <= 0) return 0;
    hipLaunchKernelGGL(gemv_t_kernel, dim3((k + 3) / 4), dim3(256), 0, st, m, k, A, lda, x, y);
    SGPR_CHECK_LAUNCH();
    return 0;
}

}  // namespace sgpr
