// eig.hip -- symmetric eigendecomposition by parallel cyclic Jacobi rotations.
//
// This is the positive-definiteness FAILURE path of the drivers' objective: when
// scipy.linalg.cholesky raises, nll_chol falls back to an eigen-solve of Ky
// (python/02_pert_pendulum/func.py:194-203, 01_pendulum/implicit/func.py:99-114,
// 05_tokamak/Split_SympGPR/func.py:128-166: `eigsh(Ky, neig, ...)`).  The reference takes the
// eigenpairs from ARPACK/LAPACK; here Ky stays in HBM and is diagonalised in place.
//
// Round-robin ordering: n (padded to even) indices form n/2 disjoint pairs per round, n-1 rounds
// per sweep.  The rotations of one round touch disjoint column pairs and disjoint row pairs, so a
// round is two launches: (A, V) := (A, V) J over all pairs' columns, then A := J^T A over all
// pairs' rows; the angles come from the three entries a_pp, a_qq, a_pq no other pair touches.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "common.h"

namespace sgpr {

namespace {

constexpr int JT = 256;

__device__ __forceinline__ void pair_of(int m, int r, int t, int &p, int &q)
{
    // circle method: index m-1 stays, the other m-1 rotate
    int a, b;
    if (t == 0) { a = m - 1; b = r; }
    else { a = (r + t) % (m - 1); b = (r - t + (m - 1)) % (m - 1); }
    p = min(a, b);
    q = max(a, b);
}

__global__ __launch_bounds__(JT) void jacobi_cols_kernel(int n, int m, int r, double *A, size_t lda,
                                                         double *V, size_t ldv, double *cs)
{
    __shared__ double sc, ss;
    const int t = blockIdx.x;
    int p, q;
    pair_of(m, r, t, p, q);
    if (q >= n) {   // the phantom index of an odd n
        if (threadIdx.x == 0) { cs[2 * t] = 1.0; cs[2 * t + 1] = 0.0; }
        return;
    }
    if (threadIdx.x == 0) {
        const double app = A[p + (size_t)p * lda], aqq = A[q + (size_t)q * lda], apq = A[p + (size_t)q * lda];
        double c = 1.0, s = 0.0;
        if (apq != 0.0 && (apq == apq)) {
            const double th = (aqq - app) / (2.0 * apq);
            if (fabs(th) > 1e150) {          // t = 1 / (2 theta) without overflowing theta^2
                const double tt = 0.5 / th;
                c = 1.0; s = tt;
            } else {
                const double tt = copysign(1.0, th) / (fabs(th) + sqrt(th * th + 1.0));
                c = 1.0 / sqrt(tt * tt + 1.0);
                s = tt * c;
            }
        }
        sc = c; ss = s;
        cs[2 * t] = c; cs[2 * t + 1] = s;
    }
    __syncthreads();
    const double c = sc, s = ss;
    if (s == 0.0) return;
    double *ap = A + (size_t)p * lda, *aq = A + (size_t)q * lda;
    double *vp = V + (size_t)p * ldv, *vq = V + (size_t)q * ldv;
    for (int i = threadIdx.x; i < n; i += JT) {
        const double x = ap[i], y = aq[i];
        ap[i] = c * x - s * y;
        aq[i] = s * x + c * y;
        const double u = vp[i], w = vq[i];
        vp[i] = c * u - s * w;
        vq[i] = s * u + c * w;
    }
}

__global__ __launch_bounds__(JT) void jacobi_rows_kernel(int n, int m, int r, double *A, size_t lda, const double *cs)
{
    const int t = blockIdx.x;
    int p, q;
    pair_of(m, r, t, p, q);
    if (q >= n) return;
    const double c = cs[2 * t], s = cs[2 * t + 1];
    if (s == 0.0) return;
    for (int j = threadIdx.x; j < n; j += JT) {
        double *col = A + (size_t)j * lda;
        const double x = col[p], y = col[q];
        col[p] = c * x - s * y;
        col[q] = s * x + c * y;
    }
}

// out[0] = sum of squared off-diagonal entries, out[1] = sum of squared diagonal entries
__global__ __launch_bounds__(JT) void offnorm_kernel(int n, const double *A, size_t lda, double *part)
{
    double off = 0.0, dg = 0.0;
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const double *col = A + (size_t)j * lda;
        for (int i = threadIdx.x; i < n; i += JT) {
            const double v = col[i];
            if (i == j) dg = __builtin_fma(v, v, dg);
            else off = __builtin_fma(v, v, off);
        }
    }
    __shared__ double so[JT / 64], sd[JT / 64];
    for (int o = 32; o > 0; o >>= 1) {
        off += __shfl_down(off, o, 64);
        dg += __shfl_down(dg, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { so[threadIdx.x >> 6] = off; sd[threadIdx.x >> 6] = dg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = so[0] + so[1] + so[2] + so[3];
        part[2 * blockIdx.x + 1] = sd[0] + sd[1] + sd[2] + sd[3];
    }
}

__global__ void set_identity_kernel(int n, double *V, size_t ldv)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i < n) V[i + (size_t)j * ldv] = (i == j) ? 1.0 : 0.0;
}

__global__ void gather_cols_kernel(int n, const double *V, size_t ldv, const int *perm, double *out, size_t ldo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i < n) out[i + (size_t)j * ldo] = V[i + (size_t)perm[j] * ldv];
}

}  // namespace

// A (n x n, device, full symmetric) -> eigenvectors in its columns, eigenvalues ascending in
// w_host (LAPACK dsyev 'V' convention).  V: n x n device workspace.  Returns 0, or 1 when the
// off-diagonal mass has not dropped below 1e-14 ||A||_F within max_sweeps.
int syev_jacobi(int n, double *A, size_t lda, double *V, size_t ldv, double *w_host, int max_sweeps,
                int *sweeps_done, hipStream_t st)
{
    if (sweeps_done) *sweeps_done = 0;
    if (n <= 0) return 0;
    const int m = (n + 1) & ~1;
    constexpr int NB_OFF = 128;
    double *cs = nullptr, *part = nullptr;
    int *perm = nullptr;
    SGPR_HIP(hipMalloc((void **)&cs, (size_t)m * sizeof(double)));
    SGPR_HIP(hipMalloc((void **)&part, 2 * NB_OFF * sizeof(double)));
    SGPR_HIP(hipMalloc((void **)&perm, (size_t)n * sizeof(int)));
    auto cleanup = [&]() { (void)hipFree(cs); (void)hipFree(part); (void)hipFree(perm); };
#define EIG_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { cleanup(); return hip_fail(e__, #call, __FILE__, __LINE__); } } while (0)
    hipLaunchKernelGGL(set_identity_kernel, dim3((n + 255) / 256, n), dim3(256), 0, st, n, V, ldv);
    EIG_HIP(hipGetLastError());
    std::vector<double> hp(2 * NB_OFF);
    auto offnorm = [&](double &off, double &dg) -> hipError_t {
        hipLaunchKernelGGL(offnorm_kernel, dim3(NB_OFF), dim3(JT), 0, st, n, A, lda, part);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(hp.data(), part, 2 * NB_OFF * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(st);
        off = dg = 0.0;
        for (int b = 0; b < NB_OFF; ++b) { off += hp[2 * b]; dg += hp[2 * b + 1]; }
        return e;
    };
    int status = 1, sweep = 0;
    double off, dg;
    EIG_HIP(offnorm(off, dg));
    const double tol2 = 1e-28;   // (1e-14)^2 relative to ||A||_F^2
    if (!(off > tol2 * (off + dg))) status = 0;
    for (; status && sweep < max_sweeps; ++sweep) {
        if (m >= 2 && n >= 2)
            for (int r = 0; r < m - 1; ++r) {
                hipLaunchKernelGGL(jacobi_cols_kernel, dim3(m / 2), dim3(JT), 0, st, n, m, r, A, lda, V, ldv, cs);
                hipLaunchKernelGGL(jacobi_rows_kernel, dim3(m / 2), dim3(JT), 0, st, n, m, r, A, lda, cs);
            }
        EIG_HIP(hipGetLastError());
        EIG_HIP(offnorm(off, dg));
        if (!(off == off)) break;                       // NaN input: give up, report not converged
        if (!(off > tol2 * (off + dg))) status = 0;
    }
    if (sweeps_done) *sweeps_done = sweep;
    // eigenvalues = diagonal; sort ascending, permute the vectors into A
    std::vector<double> w(n);
    EIG_HIP(hipMemcpy2DAsync(w.data(), sizeof(double), A, (lda + 1) * sizeof(double), sizeof(double), n,
                             hipMemcpyDeviceToHost, st));
    EIG_HIP(hipStreamSynchronize(st));
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return w[a] < w[b]; });
    for (int i = 0; i < n; ++i) w_host[i] = w[idx[i]];
    EIG_HIP(hipMemcpyAsync(perm, idx.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gather_cols_kernel, dim3((n + 255) / 256, n), dim3(256), 0, st, n, V, ldv, perm, A, lda);
    EIG_HIP(hipGetLastError());
    EIG_HIP(hipStreamSynchronize(st));
#undef EIG_HIP
    cleanup();
    return status;
}

}  // namespace sgpr
