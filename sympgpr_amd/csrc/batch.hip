// batch.hip -- many small independent fits in ONE launch.
//
// The reference's only batch axis: `nphmap` independent GP pairs, one per toroidal section, and the
// CMA-ES populations that evaluate nll_chol for a set of hyper-parameter vectors over the same data
// (python/05_tokamak/Split_SympGPR/main.py:36-41,63-66,96-112), at matrix orders n = 40 ... 160.  At
// those sizes one nll_chol through a device-resident handle is pure set-up (create / upload / five
// launches / download: 0.2 - 1 ms, tools/nll_call_cost.py); here one workgroup runs the whole body of
// nll_chol (python/functions/func.py:189-196) for one problem -- Gram build, Cholesky, both triangular
// solves, the negative log-likelihood -- and a launch covers the batch.
//
// Per problem (order n <= 256 = two leaves): Ky -> scratch (global, L2-resident, 512 KiB per
// workgroup);  L11 = leaf(A11) in LDS (leaf.h), X11 = inv(L11);  L21 = A21 X11^T;  A22 -= L21 L21^T;
// L22 = leaf(A22);  y = L^-1 z and alpha = L^-T y through the leaf inverses.  Latency-bound by
// construction: throughput comes from the number of problems in flight, not from the matrix cores.
#include <cmath>
#include <cstring>

#include "common.h"
#include "leaf.h"
#include "pair_eval.h"

namespace sgpr {

namespace {

using namespace leaf;
using namespace pairf;

constexpr int BMAX = 2 * LEAF;            // largest order per problem

struct BatchArgs {
    int nbatch, npts, n, reg;
    const double *x, *y, *z;              // nbatch x npts, nbatch x npts, nbatch x n
    const KConst *kc;                     // per problem
    const double *noise;                  // per problem, >= 0
    double *scratch;                      // per workgroup: BMAX*BMAX (Ky / L) + 2*LEAF*LEAF (leaf inverses) + 2*BMAX
    double *alpha, *nll;                  // outputs (alpha may be null)
    int *info;                            // per problem, zero on entry
};

template <int FAM>
__global__ __launch_bounds__(LT) void fit_batch_kernel(const BatchArgs a)
{
    __shared__ double s[LEAF_LDS];
    __shared__ double red[LT / 64];
    const int tid = threadIdx.x;
    const int n = a.n, N = a.npts;
    const size_t per_wg = (size_t)BMAX * BMAX + 2 * (size_t)LEAF * LEAF + 2 * BMAX;
    double *A = a.scratch + (size_t)blockIdx.x * per_wg;     // column-major, ld = BMAX
    double *inv = A + (size_t)BMAX * BMAX;
    double *v = inv + 2 * (size_t)LEAF * LEAF;                // y, then alpha
    constexpr size_t ld = BMAX;
    const int n1 = min(n, (int)LEAF), n2 = n - n1;
    for (int b = blockIdx.x; b < a.nbatch; b += gridDim.x) {
        const KConst kc = a.kc[b];
        const double *x = a.x + (size_t)b * N, *y = a.y + (size_t)b * N, *z = a.z + (size_t)b * n;
        const double noise = a.noise[b];
        // ---- Ky = build_K(x, x) + |sig2n| I (lower triangle; func.py:191-192) or buildKreg (func.py:182-183)
        if (a.reg) {
            for (int e = tid; e < N * N; e += LT) {
                const int i = e % N, j = e / N;
                if (i < j) continue;
                double k = kc.sig * kern_eval<FAM, false>(x[j], y[j], x[i], y[i], kc);
                if (i == j) k += noise;
                A[i + j * ld] = k;
            }
        } else {
            for (int e = tid; e < N * N; e += LT) {
                const int i = e % N, j = e / N;               // pair (row point i, column point j)
                double kxx, kxy, kyy;
                pair_eval<FAM, false>(x[j], y[j], x[i], y[i], kc, kxx, kxy, kyy);
                if (i == j) { kxx += noise; kyy += noise; }
                A[(N + i) + j * ld] = kxy;                    // Pq block: always below the diagonal
                if (i >= j) {
                    A[i + j * ld] = kxx;
                    A[(N + i) + (N + j) * ld] = kyy;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- L11, X11 = inv(L11)
        leaf_body(s, s + LEAF * LLD, s + LEAF * LLD + PW * (PW + 1), n1, A, ld, inv, a.info + b, 0, (int)LEAF_FACTOR,
                  nullptr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (n2 > 0) {
            // L21 = A21 X11^T  (X11 lower: k >= c), staged through LDS so that A21 can be overwritten
            for (int e = tid; e < n2 * n1; e += LT) {
                const int i = e % n2, c = e / n2;
                double acc = 0.0;
                for (int k = 0; k <= c; ++k) acc = __builtin_fma(A[(n1 + i) + k * ld], inv[c + k * LEAF], acc);
                s[e] = acc;
            }
            __syncthreads();
            for (int e = tid; e < n2 * n1; e += LT) A[(n1 + e % n2) + (e / n2) * ld] = s[e];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // A22 -= L21 L21^T (lower)
            for (int e = tid; e < n2 * n2; e += LT) {
                const int i = e % n2, j = e / n2;
                if (i < j) continue;
                double acc = A[(n1 + i) + (n1 + j) * ld];
                for (int k = 0; k < n1; ++k) acc = __builtin_fma(-A[(n1 + i) + k * ld], A[(n1 + j) + k * ld], acc);
                A[(n1 + i) + (n1 + j) * ld] = acc;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            leaf_body(s, s + LEAF * LLD, s + LEAF * LLD + PW * (PW + 1), n2, A + n1 + n1 * ld, ld, inv + LEAF * LEAF,
                      a.info + b, n1, (int)LEAF_FACTOR, nullptr);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // ---- y = L^-1 z:  y1 = X11 z1;  y2 = X22 (z2 - L21 y1)
        double *yv = v, *al = v + BMAX;
        if (tid < n1) {
            double acc = 0.0;
            for (int k = 0; k <= tid; ++k) acc = __builtin_fma(inv[tid + k * LEAF], z[k], acc);
            yv[tid] = acc;
        }
        __syncthreads();
        if (n2 > 0) {
            if (tid < n2) {
                double acc = z[n1 + tid];
                for (int k = 0; k < n1; ++k) acc = __builtin_fma(-A[(n1 + tid) + k * ld], yv[k], acc);
                s[tid] = acc;
            }
            __syncthreads();
            if (tid < n2) {
                double acc = 0.0;
                const double *X22 = inv + LEAF * LEAF;
                for (int k = 0; k <= tid; ++k) acc = __builtin_fma(X22[tid + k * LEAF], s[k], acc);
                yv[n1 + tid] = acc;
            }
            __syncthreads();
            // ---- alpha = L^-T y:  a2 = X22^T y2;  a1 = X11^T (y1 - L21^T a2)
            if (tid < n2) {
                double acc = 0.0;
                const double *X22 = inv + LEAF * LEAF;
                for (int k = tid; k < n2; ++k) acc = __builtin_fma(X22[k + tid * LEAF], yv[n1 + k], acc);
                al[n1 + tid] = acc;
            }
            __syncthreads();
            if (tid < n1) {
                double acc = yv[tid];
                for (int k = 0; k < n2; ++k) acc = __builtin_fma(-A[(n1 + k) + tid * ld], al[n1 + k], acc);
                s[tid] = acc;
            }
            __syncthreads();
        } else {
            if (tid < n1) s[tid] = yv[tid];
            __syncthreads();
        }
        if (tid < n1) {
            double acc = 0.0;
            for (int k = tid; k < n1; ++k) acc = __builtin_fma(inv[k + tid * LEAF], s[k], acc);
            al[tid] = acc;
        }
        __syncthreads();
        // ---- nll = z.alpha / 2 + sum log L_ii  (func.py:195)
        double q = 0.0;
        for (int i = tid; i < n; i += LT) q += 0.5 * z[i] * al[i] + log(A[i + i * ld]);
        for (int o = 32; o > 0; o >>= 1) q += __shfl_down(q, o, 64);
        if ((tid & 63) == 0) red[tid >> 6] = q;
        __syncthreads();
        if (tid == 0) a.nll[b] = red[0] + red[1] + red[2] + red[3];
        if (a.alpha)
            for (int i = tid; i < n; i += LT) a.alpha[(size_t)b * n + i] = al[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// Device + pinned-host staging of one calling thread, grown on demand and kept: a call is then one H2D copy,
// one launch and one D2H copy (nine hipMalloc / hipFree pairs and eight small pageable copies per call were
// most of a single small fit's 230 us through a handle).
struct Arena {
    char *dev = nullptr, *host = nullptr;
    size_t cap_dev = 0, cap_host = 0;
    int device = -1;
    // no destructor: at thread / process exit the HIP runtime may already be gone; the OS reclaims
    void release()
    {
        if (dev) (void)hipFree(dev);
        if (host) (void)hipHostFree(host);
        dev = host = nullptr;
        cap_dev = cap_host = 0;
    }
    int reserve(size_t need_dev, size_t need_host)
    {
        int cur = 0;
        SGPR_HIP(hipGetDevice(&cur));
        if (cur != device) { release(); device = cur; }
        if (need_dev > cap_dev) {
            if (dev) (void)hipFree(dev);
            dev = nullptr; cap_dev = 0;
            SGPR_HIP(hipMalloc((void **)&dev, need_dev));
            cap_dev = need_dev;
        }
        if (need_host > cap_host) {
            if (host) (void)hipHostFree(host);
            host = nullptr; cap_host = 0;
            SGPR_HIP(hipHostMalloc((void **)&host, need_host, hipHostMallocDefault));
            cap_host = need_host;
        }
        return 0;
    }
};
thread_local Arena t_arena;

inline size_t up256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

int fit_batch_max_order() { return BMAX; }

// host buffers in, host buffers out; see include/sympgpr_hip.h (sgpr_fit_batch)
int fit_batch(int family, int nbatch, int npts, const double *x, const double *y, const double *z, const double *hyp,
              int nhyp, const double *sig2n, unsigned flags, double *alpha, double *nll, int *info)
{
    const int reg = (flags & SGPR_FIT_REG) ? 1 : 0;
    const int n = reg ? npts : 2 * npts;
    if (nbatch < 0 || npts <= 0 || n > BMAX || !x || !y || !z || !hyp || !sig2n || !nll || !info ||
        (flags & ~(unsigned)SGPR_FIT_REG)) {
        set_error("fit_batch: bad arguments (order per problem at most 256)");
        return SGPR_E_ARG;
    }
    if (nbatch == 0) return 0;
    if (family < SGPR_FAM_A || family > SGPR_FAM_D) { set_error("fit_batch: unknown kernel family"); return SGPR_E_ARG; }
    const size_t B = (size_t)nbatch;
    const int grid = nbatch < 1024 ? nbatch : 1024;
    const size_t per_wg = (size_t)BMAX * BMAX + 2 * (size_t)LEAF * LEAF + 2 * BMAX;
    // input block (one H2D): x | y | z | KConst | noise ; output block (one D2H): alpha | nll | info
    const size_t o_x = 0, o_y = o_x + up256(B * npts * 8), o_z = o_y + up256(B * npts * 8), o_kc = o_z + up256(B * n * 8),
                 o_no = o_kc + up256(B * sizeof(KConst)), in_bytes = o_no + up256(B * 8);
    const size_t o_al = 0, o_nll = o_al + up256(B * n * 8), o_info = o_nll + up256(B * 8), out_bytes = o_info + up256(B * sizeof(int));
    const size_t scr_bytes = (size_t)grid * per_wg * 8;
    Arena &ar = t_arena;
    int rc = ar.reserve(in_bytes + out_bytes + scr_bytes, in_bytes + out_bytes);
    if (rc) return rc;
    char *hin = ar.host, *hout = ar.host + in_bytes;
    char *din = ar.dev, *dout = ar.dev + in_bytes, *dscr = ar.dev + in_bytes + out_bytes;
    memcpy(hin + o_x, x, B * npts * 8);
    memcpy(hin + o_y, y, B * npts * 8);
    memcpy(hin + o_z, z, B * n * 8);
    KConst *kcs = reinterpret_cast<KConst *>(hin + o_kc);
    double *noise = reinterpret_cast<double *>(hin + o_no);
    for (int b = 0; b < nbatch; ++b) {
        if ((rc = make_kconst(family, hyp + (size_t)b * nhyp, nhyp, &kcs[b]))) return rc;
        noise[b] = std::fabs(sig2n[b]);
    }
    hipStream_t st = nullptr;
    SGPR_HIP(hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, st));
    SGPR_HIP(hipMemsetAsync(dout + o_info, 0, B * sizeof(int), st));
    BatchArgs a{nbatch, npts, n, reg, reinterpret_cast<double *>(din + o_x), reinterpret_cast<double *>(din + o_y),
                reinterpret_cast<double *>(din + o_z), reinterpret_cast<KConst *>(din + o_kc),
                reinterpret_cast<double *>(din + o_no), reinterpret_cast<double *>(dscr),
                reinterpret_cast<double *>(dout + o_al), reinterpret_cast<double *>(dout + o_nll),
                reinterpret_cast<int *>(dout + o_info)};
    switch (family) {
    case SGPR_FAM_A: hipLaunchKernelGGL(fit_batch_kernel<SGPR_FAM_A>, dim3(grid), dim3(LT), 0, st, a); break;
    case SGPR_FAM_B: hipLaunchKernelGGL(fit_batch_kernel<SGPR_FAM_B>, dim3(grid), dim3(LT), 0, st, a); break;
    case SGPR_FAM_C: hipLaunchKernelGGL(fit_batch_kernel<SGPR_FAM_C>, dim3(grid), dim3(LT), 0, st, a); break;
    default:         hipLaunchKernelGGL(fit_batch_kernel<SGPR_FAM_D>, dim3(grid), dim3(LT), 0, st, a); break;
    }
    SGPR_CHECK_LAUNCH();
    // without alpha only the tail of the output block comes back
    const size_t from = alpha ? 0 : o_nll;
    SGPR_HIP(hipMemcpyAsync(hout + from, dout + from, out_bytes - from, hipMemcpyDeviceToHost, st));
    SGPR_HIP(hipStreamSynchronize(st));
    if (alpha) memcpy(alpha, hout + o_al, B * n * 8);
    memcpy(nll, hout + o_nll, B * 8);
    memcpy(info, hout + o_info, B * sizeof(int));
    return 0;
}

}  // namespace sgpr
