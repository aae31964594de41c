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
#include "gemm_tile.h"
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
        leaf_body(s, n1, A, ld, inv, a.info + b, 0, (int)LEAF_FACTOR,
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
            leaf_body(s, n2, A + n1 + n1 * ld, ld, inv + LEAF * LEAF,
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

// ---- mid-size problems: 256 < n <= 2048 (a CMA-ES generation at N = 512 ... 1024 points) --------------------------
// Three launches for the whole batch: (1) every Ky_b, lower triangle, into an npad x npad image (npad = n rounded up to
// 128; the padding is an identity block: det 1, alpha 0 there); (2) chol.hip's panel_batch_kernel: W = npad / 128
// workgroups per problem run the leaf chain on their own matrix, problems side by side; (3) one workgroup per problem:
// y = L^-1 z and alpha = L^-T y through the leaf inverses, and the negative log-likelihood.
constexpr int MT = 256;                   // build: pair rows per tile (one per thread)
constexpr int MJ = 16;                    // build: pair columns per tile
constexpr int ST = 512;                   // solve: threads per problem

struct MidArgs {
    int nbatch, npts, n, npad, reg;
    const double *x, *y, *z;              // nbatch x npts, nbatch x npts, nbatch x n
    const KConst *kc;
    const double *noise;
    double *A;                            // problem b at A + b * npad * npad, ld = npad
    const double *inv;                    // leaf inverses: problem b at inv + b * (npad / 128) * 128 * 128
    double *alpha, *nll;
    const int *info;
};

template <int FAM>
__global__ __launch_bounds__(MT) void mid_build_kernel(const MidArgs a)
{
    __shared__ double sx[MJ], sy[MJ];
    const int b = blockIdx.z, N = a.npts, t = threadIdx.x;
    const int i0 = blockIdx.x * MT, j0 = blockIdx.y * MJ;
    const size_t ld = (size_t)a.npad;
    double *A = a.A + (size_t)b * ld * ld;
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int r = a.n + t; r < a.npad; r += MT) A[r + r * ld] = 1.0;     // the padding's diagonal (the rest was cleared)
    const bool tile_lower = i0 + MT - 1 >= j0;        // some (i, j) of the tile has i >= j
    if (a.reg && !tile_lower) return;
    const double *x = a.x + (size_t)b * N, *y = a.y + (size_t)b * N;
    const int nj = min(MJ, N - j0);
    if (t < nj) { sx[t] = x[j0 + t]; sy[t] = y[j0 + t]; }
    __syncthreads();
    const int i = i0 + t;
    if (i >= N) return;
    const KConst kc = a.kc[b];
    const double noise = a.noise[b];
    const double xi = x[i], yi = y[i];
    if (a.reg) {
        for (int jj = 0; jj < nj; ++jj) {
            const int j = j0 + jj;
            if (i < j) break;
            double k = kc.sig * kern_eval<FAM, false>(sx[jj], sy[jj], xi, yi, kc);
            if (i == j) k += noise;
            A[i + j * ld] = k;
        }
        return;
    }
    for (int jj = 0; jj < nj; ++jj) {
        const int j = j0 + jj;
        double kxx, kxy, kyy;
        pair_eval<FAM, false>(sx[jj], sy[jj], xi, yi, kc, kxx, kxy, kyy);
        if (i == j) { kxx += noise; kyy += noise; }
        A[(N + i) + j * ld] = kxy;                     // Pq block: always below the diagonal
        if (i >= j) {
            A[i + j * ld] = kxx;
            A[(N + i) + (N + j) * ld] = kyy;
        }
    }
}

// Above order 1024 a problem is factored as two panels; between them, for every problem, the rank-k update of the block that
// is left: A22 -= L21 L21^T (lower tiles only), 128 x 128 tiles of the grid-wide MFMA kernel's LDS-DMA body (gemm_tile.h).
struct MidSyrk {
    double *A;            // problem b at A + b * stride (column-major, ld)
    size_t stride, ld;
    int k, m2;            // width of the first panel; order of the block behind it (multiples of 128)
};

__global__ __launch_bounds__(256, 2) void mid_syrk_kernel(const MidSyrk a)
{
    __shared__ double smem[2 * tile::BK * (2 * (128 + tile::PAD))];
    int t = (int)blockIdx.x, r = 0;            // lower tiles row by row: row r holds r + 1 of them
    while (t > r) { t -= r + 1; ++r; }
    double *base = a.A + (size_t)blockIdx.y * a.stride;
    tile::GemmArgs g{};
    g.m = a.m2; g.n = a.m2; g.k = a.k;
    g.alpha = -1.0; g.beta = 1.0;
    g.A = base + a.k; g.lda = a.ld;
    g.B = base + a.k; g.ldb = a.ld;
    g.C = base + a.k + (size_t)a.k * a.ld; g.ldc = a.ld;
    tile::gemm_body_dma<128, 128, 2>(g, smem, r, t);
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;      // lane 0 holds the sum
}

// The solves of the batch: the right-hand sides padded to npad ...
__global__ __launch_bounds__(256) void mid_rhs_kernel(const MidArgs a, double *r)
{
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.npad) r[(size_t)b * a.npad + i] = i < a.n ? a.z[(size_t)b * a.n + i] : 0.0;
}
// ... two launches of trsv.hip's strip solve over all problems (one workgroup per 128-row strip and problem, W x nbatch of
// them; round 3 ran both solves of a problem in ONE workgroup: 64 problems of order 1024 kept 64 CUs busy for 0.6 ms, a third
// of the whole call) ... and, per problem, nll = z.alpha / 2 + sum log L_ii (func.py:195) and alpha.
__global__ __launch_bounds__(ST) void mid_finish_kernel(const MidArgs a, const double *r, const int *state, int *info)
{
    __shared__ double red[ST / 64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = a.n;
    const size_t ld = (size_t)a.npad;
    const double *A = a.A + (size_t)b * ld * ld;
    const double *z = a.z + (size_t)b * n;
    const double *al = r + (size_t)b * ld;
    // a strip solve that gave up on a hand-off (state word 2 of either pass): an error of the call, not a result
    if (a.info[b] == 0 && (state[8 * b + 2] | state[8 * b + 6]) != 0) {
        if (t == 0) { info[b] = SOLVE_HANDOFF_TIMEOUT; a.nll[b] = __builtin_nan(""); }
        return;
    }
    if (a.info[b] != 0) {                  // not positive definite: the host turns info into NaN / +inf
        if (t == 0) a.nll[b] = __builtin_nan("");
        if (a.alpha)
            for (int i = t; i < n; i += ST) a.alpha[(size_t)b * n + i] = __builtin_nan("");
        return;
    }
    double q = 0.0;
    for (int i = t; i < n; i += ST) q += 0.5 * z[i] * al[i] + log(A[i + i * ld]);
    q = wave_sum(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    if (t == 0) {
        double v = 0.0;
        for (int w = 0; w < ST / 64; ++w) v += red[w];
        a.nll[b] = v;
    }
    if (a.alpha)
        for (int i = t; i < n; i += ST) a.alpha[(size_t)b * n + i] = al[i];
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
        // grown for a large batch once, asked for a small one now: give the difference back (a CMA-ES population at order 2048
        // must not keep gigabytes pinned under the full-size fit that follows it)
        if (cap_dev > (256ull << 20) && need_dev * 4 < cap_dev) { if (dev) (void)hipFree(dev); dev = nullptr; cap_dev = 0; }
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

int fit_batch_max_order() { return potrf_batch_max_order(); }
int fit_batch_trim() { t_arena.release(); return 0; }     // the calling thread's device arena and pinned staging block

namespace {

// problems of order 256 < n <= 2048: see the kernels above.  Chunks of problems share the scratch images.
int fit_batch_mid(int family, int nbatch, int npts, int n, int reg, const double *x, const double *y, const double *z,
                  const double *hyp, int nhyp, const double *sig2n, double *alpha, double *nll, int *info)
{
    const int npad = (n + (int)LEAF - 1) / (int)LEAF * (int)LEAF, W = npad / (int)LEAF;
    const size_t img = (size_t)npad * npad * 8, invb = (size_t)W * LEAF * LEAF * 8;
    // at most ~2 GiB of images per chunk (64 problems of order 2048; round 3: 6 GiB), at least one chip-full of strips (256 workgroups)
    int chunk = (int)std::min<size_t>((size_t)nbatch, std::max<size_t>((256 + W - 1) / W, (2ull << 30) / img));
    if (chunk > 16384) chunk = 16384;      // grid.z of the build launch
    const size_t C = (size_t)chunk;
    const size_t o_x = 0, o_y = o_x + up256(C * npts * 8), o_z = o_y + up256(C * npts * 8), o_kc = o_z + up256(C * n * 8),
                 o_no = o_kc + up256(C * sizeof(KConst)), in_bytes = o_no + up256(C * 8);
    const size_t o_al = 0, o_nll = o_al + up256(C * n * 8), o_info = o_nll + up256(C * 8), out_bytes = o_info + up256(C * sizeof(int));
    // (+ the solves: right-hand sides / solution padded to npad, the two publication buffers, 8 state words per problem)
    const size_t o_A = 0, o_inv = o_A + up256(C * img), o_fl = o_inv + up256(C * invb), o_fl2 = o_fl + up256(potrf_batch_flag_bytes(chunk)),
                 o_r = o_fl2 + up256(potrf_batch_flag_bytes(chunk)), o_pub = o_r + up256(C * npad * 8), o_st = o_pub + up256(2 * C * npad * 8),
                 scr_bytes = o_st + up256(C * 8 * sizeof(int));
    Arena &ar = t_arena;
    int rc = ar.reserve(in_bytes + out_bytes + scr_bytes, in_bytes + out_bytes);
    if (rc) return rc;
    char *hin = ar.host, *hout = ar.host + in_bytes;
    char *din = ar.dev, *dout = ar.dev + in_bytes, *dscr = ar.dev + in_bytes + out_bytes;
    hipStream_t st = nullptr;
    for (int b0 = 0; b0 < nbatch; b0 += chunk) {
        const int nb = std::min(chunk, nbatch - b0);
        const size_t B = (size_t)nb;
        memcpy(hin + o_x, x + (size_t)b0 * npts, B * npts * 8);
        memcpy(hin + o_y, y + (size_t)b0 * npts, B * npts * 8);
        memcpy(hin + o_z, z + (size_t)b0 * n, B * n * 8);
        KConst *kcs = reinterpret_cast<KConst *>(hin + o_kc);
        double *noise = reinterpret_cast<double *>(hin + o_no);
        for (int b = 0; b < nb; ++b) {
            if ((rc = make_kconst(family, hyp + (size_t)(b0 + b) * nhyp, nhyp, &kcs[b]))) return rc;
            noise[b] = std::fabs(sig2n[b0 + b]);
        }
        SGPR_HIP(hipMemcpyAsync(din, hin, in_bytes, hipMemcpyHostToDevice, st));
        double *dA = reinterpret_cast<double *>(dscr + o_A), *dinv = reinterpret_cast<double *>(dscr + o_inv);
        int *dinfo = reinterpret_cast<int *>(dout + o_info);
        if (npad != n) SGPR_HIP(hipMemsetAsync(dA, 0, B * img, st));
        MidArgs a{nb, npts, n, npad, reg, reinterpret_cast<double *>(din + o_x), reinterpret_cast<double *>(din + o_y),
                  reinterpret_cast<double *>(din + o_z), reinterpret_cast<KConst *>(din + o_kc),
                  reinterpret_cast<double *>(din + o_no), dA, dinv, alpha ? reinterpret_cast<double *>(dout + o_al) : nullptr,
                  reinterpret_cast<double *>(dout + o_nll), dinfo};
        const dim3 grid((npts + MT - 1) / MT, (npts + MJ - 1) / MJ, nb);
        switch (family) {
        case SGPR_FAM_A: hipLaunchKernelGGL(mid_build_kernel<SGPR_FAM_A>, grid, dim3(MT), 0, st, a); break;
        case SGPR_FAM_B: hipLaunchKernelGGL(mid_build_kernel<SGPR_FAM_B>, grid, dim3(MT), 0, st, a); break;
        case SGPR_FAM_C: hipLaunchKernelGGL(mid_build_kernel<SGPR_FAM_C>, grid, dim3(MT), 0, st, a); break;
        case SGPR_FAM_USER: hipLaunchKernelGGL(mid_build_kernel<SGPR_FAM_USER>, grid, dim3(MT), 0, st, a); break;
        default:         hipLaunchKernelGGL(mid_build_kernel<SGPR_FAM_D>, grid, dim3(MT), 0, st, a); break;
        }
        SGPR_CHECK_LAUNCH();
        int *fl = reinterpret_cast<int *>(dscr + o_fl);
        static const int two_min = (int)tune("batch_two_min", 512);
        if (npad <= two_min) {
            if ((rc = potrf_batch(nb, npad, dA, (size_t)npad * npad, (size_t)npad, dinv, (size_t)W * LEAF * LEAF, fl, dinfo, st)))
                return rc;
        } else {
            // two panels above order 512 (SGPR_BATCH_TWO_MIN): in ONE panel the last strip alone has ~W^2 / 2 products of 128^3 to
            // do in a row (n = 2048: 3.0 ms for the launch); as W/2 + W/2 leaf columns (at most 8 first) with the update of the
            // second half on the matrix cores between them it is ~2 ms (per fit at 16 per batch: n = 2048 947 -> 326 us,
            // n = 1024 85 -> 75 us)
            const int W1 = std::min(8, (W + 1) / 2), k1 = W1 * (int)LEAF, m2 = npad - k1;
            SGPR_HIP(hipMemsetAsync(dinfo, 0, B * sizeof(int), st));
            if ((rc = potrf_batch_panel(nb, npad, 0, W1, dA, (size_t)npad * npad, (size_t)npad, dinv, (size_t)W * LEAF * LEAF, fl,
                                        dinfo, st)))
                return rc;
            const int t2 = m2 / (int)LEAF;
            hipLaunchKernelGGL(mid_syrk_kernel, dim3((unsigned)(t2 * (t2 + 1) / 2), (unsigned)nb), dim3(256), 0, st,
                               MidSyrk{dA, (size_t)npad * npad, (size_t)npad, k1, m2});
            SGPR_CHECK_LAUNCH();
            if ((rc = potrf_batch_panel(nb, npad, k1, W - W1, dA, (size_t)npad * npad, (size_t)npad, dinv, (size_t)W * LEAF * LEAF,
                                        reinterpret_cast<int *>(dscr + o_fl2), dinfo, st)))
                return rc;
        }
        // y = L^-1 z, alpha = L^-T y for every problem: the strip solve of trsv.hip, W x nb workgroups per launch
        double *dr = reinterpret_cast<double *>(dscr + o_r), *dpub = reinterpret_cast<double *>(dscr + o_pub);
        int *dst = reinterpret_cast<int *>(dscr + o_st);
        SGPR_HIP(hipMemsetAsync(dst, 0, B * 8 * sizeof(int), st));
        SGPR_HIP(hipMemsetAsync(dpub, 0xFF, 2 * B * npad * 8, st));
        hipLaunchKernelGGL(mid_rhs_kernel, dim3((unsigned)((npad + 255) / 256), (unsigned)nb), dim3(256), 0, st, a, dr);
        SGPR_CHECK_LAUNCH();
        if ((rc = trsv_strips_batch(nb, npad, dA, (size_t)npad * npad, (size_t)npad, dinv, (size_t)W * LEAF * LEAF, dr, (size_t)npad, 0,
                                    dst, 8, dpub, st)))
            return rc;
        if ((rc = trsv_strips_batch(nb, npad, dA, (size_t)npad * npad, (size_t)npad, dinv, (size_t)W * LEAF * LEAF, dr, (size_t)npad, 1,
                                    dst + 4, 8, dpub + B * npad, st)))
            return rc;
        hipLaunchKernelGGL(mid_finish_kernel, dim3(nb), dim3(ST), 0, st, a, (const double *)dr, (const int *)dst, dinfo);
        SGPR_CHECK_LAUNCH();
        const size_t from = alpha ? 0 : o_nll;
        SGPR_HIP(hipMemcpyAsync(hout + from, dout + from, out_bytes - from, hipMemcpyDeviceToHost, st));
        SGPR_HIP(hipStreamSynchronize(st));
        if (alpha) memcpy(alpha + (size_t)b0 * n, hout + o_al, B * n * 8);
        memcpy(nll + b0, hout + o_nll, B * 8);
        memcpy(info + b0, hout + o_info, B * sizeof(int));
        for (int b = 0; b < nb; ++b)
            if (info[b0 + b] < 0) return info_status(info[b0 + b]);     // a hand-off timed out: an error of the call
    }
    return 0;
}

}  // namespace

// host buffers in, host buffers out; see include/sympgpr_hip.h (sgpr_fit_batch)
int fit_batch(int family, int nbatch, int npts, const double *x, const double *y, const double *z, const double *hyp,
              int nhyp, const double *sig2n, unsigned flags, double *alpha, double *nll, int *info)
{
    const int reg = (flags & SGPR_FIT_REG) ? 1 : 0;
    const int n = reg ? npts : 2 * npts;
    if (nbatch < 0 || npts <= 0 || n > potrf_batch_max_order() || !x || !y || !z || !hyp || !sig2n || !nll || !info ||
        (flags & ~(unsigned)SGPR_FIT_REG)) {
        set_error("fit_batch: bad arguments (order per problem at most 2048)");
        return SGPR_E_ARG;
    }
    if (nbatch == 0) return 0;
    if (family < SGPR_FAM_A || family > SGPR_FAM_USER) { set_error("fit_batch: unknown kernel family"); return SGPR_E_ARG; }
    if (n > BMAX) return fit_batch_mid(family, nbatch, npts, n, reg, x, y, z, hyp, nhyp, sig2n, alpha, nll, info);
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
    case SGPR_FAM_USER: hipLaunchKernelGGL(fit_batch_kernel<SGPR_FAM_USER>, dim3(grid), dim3(LT), 0, st, a); break;
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
